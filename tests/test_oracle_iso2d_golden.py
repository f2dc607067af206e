"""CPU tests: the plain-C iso2d oracle (oracle/mara_oracle_iso2d.c) against golden vectors produced by the
reference's physics_iso2d.hpp (oracle/ref_drivers/funcs_iso2d_ref.cpp), incl. the reference's own known-answer
tests (src/physics_test.cpp:101-153). Bit-exact."""
import numpy as np
from conftest import golden, bits_equal


def test_iso2d_conversions_bit_exact(oracle):
    g = golden("iso2d_functions")
    assert bits_equal(oracle.iso2d_to_conserved(g["Pl"]), g["U"])
    P, threw = oracle.iso2d_recover_primitive(g["U"])
    assert not threw.any() and bits_equal(P, g["c2p"])
    P, threw = oracle.iso2d_recover_primitive(g["Uneg"])
    ref_threw = g["c2p_neg_threw"] != 0
    assert np.array_equal(threw != 0, ref_threw) and ref_threw.any()
    assert bits_equal(P[~ref_threw], g["c2p_neg"][~ref_threw])
    assert bits_equal(oracle.iso2d_to_conserved_angmom(g["Pl"], g["x"]), g["Q"])
    P, threw = oracle.iso2d_recover_primitive_angmom(g["Q"], g["x"])
    assert not threw.any() and bits_equal(P, g["q2p"])


def test_reference_known_answers(oracle):
    """src/physics_test.cpp:101-141: P <-> U exact round trip; P <-> Q exact at x = (1, 2), 1e-14 at x = (1e-8, 1e-8);
    :143-153: HLLC contact speed is exactly 0 for Pl = (1,0,0), Pr = (2,0,0), cs2 = (1, 1/2)."""
    P = np.array([[1.0, 2.0, 3.0]])
    Pb, _ = oracle.iso2d_recover_primitive(oracle.iso2d_to_conserved(P))
    assert np.array_equal(Pb, P)
    for x, tol in (((1.0, 2.0), 0.0), ((1e-8, 1e-8), 1e-14)):
        xx = np.array([x])
        Pq, _ = oracle.iso2d_recover_primitive_angmom(oracle.iso2d_to_conserved_angmom(P, xx), xx)
        assert np.abs(Pq - P).max() <= tol
    F, contact, threw = oracle.iso2d_riemann([[1.0, 0, 0]], [[2.0, 0, 0]], [1.0], [0.5], 0, oracle.RIEMANN_HLLC)
    assert contact[0] == 0.0 and not threw[0]


def test_iso2d_fluxes_and_riemann_bit_exact(oracle):
    g = golden("iso2d_functions")
    for axis in range(2):
        assert bits_equal(oracle.iso2d_flux(g["Pl"], g["cs2l"], axis), g["flux_%d" % axis])
        assert bits_equal(oracle.iso2d_wavespeeds(g["Pl"], g["cs2l"], axis), g["lam_%d" % axis])
        F, _, _ = oracle.iso2d_riemann(g["Pl"], g["Pr"], g["cs2l"], g["cs2r"], axis, oracle.RIEMANN_HLLE)
        assert bits_equal(F, g["hlle_%d" % axis])
        F, contact, threw = oracle.iso2d_riemann(g["Pl"], g["Pr"], g["cs2l"], g["cs2r"], axis, oracle.RIEMANN_HLLC)
        ref_threw = g["hllc_threw_%d" % axis] != 0
        assert np.array_equal(threw != 0, ref_threw)
        assert bits_equal(contact, g["hllc_contact_%d" % axis])
        assert bits_equal(F[~ref_threw], g["hllc_%d" % axis][~ref_threw])
