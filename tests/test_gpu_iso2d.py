"""GPU parity tests for mara::iso2d through the C ABI: bit-exact against golden
vectors produced by the reference's physics_iso2d.hpp, including its own known-answer tests."""
import numpy as np
import pytest
from conftest import golden, bits_equal

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    import mara3_amd
    from mara3_amd import engine
    assert mara3_amd.load_library().mh_device_count() >= 1
    return engine


def test_iso2d_conversions_bit_exact(eng):
    g = golden("iso2d_functions")
    assert bits_equal(eng.iso2d_to_conserved(g["Pl"]), g["U"])
    P, threw = eng.iso2d_recover_primitive(g["U"])
    assert not threw.any() and bits_equal(P, g["c2p"])
    P, threw = eng.iso2d_recover_primitive(g["Uneg"])
    ref_threw = g["c2p_neg_threw"] != 0
    assert np.array_equal(threw != 0, ref_threw)
    assert bits_equal(P[~ref_threw], g["c2p_neg"][~ref_threw])
    assert bits_equal(eng.iso2d_to_conserved_angmom(g["Pl"], g["x"]), g["Q"])
    P, threw = eng.iso2d_recover_primitive_angmom(g["Q"], g["x"])
    assert not threw.any() and bits_equal(P, g["q2p"])


def test_iso2d_fluxes_and_riemann_bit_exact(eng):
    g = golden("iso2d_functions")
    for axis in range(2):
        assert bits_equal(eng.iso2d_flux(g["Pl"], g["cs2l"], axis), g["flux_%d" % axis])
        assert bits_equal(eng.iso2d_wavespeeds(g["Pl"], g["cs2l"], axis), g["lam_%d" % axis])
        F, _, _ = eng.iso2d_riemann(g["Pl"], g["Pr"], g["cs2l"], g["cs2r"], axis, "hlle")
        assert bits_equal(F, g["hlle_%d" % axis])
        F, contact, threw = eng.iso2d_riemann(g["Pl"], g["Pr"], g["cs2l"], g["cs2r"], axis, "hllc")
        assert not threw.any()
        assert bits_equal(contact, g["hllc_contact_%d" % axis])
        assert bits_equal(F, g["hllc_%d" % axis])


def test_reference_hllc_known_answer(eng):
    """src/physics_test.cpp:143-153: contact speed exactly 0 for Pl = (1,0,0), Pr = (2,0,0), cs2 = (1, 1/2)."""
    F, contact, threw = eng.iso2d_riemann([[1.0, 0, 0]], [[2.0, 0, 0]], [1.0], [0.5], 0, "hllc")
    assert contact[0] == 0.0 and not threw[0]
