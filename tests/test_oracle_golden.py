"""CPU tests: the plain-C oracle (oracle/mara_oracle.c) against golden vectors that were
produced by the reference's own headers (oracle/gen_golden.py). Bit-exact everywhere."""
import glob
import os
import numpy as np
import pytest
from conftest import golden, bits_equal, GOLDEN


def test_plm_gradient_bit_exact(oracle):
    g = golden("plm_gradient")
    y = g["y"]
    for key in g.files:
        if key.startswith("g_"):
            theta = float(key[2:])
            got = oracle.plm_gradient(y[:, 0], y[:, 1], y[:, 2], theta)
            assert bits_equal(got, g[key]), key


@pytest.mark.parametrize("gname,gamma", [("53", 5.0 / 3), ("43", 4.0 / 3), ("14", 1.4)])
def test_euler_functions_bit_exact(oracle, gname, gamma):
    g = golden("euler_functions")
    assert bits_equal(oracle.euler_to_conserved_density(g["Pl"], gamma), g["U_" + gname])
    assert bits_equal(oracle.euler_recover_primitive(g["U_" + gname], gamma), g["c2p_" + gname])
    for axis in range(3):
        got = oracle.euler_riemann(g["Pl"], g["Pr"], axis, gamma, oracle.RIEMANN_HLLE)
        assert bits_equal(got, g["hlle_%s_%d" % (gname, axis)]), axis


def test_euler_temperature_floor(oracle):
    g = golden("euler_functions")
    assert bits_equal(oracle.euler_recover_primitive(g["Uneg"], 5.0 / 3, 1e-3), g["c2p_floor_53"])
    assert bits_equal(oracle.euler_recover_primitive(g["Uneg"], 5.0 / 3, 0.0), g["c2p_nofloor_53"])
    assert (g["c2p_nofloor_53"][:, 4] < 0).any()


def test_hllc_reduces_to_exact_flux_for_equal_states(oracle):
    """Euler HLLC has no upstream counterpart (parity unpinned): pin it by properties instead."""
    g = golden("euler_functions")
    P = g["Pl"]
    for axis in range(3):
        F = oracle.euler_riemann(P, P, axis, 1.4, oracle.RIEMANN_HLLC)
        Fx = g["flux_14_%d" % axis]
        assert np.allclose(F, Fx, rtol=1e-11, atol=1e-11 * np.abs(Fx).max(axis=1, keepdims=True))


def test_hllc_resolves_stationary_contact_exactly(oracle):
    Pl = np.array([[1.0, 0.0, 0.3, -0.2, 1.0]])
    Pr = np.array([[0.125, 0.0, 0.3, -0.2, 1.0]])
    F = oracle.euler_riemann(Pl, Pr, 0, 1.4, oracle.RIEMANN_HLLC)
    assert np.array_equal(F, np.array([[0.0, 1.0, 0.0, 0.0, 0.0]]))
    Fe = oracle.euler_riemann(Pl, Pr, 0, 1.4, oracle.RIEMANN_HLLE)
    assert abs(Fe[0, 0]) > 1e-3  # HLLE smears the contact


def test_hllc_mirror_symmetry(oracle):
    g = golden("euler_functions")
    # near-equal pairs: wave ordering sl < s* < sr holds, so the if-chain of the iso2d template is mirror symmetric
    Pl, Pr = g["Pl"][:512].copy(), g["Pr"][:512].copy()
    F = oracle.euler_riemann(Pl, Pr, 0, 5.0 / 3, oracle.RIEMANN_HLLC)
    Ql, Qr = Pr.copy(), Pl.copy()
    Ql[:, 1] *= -1
    Qr[:, 1] *= -1
    G = oracle.euler_riemann(Ql, Qr, 0, 5.0 / 3, oracle.RIEMANN_HLLC)
    G[:, [0, 2, 3, 4]] *= -1
    scale = np.abs(F).max(axis=1, keepdims=True) + 1e-300
    assert (np.abs(F - G) / scale).max() < 1e-12


STEP_CASES = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, "euler[123]d_*.npz")))


@pytest.mark.parametrize("case", STEP_CASES)
def test_euler_cart_steps_bit_exact(oracle, case):
    g = golden(case)
    for ns in g["nsteps"]:
        got = oracle.euler_cart_run(g["u0"], g["dl"], float(g["dt"]), int(ns), float(g["gamma"]),
                                    float(g["theta"]), int(g["rk"]), oracle.RIEMANN_HLLE, int(g["bc"]), nthreads=1)
        assert bits_equal(got, g["u_%d" % ns]), (case, ns)


def test_euler_cart_threads_do_not_change_results(oracle):
    g = golden("euler2d_wave33x70_plm12_rk2_outflow")
    a = oracle.euler_cart_run(g["u0"], g["dl"], float(g["dt"]), 3, 1.4, 1.2, 2, nthreads=1)
    b = oracle.euler_cart_run(g["u0"], g["dl"], float(g["dt"]), 3, 1.4, 1.2, 2, nthreads=5)
    assert bits_equal(a, b) and bits_equal(a, g["u_3"])


def test_sedov_bit_exact(oracle):
    g = golden("sedov_newtonian_nr256")
    v = oracle.sedov_vertices(256, 100.0)
    assert bits_equal(v, g["vertices"])
    u = oracle.sedov_initial(v)
    assert bits_equal(u, g["u0"])
    dt = oracle.sedov_timestep(v)
    for n in range(1, 101):
        u = oracle.sedov_advance(v, u, dt)
        if "u_%d" % n in g.files:
            assert bits_equal(u, g["u_%d" % n]), n


def test_decomposition_integers(oracle):
    g = golden("decomposition")
    for rank in (1, 2, 3):
        table = g["decomp_rank%d" % rank]
        for n in range(1, 33):
            assert oracle.propose_block_decomposition(rank, n) == tuple(table[n - 1]), (rank, n)
    assert oracle.propose_block_decomposition(3, 8) == (2, 2, 2)
    assert oracle.propose_block_decomposition(2, 8) == (2, 4)
    for key in g.files:
        if key.startswith("partition_") or key.startswith("blocks_"):
            kind, count, nparts = key.split("_")
            count, nparts = int(count), int(nparts)
            for p in range(nparts):
                a, b = oracle.partition_rows(count, nparts, p)
                if kind == "partition":
                    start, size = g[key][p]
                    assert b - a == size and (size == 0 or a == start), key
                else:
                    assert (a, b) == tuple(g[key][p]), key


@pytest.mark.parametrize("name,srhd", [("sedovdiag_newtonian_nr256", False), ("sedovdiag_srhd_nr256", True)])
def test_sedov_diagnostics_bit_exact(oracle, name, srhd):
    """SedovProblem::make_diagnostic_fields / compute_time_series_data (subprog_sedov.cpp:252-308: entropy, shock locator, parabola vertex,
    shock velocity) against the reference's own functions composed the same way (sedov_ref.cpp), both hydro systems, two epochs."""
    g = golden(name)
    for ns in (100, 400):
        fields, indices, series, status = oracle.sedov_diagnostics(g["vertices"], g["u_%d" % ns], g["series_%d" % ns][0], srhd)
        assert status == 0
        assert bits_equal(fields, g["fields_%d" % ns]) and np.array_equal(indices, g["indices_%d" % ns]) and bits_equal(series, g["series_%d" % ns])
