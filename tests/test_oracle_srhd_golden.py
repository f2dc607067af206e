"""CPU tests: the plain-C SRHD / cloud oracle (oracle/mara_oracle_srhd.c) against golden vectors produced by
the reference's own headers (oracle/ref_drivers/funcs_srhd_ref.cpp, cloud_ref.cpp). Bit-exact."""
import glob
import os
import numpy as np
import pytest
from conftest import golden, bits_equal, GOLDEN

G = 4.0 / 3


def test_srhd_functions_bit_exact(oracle):
    g = golden("srhd_functions")
    assert bits_equal(oracle.srhd_to_conserved_density(g["Pl"], G), g["U"])
    P, st = oracle.srhd_recover_primitive(g["U"], G, 0.0)
    assert not st.any() and not g["c2p_threw"].any()
    assert bits_equal(P, g["c2p"])
    for axis in range(3):
        assert bits_equal(oracle.srhd_riemann_hlle(g["Pl"], g["Pr"], axis, G), g["hlle_%d" % axis]), axis
    assert bits_equal(oracle.srhd_source_terms(g["Pl"], g["src_r"], g["src_q"], G), g["src"])


@pytest.mark.parametrize("name,floor", [("floor", 1e-8), ("nofloor", 0.0)])
def test_srhd_recover_primitive_failures_match_reference_exceptions(oracle, name, floor):
    """Where the reference throws (not converged / rho <= 0 / p <= 0 / NaN W) the restatement reports status bits;
    everywhere else the primitives are bit-identical, with the temperature floor active in many cells."""
    g = golden("srhd_functions")
    P, st = oracle.srhd_recover_primitive(g["Ubad"], G, floor)
    threw = g["c2p_bad_%s_threw" % name] != 0
    assert np.array_equal(st != 0, threw)
    assert threw.any() and (~threw).any()
    assert bits_equal(P[~threw], g["c2p_bad_" + name][~threw])


CLOUD_CASES = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, "cloud_*.npz")))


@pytest.mark.parametrize("case", CLOUD_CASES)
def test_cloud_steps_bit_exact(oracle, case):
    g = golden(case)
    theta = float(g["theta"]) if int(g["method"]) == 2 else -1.0
    u, st = oracle.cloud_run(g["u0"], g["rv"], g["qv"], g["inflow"], float(g["dt"]), int(g["nsteps"]), int(g["rk"]), theta, float(g["tfloor"]))
    assert st == 0
    assert bits_equal(u, g["un"]), np.abs(u - g["un"]).max()


def test_sedov_srhd_bit_exact(oracle):
    """`mara sedov` with its default system (newtonian=0 -> mara::srhd), 512 zones, 100 steps."""
    g = golden("sedov_srhd_nr256")
    v = oracle.sedov_vertices(256, 100.0)
    assert bits_equal(v, g["vertices"])
    u = oracle.sedov_initial_srhd(v)
    assert bits_equal(u, g["u0"])
    dt = oracle.sedov_timestep(v)
    for n in range(1, 101):
        u, st = oracle.sedov_advance_srhd(v, u, dt)
        assert st == 0
        if "u_%d" % n in g.files:
            assert bits_equal(u, g["u_%d" % n]), n


CLOUD_DIAG_CASES = ["clouddiag_nr48_150steps", "clouddiag_nr40_pcm_60steps", "clouddiag_nr32_4steps"]


@pytest.mark.parametrize("case", CLOUD_DIAG_CASES)
def test_cloud_diagnostic_fields_bit_exact(oracle, case):
    """CloudProblem::make_diagnostic_fields (subprog_cloud.cpp:334-433): the five 2-D fields and the fifteen per-angle arrays (shock
    locator of post_shock_locator.hpp included) against the reference's own operators composed the same way (cloud_ref.cpp)."""
    g = golden(case)
    fields, columns, status = oracle.cloud_diagnostics(g["un"], g["rv"], g["qv"], g["diag_meta"][1:4], float(g["tfloor"]))
    assert status == 0
    assert bits_equal(fields, g["diag_fields"])
    assert bits_equal(columns, g["diag_columns"])
