"""GPU parity tests for mara::srhd and the `cloud` stage (BASELINE config 4) through the C ABI, against golden
vectors produced by the reference's own headers. MH_ARITH_STRICT: bit-exact; north_star's bound (conserved L1
<= 1e-12) is asserted too."""
import glob
import math
import os
import numpy as np
import pytest
from conftest import golden, bits_equal, l1, GOLDEN

pytestmark = pytest.mark.gpu
G = 4.0 / 3


@pytest.fixture(scope="module")
def eng():
    import mara3_amd
    from mara3_amd import engine
    assert mara3_amd.load_library().mh_device_count() >= 1
    return engine


def test_srhd_functions_bit_exact(eng):
    g = golden("srhd_functions")
    assert bits_equal(eng.srhd_to_conserved(g["Pl"], G), g["U"])
    P, st = eng.srhd_recover_primitive(g["U"], G, 0.0)
    assert not st.any()
    assert bits_equal(P, g["c2p"])
    for axis in range(3):
        assert bits_equal(eng.srhd_riemann_hlle(g["Pl"], g["Pr"], axis, G), g["hlle_%d" % axis]), axis
    cot = np.array([math.tan(math.pi / 2 - q) for q in g["src_q"]])       # libm tan, as the host geometry code
    assert bits_equal(eng.srhd_source_terms(g["Pl"], g["src_r"], cot, G), g["src"])


@pytest.mark.parametrize("name,floor", [("floor", 1e-8), ("nofloor", 0.0)])
def test_srhd_recover_primitive_status_matches_reference_exceptions(eng, name, floor):
    g = golden("srhd_functions")
    P, st = eng.srhd_recover_primitive(g["Ubad"], G, floor)
    threw = g["c2p_bad_%s_threw" % name] != 0
    assert np.array_equal(st != 0, threw)
    assert bits_equal(P[~threw], g["c2p_bad_" + name][~threw])


CLOUD_CASES = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, "cloud_*.npz")))


@pytest.mark.parametrize("chunk", [0, 5])
@pytest.mark.parametrize("case", CLOUD_CASES)
def test_cloud_steps_vs_reference_golden(eng, case, chunk):
    g = golden(case)
    theta = float(g["theta"]) if int(g["method"]) == 2 else -1.0
    s = eng.CloudSolver(g["rv"], g["qv"], int(g["rk"]), theta, float(g["tfloor"]), chunk_rows=chunk)
    s.upload(g["u0"])
    assert abs(s.timestep() - float(g["dt"])) == 0.0
    for n in range(int(g["nsteps"])):
        s.set_inflow(g["inflow"][n])          # nozzle row at the step-start time, used by both RK stages
        s.step(float(g["dt"]), 1)
    got = s.download()
    assert s.status() == 0
    assert l1(got, g["un"]) <= 1e-12
    assert bits_equal(got, g["un"]), np.abs(got - g["un"]).max()


def test_cloud_reports_c2p_failure_in_status_word(eng):
    g = golden("cloud_nr32_plm_rk1")
    u = g["u0"].copy()
    u[10, 12, 4] = -abs(u[10, 12, 4])          # negative tau: the reference's recover_primitive throws
    s = eng.CloudSolver(g["rv"], g["qv"], 1, 1.2, 0.0)
    s.upload(u)
    s.set_inflow(g["inflow"][0])
    s.step(float(g["dt"]), 1)
    assert s.status() != 0


def test_sedov_srhd_bit_exact_vs_reference(eng):
    """`mara sedov` with its default system mara::srhd (512 zones, PCM + HLLE + forward Euler) after 1/10/100 steps."""
    g = golden("sedov_srhd_nr256")
    s = eng.SedovSolver(g["vertices"], system="srhd")
    s.upload(g["u0"])
    dt = s.timestep()
    done = 0
    for n in (1, 10, 100):
        s.step(dt, n - done)
        done = n
        assert bits_equal(s.download(), g["u_%d" % n]), n
    assert s.status() == 0
