"""GPU parity tests for the 3-D Euler stage (BASELINE config 5 at fixture size) through the C ABI: bit-exact
against reference-generated golden vectors and the oracle on ragged shapes; FAST within L1 <= 1e-12."""
import glob
import os
import numpy as np
import pytest
from conftest import golden, bits_equal, l1, GOLDEN

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    import mara3_amd
    from mara3_amd import engine
    assert mara3_amd.load_library().mh_device_count() >= 1
    return engine


CASES = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, "euler3d_*.npz")))


@pytest.mark.parametrize("case", CASES)
def test_euler3d_steps_vs_reference_golden(eng, case):
    g = golden(case)
    bc = "periodic" if int(g["bc"]) == 1 else "outflow"
    for ns in g["nsteps"]:
        s = eng.EulerCartSolver(g["u0"].shape[:3], g["dl"], float(g["gamma"]), float(g["theta"]), "hlle", int(g["rk"]), bc)
        s.upload(g["u0"])
        s.step(float(g["dt"]), int(ns))
        got = s.download()
        assert s.status() == 0
        want = g["u_%d" % ns]
        assert l1(got, want) <= 1e-12
        assert bits_equal(got, want), (case, ns, np.abs(got - want).max())


@pytest.mark.parametrize("shape,chunk", [((5, 3, 4), 0), ((9, 8, 60), 4), ((12, 17, 61), 5), ((20, 9, 130), 7), ((6, 25, 7), 0)])
@pytest.mark.parametrize("riemann,bc,theta", [("hlle", "outflow", 1.5), ("hllc", "periodic", 2.0), ("hlle", "periodic", -1.0)])
def test_euler3d_ragged_shapes_vs_oracle(eng, oracle, shape, chunk, riemann, bc, theta):
    """Tile edges on all three axes: shapes that are not multiples of the 8-row tile, the 60-column strip or chunk_rows."""
    from mara3_amd import setups
    gamma = 1.4
    u0 = setups.wave_ic(shape, gamma, seed=sum(shape))
    dl = (1.0 / shape[0], 0.8 / shape[1], 1.3 / shape[2])
    dt = 0.08 * min(dl)
    s = eng.EulerCartSolver(shape, dl, gamma, theta, riemann, 2, bc, chunk_rows=chunk)
    s.upload(u0)
    s.step(dt, 2)
    got = s.download()
    kind = oracle.RIEMANN_HLLC if riemann == "hllc" else oracle.RIEMANN_HLLE
    obc = oracle.BC_PERIODIC if bc == "periodic" else oracle.BC_OUTFLOW
    want = oracle.euler_cart_run(u0, dl, dt, 2, gamma, theta, 2, kind, obc)
    assert bits_equal(got, want), np.abs(got - want).max()


def test_euler3d_fast_within_tolerance(eng):
    g = golden("euler3d_blast24_plm15_rk2")
    s = eng.EulerCartSolver((24, 24, 24), g["dl"], float(g["gamma"]), 1.5, "hlle", 2, "outflow", arith="fast")
    s.upload(g["u0"])
    s.step(float(g["dt"]), 4)
    assert l1(s.download(), g["u_4"]) <= 1e-12


@pytest.mark.parametrize("arith", ["strict", "fast"])
def test_euler3d_result_does_not_depend_on_the_cut_along_axis_0(eng, arith):
    """the launcher's own choice of planes per work item (csrc/euler3d.hip: euler3d_default_chunk) against
    explicit cuts, ragged ones included: bit-identical"""
    from mara3_amd import setups
    shape, gamma = (130, 24, 70), 5.0 / 3
    dl = (1.0 / 130,) * 3
    u0 = setups.blast_ic(shape, gamma)
    dt = setups.baseline_dt(130)
    out = {}
    for chunk in (0, 32, 7, 130):
        s = eng.EulerCartSolver(shape, dl, gamma, 1.5, "hlle", 2, "outflow", arith=arith, chunk_rows=chunk)
        s.upload(u0)
        s.step(dt, 3)
        out[chunk] = s.download()
        assert s.status() == 0
        s.close()
    for chunk in (32, 7, 130):
        assert bits_equal(out[0], out[chunk]), chunk


def test_euler3d_blast_128_properties(eng, oracle):
    """A 128^3 blast (config-5 IC at a size one GPU test can hold): conservation, and a sub-block equal to the
    oracle run on that block plus context (domain of dependence), bit for bit."""
    from mara3_amd import setups
    n, gamma = 128, 5.0 / 3
    dl = (1.0 / n,) * 3
    dt = setups.baseline_dt(n)
    u0 = setups.blast_ic((n, n, n), gamma)
    s = eng.EulerCartSolver((n, n, n), dl, gamma, 1.5, "hlle", 2, "outflow")
    s.upload(u0)
    s.step(dt, 3)
    got = s.download()
    assert s.status() == 0
    for q in (0, 4):
        assert abs(got[..., q].sum() - u0[..., q].sum()) <= 1e-11 * abs(u0[..., q].sum())
    pad = 4 * 3 + 2
    a, b = 44, 60
    sub = oracle.euler_cart_run(u0[a - pad:b + pad, a - pad:b + pad, 30 - pad:100 + pad], dl, dt, 3, gamma, 1.5, 2,
                                oracle.RIEMANN_HLLE, oracle.BC_OUTFLOW, nthreads=8)
    assert bits_equal(got[a:b, a:b, 30:100], sub[pad:-pad, pad:-pad, pad:-pad])


@pytest.mark.timeout(600)
def test_config5_full_rank_size_symmetry_and_conservation(eng):
    """BASELINE config 5 at one rank's full size (512^3 of the 1024^3 / 8-GPU case; PLM 1.5 + HLLE, RK2, fast arithmetic): a centred blast
    stays mirror-symmetric about the three mid-planes (to rounding: a reflection swaps the left / right roles inside the Riemann solver,
    so the last bits may differ) and, as long as the wave is far from the outflow boundaries, mass and energy are conserved to rounding."""
    from mara3_amd import setups
    n, gamma = 512, 5.0 / 3
    u0 = setups.blast_ic((n, n, n), gamma)
    s = eng.EulerCartSolver((n, n, n), (1.0 / n,) * 3, gamma, 1.5, "hlle", 2, "outflow", arith="fast")
    s.upload(u0)
    mass0, energy0 = float(u0[..., 0].sum()), float(u0[..., 4].sum())
    del u0
    s.step(0.3 / n / 6, 3)
    u = s.download()
    assert s.status() == 0
    s.close()
    assert abs(float(u[..., 0].sum()) - mass0) <= 1e-12 * mass0 and abs(float(u[..., 4].sum()) - energy0) <= 1e-12 * energy0
    scale = np.abs(u[..., 4]).max()
    for axis in range(3):
        sign = np.ones(5)
        sign[1 + axis] = -1.0                                   # the momentum along the mirrored axis changes sign
        mirror = np.flip(u, axis=axis) * sign
        assert np.abs(u - mirror).max() <= 1e-12 * scale, axis
        del mirror
    assert (u[..., 0] > 0).all() and np.abs(u[n // 2, n // 2, :, 3]).max() > 0      # the blast edge on the central line is moving


@pytest.mark.parametrize("arith", ["strict", "fast"])
@pytest.mark.parametrize("poison", ["nan", "negative_density", "negative_energy"])
def test_bad_states_raise_the_status_word_3d(eng, poison, arith):
    from mara3_amd import setups
    shape, gamma = (24, 20, 70), 5.0 / 3
    u0 = setups.blast_ic(shape, gamma, radius=0.3)
    bad = u0.copy()
    if poison == "nan":
        bad[11, 9, 33, 0] = np.nan
    elif poison == "negative_density":
        bad[0, 3, 64, 0] = -1.0
    else:
        bad[23, 19, 1, 4] = -50.0
    s = eng.EulerCartSolver(shape, tuple(1.0 / n for n in shape), gamma, 1.5, "hlle", 2, "outflow", arith=arith)
    s.upload(bad)
    s.step(1e-3, 1)
    assert s.status() != 0
    s.upload(u0)
    s.step(1e-3, 2)
    assert s.status() == 0
    s.close()
