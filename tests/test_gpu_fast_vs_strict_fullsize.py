"""FAST against STRICT arithmetic at the BASELINE size on a workload without quiescent regions (the blast leaves ~97 % of a
4096^2 grid untouched after 25 steps, so a mean over all cells flatters the L1). north_star's bound: conserved-variable L1 <= 1e-12;
here additionally a max-norm bound, for HLLE (STRICT is bit-identical to the reference for it) and HLLC, and the blast L1 restricted to
the cells the wave has reached."""
import numpy as np
import pytest

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(900)]

L1_TOL = 1e-12          # north_star
MAX_TOL = 1e-11         # max-norm over all cells and variables (measured: a few 1e-14 after 20 steps)


@pytest.fixture(scope="module")
def mods():
    import mara3_amd
    from mara3_amd import engine, setups
    assert mara3_amd.load_library().mh_device_count() >= 1
    return engine, setups


def run(engine, u0, n, riemann, arith, bc, nsteps, dt):
    s = engine.EulerCartSolver((n, n), (1.0 / n, 1.0 / n), 5.0 / 3, 1.5, riemann, 2, bc, arith=arith)
    s.upload(u0)
    s.step(dt, nsteps)
    out = s.download()
    assert s.status_result() == (0, None)
    s.close()
    return out


@pytest.mark.parametrize("riemann", ["hlle", "hllc"])
def test_smooth_wave_4096_fast_vs_strict(mods, riemann):
    engine, setups = mods
    n = 4096
    u0 = setups.smooth_wave_ic((n, n), 5.0 / 3)
    dt = setups.baseline_dt(n)
    strict = run(engine, u0, n, riemann, "strict", "periodic", 20, dt)
    fast = run(engine, u0, n, riemann, "fast", "periodic", 20, dt)
    moved = np.abs(strict - u0).reshape(-1, 5).max(axis=1) > 0
    assert moved.mean() > 0.99                     # every cell takes part
    diff = np.abs(fast - strict)
    assert diff.mean() <= L1_TOL, diff.mean()
    assert diff.max() <= MAX_TOL, diff.max()
    for q in range(5):
        assert diff[..., q].mean() <= L1_TOL


@pytest.mark.parametrize("riemann", ["hlle", "hllc"])
def test_blast_4096_fast_vs_strict_on_the_cells_the_wave_reached(mods, riemann):
    engine, setups = mods
    n = 4096
    u0 = setups.blast_ic((n, n), 5.0 / 3)
    dt = setups.baseline_dt(n)
    strict = run(engine, u0, n, riemann, "strict", "outflow", 25, dt)
    fast = run(engine, u0, n, riemann, "fast", "outflow", 25, dt)
    active = np.abs(strict - u0).reshape(n, n, 5).max(axis=2) > 0
    assert 0.001 < active.mean() < 0.2              # a thin shell around the charge
    diff = np.abs(fast - strict)
    assert diff[active].mean() <= L1_TOL, diff[active].mean()
    assert diff.max() <= MAX_TOL, diff.max()
    # static gas: STRICT leaves it bit-identical; FAST re-forms the conserved state from the primitives at the update, which may move the
    # energy by an ulp (4e-17 here) once - it does not accumulate (the re-formed state is a fixed point of the round trip)
    assert diff[~active].max() <= 1e-15
