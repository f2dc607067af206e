"""The error contract of SURVEY.md §8b on a live device: a step reports {status bits, first failing flat cell index}, and
mh_step_checked is a transaction - a rejected step leaves the previous solution downloadable bit for bit, as the reference's callers
expect when they catch the exception and retry from the old solution (src/subprog_binary.cpp:285-292); the bits are those of
include/mara_hip.h (src/physics_srhd.hpp:430-449 for the relativistic system)."""
import numpy as np
import pytest
from conftest import golden, bits_equal

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def mods():
    import mara3_amd
    from mara3_amd import engine, setups, _lib
    assert mara3_amd.load_library().mh_device_count() >= 1
    return engine, setups, _lib


@pytest.mark.parametrize("arith", ["strict", "fast"])
@pytest.mark.parametrize("rk", [1, 2])
@pytest.mark.parametrize("shape", [(96, 130), (24, 20, 70)])
def test_checked_steps_equal_plain_steps_when_healthy(mods, shape, rk, arith):
    engine, setups, L = mods
    gamma = 1.4
    dl = tuple(1.0 / n for n in shape)
    u0 = setups.wave_ic(shape, gamma, seed=31)
    a = engine.EulerCartSolver(shape, dl, gamma, 1.5, "hllc", rk, "periodic", arith=arith)
    b = engine.EulerCartSolver(shape, dl, gamma, 1.5, "hllc", rk, "periodic", arith=arith)
    a.upload(u0); b.upload(u0)
    a.step(5e-4, 5)
    for _ in range(5):
        assert b.step_checked(5e-4) == (0, None)
    assert bits_equal(a.download(), b.download())
    # and the two forms can be mixed
    a.step(5e-4, 2); b.step(5e-4, 1); assert b.step_checked(5e-4) == (0, None)
    assert bits_equal(a.download(), b.download())


@pytest.mark.parametrize("arith", ["strict", "fast"])
@pytest.mark.parametrize("rk", [1, 2])
@pytest.mark.parametrize("shape,cell", [((96, 130), (40, 77)), ((96, 130), (0, 0)), ((96, 130), (95, 129)), ((24, 20, 70), (13, 7, 33))])
@pytest.mark.parametrize("poison", ["density", "energy", "nan"])
def test_rejected_step_leaves_the_previous_solution(mods, shape, cell, rk, arith, poison):
    """A poisoned state - negative density, negative total energy (-> negative pressure) or a NaN - in the interior, in a corner and at
    the far edge: the step is rejected with the right kind of bit and a first failing cell inside the poisoned cell's stencil; the state
    that was uploaded comes back bit for bit; after repairing the cell the context steps on as if nothing had happened."""
    engine, setups, L = mods
    gamma = 1.4
    dl = tuple(1.0 / n for n in shape)
    good = setups.wave_ic(shape, gamma, seed=32)
    bad = good.copy()
    if poison == "density":
        bad[cell + (0,)] = -1.0
    elif poison == "energy":
        bad[cell + (4,)] = -5.0
    else:
        bad[cell + (0,)] = np.nan
    s = engine.EulerCartSolver(shape, dl, gamma, 1.5, "hllc", rk, "outflow", arith=arith)
    s.upload(bad)
    bits, first = s.step_checked(5e-4)
    assert bits != 0
    if poison == "energy":
        assert bits & L.STATUS_NEG_PRESSURE
    if poison == "nan":
        assert bits & L.STATUS_NAN
    if poison == "density":
        assert bits & (L.STATUS_NEG_DENSITY | L.STATUS_NEG_PRESSURE | L.STATUS_NAN)
    flat = int(np.ravel_multi_index(cell, shape))
    stride0 = int(np.prod(shape[1:]))
    assert max(0, flat - 2 * stride0) <= first <= flat, (first, flat)      # no cell before the stencil's first row; the cell itself at the latest
    assert bits_equal(s.download(), bad)                                   # untouched, NaN payloads and all
    assert s.status_result() == (0, None)
    s.upload(good)
    ref = engine.EulerCartSolver(shape, dl, gamma, 1.5, "hllc", rk, "outflow", arith=arith)
    ref.upload(good)
    ref.step(5e-4, 3)
    for _ in range(3):
        assert s.step_checked(5e-4) == (0, None)
    assert bits_equal(s.download(), ref.download())


def test_plain_step_reports_distinct_bits_and_first_cell(mods):
    engine, setups, L = mods
    shape, gamma = (64, 64), 5.0 / 3
    u = setups.blast_ic(shape, gamma, radius=0.25)
    u[20, 30, 4] = -1.0                      # negative total energy: recovered pressure < 0
    s = engine.EulerCartSolver(shape, (1 / 64, 1 / 64), gamma, -1.0, "hlle", 1, "outflow")      # PCM + RK1: the cell's own check fires
    s.upload(u)
    s.step(1e-4, 1)
    bits, first = s.status_result()
    assert bits & L.STATUS_NEG_PRESSURE
    assert 19 * 64 + 30 <= first <= 20 * 64 + 30
    assert s.status_result() == (0, None)    # reading clears


@pytest.mark.parametrize("rk", [1, 2])
def test_cloud_checked_step_is_transactional_and_names_the_cell(mods, rk):
    """mara::srhd::recover_primitive throws for a negative tau (src/physics_srhd.hpp:430-449); on the device: the status bits of that
    throw, the flat index of exactly that cell, and the solution untouched."""
    engine, setups, L = mods
    g = golden("cloud_nr32_plm_rk%d" % rk)
    nq = g["u0"].shape[1]
    u = g["u0"].copy()
    u[10, 12, 4] = -abs(u[10, 12, 4])
    s = engine.CloudSolver(g["rv"], g["qv"], rk, 1.2, 0.0)
    s.upload(u)
    s.set_inflow(g["inflow"][0])
    bits, first = s.step_checked(float(g["dt"]))
    assert bits != 0
    if rk == 1:
        assert first == 10 * nq + 12                          # one stage: only the poisoned cell's own recover_primitive fails
    else:
        assert 8 * nq + 10 <= first <= 10 * nq + 12           # the second stage also sees what the first made of its neighbours
    assert bits_equal(s.download(), u)
    # healthy: checked steps reproduce the reference vector
    t = engine.CloudSolver(g["rv"], g["qv"], rk, float(g["theta"]), float(g["tfloor"]))
    t.upload(g["u0"])
    for n in range(int(g["nsteps"])):
        t.set_inflow(g["inflow"][n])
        assert t.step_checked(float(g["dt"])) == (0, None)
    assert bits_equal(t.download(), g["un"])


def test_sedov_checked_step(mods):
    engine, setups, L = mods
    g = golden("sedov_srhd_nr256")
    s = engine.SedovSolver(g["vertices"], system="srhd")
    s.upload(g["u0"])
    dt = s.timestep()
    for _ in range(10):
        assert s.step_checked(dt) == (0, None)
    assert bits_equal(s.download(), g["u_10"])
    u = s.download()
    u[100, 4] = -abs(u[100, 4]) - 1.0
    s.upload(u)
    bits, first = s.step_checked(dt)
    assert bits != 0 and first is not None
    assert bits_equal(s.download(), u)
