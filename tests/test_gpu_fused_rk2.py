"""The fused RK2 step (mara3_amd/csrc/euler2d_fused.hip: both stages of `s0 * 0.5 + advance(advance(s0)) * 0.5`,
src/subprog_cloud.cpp:676-697, in ONE launch - the first-stage field lives in an LDS ring between a producer and a consumer wave)
against the two launches it replaces. Same FastArith functions on the same values in the same order, so the requirement is
BIT-IDENTITY with the two-launch FAST path - which the other tests hold to the reference (L1 <= 1e-12, tests/test_gpu_parity.py
::test_fast_*). Covered: outflow and periodic on either axis (the ghost cells of the first-stage field are made in LDS, not in
memory), ragged shapes (strips and chunks that do not divide the grid, chunks of 1-3 rows at the end), chunk lengths down to 2,
both Riemann solvers, the context API, the native slab stepper with and without graph replay, odd and even step counts (the two
fields swap every step), the status contract and the transactional step."""
import numpy as np
import pytest
from conftest import bits_equal

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(600)]


@pytest.fixture(scope="module")
def eng():
    import mara3_amd
    from mara3_amd import engine
    assert mara3_amd.load_library().mh_device_count() >= 1
    return engine


def run(eng, shape, riemann, bc, fuse, u0, dt, pieces, chunk_rows=0, bc0=None):
    dl = (1.0 / shape[0], 1.0 / shape[1])
    s = eng.EulerCartSolver(shape, dl, 1.4, 1.5, riemann, 2, bc, bc_lo0=bc0, bc_hi0=bc0, arith="fast", fuse=fuse, chunk_rows=chunk_rows)
    s.upload(u0)
    out = []
    for n in pieces:
        s.step(dt, n)
        out.append(s.download())
    st = s.status()
    s.close()
    return out, st


@pytest.mark.parametrize("riemann", ["hllc", "hlle"])
@pytest.mark.parametrize("bc", ["outflow", "periodic"])
@pytest.mark.parametrize("shape,chunk", [((250, 300), 0), ((64, 56), 0), ((129, 113), 16), ((67, 200), 7), ((40, 500), 2), ((200, 64), 3)])
def test_fused_step_is_bit_identical_to_the_two_launches(eng, shape, chunk, bc, riemann):
    from mara3_amd import setups
    u0 = setups.wave_ic(shape, 1.4, seed=31)
    pieces = (1, 2, 3)                       # odd and even step counts: the result lands in either field
    two, st2 = run(eng, shape, riemann, bc, False, u0, 4e-4, pieces, chunk_rows=chunk)
    one, st1 = run(eng, shape, riemann, bc, True, u0, 4e-4, pieces, chunk_rows=chunk)
    assert st1 == 0 and st2 == 0
    for a, b, n in zip(one, two, pieces):
        assert bits_equal(a, b), (shape, chunk, bc, riemann, n, np.abs(a - b).max())


@pytest.mark.parametrize("bc0,bc1", [("outflow", "periodic"), ("periodic", "outflow")])
def test_fused_step_with_different_boundary_kinds_per_axis(eng, bc0, bc1):
    from mara3_amd import setups
    shape = (150, 170)
    u0 = setups.wave_ic(shape, 1.4, seed=37)
    two, _ = run(eng, shape, "hllc", bc1, False, u0, 4e-4, (3,), bc0=bc0)
    one, _ = run(eng, shape, "hllc", bc1, True, u0, 4e-4, (3,), bc0=bc0)
    assert bits_equal(one[0], two[0])


def test_fused_step_on_the_blast_with_the_default_chunks(eng):
    """a grid of several chunks and strips at the defaults (64 rows, 56 columns), the workload of the headline: blast, outflow"""
    from mara3_amd import setups
    shape = (520, 600)
    u0 = setups.blast_ic(shape, 5.0 / 3)
    dt = setups.baseline_dt(shape[0])
    dl = (1.0 / shape[0], 1.0 / shape[1])
    res = []
    for fuse in (False, True):
        s = eng.EulerCartSolver(shape, dl, 5.0 / 3, 1.5, "hllc", 2, "outflow", arith="fast", fuse=fuse)
        s.upload(u0); s.step(dt, 25)
        res.append(s.download()); assert s.status() == 0
        s.close()
    assert bits_equal(res[0], res[1])


@pytest.mark.parametrize("graph", [False, True])
def test_native_slab_stepper_takes_the_fused_step_and_replays_it_from_graphs(eng, graph, monkeypatch):
    """(the stepper issues the fused step as a plain launch - a one-node graph replay is 29 us per step slower on this stack - and keeps
    the replay behind MH_SLAB_FUSED_GRAPH=1, which the graph=True case sets: both ways, the same bits)"""
    from mara3_amd import setups
    if graph:
        monkeypatch.setenv("MH_SLAB_FUSED_GRAPH", "1")
    from mara3_amd.slab import NativeSlabStepper
    shape, gamma = (256, 300), 1.4
    dl = (1.0 / shape[0], 1.0 / shape[1])
    u0 = setups.wave_ic(shape, gamma, seed=41)
    ref = eng.EulerCartSolver(shape, dl, gamma, 1.5, "hllc", 2, "periodic", arith="fast", fuse=False)
    ref.upload(u0)
    st = NativeSlabStepper(shape, dl, gamma, 1.5, "hllc", 2, "periodic", arith="fast", fuse=True)
    st.load_slab(u0)
    for nsteps in (1, 4, 3, 2, 5):          # replays alternate between the two captured directions; a download in between
        ref.step(1e-3, nsteps)
        st.step(1e-3, nsteps, graph=graph)
        st.synchronize()
        assert bits_equal(st.slab_host(), ref.download()), nsteps
    assert st.status() == 0
    st.profile(True)
    st.step(1e-3, 3, graph=graph); st.synchronize()
    (ms1, ms2), (n1, n2), rows = st.profile_read()
    assert n1 == 0 and n2 == 3 and ms2 > 0          # the step's one launch is reported in the second-stage slot
    st.close(); ref.close()


def test_fused_is_refused_where_it_does_not_exist_and_skipped_where_it_is_optional(eng):
    import mara3_amd
    shape, dl = (64, 64), (1.0 / 64, 1.0 / 64)
    with pytest.raises(mara3_amd.MaraHipError, match="fuse_stages"):
        eng.EulerCartSolver(shape, dl, 1.4, 1.5, "hlle", 2, "outflow", arith="strict", fuse=True)
    with pytest.raises(mara3_amd.MaraHipError, match="fuse_stages"):
        eng.EulerCartSolver(shape, dl, 1.4, -1.0, "hlle", 2, "outflow", arith="fast", fuse=True)
    with pytest.raises(mara3_amd.MaraHipError, match="fuse_stages"):
        eng.EulerCartSolver(shape, dl, 1.4, 1.5, "hlle", 1, "outflow", arith="fast", fuse=True)
    s = eng.EulerCartSolver(shape, dl, 1.4, 1.5, "hlle", 2, "outflow", arith="strict")      # optional: the two launches
    s.close()


def test_fused_step_reports_the_failing_cell_and_the_checked_step_keeps_the_old_state(eng):
    """the error contract of include/mara_hip.h on the fused path: a poisoned cell raises the same bits and the same first index as the
    two launches, and mh_step_checked rejects the step with the previous solution still in place"""
    import ctypes as C
    from mara3_amd import setups, _lib as L
    shape = (96, 120)
    dl = (1.0 / shape[0], 1.0 / shape[1])
    u0 = setups.wave_ic(shape, 1.4, seed=43)
    u0[40, 57, 4] = -5.0                      # negative energy: the recovered pressure is negative
    res = []
    for fuse in (False, True):
        s = eng.EulerCartSolver(shape, dl, 1.4, 1.5, "hllc", 2, "outflow", arith="fast", fuse=fuse)
        s.upload(u0)
        r = L.StepResult()
        rc = s.lib.mh_step_checked(s.ctx, 4e-4, C.byref(r))
        assert rc == L.E_PHYSICS if hasattr(L, "E_PHYSICS") else rc != 0
        assert bits_equal(s.download(), u0)
        res.append((r.status, int(r.first_bad_index)))
        s.close()
    assert res[0] == res[1] and res[0][0] & L.STATUS_NEG_PRESSURE and res[0][1] == 40 * shape[1] + 57


def test_every_position_of_the_domain_edge_within_a_workgroup(eng):
    """A workgroup is two producer / consumer pairs that share the first stage's halo through LDS (116 output columns): every column count
    from 8 to 260, and a few around the next multiples, so that the domain's edge falls on every lane of either pair - both boundary kinds,
    both Riemann solvers, the library's chunking and short chunks - against the two launches, bit for bit (262 cases, 3 steps each)."""
    from mara3_amd import setups
    gamma = 1.4
    rng = np.random.default_rng(7)
    for n1 in list(range(8, 261)) + [347, 348, 349, 463, 464, 465, 579, 580, 581]:
        n0 = int(rng.choice([8, 9, 13, 31, 64, 101]))
        bc = ("outflow", "periodic")[n1 % 2]
        riemann = ("hllc", "hlle")[(n1 // 2) % 2]
        chunk = int(rng.choice([0, 0, 2, 5, 11]))
        shape, dl = (n0, n1), (1.0 / n0, 1.0 / n1)
        u0 = setups.wave_ic(shape, gamma, seed=n1)
        res = []
        for fuse in (False, True):
            s = eng.EulerCartSolver(shape, dl, gamma, 1.5, riemann, 2, bc, arith="fast", fuse=fuse, chunk_rows=chunk)
            s.upload(u0); s.step(0.2 * min(dl) / 2.0, 3)
            res.append(s.download())
            assert s.status() == 0
            s.close()
        assert bits_equal(res[0], res[1]), (n0, n1, bc, riemann, chunk)
