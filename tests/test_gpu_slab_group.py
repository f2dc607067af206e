"""The NATIVE multi-rank slab stepper (mara3_amd/csrc/slab.hip: mh_slab_*) executed for real on ONE GPU.

RCCL refuses two ranks on one device ("Duplicate GPU detected"), so the ranks of a decomposition are created as objects of one
process and exchange their ghost rows through the stepper's LOOPBACK backend: stream-ordered device-to-device copies under the very
event protocol the RCCL ranks use. Everything else is the code `bench.py --gpus N` runs on N GPUs: the nd::partition_shape cut
(src/core_ndarray.hpp:820-836; src/app_parallel.hpp:76-103 is the evaluator the slabs replace), ranks with lo != hi, ranks with one
physical and one external side, uneven cuts, 2-row edge strips, staggered edges, edge / interior streams, RK1 and RK2.

Requirement: per-cell arithmetic does not depend on the partition, so the union of the slabs is BIT-IDENTICAL to
the single-domain run - and, where a reference vector exists, to the reference."""
import numpy as np
import pytest
from conftest import golden, bits_equal

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(300)]


@pytest.fixture(scope="module")
def eng():
    import mara3_amd
    from mara3_amd import engine
    assert mara3_amd.load_library().mh_device_count() >= 1
    return engine


def run_group(shape, dl, gamma, theta, riemann, rk, bc, world, u0, dt, nsteps, arith="strict", chunk_rows=0, pieces=None):
    from mara3_amd.slab import NativeSlabGroup
    g = NativeSlabGroup(shape, dl, gamma, theta, riemann, rk, bc, world=world, arith=arith, chunk_rows=chunk_rows)
    g.upload(u0)
    for n in (pieces or [nsteps]):
        g.step(dt, n)
    g.synchronize()
    out = g.download()
    status = g.status()
    rows = list(g.rows)
    g.close()
    return out, status, rows


@pytest.mark.parametrize("world", [2, 3, 8])
@pytest.mark.parametrize("case", ["euler2d_wave33x70_plm12_rk2_outflow", "euler2d_blast128_plm15_rk2", "euler2d_wave48x40_plm15_rk2_periodic",
                                  "euler2d_blast64_plm20_rk1", "euler2d_blast64_pcm_rk1"])
def test_group_equals_reference_golden(eng, case, world):
    """2, 3 and 8 native slabs against the REFERENCE's vectors (strict arithmetic, HLLE): 33 rows over 8 ranks gives slabs of 4 and 5
    rows (interior launch empty or one row), 128 over 3 an uneven cut; outflow (rank 0 has no lo, the last rank no hi) and periodic
    (every rank has both; world 2: lo == hi)."""
    g = golden(case)
    u0 = g["u0"]
    bc = "periodic" if int(g["bc"]) == 1 else "outflow"
    done, pieces = 0, []
    for ns in sorted(int(n) for n in g["nsteps"]):
        pieces.append(ns - done)
        done = ns
    theta = float(g["theta"])
    got, status, rows = run_group(u0.shape[:2], g["dl"], float(g["gamma"]), theta, "hlle", int(g["rk"]), bc, world, u0, float(g["dt"]), done, pieces=pieces)
    assert status == (0, None)
    assert rows[0][0] == 0 and rows[-1][1] == u0.shape[0] and all(rows[r][1] == rows[r + 1][0] for r in range(world - 1))
    assert bits_equal(got, g["u_%d" % done]), (case, world, np.abs(got - g["u_%d" % done]).max())


@pytest.mark.parametrize("bc", ["outflow", "periodic"])
@pytest.mark.parametrize("world", [2, 3, 8])
@pytest.mark.parametrize("riemann,arith,rk", [("hllc", "fast", 2), ("hlle", "strict", 2), ("hllc", "strict", 1)])
def test_group_equals_single_domain_with_staggered_edges(eng, world, bc, riemann, arith, rk):
    """Slabs thick enough for the staggered edge schedule (>= 20 rows per rank: 4-stage period, edges of 2, 4, 6, 8 rows), an uneven
    cut (250 rows), step counts that end in the middle of a period and a download in between."""
    from mara3_amd import setups
    from mara3_amd.slab import NativeSlabGroup
    shape, gamma = (250, 300), 1.4
    dl = (1.0 / shape[0], 1.0 / shape[1])
    u0 = setups.wave_ic(shape, gamma, seed=21)
    ref = eng.EulerCartSolver(shape, dl, gamma, 1.5, riemann, rk, bc, arith=arith)
    ref.upload(u0)
    grp = NativeSlabGroup(shape, dl, gamma, 1.5, riemann, rk, bc, world=world, arith=arith)
    grp.upload(u0)
    for nsteps in (1, 2, 3, 1, 5):
        ref.step(5e-4, nsteps)
        grp.step(5e-4, nsteps)
        grp.synchronize()
        assert bits_equal(grp.download(), ref.download()), (world, bc, nsteps)
    assert grp.status() == (0, None)
    grp.close()


@pytest.mark.parametrize("cuts", ["0", "1"])
@pytest.mark.parametrize("riemann", ["hllc", "hlle"])
@pytest.mark.parametrize("shape, world, bc", [((250, 300), 2, "outflow"), ((250, 300), 3, "periodic"), ((41, 100), 5, "outflow"), ((61, 130), 5, "periodic"), ((96, 130), 8, "periodic"),
                                              ((97, 61), 4, "outflow")])
def test_group_fused_step_across_cuts_and_the_two_launch_schedule(eng, shape, world, bc, riemann, cuts, monkeypatch):
    """Round 3: FAST RK2 slabs WITH neighbours take the fused step too - four rows of each neighbour, one exchange per step, the
    neighbours' first-stage rows recomputed inside the launch (slab.hip: group_fused_cut_step) - unless MH_SLAB_FUSED_CUTS=0 keeps the
    two-launch schedule with its exchange per stage. Both are the one-domain result bit for bit: slabs down to twelve rows (thinner ones keep the two-launch schedule), ranks with one
    and with two neighbours, periodic cuts (both sides external on every rank), ragged column counts, a download in between."""
    from mara3_amd import setups
    from mara3_amd.slab import NativeSlabGroup
    monkeypatch.setenv("MH_SLAB_FUSED_CUTS", cuts)
    gamma = 1.4
    dl = (1.0 / shape[0], 1.0 / shape[1])
    u0 = setups.wave_ic(shape, gamma, seed=23)
    ref = eng.EulerCartSolver(shape, dl, gamma, 1.5, riemann, 2, bc, arith="fast")
    ref.upload(u0)
    grp = NativeSlabGroup(shape, dl, gamma, 1.5, riemann, 2, bc, world=world, arith="fast")
    grp.upload(u0)
    for nsteps in (1, 2, 4):
        ref.step(4e-4, nsteps)
        grp.step(4e-4, nsteps)
        grp.synchronize()
        assert bits_equal(grp.download(), ref.download()), (shape, world, bc, cuts, nsteps)
    assert grp.status() == (0, None)
    grp.close(); ref.close()


@pytest.mark.parametrize("cuts", ["1", "0"])
@pytest.mark.parametrize("on_launch", [1, 0])
@pytest.mark.parametrize("delay", [1, 2, 3])
@pytest.mark.parametrize("stagger", [0, 4])
def test_group_dependencies_hold_under_shifted_timing(eng, delay, stagger, on_launch, cuts, monkeypatch):
    """Three slabs (outflow: a rank with only a hi neighbour, one with both, one with only a lo neighbour) with a ~150 us sleeping wave
    queued in front of every edge launch (1), every interior launch (2) or both (3): the events alone must order the two chains of every
    rank AND the copies between ranks. With and without staggered edges, events carried by the launches or recorded behind them."""
    from mara3_amd import setups
    from mara3_amd.slab import NativeSlabGroup
    if cuts == "1" and (stagger or not on_launch):
        pytest.skip("the fused step across cuts has one schedule: the stagger / event switches belong to the two-launch one")
    monkeypatch.setenv("MH_SLAB_FUSED_CUTS", cuts)
    monkeypatch.setenv("MH_SLAB_TEST_DELAY", str(delay))
    monkeypatch.setenv("MH_SLAB_STAGGER", str(stagger))
    monkeypatch.setenv("MH_SLAB_EVENT_ON_LAUNCH", str(on_launch))
    shape, gamma = (192, 260), 1.4
    dl = (1.0 / shape[0], 1.0 / shape[1])
    u0 = setups.wave_ic(shape, gamma, seed=22)
    ref = eng.EulerCartSolver(shape, dl, gamma, 1.5, "hllc", 2, "outflow", arith="fast")
    ref.upload(u0)
    ref.step(5e-4, 7)
    grp = NativeSlabGroup(shape, dl, gamma, 1.5, "hllc", 2, "outflow", world=3, arith="fast")
    grp.upload(u0)
    grp.step(5e-4, 7)
    grp.synchronize()
    assert bits_equal(grp.download(), ref.download())
    grp.close()


@pytest.mark.parametrize("world", [8, 2])
def test_group_baseline_cut_4096_over_8(eng, world):
    """The BASELINE cut itself: 4096^2 over 8 ranks (512 x 4096 cells per rank, the staggered schedule and the one-residency-round
    chunk heuristic active), PLM + HLLC RK2 FAST as bench.py runs it, 6 steps, bit-identical to the one-domain run; per-rank bit
    fingerprints as bench.py's partition_check forms them. And over 2 ranks. Since late round 3 both cuts take the fused step across
    their cuts by the library's own choice (slabs of 384 rows and more); the two-launch schedule at this size is covered by
    tests/test_gpu_bench_contract.py with MH_SLAB_FUSED_CUTS=0 and by the parametrised tests above."""
    import torch
    from mara3_amd import setups
    from mara3_amd.slab import NativeSlabGroup, slab_fingerprint, partition_rows
    n, gamma = 4096, 5.0 / 3
    dl = (1.0 / n, 1.0 / n)
    dt = setups.baseline_dt(n)
    u0 = setups.blast_ic((n, n), gamma)
    ref = eng.EulerCartSolver((n, n), dl, gamma, 1.5, "hllc", 2, "outflow", arith="fast")
    ref.upload(u0)
    ref.step(dt, 6)
    whole = ref.download()
    ref.close()
    grp = NativeSlabGroup((n, n), dl, gamma, 1.5, "hllc", 2, "outflow", world=world, arith="fast")
    grp.upload(u0)
    grp.step(dt, 6)
    grp.synchronize()
    assert grp.status() == (0, None)
    for r in range(world):
        a, b = partition_rows(n, world, r)
        assert (a, b) == grp.rows[r] == (n // world * r, n // world * (r + 1))
        mine = grp.member_host(r)
        assert slab_fingerprint(torch.from_numpy(mine)) == slab_fingerprint(torch.from_numpy(whole[a:b])), r
        assert bits_equal(mine, whole[a:b]), r
    grp.close()


@pytest.mark.parametrize("world", [2, 3])
@pytest.mark.parametrize("case", ["euler3d_blast24_plm15_rk2", "euler3d_wave20x12x16_plm15_rk2_periodic"])
def test_group_3d_axis0_slabs_equal_reference_golden(eng, case, world):
    g = golden(case)
    u0 = g["u0"]
    bc = "periodic" if int(g["bc"]) == 1 else "outflow"
    last = max(int(n) for n in g["nsteps"])
    got, status, rows = run_group(u0.shape[:3], g["dl"], float(g["gamma"]), float(g["theta"]), "hlle", int(g["rk"]), bc, world, u0, float(g["dt"]), last)
    assert status == (0, None)
    assert bits_equal(got, g["u_%d" % last]), (case, world)


def test_group_reports_first_failing_cell_in_global_index(eng):
    """Error contract: the status comes back as {bits, first failing flat cell index} - here a negative density planted in
    rank 2's share; the index is in the order of the GLOBAL host array."""
    from mara3_amd import setups
    from mara3_amd import _lib as L
    from mara3_amd.slab import NativeSlabGroup
    shape, gamma = (96, 130), 1.4
    dl = (1.0 / shape[0], 1.0 / shape[1])
    u0 = setups.wave_ic(shape, gamma, seed=23)
    u0[70, 45, 0] = -1.0
    grp = NativeSlabGroup(shape, dl, gamma, 1.5, "hllc", 2, "outflow", world=4, arith="fast")
    grp.upload(u0)
    grp.step(1e-4, 1)
    bits, first = grp.status()
    assert bits != 0
    # the poisoned cell fails first in row-major order among everything it contaminates except its own stencil rows above it
    assert first is not None and 68 * shape[1] <= first <= 70 * shape[1] + 45
    assert grp.status() == (0, None)          # reading clears
    grp.close()


CLOUD_GROUP_CASES = ["cloud_nr70_plm_rk2", "cloud_nr32_plm_rk2", "cloud_nr32_plm_rk1", "cloud_nr24_pcm_rk1", "cloud_nr20x2dec_plm_rk2"]


@pytest.mark.parametrize("world", [2, 3, 4])
@pytest.mark.parametrize("case", CLOUD_GROUP_CASES)
def test_cloud_group_equals_reference_golden(eng, case, world):
    """BASELINE config 4's decomposition in the NATIVE stepper: radial slabs of the `cloud` grid (CloudProblem::advance,
    src/subprog_cloud.cpp:511-584, under evaluate_on<N>, :527-581), nozzle row per step at the step-start time, rank 0 with the inflow
    boundary, the last rank with the zero-gradient one, two-row exchange. Bit-identical to the REFERENCE's single-domain vectors."""
    from mara3_amd.slab import NativeSlabGroup
    g = golden(case)
    theta = float(g["theta"]) if int(g["method"]) == 2 else -1.0
    grp = NativeSlabGroup(r_vertices=g["rv"], q_vertices=g["qv"], gamma=4.0 / 3, plm_theta=theta, rk_order=int(g["rk"]), world=world,
                          temperature_floor=float(g["tfloor"]))
    grp.upload(g["u0"])
    for n in range(int(g["nsteps"])):
        grp.set_inflow(g["inflow"][n])
        grp.step(float(g["dt"]), 1)
    grp.synchronize()
    got = grp.download()
    assert grp.status() == (0, None)
    assert bits_equal(got, g["un"]), (case, world, np.abs(got - g["un"]).max())
    grp.close()


def test_cloud_group_fast_arith_equals_single_domain(eng):
    from mara3_amd.slab import NativeSlabGroup
    g = golden("cloud_nr70_plm_rk2")
    one = eng.CloudSolver(g["rv"], g["qv"], 2, 1.2, float(g["tfloor"]), arith="fast")
    one.upload(g["u0"])
    grp = NativeSlabGroup(r_vertices=g["rv"], q_vertices=g["qv"], gamma=4.0 / 3, plm_theta=1.2, rk_order=2, world=4,
                          temperature_floor=float(g["tfloor"]), arith="fast")
    grp.upload(g["u0"])
    for n in range(int(g["nsteps"])):
        one.set_inflow(g["inflow"][n]); one.step(float(g["dt"]), 1)
        grp.set_inflow(g["inflow"][n]); grp.step(float(g["dt"]), 1)
    grp.synchronize()
    assert bits_equal(grp.download(), one.download())
    grp.close()


@pytest.mark.timeout(900)
@pytest.mark.parametrize("arith", ["strict", "fast"])
def test_cloud_config4_as_configured_4096_over_4_slabs(eng, arith):
    """BASELINE config 4 as stated - `cloud` at 4096^2 (nr = 4096, one decade), RK2, PLM theta = 1.2, 4 radial slabs with the two-row halo -
    on one GPU through the loopback backend, against the single-domain run: 3 steps with a time-dependent nozzle row, bit-identical, status
    clean. (The state is a synthetic admissible SRHD flow on the sub-program's grid; the sub-program's own initial condition lives in the
    compiled host, tests/test_gpu_host_subprograms.py.)"""
    from mara3_amd.slab import NativeSlabGroup
    nr = 4096
    rv = 10.0 ** np.linspace(0.0, 1.0, nr + 1)
    qv = np.linspace(0.0, np.pi, nr + 1)
    rc, qc = 0.5 * (rv[1:] + rv[:-1]), 0.5 * (qv[1:] + qv[:-1])
    gamma = 4.0 / 3
    rho = rc[:, None] ** -2.0 * (1.0 + 0.3 * np.cos(3 * qc)[None, :])
    ur = 0.4 * np.exp(-((rc[:, None] - 3.0) / 1.5) ** 2) * (1.0 + 0.5 * np.cos(qc)[None, :] ** 2)
    uq = 0.05 * np.sin(2 * qc)[None, :] * np.ones_like(rho)
    pre = 1e-3 * rho
    W = np.sqrt(1.0 + ur ** 2 + uq ** 2)
    h = 1.0 + pre / rho * (gamma / (gamma - 1.0))
    D = rho * W
    dv = (rv[1:] ** 3 - rv[:-1] ** 3)[:, None] * (-np.cos(qv[1:]) + np.cos(qv[:-1]))[None, :] * 2 * np.pi / 3.0
    u0 = np.zeros((nr, nr, 5))
    u0[..., 0] = D * dv
    u0[..., 1] = D * h * ur * dv
    u0[..., 2] = D * h * uq * dv
    u0[..., 4] = (D * h * W - pre - D) * dv
    dt = 0.4 * (rv[1] - rv[0])
    def nozzle(step):
        row = np.zeros((nr, 5))
        row[:, 0] = 1.0
        row[:, 1] = 0.6 * np.exp(-0.5 * (qc / 0.3) ** 2) * np.exp(-0.1 * step) + 0.6 * np.exp(-0.5 * ((np.pi - qc) / 0.3) ** 2) * np.exp(-0.1 * step)
        row[:, 4] = 1e-3
        return row
    one = eng.CloudSolver(rv, qv, 2, 1.2, 1e-8, arith=arith)
    one.upload(u0)
    grp = NativeSlabGroup(r_vertices=rv, q_vertices=qv, gamma=gamma, plm_theta=1.2, rk_order=2, world=4, temperature_floor=1e-8, arith=arith)
    assert grp.rows == [(1024 * r, 1024 * (r + 1)) for r in range(4)]
    grp.upload(u0)
    for step in range(3):
        one.set_inflow(nozzle(step)); one.step(dt, 1)
        grp.set_inflow(nozzle(step)); grp.step(dt, 1)
    grp.synchronize()
    assert one.status() == 0 and grp.status() == (0, None)
    assert bits_equal(grp.download(), one.download())
    grp.close(); one.close()


@pytest.mark.timeout(600)
@pytest.mark.parametrize("cuts", ["0", "1"])
@pytest.mark.parametrize("world,shape", [(8, (1000, 600)), (5, (333, 1030))])
def test_group_long_run_stays_bit_identical(eng, world, shape, cuts, monkeypatch):
    """300 steps (600 stages, 75 stagger periods) of an uneven cut: an ordering hole between the two chains of a rank, or between ranks,
    that only opens now and then would show up here as a single differing bit. Both schedules: two launches with an exchange per stage,
    and the fused step across the cuts with one exchange per step."""
    from mara3_amd import setups
    from mara3_amd.slab import NativeSlabGroup
    monkeypatch.setenv("MH_SLAB_FUSED_CUTS", cuts)
    gamma = 1.4
    dl = (1.0 / shape[0], 1.0 / shape[1])
    u0 = setups.wave_ic(shape, gamma, seed=51)
    dt = 0.2 * min(dl) / 2.0
    ref = eng.EulerCartSolver(shape, dl, gamma, 1.5, "hllc", 2, "periodic", arith="fast")
    ref.upload(u0)
    grp = NativeSlabGroup(shape, dl, gamma, 1.5, "hllc", 2, "periodic", world=world, arith="fast")
    grp.upload(u0)
    for _ in range(6):
        ref.step(dt, 50)
        grp.step(dt, 50)
    grp.synchronize()
    assert grp.status() == (0, None) and ref.status() == 0
    assert bits_equal(grp.download(), ref.download())
    grp.close()


@pytest.mark.timeout(600)
def test_group_fused_step_across_cuts_on_random_shapes(eng, monkeypatch):
    """Twenty seeded random decompositions (rows 24 .. 400, columns 8 .. 300, 2 .. 9 slabs of at least 12 rows, either boundary kind, either
    Riemann solver, chunk lengths from the library's choice down to 2 rows): the fused step across the cuts against the one-domain run, bit
    for bit, after an odd and an even number of steps."""
    from mara3_amd import setups
    from mara3_amd.slab import NativeSlabGroup
    monkeypatch.setenv("MH_SLAB_FUSED_CUTS", "1")
    rng = np.random.default_rng(20261004)
    for case in range(20):
        world = int(rng.integers(2, 10))
        n0 = int(rng.integers(12 * world, max(12 * world + 1, 401)))
        n1 = int(rng.integers(8, 301))
        bc = ("outflow", "periodic")[int(rng.integers(0, 2))]
        riemann = ("hllc", "hlle")[int(rng.integers(0, 2))]
        chunk = int(rng.choice([0, 0, 2, 3, 7, 16]))
        shape, gamma = (n0, n1), 1.4
        dl = (1.0 / n0, 1.0 / n1)
        u0 = setups.wave_ic(shape, gamma, seed=100 + case)
        ref = eng.EulerCartSolver(shape, dl, gamma, 1.5, riemann, 2, bc, arith="fast")
        ref.upload(u0)
        grp = NativeSlabGroup(shape, dl, gamma, 1.5, riemann, 2, bc, world=world, arith="fast", chunk_rows=chunk)
        grp.upload(u0)
        dt = 0.2 * min(dl) / 2.0
        for nsteps in (3, 2):
            ref.step(dt, nsteps); grp.step(dt, nsteps); grp.synchronize()
            assert bits_equal(grp.download(), ref.download()), (case, shape, world, bc, riemann, chunk, nsteps)
        assert grp.status() == (0, None)
        grp.close(); ref.close()


def test_group_create_on_leaves_no_slab_behind_when_a_device_is_missing_or_refuses_peer_access(eng, monkeypatch):
    """mh_slab_group_create_on (one process driving several GPUs; the reference's thread slabs, src/app_parallel.hpp:75-103) must not leak members
    when it fails half-way. (i) a device id this box does not have: the members created before it are destroyed, every handle of the caller's
    array is null, the text names the device. (ii) the branch behind hipDeviceEnablePeerAccess - unreachable with one GPU, so the library takes
    it on request (MH_SLAB_TEST_PEER_FAIL=1): ALL members are destroyed, the array nulled, and the text is the peer-access failure, not a
    later destructor's."""
    import ctypes as C
    from mara3_amd import _lib as L
    from mara3_amd.slab import euler_cart_desc
    lib = L.load_library()
    world, shape = 3, (48, 40)
    d = euler_cart_desc(shape, (1.0 / 48, 1.0 / 40), 1.4, 1.5, "hllc", "outflow", 0, "fast")
    handles = (C.c_void_p * world)(*[C.c_void_p(0xdead)] * world)
    ids = (C.c_int * world)(0, 0, lib.mh_device_count() + 5)
    rc = lib.mh_slab_group_create_on(handles, C.byref(d), 2, world, ids)
    assert rc != 0 and all(not h for h in handles), (rc, list(handles))
    assert b"device" in lib.mh_last_error(None)
    # the PRODUCT library has no such hook (advisor finding, round 4): with the variable set its groups are created as ever
    monkeypatch.setenv("MH_SLAB_TEST_PEER_FAIL", "1")
    g = eng_slab_group(shape, world)
    g.close()
    monkeypatch.delenv("MH_SLAB_TEST_PEER_FAIL")
    # the CHECK library (-DMH_TEST_HOOKS) takes the branch on request, in a child process of its own
    import os, subprocess, sys
    from conftest import ROOT
    check = os.path.join(ROOT, "mara3_amd", "libmara_hip_check.so")
    assert os.path.exists(check), "build the check libraries: make -C mara3_amd/csrc check (__graft_entry__.build() does)"
    code = ("import sys, ctypes as C; sys.path.insert(0, %r)\n"
            "from mara3_amd import _lib as L\n"
            "from mara3_amd.slab import euler_cart_desc, NativeSlabGroup\n"
            "lib = L.load_library(); world, shape = 3, (48, 40)\n"
            "d = euler_cart_desc(shape, (1.0 / 48, 1.0 / 40), 1.4, 1.5, 'hllc', 'outflow', 0, 'fast')\n"
            "handles = (C.c_void_p * world)(*[C.c_void_p(0xdead)] * world)\n"
            "ids = (C.c_int * world)(0, 0, 0)\n"
            "rc = lib.mh_slab_group_create_on(handles, C.byref(d), 2, world, ids)\n"
            "assert rc != 0 and all(not h for h in handles), (rc, list(handles))\n"
            "assert b'hipDeviceEnablePeerAccess' in lib.mh_last_error(None), lib.mh_last_error(None)\n"
            "import os; del os.environ['MH_SLAB_TEST_PEER_FAIL']\n"
            "g = NativeSlabGroup(shape, (1.0 / 48, 1.0 / 40), 1.4, 1.5, 'hllc', 2, 'outflow', world=world, arith='fast'); g.close()\n"      # still usable
            "print('peer-fail branch ok')\n" % ROOT)
    env = dict(os.environ, MARA_HIP_LIBRARY=check, MH_SLAB_TEST_PEER_FAIL="1")
    p = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0 and "peer-fail branch ok" in p.stdout, (p.stdout[-800:], p.stderr[-1500:])


def eng_slab_group(shape, world):
    from mara3_amd.slab import NativeSlabGroup
    from mara3_amd import setups
    g = NativeSlabGroup(shape, (1.0 / shape[0], 1.0 / shape[1]), 1.4, 1.5, "hllc", 2, "outflow", world=world, arith="fast")
    g.upload(setups.wave_ic(shape, 1.4, seed=2))
    g.step(2e-4, 2)
    g.synchronize()
    assert g.status()[0] == 0
    return g


@pytest.mark.parametrize("cuts", ["1", "0"])
def test_native_slabs_carry_the_plan_the_torch_stepper_reads(eng, cuts, monkeypatch):
    """round 5: ONE place decides rows, neighbours, ghost rows and message order (csrc/slab_plan.hpp). Every member of a native group holds
    exactly mh_slab_plan_make's plan - four ghost rows once per step where it takes the one-launch step across its cuts, two per stage otherwise -
    and mara3_amd.slab's torch.distributed stepper (the gloo tests, bench.py's fallback) binds the same function."""
    import ctypes as C
    from mara3_amd import _lib as L
    from mara3_amd.slab import NativeSlabGroup, slab_plan
    monkeypatch.setenv("MH_SLAB_FUSED_CUTS", cuts)
    shape, world = (97, 61), 4
    for bc in ("outflow", "periodic"):
        g = NativeSlabGroup(shape, (1.0 / shape[0], 1.0 / shape[1]), 1.4, 1.5, "hllc", 2, bc, world=world, arith="fast")
        fused = g.launches_per_step() == [1] * world
        assert fused == (cuts == "1")
        for r in range(world):
            have = L.SlabPlan()
            L.check(g.lib.mh_slab_plan_of(C.c_void_p(g.handles[r]), C.byref(have)))
            want = slab_plan(shape[0], world, r, bc == "periodic", False, 2, fused)
            assert bytes(have) == bytes(want), (bc, r)
            assert have.ghost_rows == (4 if fused else 2) and have.exchanges_per_step == (1 if fused else 2)
            assert (have.row0, have.row1) == g.rows[r]
        g.close()


def last_fused_cut(lib, family):
    import ctypes as C
    out = (C.c_int32 * 4)()
    assert lib.mh_debug_last_fused_cut(family, out) == 0
    return tuple(out)


@pytest.mark.parametrize("world,n0,bc", [(2, 620, "outflow"), (3, 930, "periodic")])
def test_tapered_interior_launch_is_the_same_step(eng, world, n0, bc, monkeypatch):
    """round 5 (euler2d_fused.hip: TAPER): the interior launch of a slab with neighbours ends in SHORTER chunks, launched last (a chunk length per
    segment, segments in launch order) - by default from 24-row short chunks on, here on request on a grid whose 13 strips make 39 chunks per strip
    and residency round. The launch does cut its rows that way (mh_debug_last_fused_cut), and whatever the cut, the step is the one-domain step."""
    from mara3_amd import setups
    from mara3_amd.slab import NativeSlabGroup
    monkeypatch.setenv("MH_SLAB_FUSED_CUTS", "1")
    shape, gamma = (n0, 1400), 1.4
    dl = (1.0 / shape[0], 1.0 / shape[1])
    u0 = setups.wave_ic(shape, gamma, seed=31)
    ref = eng.EulerCartSolver(shape, dl, gamma, 1.5, "hllc", 2, bc, arith="fast")
    ref.upload(u0)
    ref.step(3e-4, 3)
    want = ref.download()
    ref.close()
    for taper, least, cut in (("3", "2", (8, 5)), ("5", "1", (8, 3)), ("0", "24", None)):
        monkeypatch.setenv("MH_FUSED_TAPER_ROWS", taper)
        monkeypatch.setenv("MH_FUSED_TAPER_MIN", least)
        grp = NativeSlabGroup(shape, dl, gamma, 1.5, "hllc", 2, bc, world=world, arith="fast")
        assert grp.launches_per_step() == [1] * world
        grp.upload(u0)
        grp.step(3e-4, 3)
        grp.synchronize()
        got = last_fused_cut(grp.lib, 1)          # the last launch issued: the last member's interior
        if cut is not None:
            assert got[:2] == cut and got[3] > got[2] > 0, got          # long chunks, then shorter ones
        else:
            assert got[0] == got[1], got
        assert grp.status() == (0, None)
        assert bits_equal(grp.download(), want), (world, bc, taper, least)
        grp.close()
