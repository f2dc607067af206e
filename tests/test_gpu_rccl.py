"""GPU test of the RCCL point-to-point path on ONE GPU: a periodic domain whose wrap-around is done by the
rank sending its two edge row-blocks to ITSELF (ncclSend/ncclRecv to self inside one group). This drives the
real mara3_amd.slab exchange code - message layout, group ordering when both neighbours are the same rank,
MH_BC_EXTERNAL sides, edge/interior launch split and the two-stream overlap - and must equal the kernel's own
local periodic handling bit for bit. (The 8-GPU run itself is the driver's; world_size > 1 is covered on CPU
by tests/test_slab_gloo.py.)"""
import os
import pytest
import torch

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(180)]


@pytest.fixture(scope="module")
def nccl_world1():
    import torch.distributed as dist
    assert torch.cuda.is_available()
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", str(29600 + os.getpid() % 300))
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    yield dist
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("overlap", [False, True])
@pytest.mark.parametrize("riemann,arith", [("hlle", "strict"), ("hllc", "fast")])
def test_self_exchange_equals_local_periodic(nccl_world1, overlap, riemann, arith):
    from mara3_amd import setups
    from mara3_amd.slab import SlabEulerStepper, TorchDistExchange
    from mara3_amd import _lib as L
    shape, gamma = (256, 300), 1.4
    dl = (1.0 / shape[0], 1.0 / shape[1])
    u0 = setups.wave_ic(shape, gamma, seed=3)
    ref = SlabEulerStepper(shape, dl, gamma, 1.5, riemann, 2, "periodic", arith=arith)
    ref.load_slab(u0)
    ref.step(1e-3, 6)
    st = SlabEulerStepper(shape, dl, gamma, 1.5, riemann, 2, "periodic", arith=arith, overlap=overlap,
                          exchange=TorchDistExchange(0, 1, True, self_exchange=True))
    assert st.has_neighbours and st.desc.bc_lo0 == L.BC_EXTERNAL and st.desc.bc_hi0 == L.BC_EXTERNAL
    st.load_slab(u0)
    st.step(1e-3, 6)
    torch.cuda.synchronize()
    assert torch.equal(st.slab(), ref.slab())
    assert st.status() == 0


@pytest.mark.parametrize("graph", [False, True])
@pytest.mark.parametrize("riemann,arith", [("hlle", "strict"), ("hllc", "fast")])
def test_native_slab_self_exchange_equals_local_periodic(graph, riemann, arith):
    """The native (C++/RCCL) slab stepper of libmara_hip.so on one GPU: periodic wrap through ncclSend/ncclRecv to self against the
    kernel's own local periodic handling. Bit-identical. A step with neighbours is always issued eagerly (RCCL point-to-point inside a
    stream capture crashes this stack, slab.hip: mh_slab_step), so `graph` only checks that asking for graph replay is harmless here;
    the replayed graph itself is covered by test_native_slab_without_neighbours_matches_context_api."""
    import numpy as np
    from mara3_amd import setups
    from mara3_amd.slab import NativeSlabStepper, native_comm_id
    from mara3_amd.engine import EulerCartSolver
    shape, gamma = (256, 300), 1.4
    dl = (1.0 / shape[0], 1.0 / shape[1])
    u0 = setups.wave_ic(shape, gamma, seed=5)
    ref = EulerCartSolver(shape, dl, gamma, 1.5, riemann, 2, "periodic", arith=arith)
    ref.upload(u0)
    ref.step(1e-3, 6)
    st = NativeSlabStepper(shape, dl, gamma, 1.5, riemann, 2, "periodic", rank=0, world=1, arith=arith,
                           comm_id=native_comm_id(0, 1), self_exchange=True)
    assert (st.row0, st.row1) == (0, shape[0])
    st.load_slab(u0)
    st.step(1e-3, 6, graph=graph)
    st.synchronize()
    assert np.array_equal(st.slab_host().view(np.uint64), ref.download().view(np.uint64))
    assert st.status() == 0
    st.close()


def test_native_slab_deferred_connect_as_bench_py_does_it():
    """bench.py --gpus N creates every rank's slab WITHOUT a communicator, lets the ranks agree that creation succeeded, and only then
    enters the collective ncclCommInitRank (mh_slab_connect). The same sequence here with the exchange going to self; stepping before
    the connection is refused, not crashed."""
    import numpy as np
    import mara3_amd
    from mara3_amd import setups
    from mara3_amd.slab import NativeSlabStepper, native_comm_id
    from mara3_amd.engine import EulerCartSolver
    shape, gamma = (128, 200), 1.4
    dl = (1.0 / shape[0], 1.0 / shape[1])
    u0 = setups.wave_ic(shape, gamma, seed=7)
    ref = EulerCartSolver(shape, dl, gamma, 1.5, "hllc", 2, "periodic", arith="fast")
    ref.upload(u0); ref.step(1e-3, 4)
    st = NativeSlabStepper(shape, dl, gamma, 1.5, "hllc", 2, "periodic", rank=0, world=1, arith="fast", comm_id=None, self_exchange=True)
    with pytest.raises(mara3_amd.MaraHipError, match="connect"):
        st.load_slab(u0)
    st.connect(native_comm_id(0, 1))
    st.load_slab(u0); st.step(1e-3, 4); st.synchronize()
    assert np.array_equal(st.slab_host().view(np.uint64), ref.download().view(np.uint64))
    st.close()


def test_native_slab_without_neighbours_matches_context_api():
    import numpy as np
    from mara3_amd import setups
    from mara3_amd.slab import NativeSlabStepper
    from mara3_amd.engine import EulerCartSolver
    shape, gamma = (130, 200), 5.0 / 3
    dl = (1.0 / shape[0], 1.0 / shape[1])
    u0 = setups.blast_ic(shape, gamma, radius=0.3)
    ref = EulerCartSolver(shape, dl, gamma, 1.5, "hllc", 2, "outflow")
    ref.upload(u0); ref.step(1e-3, 5)
    st = NativeSlabStepper(shape, dl, gamma, 1.5, "hllc", 2, "outflow")
    st.load_slab(u0); st.step(1e-3, 5, graph=True); st.synchronize()
    assert np.array_equal(st.slab_host().view(np.uint64), ref.download().view(np.uint64))


def test_native_slab_3d_self_exchange_equals_local_periodic():
    """3-D (config-5 scheme) in the native slab stepper: axis-0 planes exchanged through RCCL to self."""
    import numpy as np
    from mara3_amd import setups
    from mara3_amd.slab import NativeSlabStepper, native_comm_id
    from mara3_amd.engine import EulerCartSolver
    shape, gamma = (40, 20, 70), 1.4
    dl = (1.0 / 40, 1.0 / 20, 1.0 / 70)
    u0 = setups.wave_ic(shape, gamma, seed=6)
    ref = EulerCartSolver(shape, dl, gamma, 1.5, "hlle", 2, "periodic")
    ref.upload(u0); ref.step(2e-3, 4)
    st = NativeSlabStepper(shape, dl, gamma, 1.5, "hlle", 2, "periodic", comm_id=native_comm_id(0, 1), self_exchange=True)
    st.load_slab(u0); st.step(2e-3, 4); st.synchronize()
    assert np.array_equal(st.slab_host().view(np.uint64), ref.download().view(np.uint64))


@pytest.mark.parametrize("rk_order,shape", [(1, (256, 300)), (2, (256, 300)), (2, (18, 130)), (1, (36, 70))])
def test_native_slab_staggered_edges_survive_interruptions(rk_order, shape):
    """The native stepper's staggered edge strips (period of 4 stages, mara3_amd/csrc/slab.hip): step counts that end in the middle
    of a period, a download in between, RK1 as well as RK2, and slabs too thin for staggering (< 20 rows) - always bit-identical to
    the kernel's own periodic handling."""
    import numpy as np
    from mara3_amd import setups
    from mara3_amd.slab import NativeSlabStepper, native_comm_id
    from mara3_amd.engine import EulerCartSolver
    gamma = 1.4
    dl = (1.0 / shape[0], 1.0 / shape[1])
    u0 = setups.wave_ic(shape, gamma, seed=11)
    ref = EulerCartSolver(shape, dl, gamma, 1.5, "hllc", rk_order, "periodic", arith="fast")
    ref.upload(u0)
    st = NativeSlabStepper(shape, dl, gamma, 1.5, "hllc", rk_order, "periodic", rank=0, world=1, arith="fast",
                           comm_id=native_comm_id(0, 1), self_exchange=True)
    st.load_slab(u0)
    done = 0
    for nsteps in (1, 2, 3, 1, 5):
        st.step(5e-4, nsteps)
        st.synchronize()
        done += nsteps
        ref.step(5e-4, nsteps)
        assert np.array_equal(st.slab_host().view(np.uint64), ref.download().view(np.uint64)), (done, nsteps)
    assert st.status() == 0
    st.close()


@pytest.mark.parametrize("on_launch", [1, 0])
@pytest.mark.parametrize("delay", [1, 2, 3])
@pytest.mark.parametrize("stagger", [0, 4])
def test_native_slab_dependencies_hold_under_shifted_timing(delay, stagger, on_launch, monkeypatch):
    """The two-chain schedule of the native stepper (edge -> exchange on one stream, interior on the other) with a ~150 us sleeping
    wave queued in front of the edge launches (1), the interior launches (2) or both (3): whatever the relative timing of the chains,
    the events alone must order them. With and without staggered edges."""
    import numpy as np
    from mara3_amd import setups
    from mara3_amd.slab import NativeSlabStepper, native_comm_id
    from mara3_amd.engine import EulerCartSolver
    monkeypatch.setenv("MH_SLAB_TEST_DELAY", str(delay))
    monkeypatch.setenv("MH_SLAB_STAGGER", str(stagger))
    monkeypatch.setenv("MH_SLAB_EVENT_ON_LAUNCH", str(on_launch))      # events carried by the launches, or recorded behind them
    shape, gamma = (192, 260), 1.4
    dl = (1.0 / shape[0], 1.0 / shape[1])
    u0 = setups.wave_ic(shape, gamma, seed=12)
    ref = EulerCartSolver(shape, dl, gamma, 1.5, "hllc", 2, "periodic", arith="fast")
    ref.upload(u0)
    ref.step(5e-4, 7)
    st = NativeSlabStepper(shape, dl, gamma, 1.5, "hllc", 2, "periodic", rank=0, world=1, arith="fast",
                           comm_id=native_comm_id(0, 1), self_exchange=True)
    st.load_slab(u0)
    st.step(5e-4, 7)
    st.synchronize()
    assert np.array_equal(st.slab_host().view(np.uint64), ref.download().view(np.uint64))
    st.close()


@pytest.mark.parametrize("shape", [(40, 24, 70), (16, 16, 130)])
def test_native_block_rccl_exchange_to_self_on_all_three_axes(shape):
    """The block stepper's RCCL path (mara3_amd/csrc/block3d.hip: six ncclSend / ncclRecv in one group, packed axis-1 / axis-2 faces,
    shell / interior streams) on one GPU: one block of a periodic domain whose wrap-around goes through the exchange, to itself, on ALL
    THREE axes - the message order that pairs a rank with the same neighbour on both sides of an axis. Bit-identical to the kernel's own
    periodic handling."""
    import numpy as np
    from mara3_amd import setups
    from mara3_amd.slab import native_comm_id
    from mara3_amd.block import NativeBlock
    from mara3_amd.engine import EulerCartSolver
    gamma = 1.4
    dl = tuple(1.0 / n for n in shape)
    u0 = setups.wave_ic(shape, gamma, seed=13)
    ref = EulerCartSolver(shape, dl, gamma, 1.5, "hllc", 2, "periodic", arith="fast")
    ref.upload(u0); ref.step(1e-3, 5)
    blk = NativeBlock(shape, dl, gamma, 1.5, "hllc", 2, "periodic", rank=0, world=1, comm_id=native_comm_id(0, 1), arith="fast", self_exchange=True)
    assert blk.neighbours == (0, 0, 0, 0, 0, 0)
    blk.upload(u0); blk.step(1e-3, 5); blk.synchronize()
    assert blk.status()[0] == 0
    assert np.array_equal(blk.download().view(np.uint64), ref.download().view(np.uint64))
    blk.close()


@pytest.mark.parametrize("edge_rows", [None, -1, 5])
def test_binary_band_rccl_halo_and_allreduce_to_self(edge_rows):
    """`binary` over bands, RCCL backend, on one GPU: a single band whose periodic ghost rows travel through ncclSend / ncclRecv to self and
    whose totals, wavespeed and status go through the three ncclAllReduce calls (over one rank: the identity). edge_rows None: one launch per
    stage, the exchange behind it - field AND scalars bit-identical to the whole-mesh solver, CFL time step included. Otherwise (round 3) the
    band's edge rows are stepped first and their exchange runs on a second stream beside the interior launch: the field is still
    bit-identical, the scalars agree to the order in which the waves' partial sums are added (edge waves first)."""
    import numpy as np
    from mara3_amd import binary
    from mara3_amd.slab import native_comm_id
    cfg = binary.config(depth=2, block_size=16)
    one = binary.BinarySolver(cfg)
    band = binary.BinaryBand(cfg, 0, 1, native_comm_id(0, 1), self_exchange=True, edge_rows=edge_rows)
    assert (band.row0, band.row1) == (0, 64)
    for nsteps in (1, 3):
        assert one.next(nsteps) == 0 and band.next(nsteps) == 0
        assert band.last_dt == one.last_dt
        assert np.array_equal(band.solution().view(np.uint64), one.solution().view(np.uint64))
        a, b = binary.state_as_dict(band.state()), binary.state_as_dict(one.state())
        if edge_rows is None:
            assert a == b
        else:
            assert a["time"] == b["time"] and a["iteration"] == b["iteration"]
            for k in ("mass_accreted_on", "integrated_torque_on", "work_done_on", "orbital_elements"):
                assert np.allclose(a[k], b[k], rtol=1e-9, atol=1e-13 * max(1.0, float(np.abs(b[k]).max()))), k
    one.close(); band.close()


def test_binary_band_edge_rows_beside_the_interior_at_a_size_where_they_overlap():
    """1024^2 (the launches last long enough to run side by side): one RCCL band whose edge launch and exchange run on the exchange stream
    beside the interior launch - the field is the single-domain field bit for bit after every call, from a set solution too (whose ghost
    exchange must be complete before the first stage reads it)."""
    import numpy as np
    from mara3_amd import binary
    from mara3_amd.slab import native_comm_id
    cfg = binary.config(depth=4, block_size=64, fixed_dt=1, plm_theta=1.8)
    one = binary.BinarySolver(cfg, arith="fast")
    band = binary.BinaryBand(cfg, 0, 1, native_comm_id(0, 1), arith="fast", self_exchange=True, edge_rows=-1)
    for nsteps in (1, 4):
        assert one.next(nsteps) == 0 and band.next(nsteps) == 0
        assert np.array_equal(band.solution().view(np.uint64), one.solution().view(np.uint64))
    u, st = one.solution(), one.state()
    u[3, :, 0] *= 1.001
    one.set_solution(u, st); band.set_solution(u, st)
    assert one.next(3) == 0 and band.next(3) == 0
    assert np.array_equal(band.solution().view(np.uint64), one.solution().view(np.uint64))
    one.close(); band.close()


def test_one_communicator_per_process_is_lent_to_every_stepper():
    """bench.py at N > 1 enters ncclCommInitRank ONCE per process (mh_comm_create) and lends the communicator to the stepper of every leg
    (mh_slab_use_comm / mh_block_use_comm / mh_binary_band_use_comm). Here with the exchange going to self: two slabs one after the other,
    a block and a band all run on the same communicator, each bit-identical to the single-domain solver, and the communicator outlives
    them; a stepper of another rank / world is refused."""
    import numpy as np
    import mara3_amd
    from mara3_amd import setups, binary
    from mara3_amd.slab import NativeSlabStepper, NativeComm, native_comm_id
    from mara3_amd.block import NativeBlock
    from mara3_amd.engine import EulerCartSolver
    comm = NativeComm(native_comm_id(0, 1), 0, 1, device=0)
    shape, gamma = (128, 200), 1.4
    dl = (1.0 / shape[0], 1.0 / shape[1])
    u0 = setups.wave_ic(shape, gamma, seed=11)
    for riemann, arith in (("hllc", "fast"), ("hlle", "strict")):
        ref = EulerCartSolver(shape, dl, gamma, 1.5, riemann, 2, "periodic", arith=arith)
        ref.upload(u0); ref.step(1e-3, 4)
        st = NativeSlabStepper(shape, dl, gamma, 1.5, riemann, 2, "periodic", rank=0, world=1, arith=arith, comm_id=None, self_exchange=True)
        st.use_comm(comm)
        with pytest.raises(mara3_amd.MaraHipError, match="already"):
            st.use_comm(comm)
        st.load_slab(u0); st.step(1e-3, 4); st.synchronize()
        assert np.array_equal(st.slab_host().view(np.uint64), ref.download().view(np.uint64))
        st.close(); ref.close()
    shape3 = (24, 16, 70)
    dl3 = tuple(1.0 / n for n in shape3)
    u3 = setups.wave_ic(shape3, gamma, seed=13)
    ref = EulerCartSolver(shape3, dl3, gamma, 1.5, "hllc", 2, "periodic", arith="fast")
    ref.upload(u3); ref.step(1e-3, 3)
    blk = NativeBlock(shape3, dl3, gamma, 1.5, "hllc", 2, "periodic", rank=0, world=1, comm_id=None, arith="fast", self_exchange=True)
    blk.use_comm(comm)
    blk.upload(u3); blk.step(1e-3, 3); blk.synchronize()
    assert np.array_equal(blk.download().view(np.uint64), ref.download().view(np.uint64))
    blk.close(); ref.close()
    cfg = binary.config(depth=2, block_size=16)
    one = binary.BinarySolver(cfg)
    band = binary.BinaryBand(cfg, 0, 1, None, self_exchange=True, defer=True)
    band.attach(comm)
    assert one.next(3) == 0 and band.next(3) == 0
    assert np.array_equal(band.solution().view(np.uint64), one.solution().view(np.uint64))
    assert binary.state_as_dict(band.state()) == binary.state_as_dict(one.state())
    one.close(); band.close()
    comm.close()


def test_binary_bands_report_the_failing_cell_of_the_whole_mesh():
    """The status of a failed attempt is the OR of every band's bits with the first failing cell as an index into the WHOLE-mesh host
    array (mh_binary_last_failure) - the same from the single-domain solver, through the RCCL gather of the bands' status words (here
    over one rank) and through the loopback merge of four bands; the collective safe-mode retry then lands on the same bits."""
    import numpy as np
    from mara3_amd import binary
    from mara3_amd import _lib as L
    from mara3_amd.slab import native_comm_id
    cfg = binary.config(depth=2, block_size=16, fixed_dt=1)
    one = binary.BinarySolver(cfg)
    band = binary.BinaryBand(cfg, 0, 1, native_comm_id(0, 1), self_exchange=True)
    grp = binary.BinaryBandGroup(cfg, world=4)
    assert one.last_failure() == (0, None)
    base = one.solution()
    s = one.state()
    for boost in (10.0, 30.0, 100.0, 300.0):      # a hole next to a fast stream (as test_gpu_binary.py::test_safe_mode_retry), in band 3 of 4
        u = base.copy()
        u[50, 20, 0] *= 1e-2
        u[50, 21, 1:] *= boost
        u[51, 20, 1:] *= boost
        one.set_solution(u, s)
        try:
            if one.next(1) == 1:
                break
        except Exception:
            pass
    else:
        pytest.fail("no test state made the ordinary step fail")
    for solver in (one, band, grp):
        solver.set_solution(u, s)
    safe = [solver.next(1) for solver in (one, band, grp)]
    assert safe == [1, 1, 1]
    fails = [solver.last_failure() for solver in (one, band, grp)]
    assert fails[0][0] == L.STATUS_NEG_DENSITY and fails[0][1] is not None and fails[0][1] // 64 in range(48, 53)
    assert fails[1] == fails[0] and fails[2] == fails[0]
    assert np.array_equal(band.solution().view(np.uint64), one.solution().view(np.uint64))
    assert np.array_equal(grp.solution().view(np.uint64), one.solution().view(np.uint64))
    one.set_solution(base, s)                   # a clean call clears the record
    assert one.next(1) == 0 and one.last_failure() == (0, None)
    one.close(); band.close(); grp.close()


@pytest.mark.parametrize("cuts", [None, "0"])
def test_native_slab_at_the_eight_gpu_cut_size_through_rccl_to_self(cuts, monkeypatch):
    """One rank's slab of the 8-GPU cut (512 x 4096, FAST HLLC RK2) with both sides cut and its ghost rows travelling through ncclSend /
    ncclRecv to itself: the library's own schedule at this size (the fused step across the cuts: four rows, one exchange per step) and the
    two-launch schedule (MH_SLAB_FUSED_CUTS=0), 12 steps, bit-identical to the kernel's own periodic handling."""
    import numpy as np
    from mara3_amd import setups
    from mara3_amd.slab import NativeSlabStepper, native_comm_id
    if cuts is not None:
        monkeypatch.setenv("MH_SLAB_FUSED_CUTS", cuts)
    shape, gamma = (512, 4096), 5.0 / 3
    dl = (1.0 / 4096, 1.0 / 4096)
    u0 = setups.wave_ic(shape, gamma, seed=77)
    ref = NativeSlabStepper(shape, dl, gamma, 1.5, "hllc", 2, "periodic", arith="fast")
    ref.load_slab(u0); ref.step(2e-5, 12); ref.synchronize()
    st = NativeSlabStepper(shape, dl, gamma, 1.5, "hllc", 2, "periodic", rank=0, world=1, arith="fast", comm_id=native_comm_id(0, 1), self_exchange=True)
    st.load_slab(u0); st.step(2e-5, 5); st.step(2e-5, 7); st.synchronize()
    assert st.status() == 0
    assert np.array_equal(st.slab_host().view(np.uint64), ref.slab_host().view(np.uint64))
    st.profile(True); st.step(2e-5, 3); st.synchronize()
    (ms1, ms2), (n1, n2), rows = st.profile_read()
    assert (n1, n2, rows) == ((0, 3, 504) if cuts is None else (3, 3, rows))          # fused: one interior launch per step over rows 4 .. n0 - 4
    st.close(); ref.close()
