"""Multi-GPU decomposition of `binary` on a GRADED block tree (mara3_amd/csrc/binary_api.hip: tree_stage_distributed),
executed on ONE GPU: the leaves ordered along the Hilbert curve through them, member r of N running the block kernels on the r-th run of
that order, the members' results gathered behind each kernel - as 2, 3, 4 and 7 member objects of one process (LOOPBACK), and as one RCCL
member whose status words travel through ncclAllGather. The reference hands the leaves to its thread pool in traversal order
(tree.map(fn, pool), src/core_tree.hpp:615-625).

Every member forms the totals over ALL blocks in the caller's block order, so nothing depends on the partition: field, accumulators,
orbital elements and time steps are the single-domain solver's BIT FOR BIT - with a live binary too (where bands of the uniform mesh only
agree to the order of summation)."""
import numpy as np
import pytest
from conftest import bits_equal

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(600)]


@pytest.fixture(scope="module")
def binary():
    import mara3_amd
    from mara3_amd import binary
    assert mara3_amd.load_library().mh_device_count() >= 1
    return binary


def state_bits(binary, s):
    d = binary.state_as_dict(s)
    flat = [d["time"], float(d["iteration"]), d["mass_ejected"], d["angular_momentum_ejected"]]
    for k in ("mass_accreted_on", "angular_momentum_accreted_on", "integrated_torque_on", "work_done_on", "orbital_elements_acc", "orbital_elements_grav", "orbital_elements"):
        flat += list(d[k])
    return np.array(flat)


@pytest.mark.parametrize("world", [2, 3, 4, 7])
@pytest.mark.parametrize("overrides", [
    dict(depth=3, block_size=8),                                                         # the graded default focus, CFL time step
    dict(depth=4, block_size=8, focus_factor=2.0, rk_order=1),
    dict(depth=3, block_size=8, conserve_linear_p=0),                                    # advance_q
    dict(depth=3, block_size=12, nu=1e-3, alpha=0.0, fixed_dt=1),
    dict(depth=3, block_size=8, begin_live_binary=0.0, mass_ratio=0.7),                  # live binary: the second stage depends on the first stage's totals
])
@pytest.mark.parametrize("arith", ["strict", "fast"])
def test_distributed_tree_equals_single_domain_bit_for_bit(binary, overrides, world, arith):
    cfg = binary.config(**overrides)
    one = binary.BinaryTreeSolver(cfg, arith=arith)
    grp = binary.BinaryTreeGroup(cfg, world=world, arith=arith)
    nb = len(one.blocks)
    owned = np.concatenate(grp.owned)
    assert sorted(owned.tolist()) == list(range(nb)) and bits_equal(owned, binary.tree_curve_order(one.blocks))
    assert max(len(o) for o in grp.owned) - min(len(o) for o in grp.owned) <= 1
    for nsteps in (1, 3):
        assert one.next(nsteps) == grp.next(nsteps)
        assert grp.last_dt == one.last_dt
        want = one.solution()
        assert bits_equal(grp.solution(), want), (overrides, world, nsteps)
        assert bits_equal(state_bits(binary, grp.state()), state_bits(binary, one.state()))
    for r in range(world):          # every member holds the whole tree
        assert bits_equal(grp.member_solution(r), want)
    one.close(); grp.close()


def test_distributed_tree_default_configuration_20_steps(binary):
    """the sub-program's default mesh (depth 4, 24-zone blocks, 64 leaves) on 4 members"""
    cfg = binary.config()
    one = binary.BinaryTreeSolver(cfg)
    grp = binary.BinaryTreeGroup(cfg, world=4)
    assert one.next(20) == grp.next(20)
    assert bits_equal(grp.solution(), one.solution())
    assert bits_equal(state_bits(binary, grp.state()), state_bits(binary, one.state()))
    one.close(); grp.close()


def test_distributed_tree_safe_mode_retry_and_failing_cell(binary):
    """a nearly empty cell in ONE member's block: every member retries (dt * 0.1, theta = 0) and the failing cell is reported as the flat
    index into the caller's array - the same index the single-domain solver reports, although the members store the blocks in curve order"""
    cfg = binary.config(depth=3, block_size=8, fixed_dt=1)
    one = binary.BinaryTreeSolver(cfg)
    grp = binary.BinaryTreeGroup(cfg, world=3)
    u0, s0 = one.solution(), one.state()
    blk = int(grp.owned[2][1])          # a block of the last member
    for boost in (10.0, 100.0, 1e3, 1e4, 1e5):          # a hole next to a fast stream, fast enough for the ordinary step to overshoot below zero
        u = u0.copy()
        u[blk, 3, 4, 0] *= 1e-2
        u[blk, 3, 5, 1:] *= boost
        u[blk, 4, 4, 1:] *= boost
        one.set_solution(u, s0)
        sa = one.next(1)
        if sa == 1:
            break
    assert sa == 1, "the test state must fail the ordinary step"
    grp.set_solution(u, s0)
    assert grp.next(1) == 1
    assert grp.last_failure() == one.last_failure() and one.last_failure()[1] is not None
    assert one.last_failure()[1] // (8 * 8) == blk
    assert bits_equal(grp.solution(), one.solution())
    one.close(); grp.close()


def test_distributed_tree_member_over_rccl(binary):
    """world 1 through the RCCL form: the member's status words go through ncclAllGather (over one rank), the rest is the distributed
    stage with a run that is the whole tree - stored in curve order, totals added in the caller's order: bit-identical to BinaryTreeSolver"""
    from mara3_amd.slab import native_comm_id, NativeComm
    cfg = binary.config(depth=3, block_size=8)
    one = binary.BinaryTreeSolver(cfg)
    for how in ("id", "comm"):
        comm = NativeComm(native_comm_id(0, 1), 0, 1) if how == "comm" else None
        band = binary.BinaryTreeBand(cfg, 0, 1, native_comm_id(0, 1) if how == "id" else None, comm=comm)
        one2 = binary.BinaryTreeSolver(cfg)
        assert one2.next(3) == band.next(3)
        assert bits_equal(band.solution(), one2.solution())
        assert bits_equal(state_bits(binary, band.state()), state_bits(binary, one2.state()))
        band.close(); one2.close()
        if comm is not None:
            comm.close()
    one.close()


def test_boundary_refuses_what_it_cannot_run(binary):
    """error behaviour of the round-3 entry points: a message and a negative code, never a launch"""
    import ctypes as C
    from mara3_amd import _lib as L
    cfg = binary.config(depth=3, block_size=8)
    t = binary._TreeSetup(cfg)
    lib = t.lib
    h = C.c_void_p()
    # a rank outside the world, a world of zero
    for rank, world in ((3, 3), (-1, 2), (0, 0)):
        assert lib.mh_binary_tree_band_create(C.byref(h), 0, C.byref(t.desc), C.byref(t.run), *t.pointers(), rank, world, None) < 0
        assert not h.value
    # overlapping leaves cannot be ordered along the curve
    bad = np.array([[1, 0, 0], [1, 0, 1], [1, 1, 0], [1, 1, 1], [2, 1, 1]], dtype=np.int32)
    order = np.empty(5, dtype=np.int32)
    assert lib.mh_binary_tree_curve_order(bad.ctypes.data_as(C.c_void_p), 5, order.ctypes.data_as(C.c_void_p)) < 0
    assert lib.mh_binary_tree_curve_order(None, 5, order.ctypes.data_as(C.c_void_p)) < 0
    # edge rows are a property of bands of the uniform mesh
    one = binary.BinaryTreeSolver(cfg)
    assert lib.mh_binary_band_set_edge_rows(one.handle, 2) < 0
    n = C.c_int(-1)
    assert lib.mh_binary_tree_owned_blocks(one.handle, None, C.byref(n)) == 0 and n.value == len(one.blocks)          # one domain: every block
    uni = binary.BinarySolver(binary.config(depth=2, block_size=16))
    assert lib.mh_binary_tree_owned_blocks(uni.handle, None, C.byref(n)) < 0
    assert lib.mh_binary_band_set_edge_rows(uni.handle, 2) < 0          # not a band with neighbours
    # the diagnostics of a distributed tree are refused (they would come out in curve order)
    grp = binary.BinaryTreeGroup(cfg, world=2)
    dm, dl = C.c_double(), C.c_double()
    assert lib.mh_binary_disk_totals(C.c_void_p(grp.handles[0]), C.byref(dm), C.byref(dl)) < 0
    one.close(); uni.close(); grp.close()
