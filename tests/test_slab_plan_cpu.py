"""The device-free decisions of a slab decomposition live in ONE place (mara3_amd/csrc/slab_plan.hpp, exported as mh_slab_plan_make - host code,
callable without a GPU): the native stepper issues the plan's messages (csrc/slab.hip: exchange_rccl), and the torch.distributed stepper that
bench.py falls back to and the gloo tests drive (mara3_amd/slab.py) reads the same plan. Checked here: the cut is nd::partition_shape
(src/core_ndarray.hpp:820-836), and the messages of all ranks PAIR UP - per ordered pair of ranks the k-th send meets the k-th receive, a block of a
rank's low rows lands in its lower neighbour's high ghost rows and vice versa - for 1 .. 8 ranks, outflow and periodic axes (incl. two ranks on
a periodic axis, where both neighbours are the same rank, and a rank exchanging with itself), two ghost rows per stage or four once per step."""
import pytest


def plans(nrows, world, periodic, self_exchange=False, rk_order=2, fused_cut=False):
    from mara3_amd.slab import slab_plan
    return [slab_plan(nrows, world, r, periodic, self_exchange, rk_order, fused_cut) for r in range(world)]


@pytest.mark.parametrize("fused_cut", [False, True])
@pytest.mark.parametrize("periodic", [False, True])
@pytest.mark.parametrize("world", [1, 2, 3, 4, 5, 8])
@pytest.mark.parametrize("nrows", [4096, 4097, 1000, 97])
def test_messages_pair_up_and_rows_are_the_reference_cut(nrows, world, periodic, fused_cut):
    from mara3_amd.slab import partition_rows
    P = plans(nrows, world, periodic, fused_cut=fused_cut)
    G = 4 if fused_cut else 2
    covered = 0
    for r, p in enumerate(P):
        assert (p.row0, p.row1) == partition_rows(nrows, world, r) and p.row0 == covered      # nd::partition_shape, contiguous
        covered = p.row1
        assert p.ghost_rows == G and p.exchanges_per_step == (1 if fused_cut else 2)
        wrap = periodic and world > 1
        assert p.lo == (r - 1 if r > 0 else (world - 1 if wrap else -1)) and p.hi == (r + 1 if r < world - 1 else (0 if wrap else -1))
        assert p.edge_rows == (G if (p.lo >= 0 or p.hi >= 0) else 0)
        assert p.nmsg == 2 * ((p.lo >= 0) + (p.hi >= 0))
    assert covered == nrows
    # simulate the exchange on row LABELS: every rank's field holds (rank, local row) tags; after the exchange a ghost row must hold the tag
    # of the neighbour row that is adjacent in the global grid
    sent = {}                                            # (src, dst) -> list of blocks in issue order
    for r, p in enumerate(P):
        for k in range(p.nmsg):
            m = p.msg[k]
            if m.send:
                sent.setdefault((r, m.peer), []).append([(r, m.first_row + i) for i in range(m.rows)])
    taken = {}
    for r, p in enumerate(P):
        n0 = p.row1 - p.row0
        for k in range(p.nmsg):
            m = p.msg[k]
            if not m.send:
                i = taken.get((m.peer, r), 0)
                taken[(m.peer, r)] = i + 1
                block = sent[(m.peer, r)][i]             # messages between one pair of ranks match in issue order
                assert len(block) == m.rows
                for j, (src, src_row) in enumerate(block):
                    ghost = m.first_row + j              # local ghost row index: negative, or >= n0
                    want_global = (p.row0 + ghost) % nrows
                    assert P[src].row0 + src_row == want_global, (r, k, ghost, src, src_row)
    assert all(len(v) == taken.get(k, 0) for k, v in sent.items())


def test_one_rank_exchanging_with_itself_and_the_plan_refuses_nonsense():
    import mara3_amd
    p = plans(64, 1, True, self_exchange=True)[0]
    assert (p.lo, p.hi, p.nmsg) == (0, 0, 4) and [p.msg[k].send for k in range(4)] == [1, 1, 0, 0]
    # low rows first on the way out; they arrive in the HIGH ghosts first
    assert [(p.msg[k].first_row, p.msg[k].rows) for k in range(4)] == [(0, 2), (62, 2), (64, 2), (-2, 2)]
    assert plans(64, 1, True)[0].nmsg == 0 and plans(64, 1, False)[0].nmsg == 0
    for bad in ((3, 4, 0), (64, 2, 2), (64, 0, 0)):
        with pytest.raises(mara3_amd.MaraHipError):
            from mara3_amd.slab import slab_plan
            slab_plan(bad[0], bad[1], bad[2], False)


def test_plan_matches_the_reference_tables_of_partition_shape():
    from mara3_amd.slab import slab_plan
    assert [(slab_plan(4097, 8, r, False).row0, slab_plan(4097, 8, r, False).row1) for r in range(8)] == \
        [(0, 512), (512, 1024), (1024, 1536), (1536, 2048), (2048, 2560), (2560, 3072), (3072, 3584), (3584, 4097)]
