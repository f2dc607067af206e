"""The 3-axis block decomposition of BASELINE config 5 (1024^3 as (2,2,2) blocks on 8 GPUs) in the native block stepper
(mara3_amd/csrc/block3d.hip), executed on ONE GPU through its LOOPBACK backend: the blocks of mara::propose_block_decomposition<3>
with the extents of create_access_pattern_array (src/app_parallel.hpp:119-131, :148-179) as objects of one process; per stage and cut
side one message (axis 0: two planes straight out of the field; axes 1, 2: packed rows / columns), boundary shell and interior as two
launches. The union of the blocks must be BIT-IDENTICAL to the single-domain run and to the reference's vectors."""
import numpy as np
import pytest
from conftest import golden, bits_equal

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(600)]


@pytest.fixture(scope="module")
def eng():
    import mara3_amd
    from mara3_amd import engine
    assert mara3_amd.load_library().mh_device_count() >= 1
    return engine


@pytest.mark.parametrize("world", [2, 4, 8, 12])
@pytest.mark.parametrize("case", ["euler3d_blast24_plm15_rk2", "euler3d_wave20x12x16_plm15_rk2_periodic"])
def test_blocks_equal_reference_golden(eng, case, world):
    """(1,1,2), (1,2,2), (2,2,2) and (2,2,3) blocks against the REFERENCE's 3-D vectors (strict, HLLE): outflow (blocks with physical
    and cut sides mixed) and periodic (two blocks on an axis are each other's neighbour on both sides); 20 x 12 x 16 over (2,2,3)
    gives axis-2 extents 5, 5, 6."""
    from mara3_amd.block import NativeBlockGroup
    g = golden(case)
    u0 = g["u0"]
    bc = "periodic" if int(g["bc"]) == 1 else "outflow"
    grp = NativeBlockGroup(u0.shape[:3], g["dl"], float(g["gamma"]), float(g["theta"]), "hlle", int(g["rk"]), bc, world=world)
    grp.upload(u0)
    done = 0
    for ns in sorted(int(n) for n in g["nsteps"]):
        grp.step(float(g["dt"]), ns - done)
        done = ns
        grp.synchronize()
        got = grp.download()
        assert bits_equal(got, g["u_%d" % ns]), (case, world, ns, np.abs(got - g["u_%d" % ns]).max())
    assert grp.status() == (0, None)
    grp.close()


@pytest.mark.parametrize("bc", ["outflow", "periodic"])
@pytest.mark.parametrize("shape", [(48, 48, 48), (64, 40, 56), (20, 150, 18)])
@pytest.mark.parametrize("riemann,arith,rk", [("hlle", "strict", 2), ("hllc", "fast", 2), ("hllc", "strict", 1)])
def test_eight_blocks_equal_single_domain(eng, shape, bc, riemann, arith, rk):
    """(2,2,2) blocks at 48^3 and 64 x 40 x 56 (and a slab-like 20 x 150 x 18, where the shell is the whole block on two axes): several
    steps, a download in between, bit-identical to the one-domain run of the context API."""
    from mara3_amd import setups
    from mara3_amd.block import NativeBlockGroup
    gamma = 1.4
    dl = tuple(1.0 / n for n in shape)
    u0 = setups.wave_ic(shape, gamma, seed=41)
    ref = eng.EulerCartSolver(shape, dl, gamma, 1.5, riemann, rk, bc, arith=arith)
    ref.upload(u0)
    grp = NativeBlockGroup(shape, dl, gamma, 1.5, riemann, rk, bc, world=8, arith=arith)
    assert grp.members[0].blocks == (2, 2, 2)
    grp.upload(u0)
    for nsteps in (1, 2, 3):
        ref.step(4e-4, nsteps)
        grp.step(4e-4, nsteps)
        grp.synchronize()
        assert bits_equal(grp.download(), ref.download()), (shape, bc, nsteps)
    assert grp.status() == (0, None)
    grp.close()


def test_block_extents_neighbours_and_message_sizes(eng):
    """Integer work: the (2,2,2) cut of 64 x 40 x 56, neighbour ranks (row-major block order) and the documented message sizes."""
    from mara3_amd.block import NativeBlockGroup
    shape = (64, 40, 56)
    grp = NativeBlockGroup(shape, (1 / 64, 1 / 40, 1 / 56), 1.4, world=8)
    seen = np.zeros(shape, dtype=int)
    for r, m in enumerate(grp.members):
        assert m.blocks == (2, 2, 2) and m.coords == (r // 4, (r // 2) % 2, r % 2)
        assert m.count == (32, 20, 28) and m.start == tuple(c * k for c, k in zip(m.coords, m.count))
        seen[m.slices()] += 1
        lo = lambda a: -1 if m.coords[a] == 0 else r - (4, 2, 1)[a]
        hi = lambda a: -1 if m.coords[a] == 1 else r + (4, 2, 1)[a]
        assert m.neighbours == (lo(0), hi(0), lo(1), hi(1), lo(2), hi(2))
        # axis 0: two planes incl. the stored transverse ghost layers; axes 1 / 2: [n0][5][2][n2] / [n0][5][n1][2]
        assert m.message_doubles == (2 * 5 * (20 + 4) * (28 + 4), 32 * 5 * 2 * 28, 32 * 5 * 20 * 2)
    assert (seen == 1).all()
    grp.close()


def test_blocks_report_first_failing_cell_in_global_index(eng):
    from mara3_amd import setups
    from mara3_amd.block import NativeBlockGroup
    shape, gamma = (32, 24, 40), 1.4
    u0 = setups.wave_ic(shape, gamma, seed=42)
    u0[20, 15, 33, 0] = np.nan
    grp = NativeBlockGroup(shape, tuple(1.0 / n for n in shape), gamma, 1.5, "hllc", 2, "outflow", world=8, arith="fast")
    grp.upload(u0)
    grp.step(1e-4, 1)
    bits, first = grp.status()
    flat = (20 * 24 + 15) * 40 + 33
    assert bits != 0 and flat - 2 * 24 * 40 <= first <= flat
    grp.close()


@pytest.mark.timeout(900)
def test_config5_per_rank_share_as_eight_blocks_of_256(eng):
    """256^3 as (2,2,2) blocks of 128^3 - a real shell / interior split (interior tiles and strips exist) - against the one-domain run:
    FAST PLM + HLLE RK2 as bench_configs.py --config c5 runs it, 3 steps, bit-identical."""
    from mara3_amd import setups
    from mara3_amd.block import NativeBlockGroup
    n, gamma = 256, 5.0 / 3
    dl = (1.0 / n,) * 3
    u0 = setups.blast_ic((n, n, n), gamma)
    dt = setups.baseline_dt(n)
    ref = eng.EulerCartSolver((n, n, n), dl, gamma, 1.5, "hlle", 2, "outflow", arith="fast")
    ref.upload(u0)
    ref.step(dt, 3)
    want = ref.download()
    ref.close()
    grp = NativeBlockGroup((n, n, n), dl, gamma, 1.5, "hlle", 2, "outflow", world=8, arith="fast")
    grp.upload(u0)
    grp.step(dt, 3)
    grp.synchronize()
    assert grp.status() == (0, None)
    assert bits_equal(grp.download(), want)
    grp.close()


@pytest.mark.timeout(1200)
def test_config5_as_configured_per_gpu_share_512_as_eight_blocks(eng):
    """The (2,2,2) decomposition at HALF the configured linear size - 512^3 as 8 blocks of 256^3, each with a real shell / interior split and
    21 MB-class faces scaled by 1/4 - on one GPU, 2 steps, FAST PLM + HLLE RK2: bit-identical to the undivided 512^3 run (one rank's share
    of the 1024^3 case), status clean, mass conserved to rounding. (1024^3 itself is 43 GB per field copy: it needs the 8 GPUs.)"""
    from mara3_amd import setups
    from mara3_amd.block import NativeBlockGroup
    n, gamma = 512, 5.0 / 3
    dl = (1.0 / n,) * 3
    u0 = setups.blast_ic((n, n, n), gamma)
    dt = setups.baseline_dt(n)
    ref = eng.EulerCartSolver((n, n, n), dl, gamma, 1.5, "hlle", 2, "outflow", arith="fast")
    ref.upload(u0)
    ref.step(dt, 2)
    want = ref.download()
    assert ref.status() == 0
    ref.close()
    grp = NativeBlockGroup((n, n, n), dl, gamma, 1.5, "hlle", 2, "outflow", world=8, arith="fast")
    assert all(m.count == (256, 256, 256) for m in grp.members)
    assert grp.members[0].message_doubles == (2 * 5 * 260 * 260, 256 * 5 * 2 * 256, 256 * 5 * 256 * 2)
    grp.upload(u0)
    mass0 = float(u0[..., 0].sum())
    del u0
    grp.step(dt, 2)
    grp.synchronize()
    assert grp.status() == (0, None)
    got = grp.download()
    grp.close()
    assert bits_equal(got, want)
    assert abs(float(got[..., 0].sum()) - mass0) <= 1e-12 * mass0


def test_blocks_long_run_stays_bit_identical(eng):
    """120 steps of (2,2,3) blocks of a periodic 40 x 36 x 66 grid (uneven axis-2 extents 22, 22, 22 -> strips of one partial tile): the
    shell / interior chains and the six-message exchange over many stages."""
    from mara3_amd import setups
    from mara3_amd.block import NativeBlockGroup
    shape, gamma = (40, 36, 66), 1.4
    dl = tuple(1.0 / n for n in shape)
    u0 = setups.wave_ic(shape, gamma, seed=52)
    dt = 0.2 * min(dl) / 2.0
    ref = eng.EulerCartSolver(shape, dl, gamma, 1.5, "hlle", 2, "periodic", arith="strict")
    ref.upload(u0)
    grp = NativeBlockGroup(shape, dl, gamma, 1.5, "hlle", 2, "periodic", world=12, arith="strict")
    assert grp.members[0].blocks == (2, 2, 3)
    grp.upload(u0)
    for _ in range(4):
        ref.step(dt, 30)
        grp.step(dt, 30)
    grp.synchronize()
    assert grp.status() == (0, None)
    assert bits_equal(grp.download(), ref.download())
    grp.close()
