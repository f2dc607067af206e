"""CPU multi-process tests of the N>1 path: world_size-2 and -3 runs over the gloo backend. The slab cut, the
ghost-row exchange (message pairing, ordering when both neighbours are the same rank, periodic wrap) and the
edge/interior launch split are the product code of mara3_amd.slab; only the stage arithmetic is replaced by
the oracle (tests/slab_helpers.py). The gathered result must be BIT-IDENTICAL to the single-domain oracle run:
per-cell arithmetic must not depend on the partition."""
import os
import sys
import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from conftest import bits_equal


def _worker(rank, world, port, shape, bc, riemann, nsteps, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from slab_helpers import OracleStage
        from mara3_amd.slab import SlabEulerStepper
        from mara3_amd import setups
        gamma, theta = 1.4, 1.5
        dl = (1.0 / shape[0], 1.0 / shape[1])
        u0 = setups.wave_ic(shape, gamma, seed=11)
        ref = []
        stage = OracleStage(dl, gamma, theta, riemann, bc == "periodic", ref)
        st = SlabEulerStepper(shape, dl, gamma, theta, riemann, 2, bc, rank=rank, world=world, device="cpu",
                              stage_fn=stage, edge_chunk_rows=3)
        ref.append(st)
        st.load_slab(u0[st.row0:st.row1])
        st.step(2e-3, nsteps)
        np.save(os.path.join(out_dir, "slab_%d.npy" % rank), st.slab().numpy())
        np.save(os.path.join(out_dir, "rows_%d.npy" % rank), np.array([st.row0, st.row1]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,shape,bc,riemann", [
    (2, (37, 20), "outflow", "hlle"),
    (2, (24, 17), "periodic", "hlle"),      # both neighbours are the same rank: ordering of the paired messages
    (3, (31, 12), "periodic", "hllc"),
    (3, (40, 9), "outflow", "hllc"),
])
def test_slabs_over_gloo_match_single_domain(tmp_path, oracle, world, shape, bc, riemann):
    port = 29500 + (os.getpid() % 2000) + world
    nsteps = 3
    mp.spawn(_worker, args=(world, port, shape, bc, riemann, nsteps, str(tmp_path)), nprocs=world, join=True)
    from mara3_amd import setups
    u0 = setups.wave_ic(shape, 1.4, seed=11)
    dl = (1.0 / shape[0], 1.0 / shape[1])
    kind = oracle.RIEMANN_HLLC if riemann == "hllc" else oracle.RIEMANN_HLLE
    obc = oracle.BC_PERIODIC if bc == "periodic" else oracle.BC_OUTFLOW
    want = oracle.euler_cart_run(u0, dl, 2e-3, nsteps, 1.4, 1.5, 2, kind, obc)
    got = np.empty_like(want)
    covered = 0
    for r in range(world):
        a, b = np.load(os.path.join(tmp_path, "rows_%d.npy" % r))
        assert (a, b) == oracle.partition_rows(shape[0], world, r)
        got[a:b] = np.load(os.path.join(tmp_path, "slab_%d.npy" % r))
        covered += b - a
    assert covered == shape[0]
    assert bits_equal(got, want), np.abs(got - want).max()


def _cloud_worker(rank, world, port, case, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from conftest import golden
        from slab_helpers import OracleCloudStage
        from mara3_amd.slab import SlabCloudStepper
        g = golden(case)
        ref = []
        stage = OracleCloudStage(g["rv"], g["qv"], float(g["theta"]), float(g["tfloor"]), ref)
        st = SlabCloudStepper(g["rv"], g["qv"], rk_order=int(g["rk"]), plm_theta=float(g["theta"]), temperature_floor=float(g["tfloor"]),
                              rank=rank, world=world, device="cpu", stage_fn=stage, edge_chunk_rows=3)
        ref.append(st)
        st.load_slab(g["u0"][st.row0:st.row1])
        for n in range(int(g["nsteps"])):
            st.set_inflow(g["inflow"][n])
            st.step(float(g["dt"]), 1)
        np.save(os.path.join(out_dir, "slab_%d.npy" % rank), st.slab().numpy())
        np.save(os.path.join(out_dir, "rows_%d.npy" % rank), np.array([st.row0, st.row1]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,case", [(2, "cloud_nr32_plm_rk2"), (3, "cloud_nr70_plm_rk2")])
def test_cloud_radial_slabs_over_gloo_match_the_reference(tmp_path, oracle, world, case):
    """BASELINE config 4's decomposition: radial slabs of the `cloud` grid, two ghost rows exchanged per stage. The gathered
    result must equal the single-domain REFERENCE-generated vector bit for bit."""
    from conftest import golden
    port = 31500 + (os.getpid() % 2000) + world
    mp.spawn(_cloud_worker, args=(world, port, case, str(tmp_path)), nprocs=world, join=True)
    g = golden(case)
    got = np.empty_like(g["un"])
    covered = 0
    for r in range(world):
        a, b = np.load(os.path.join(tmp_path, "rows_%d.npy" % r))
        got[a:b] = np.load(os.path.join(tmp_path, "slab_%d.npy" % r))
        covered += b - a
    assert covered == g["un"].shape[0]
    assert bits_equal(got, g["un"]), np.abs(got - g["un"]).max()


def test_slab_fingerprint_detects_a_single_bit():
    """bench.py's N > 1 determinism check compares fingerprints of the ranks' slabs with those of a one-GPU run."""
    import numpy as np
    import torch
    from mara3_amd.slab import slab_fingerprint
    rng = np.random.default_rng(5)
    a = torch.from_numpy(rng.standard_normal((37, 11, 5)))
    assert slab_fingerprint(a) == slab_fingerprint(a.clone())
    assert slab_fingerprint(a)[1] == int(np.bitwise_xor.reduce(a.numpy().view(np.int64).reshape(-1)))
    b = a.clone()
    b.view(torch.int64)[3, 4, 1] ^= 1                      # flip the last mantissa bit of one value
    assert slab_fingerprint(b) != slab_fingerprint(a)
    assert slab_fingerprint(a[5:9]) == slab_fingerprint(a[5:9].contiguous())
