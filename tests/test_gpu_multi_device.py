"""What needs MORE THAN ONE MI355X: RCCL between distinct devices and one process driving several devices.

On a one-GPU box every test here skips (the loopback groups of test_gpu_slab_group.py / test_gpu_block3d.py / test_gpu_binary_bands.py
execute the same stepper code on one device, and test_gpu_rccl.py runs the RCCL calls to self). On a node with 2+ GPUs these are the
first things to run: the requirement is SURVEY.md §8e's - results do not depend on the partition, bit for bit - and the reference's
analogue is mara::evaluate_on<N> (src/app_parallel.hpp:75-103), whose thread slabs leave the result unchanged."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
from conftest import bits_equal

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(900)]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def device_count():
    import mara3_amd
    return mara3_amd.load_library().mh_device_count()


def need(n):
    if device_count() < n:
        pytest.skip("needs %d GPUs, this box has %d" % (n, device_count()))


@pytest.mark.parametrize("world", [2, 4])
@pytest.mark.parametrize("riemann,arith,bc", [("hlle", "strict", "outflow"), ("hllc", "fast", "periodic")])
def test_one_process_driving_distinct_devices_equals_single_domain(world, riemann, arith, bc):
    """mh_slab_group_create_on with member r on device r: peer copies, events ordering streams of different devices"""
    need(world)
    from mara3_amd import engine, setups
    from mara3_amd.slab import NativeSlabGroup
    shape, gamma = (250, 300), 1.4
    dl = (1.0 / shape[0], 1.0 / shape[1])
    u0 = setups.wave_ic(shape, gamma, seed=5)
    ref = engine.EulerCartSolver(shape, dl, gamma, 1.5, riemann, 2, bc, arith=arith)
    ref.upload(u0)
    grp = NativeSlabGroup(shape, dl, gamma, 1.5, riemann, 2, bc, world=world, arith=arith, devices=list(range(world)))
    grp.upload(u0)
    for nsteps in (1, 3, 5, 2):
        ref.step(5e-4, nsteps)
        grp.step(5e-4, nsteps)
        grp.synchronize()
        assert bits_equal(grp.download(), ref.download()), (world, nsteps)
    assert grp.status() == (0, None)
    grp.close()
    ref.close()


def run_bench(script, args, timeout=800):
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, script)] + args, cwd=ROOT, capture_output=True, text=True, timeout=timeout, env=env)
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert p.returncode == 0 and len(lines) == 1, (p.returncode, p.stdout[-2000:], p.stderr[-4000:])
    return json.loads(lines[0]), p.stderr


@pytest.mark.parametrize("world", [2])
def test_plain_bench_command_runs_the_ranks_over_rccl_and_the_slabs_match_one_gpu(world):
    """`python bench.py --gpus N`: supervisor -> torch.distributed.run -> one rank per GPU, ghost rows over RCCL between DISTINCT devices"""
    need(world)
    out, err = run_bench("bench.py", ["--gpus", str(world), "--steps", "10", "--warmup", "3", "--blocks", "1", "--no-cpu-baseline", "--precondition", "0"])
    assert out["n_gpus"] == world and out["value"] > 0
    assert out["slabs_bit_identical_to_one_gpu_run"] is True
    assert out["summary"]["stepper"] == "native", err[-2000:]
    assert out["config"]["status_word"] == 0


def test_block_and_band_configs_start_from_a_plain_command():
    need(2)
    out, _ = run_bench("bench_configs.py", ["--config", "c5", "--gpus", "2", "--grid", "128", "--steps", "3", "--warmup", "1"])
    assert out["n_gpus"] == 2 and out["value"] > 0 and out["config"]["status_word"] == 0
    out, _ = run_bench("bench_configs.py", ["--config", "c3", "--gpus", "2", "--steps", "5", "--warmup", "2"])
    assert out["n_gpus"] == 2 and out["value"] > 0


@pytest.mark.parametrize("world", [2, 4])
def test_cloud_one_launch_step_across_radial_cuts_on_distinct_devices(world, monkeypatch):
    """round 5: `cloud`'s fused RK2 step across radial cuts (four ghost rows, one exchange per step) with member r on device r - peer copies of the
    four-row blocks, the nozzle row through mh_slab_group_set_inflow - equals the one-launch step of the whole field bit for bit"""
    need(world)
    monkeypatch.setenv("MH_SLAB_FUSED_CUTS", "1")
    from mara3_amd import engine
    from mara3_amd.slab import NativeSlabGroup
    from test_gpu_cloud_fused import smooth_cloud_state, run
    rv, qv, u0, inflow, dt = smooth_cloud_state(engine, 130, 117, seed=world)
    whole, st = run(engine, rv, qv, u0, inflow, dt, (3,), True)
    g = NativeSlabGroup(world=world, rk_order=2, plm_theta=1.2, gamma=4.0 / 3, arith="fast", r_vertices=rv, q_vertices=qv, temperature_floor=0.0,
                        devices=list(range(world)))
    assert g.launches_per_step() == [1] * world
    g.upload(u0)
    for n in range(3):
        g.set_inflow(inflow[n])
        g.step(dt, 1)
    g.synchronize()
    assert st == (0, None) and g.status() == (0, None)
    assert bits_equal(g.download(), whole[0])
    g.close()
