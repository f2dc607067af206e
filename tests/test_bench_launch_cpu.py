"""`python bench.py --gpus N` from a plain command line: the process becomes a supervisor that starts the ranks itself, as
mara::evaluate_on<N> starts its own workers (/root/reference is not read here; src/app_parallel.hpp:75-103 is the model).
These run without a GPU: they check the command line the supervisor would run, that it has not touched the GPU, that a
multi-rank run on a box without enough GPUs ends quickly with a message and a non-zero status instead of waiting in a
collective, and the two bounded waits (the supervisor's wall-clock limit, a rank's deadline)."""
import json
import os
import subprocess
import sys
import time

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench_launch  # noqa: E402


def run(cmd, timeout=180, env=None):
    e = dict(os.environ)
    e.pop("WORLD_SIZE", None)
    e.pop("RANK", None)
    e.pop("LOCAL_RANK", None)
    if env:
        e.update(env)
    return subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=timeout, env=e)


@pytest.mark.parametrize("script,extra", [("bench.py", []), ("bench_configs.py", ["--config", "c5"]), ("bench_configs.py", ["--config", "c3"])])
def test_dry_launch_prints_the_child_command_and_touches_no_gpu(script, extra):
    p = run([sys.executable, script] + extra + ["--gpus", "8", "--steps", "7", "--warmup", "2", "--dry-launch"])
    assert p.returncode == 0, p.stderr
    rec = json.loads(p.stdout.strip().splitlines()[-1])
    cmd = rec["command"]
    assert rec["dry_launch"] is True and rec["gpu_touched_by_supervisor"] is False
    # the form the driver's contract names
    assert cmd[1:3] == ["-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and cmd[cmd.index("--nproc-per-node") + 1] == "8"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    i = cmd.index(os.path.join(ROOT, script))
    rest = cmd[i + 1:]
    assert "--dry-launch" not in rest and rest[rest.index("--gpus") + 1] == "8" and rest[rest.index("--steps") + 1] == "7"
    if script == "bench.py":
        assert rec["retry_arguments"] == ["--stepper", "torch"]


def test_supervisor_module_imports_neither_torch_nor_the_library():
    p = run([sys.executable, "-c", "import sys, bench_launch; print(bench_launch.gpu_touched(), 'torch' in sys.modules)"])
    assert p.stdout.split() == ["False", "False"], p.stdout + p.stderr


def test_multi_rank_run_without_enough_gpus_ends_with_a_message_not_in_a_collective():
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("this box has the GPUs: the run would proceed")
    t0 = time.time()
    p = run([sys.executable, "bench.py", "--gpus", "2", "--steps", "1", "--warmup", "0"], timeout=240)
    assert p.returncode != 0
    assert "one process per GPU needs 2" in p.stderr
    assert "retrying once" not in p.stderr              # too few GPUs: another stepper cannot help
    assert time.time() - t0 < 120
    assert not [l for l in p.stdout.splitlines() if l.startswith("{")]


def test_supervisor_wall_clock_limit_ends_the_child_process_group():
    t0 = time.time()
    rc, lines, tail = bench_launch.run_child([sys.executable, "-c", "import time; print('{\"a\": 1}', flush=True); time.sleep(600)"], timeout_s=2)
    assert rc == 124 and lines == ['{"a": 1}']
    assert "limit of the multi-rank child expired" in tail[-1]
    assert time.time() - t0 < 30


def test_rank_deadline_leaves_with_status_3_and_names_the_phase():
    code = ("import time, bench_launch\n"
            "w = bench_launch.Deadline(600, rank=5)\n"
            "w.phase('mh_comm_create (ncclCommInitRank of the native steppers)', limit=1)\n"
            "time.sleep(60)\n")
    t0 = time.time()
    p = run([sys.executable, "-c", code], timeout=60)
    assert p.returncode == 3
    assert "rank 5 is still in phase 'mh_comm_create" in p.stderr
    assert time.time() - t0 < 20
    # a cancelled watchdog lets the process end normally
    p = run([sys.executable, "-c", "import time, bench_launch\nw = bench_launch.Deadline(1)\nw.cancel()\ntime.sleep(2.5)\n"], timeout=60)
    assert p.returncode == 0


def test_retry_is_skipped_when_the_message_says_so_and_taken_otherwise(tmp_path):
    """the supervisor's retry policy on a stand-in child script (no torch.distributed.run involved: child_command is patched)"""
    script = tmp_path / "fake.py"
    script.write_text("import sys\n"
                      "if '--stepper' in sys.argv: print('{\"stepper\": \"torch\"}'); sys.exit(0)\n"
                      "print('first attempt fails'); sys.exit(7)\n")
    code = ("import sys, bench_launch\n"
            "bench_launch.child_command = lambda script, argv, gpus, port=None: [sys.executable, script] + list(argv)\n"
            "bench_launch.supervise(%r, ['--x'], 2, timeout_s=60, retry_with=('--stepper', 'torch'))\n" % str(script))
    p = run([sys.executable, "-c", code])
    assert p.returncode == 0 and p.stdout.strip() == '{"stepper": "torch"}'
    assert "retrying once with --stepper torch" in p.stderr
    script.write_text("import sys\nprint('nothing to be done %s'); sys.exit(7)\n" % bench_launch.NO_RETRY)
    p = run([sys.executable, "-c", code])
    assert p.returncode == 7 and "retrying once" not in p.stderr
