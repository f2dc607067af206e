"""Start of a multi-rank bench run from a plain command line, and the guard that keeps such a run from hanging.

    python bench.py --gpus N ...            (no torch.distributed.run in front of it)

The reference's parallel evaluator starts its own workers - `mara::evaluate_on<N>()` forks N thread slabs inside the call and joins them
(/root/reference/src/app_parallel.hpp:75-103); the caller does nothing special. Here a worker is a process that owns one GPU, so the
equivalent is: the process the user started becomes a SUPERVISOR that has not touched the GPU (this module imports neither torch nor the
HIP library), starts `python -m torch.distributed.run --nnodes=1 --nproc-per-node N <script> <same arguments>` as a fresh child process
(subprocess, never exec), relays the one JSON line rank 0 prints and exits with the child's return code.

Nothing here can wait for ever:
  * the child runs under a wall-clock limit (--launch-timeout); at the limit its whole process group is ended and the supervisor exits 124;
  * every rank arms a deadline of its own (`Deadline`): a rank that is still inside a collective set-up call (ncclCommInitRank, a
    barrier whose peers died) when it expires prints which phase it was in and leaves with status 3 - torch.distributed.run then ends
    the other ranks;
  * with the native stepper, a failed child is retried ONCE with `--stepper torch` (the torch.distributed stepper needs nothing of the
    library's own RCCL binding); the JSON line says which stepper produced it.
"""
import json
import os
import signal
import socket
import subprocess
import sys
import threading
import time


NO_RETRY = "[no retry]"        # a rank's message carries this when another attempt cannot help (too few GPUs, no library)


def free_port():
    s = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def child_command(script, argv, gpus, port=None):
    """The command line of the multi-rank child: exactly the form the driver's contract names."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(gpus), "--master-addr", "127.0.0.1",
            "--master-port", str(port if port is not None else free_port()), os.path.abspath(script)] + list(argv)


def gpu_touched():
    """True if this process has loaded anything that could have initialised the GPU (the supervisor must not)."""
    return any(m in sys.modules for m in ("torch", "mara3_amd", "mara3_amd._lib"))


def strip_option(argv, name, has_value=True):
    out, skip = [], False
    for a in argv:
        if skip:
            skip = False
            continue
        if a == name:
            skip = has_value
            continue
        if has_value and a.startswith(name + "="):
            continue
        out.append(a)
    return out


def run_child(cmd, timeout_s):
    """Run the child in its own process group; returns (return code, stdout lines that look like JSON objects, the other output lines)."""
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("MASTER_ADDR", "127.0.0.1")
    env.setdefault("OMP_NUM_THREADS", "4")
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=env, start_new_session=True)
    lines, other = [], []

    def pump():
        for line in proc.stdout:
            line = line.rstrip("\n")
            if line.startswith("{") and line.endswith("}"):
                lines.append(line)
            else:
                other.append(line)
                print(line, file=sys.stderr, flush=True)

    t = threading.Thread(target=pump, daemon=True)
    t.start()
    try:
        rc = proc.wait(timeout=timeout_s)
    except subprocess.TimeoutExpired:
        # the exact process group this call created
        try:
            os.killpg(proc.pid, signal.SIGTERM)
            try:
                proc.wait(timeout=10)
            except subprocess.TimeoutExpired:
                pass
            os.killpg(proc.pid, signal.SIGKILL)       # whatever of the group is left
        except ProcessLookupError:
            pass
        proc.wait()
        rc = 124
        other.append("bench_launch: the %d s limit of the multi-rank child expired; its process group was ended" % timeout_s)
    t.join(timeout=10)
    return rc, lines, other


def supervise(script, argv, gpus, timeout_s=900, dry=False, retry_with=None):
    """Called by a bench script when --gpus N > 1 and WORLD_SIZE is unset. Never returns: exits with the child's status."""
    argv = strip_option(strip_option(list(argv), "--dry-launch", has_value=False), "--launch-timeout")
    cmd = child_command(script, argv, gpus)
    if dry:
        print(json.dumps({"dry_launch": True, "command": cmd, "gpu_touched_by_supervisor": gpu_touched(),
                          "retry_arguments": list(retry_with) if retry_with else None, "timeout_s": timeout_s}), flush=True)
        raise SystemExit(0)
    assert not gpu_touched(), "the supervisor must start its ranks before anything initialises the GPU"
    t0 = time.time()
    rc, lines, tail = run_child(cmd, timeout_s)
    if rc != 0 and retry_with and not any(a in argv for a in retry_with[:1]) and not any(NO_RETRY in l for l in tail):
        left = max(120, timeout_s - int(time.time() - t0))
        print("bench_launch: the %d-rank run ended with status %d; retrying once with %s" % (gpus, rc, " ".join(retry_with)), file=sys.stderr, flush=True)
        rc, lines, tail = run_child(child_command(script, argv + list(retry_with), gpus), left)
    for line in lines:
        print(line, flush=True)
    if rc == 0 and not lines:
        print("bench_launch: the child ended with status 0 but printed no JSON line", file=sys.stderr, flush=True)
        rc = 1
    raise SystemExit(rc)


class Deadline:
    """Per-rank watchdog of a multi-rank run. `phase(name)` says what the rank is about to do; if that phase lasts longer than the deadline
    (`seconds`, or the phase's own shorter `limit`), the rank prints the phase it is stuck in and leaves with status 3 (os._exit: a thread blocked
    inside ncclCommInitRank cannot be interrupted any other way). The clock restarts with every phase: a run that keeps moving from phase to
    phase is never ended for its total length - that bound is the supervisor's `--launch-timeout`."""

    def __init__(self, seconds, rank=0, enabled=True):
        self.seconds, self.rank, self.enabled = seconds, rank, enabled and seconds > 0
        self.name, self.t0, self.limit = "start", time.time(), None
        self._stop = threading.Event()
        if self.enabled:
            self.thread = threading.Thread(target=self._watch, daemon=True)
            self.thread.start()

    def phase(self, name, limit=None):
        """enter a phase: the deadline counts from here; `limit` (seconds) is a shorter bound for this phase alone"""
        self.name, self.t0, self.limit = name, time.time(), (time.time() + limit if limit else None)

    def _watch(self):
        while not self._stop.wait(1.0):
            now = time.time()
            if now - self.t0 > self.seconds or (self.limit is not None and now > self.limit):
                which = "the %d s deadline of a phase" % self.seconds if now - self.t0 > self.seconds else "the limit of this phase"
                print("bench: rank %d is still in phase '%s' after %.0f s - %s has passed; leaving with status 3 instead of waiting in a collective"
                      % (self.rank, self.name, now - self.t0, which), file=sys.stderr, flush=True)
                os._exit(3)

    def cancel(self):
        self._stop.set()


class Ranks:
    """What every rank of a bench run does first, in one place (bench.py, bench_configs.py): read the launcher's environment, check that
    there is a GPU per rank BEFORE any collective can be entered, select the device, join the torch.distributed (RCCL) group under the
    watchdog - and later hand out the process's ONE communicator of the native steppers. world == 1: no process group, no watchdog."""

    def __init__(self, gpus, deadline_s):
        import torch
        self.torch = torch
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        if self.world != gpus:
            raise SystemExit("--gpus %d does not match WORLD_SIZE %d %s" % (gpus, self.world, NO_RETRY))
        self.watch = Deadline(deadline_s, rank=self.rank, enabled=self.world > 1)
        ndev = torch.cuda.device_count()              # counts without initialising the GPU
        if self.world > 1 and ndev < self.world and not os.environ.get("MH_BENCH_SHARE_DEVICES"):
            raise SystemExit("--gpus %d: rank %d sees %d GPU(s); one process per GPU needs %d %s" % (self.world, self.rank, ndev, self.world, NO_RETRY))
        if not torch.cuda.is_available():
            raise SystemExit("this bench needs an MI355X; there is no CPU path %s" % NO_RETRY)
        self.local_rank = local_rank % max(1, ndev)   # == LOCAL_RANK unless MH_BENCH_SHARE_DEVICES (rehearsal of the error path on one GPU)
        torch.cuda.set_device(self.local_rank)
        self.comm = None
        if self.world > 1:
            import datetime
            import torch.distributed as dist
            self.dist = dist
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            self.watch.phase("torch.distributed.init_process_group (RCCL)", limit=240)
            dist.init_process_group("nccl", rank=self.rank, world_size=self.world, device_id=torch.device("cuda", self.local_rank),
                                    timeout=datetime.timedelta(seconds=300))
            self.watch.phase("first barrier", limit=240)
            dist.barrier()
            self.watch.phase("set-up")

    def agree(self, ok):
        """True only if `ok` on every rank (so that all ranks take the same path before a collective set-up call)"""
        if self.world == 1:
            return bool(ok)
        t = self.torch.tensor([1 if ok else 0], device="cuda")
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MIN)
        return int(t.item()) == 1

    def process_comm(self):
        """the ONE communicator of this process's native steppers (mh_comm_create: collective, entered once, bounded by the watchdog)"""
        if self.comm is None:
            from mara3_amd.slab import NativeComm, native_comm_id
            self.watch.phase("mh_comm_create (ncclCommInitRank of the native steppers)", limit=240)
            cid = native_comm_id(self.rank, self.world, device="cuda")
            self.comm = NativeComm(cid, self.rank, self.world, device=self.local_rank)
            self.watch.phase("after mh_comm_create")
        return self.comm

    def fence(self):
        self.torch.cuda.synchronize()
        if self.world > 1:
            self.dist.barrier()
            self.torch.cuda.synchronize()

    def max_over_ranks(self, seconds):
        if self.world == 1:
            return seconds
        t = self.torch.tensor([seconds], dtype=self.torch.float64, device="cuda")
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def close(self):
        self.watch.phase("closing", limit=120)
        if self.world > 1:
            self.dist.barrier()
            if self.comm is not None:
                self.comm.close()
                self.comm = None
            self.dist.destroy_process_group()
        self.watch.cancel()
