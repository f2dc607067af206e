#!/usr/bin/env python3
"""bench.py — zone-updates/s of the 2-D Euler PLM+HLLC RK2 sweep at 4096^2 (BASELINE.json configs[1]).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--riemann hllc|hlle] [--grid 4096]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
           bench.py --gpus N --steps K --warmup W

One process per GPU. The global 4096^2 grid is cut into axis-0 slabs with the reference's partition formula
(strong scaling: total work fixed), ghost rows travel as RCCL send/recv. A "step" is one full RK2 time step
of the whole grid; inputs are resident in HBM before the timed region. Rank 0 prints as its LAST stdout line ONE compact JSON object
(bench_report.final_line: at most 6000 characters - contract keys, config, roofline, roofline_step, cpu_baseline, cpu_reference, summary);
the full record goes to bench_details.json next to this file and to stderr:

  roofline                                   the headline leg's dominant kernel as a HARDWARE fraction: the one-launch RK2 step is fp64-issue-bound
                                             (bound "fp64": FLOP per launch over the launch time against 78.6 TFLOP/s, with VALU-busy and the HBM
                                             fraction its measured bytes make); a stage kernel of a two-launch step is bound "hbm" at 80 / 120 B per cell
  roofline_step                              throughput in SURVEY 8(d) byte-equivalents (200 B x zone-updates/s) - the unit BASELINE.md's target is in
  repeat_blocks                              4 more timed blocks of K steps of the same leg (spread of the measurement)
  legs                                       the same measurement for the other variants, each with its own rooflines:
                                             strict+HLLE (the variant pinned bit for bit to the reference), fast+HLLE, strict+HLLC, and
                                             the smooth periodic wave of SURVEY.md §8d (every face in a shock / rarefaction branch)
  extra_configs                              BASELINE configs 3, 4, 5 on this GPU (bench_configs.py), N = 1 only
  cpu_baseline / cpu_reference               N = 1 only: the oracle and the reference's own lazy-array composition on the host cores

Timing: the timed region is the product path (N = 1: the fused step's one launch per step issued as it is, the two-launch step replayed from a HIP graph;
N > 1: eager two-stream issue with RCCL), no events in it.
The per-kernel durations behind the rooflines come from HIP events riding on the launches, in a short separate pass right after it.
"""
import argparse
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import bench_report
from bench_report import HBM_PEAK_GBS, FP64_VECTOR_PEAK_TFLOPS

BYTES_STAGE1 = 2 * 5 * 8       # read U, write U1                      
BYTES_STAGE2 = 3 * 5 * 8       # read U1, read U0, write U (in place)
BYTES_STEP = BYTES_STAGE1 + BYTES_STAGE2   # 200 B per zone-update


def host_cores():
    """Usable host cores: affinity mask capped by the cgroup CPU quota (the GPU box gives a 1-GPU job a share)."""
    cores = os.cpu_count() or 1
    try:
        cores = len(os.sched_getaffinity(0))
    except Exception:
        pass
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    cores = min(cores, max(1, int(float(txt[0]) / float(txt[1]) + 0.5)))
            else:
                q = int(txt[0])
                per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if q > 0:
                    cores = min(cores, max(1, int(q / per + 0.5)))
            break
        except Exception:
            continue
    return min(cores, int(os.environ.get("MARA_BENCH_CPU_THREADS", "64")))


KERNEL_SOURCES = ("euler2d.hip", "euler2d_fused.hip", "euler2d_rows.hpp", "euler3d.hip", "euler3d_fast.hip", "euler3d_kernel.hpp", "cloud.hip", "cloud_fused.hip", "cloud_rows.hpp", "row_check.hpp", "binary.hip", "binary_fast.hip", "binary_kernel.hpp", "euler_device.hpp", "euler_device_fast.hpp", "srhd_device.hpp",
                  "srhd_device_fast.hpp", "iso2d_device.hpp", "binary_device.hpp", "status_device.hpp")


def code_only(text):
    """a source file without its comments and with runs of white space collapsed: what the fingerprint is taken over (round 5: editing a
    comment in a kernel file used to make every recorded counter 'stale')"""
    import re
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    text = re.sub(r"//[^\n]*", " ", text)
    return re.sub(r"\s+", " ", text).strip()


def csrc_fingerprint():
    """sha256 (16 hex digits) over the CODE (comments and white space aside) of the measured stage kernels and their device headers: PMC
    numbers recorded for other sources are not reported (see `traffic`). Host-side files of the library (steppers, API) do not enter."""
    h = hashlib.sha256()
    d = os.path.join(ROOT, "mara3_amd", "csrc")
    for name in KERNEL_SOURCES:
        h.update(name.encode())
        h.update(code_only(open(os.path.join(d, name), "r", errors="replace").read()).encode())
    return h.hexdigest()[:16]


def cpu_baseline(n, gamma, theta, riemann, budget_s=15.0):
    """Time the oracle (test infrastructure, used here ONLY as the reported CPU baseline) on the same workload."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import mara_oracle
    from mara3_amd import setups
    cores = host_cores()
    kind = mara_oracle.RIEMANN_HLLC if riemann == "hllc" else mara_oracle.RIEMANN_HLLE
    dl = (1.0 / n, 1.0 / n)
    dt = setups.baseline_dt(n)
    u = setups.blast_ic((n, n), gamma)
    t0 = time.perf_counter()
    u = mara_oracle.euler_cart_run(u, dl, dt, 1, gamma, theta, 2, kind, mara_oracle.BC_OUTFLOW, nthreads=cores)
    t1 = time.perf_counter() - t0
    steps = max(1, min(50, int(budget_s / max(t1, 1e-3))))
    t0 = time.perf_counter()
    mara_oracle.euler_cart_run(u, dl, dt, steps, gamma, theta, 2, kind, mara_oracle.BC_OUTFLOW, nthreads=cores)
    t = time.perf_counter() - t0
    return {"value": n * n * steps / t / 1e6, "unit": "Mcells/s", "cores": cores, "kind": "port",
            "sample": "%d RK2 steps of the same %dx%d PLM+%s workload, oracle/mara_oracle.c with %d slab threads"
                      % (steps, n, n, riemann.upper(), cores)}


def upstream_threads(cores):
    """mara::evaluate_on<N>() takes its thread count as a template parameter: the drivers instantiate 1, 2, 4, 8, 16 and 32 - the largest that the
    usable host cores cover"""
    return max(t for t in (1, 2, 4, 8, 16, 32) if t <= max(1, cores))


def cpu_reference(gamma, theta, n=1024, steps=4):
    """If the reference-composed driver was prebuilt (oracle/_ref), time Mara3's OWN CPU path on the host cores: its lazy-array composition of the
    step evaluated as upstream evaluates its `advance` - primitives and result through the threaded evaluator mara::evaluate_on<N>()
    (src/app_parallel.hpp:72-103; src/subprog_cloud.cpp:525-533, :582), gradients through nd::to_shared() (:566). The one-thread figure of the
    same composition is reported beside it (the evaluator scales poorly on this composition - that is the reference's behaviour, not this repo's)."""
    import subprocess
    import tempfile
    from mara3_amd import setups
    exe = os.path.join(ROOT, "oracle", "_ref", "euler_cart_ref")
    if not os.path.exists(exe):
        return None
    cores = host_cores()
    threads = upstream_threads(cores)
    u = setups.blast_ic((n, n), gamma)
    with tempfile.TemporaryDirectory() as d:
        fin, fout = os.path.join(d, "in"), os.path.join(d, "out")
        u.tofile(fin)
        hx = lambda x: float(x).hex()
        args = [exe, "2", str(n), str(n), "1", hx(gamma), hx(theta), "2", "0", hx(setups.baseline_dt(n)),
                hx(1.0 / n), hx(1.0 / n), hx(1.0)]

        def rate(th):
            t0 = time.perf_counter()
            subprocess.check_call(args + ["0", fin, fout, str(th)])
            tz = time.perf_counter() - t0
            t0 = time.perf_counter()
            subprocess.check_call(args + [str(steps), fin, fout, str(th)])
            return n * n * steps / (time.perf_counter() - t0 - tz) / 1e6
        try:
            many = rate(threads)
            one = rate(1) if threads > 1 else many
        except Exception:
            return None
    return {"value": many, "unit": "Mcells/s", "cores": threads, "kind": "reference", "one_thread": one, "host_cores": cores,
            "sample": "%d RK2 steps at %dx%d PLM+HLLE: the reference's headers composed and evaluated as its own `advance` is (lazy arrays, mara::evaluate_on<%d>() "
                      "where upstream pipes `| evaluate`), oracle/ref_drivers/euler_cart_ref.cpp; results bit-identical to the one-thread run" % (steps, n, n, threads)}


def extra_configs():
    """BASELINE configs 3, 4, 5 on this GPU, each as a child process of bench_configs.py (its JSON line is embedded as is). Step counts per
    config: a child starts on a GPU that idled through its set-up, so the warm-up steps cover the clock ramp that the headline leg's scratch-grid
    preconditioning covers (about 25 launches) and the timed region is tens of milliseconds."""
    import subprocess
    out = {}
    # (c4 FAST is ONE launch per step since round 4: 30 warm-up steps, where 10 two-launch steps used to cover the ramp)
    for cfg, steps, warmup in (("c3", 100, 60), ("c4", 30, 30), ("c5", 10, 4)):
        try:
            p = subprocess.run([sys.executable, os.path.join(ROOT, "bench_configs.py"), "--config", cfg, "--steps", str(steps), "--warmup", str(warmup)],
                               capture_output=True, text=True, timeout=900)
            line = [l for l in p.stdout.splitlines() if l.startswith("{")]
            out[cfg] = json.loads(line[-1]) if p.returncode == 0 and line else {"error": (p.stderr or p.stdout)[-400:]}
        except Exception as e:
            out[cfg] = {"error": repr(e)}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--grid", dest="n", type=int, default=4096, help="cells per axis of the global grid")
    ap.add_argument("--riemann", default="hllc", choices=["hllc", "hlle"])
    ap.add_argument("--theta", type=float, default=1.5)
    ap.add_argument("--chunk-rows", type=int, default=0)
    ap.add_argument("--arith", default="fast", choices=["strict", "fast"], help="arithmetic contract of the headline value")
    ap.add_argument("--stepper", default="native", choices=["native", "torch"],
                    help="native: C++ slab stepper of libmara_hip.so (RCCL called from the library); torch: Python stepper over torch.distributed")
    ap.add_argument("--single-arith", action="store_true", help="headline leg only: no other variants (legs), no extra configs")
    ap.add_argument("--blocks", type=int, default=5, help="timed blocks of K steps of the headline leg (the first one is `value`)")
    ap.add_argument("--precondition", type=int, default=60,
                    help="steps of a scratch grid run right BEFORE the W warm-up steps: the first ~25 launches after an idle period (building and "
                         "uploading the initial condition) run up to 25 %% slower while the clocks settle (kernel trace, DESIGN.md §6); 0 = off")
    ap.add_argument("--no-fuse", action="store_true",
                    help="headline leg as two launches per RK2 step instead of the fused one (euler2d_fused.hip; FAST arithmetic, one GPU)")
    ap.add_argument("--no-planar", action="store_true",
                    help="headline leg on the GENERAL fused kernel: by default the library verifies at upload that this 2-D field has no third momentum and "
                         "then skips that component (mh_euler_cart_desc.planar; same bits in the other four)")
    ap.add_argument("--no-overlap", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra-configs", action="store_true")
    ap.add_argument("--loopback-slabs", type=int, default=0,
                    help="rehearsal on ONE GPU of the N > 1 code path: the grid as this many slab objects of the native stepper exchanging through its "
                         "loopback backend (RCCL refuses two ranks on one device), incl. the partition check against the one-domain run")
    ap.add_argument("--dry-launch", action="store_true",
                    help="with --gpus N > 1 from a plain command line: print the multi-rank child's command line and stop (nothing touches the GPU)")
    ap.add_argument("--launch-timeout", type=int, default=900, help="wall-clock limit in seconds of the multi-rank child the supervisor starts")
    ap.add_argument("--deadline", type=int, default=600,
                    help="N > 1: seconds after which a rank that is still waiting (ncclCommInitRank, a barrier whose peers are gone) leaves with status 3")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # a plain `python bench.py --gpus N`: this process becomes the supervisor of N ranks (bench_launch.py) - before torch or HIP are loaded
        import bench_launch
        bench_launch.supervise(__file__, sys.argv[1:], args.gpus, timeout_s=args.launch_timeout, dry=args.dry_launch,
                               retry_with=("--stepper", "torch") if args.stepper == "native" else None)
    if args.dry_launch:
        raise SystemExit("--dry-launch is for --gpus N > 1 without torch.distributed.run in front")

    import torch
    import torch.distributed as dist
    import mara3_amd
    from mara3_amd import setups
    from mara3_amd.slab import SlabEulerStepper, NativeSlabStepper, NativeSlabGroup, native_comm_id, partition_rows
    from mara3_amd.slab import slab_fingerprint as slab_checksum

    import bench_launch
    if args.loopback_slabs and int(os.environ.get("WORLD_SIZE", "1")) != 1:
        raise SystemExit("--loopback-slabs is a one-GPU rehearsal")
    ranks = bench_launch.Ranks(args.gpus, args.deadline)      # device check, device selection, RCCL process group - under the watchdog at N > 1
    world, rank, local_rank, watch = ranks.world, ranks.rank, ranks.local_rank, ranks.watch
    mara3_amd.load_library()           # fails loudly if the HIP library is missing

    n, gamma = args.n, 5.0 / 3
    dl = (1.0 / n, 1.0 / n)
    dt = setups.baseline_dt(n)
    nslabs = args.loopback_slabs if args.loopback_slabs else world

    fence = ranks.fence
    state = {"stepper": args.stepper}

    class GroupAsStepper:
        """--loopback-slabs: the N slab objects of one process behind the stepper interface the timing code uses"""
        def __init__(self, arith, riemann, bc, planar=None):
            self.g = NativeSlabGroup((n, n), dl, gamma, args.theta, riemann, 2, bc, world=nslabs, device=local_rank, chunk_rows=args.chunk_rows, arith=arith, planar=planar)
            self.row0, self.row1 = 0, n
            self.member = NativeSlabStepper((n, n), dl, gamma, handle=torch_free_handle(self.g, nslabs // 2))      # profiled member: an inner rank
        def load_slab(self, u): self.g.upload(u)
        def step(self, dt_, k): self.g.step(dt_, k)
        def synchronize(self): self.g.synchronize()
        def profile(self, on): self.member.profile(on)
        def profile_read(self): return self.member.profile_read()
        def is_planar(self): return self.member.is_planar()
        def status(self): return self.g.status()[0]
        def slab_host(self): return self.g.download()
        def close(self):
            self.member.handle = None
            self.g.close()

    def torch_free_handle(group, r):
        import ctypes
        return ctypes.c_void_p(group.handles[r])

    def make_stepper(arith, riemann, bc, fuse=None, planar=None):
        # planar: None = the library's own check of the uploaded field (one GPU, loopback groups); with ranks in other processes the bench
        # ASSERTS what it knows of its initial conditions (each rank's upload still verifies its own rows); False = the general kernel
        if planar is None and world > 1:
            planar = True
        if args.loopback_slabs:
            return GroupAsStepper(arith, riemann, bc, planar)
        if state["stepper"] == "native":
            st, err = None, None
            try:
                # without a communicator first: ncclCommInitRank is collective, so every rank must have got this far before any enters it
                st = NativeSlabStepper((n, n), dl, gamma, args.theta, riemann, 2, bc, rank=rank, world=world,
                                       comm_id=None, device=local_rank, chunk_rows=args.chunk_rows, arith=arith, fuse=fuse, planar=planar)
            except mara3_amd.MaraHipError as e:
                err = e
            if ranks.agree(st is not None):                       # every rank must take the same path
                if world > 1:
                    st.use_comm(ranks.process_comm())
                return st
            if world == 1:
                raise err
            if rank == 0:
                print("bench.py: native slab stepper unavailable (%s); using the torch.distributed stepper" % err, file=sys.stderr)
            if st is not None:
                st.close()
            state["stepper"] = "torch"
        return SlabEulerStepper((n, n), dl, gamma, args.theta, riemann, 2, bc, rank=rank, world=world,
                                device="cuda", overlap=not args.no_overlap, chunk_rows=args.chunk_rows, arith=arith)

    def prime(arith, riemann):
        """One step of a 128 x 128 throw-away problem: first-use costs of the library (code-object load, stream and event creation) are
        initialisation, not part of a step of the workload - they must not land in the timed region when the caller asks for W = 0."""
        tiny = NativeSlabStepper((128, 128), (1.0 / 128, 1.0 / 128), gamma, args.theta, riemann, 2, "outflow", device=local_rank, arith=arith)
        tiny.load_slab(setups.blast_ic((128, 128), gamma))
        tiny.step(setups.baseline_dt(128), 1)
        tiny.synchronize()
        tiny.close()

    def initial_state(workload, row0, row1):
        if workload == "blast":
            return setups.blast_ic((n, n), gamma, row_range=(row0, row1))
        return setups.smooth_wave_ic((n, n), gamma, row_range=(row0, row1))

    def run_leg(arith, riemann, workload, nblocks=1, keep_state=False, fuse=None, planar=None):
        """W untimed + K timed steps (+ nblocks - 1 further timed blocks of K) of the whole grid, then a short profiled pass.
        fuse: None = the library's choice (one fused launch per RK2 step where it exists: FAST arithmetic on one GPU), False = two launches."""
        bc = "outflow" if workload == "blast" else "periodic"
        watch.phase("leg %s %s %s" % (arith, riemann, workload))
        native = state["stepper"] == "native" or bool(args.loopback_slabs)
        if native and args.warmup == 0:      # with W >= 1 the warm-up steps do this (and a profile of the run shows the workload's launches only)
            prime(arith, riemann)
        st = make_stepper(arith, riemann, bc, fuse, planar)
        native = not isinstance(st, SlabEulerStepper)
        st.load_slab(initial_state(workload, st.row0, st.row1))
        took_planar = bool(native and st.is_planar())
        if args.precondition > 0:
            # Clock settling, not part of the workload: building and uploading the initial condition leaves the GPU idle for a few hundred
            # ms, and the first ~25 stage launches after an idle period run up to 25 % slower (kernel trace). A scratch grid of
            # this rank's size is stepped right before the W warm-up steps of the real one.
            rows = max(64, min(n, st.row1 - st.row0))
            scratch = NativeSlabStepper((rows, n), dl, gamma, args.theta, riemann, 2, "outflow", device=local_rank, arith=arith, fuse=fuse,
                                        planar=False if planar is False else None)
            scratch.load_slab(setups.blast_ic((rows, n), gamma))
            scratch.step(dt, args.precondition)
            if os.environ.get("MH_BENCH_PRECONDITION_GAP"):      # the first version: the scratch grid is released before the warm-up steps start
                scratch.synchronize()
                scratch.close()
                scratch = None
        st.step(dt, args.warmup)
        if args.precondition > 0 and scratch is not None:
            # released only now: freeing 2 GB synchronises the device and leaves it idle for milliseconds, which is what the preconditioning is there to avoid
            scratch.synchronize()
            if native:
                st.synchronize()
            scratch_to_close = scratch
        else:
            scratch_to_close = None
        if native:
            st.synchronize()
        block_ms = []
        for b in range(nblocks):
            fence()
            t0 = time.perf_counter()
            st.step(dt, args.steps)
            if native:
                st.synchronize()
            fence()
            elapsed = ranks.max_over_ranks(time.perf_counter() - t0)
            block_ms.append(elapsed / args.steps * 1e3)
        if scratch_to_close is not None:
            scratch_to_close.close()
        # HIP events around every bulk stage launch, on the stream it is launched on: a separate short pass, so that the timed region
        # is the un-instrumented product path (without neighbours: one plain launch per fused step, or the graph replay of the two-launch step)
        nprof = 5
        # releasing the scratch grid above idles the GPU for milliseconds, and the first launches after an idle period run slow: `nlead`
        # un-instrumented steps lead straight into the instrumented ones (no synchronisation between them)
        nlead = 40 if native else 0          # (a fused step is ONE launch: 15 steps used to be 30 launches, which covered the ~25 slow ones after an idle period)
        if native:
            st.step(dt, nlead)
            st.profile(True)
        else:
            st.timers = []
        st.step(dt, nprof)
        if native:
            st.synchronize()
        fence()
        if native:
            (avg1, avg2), (nl1, nl2), bulk_rows = st.profile_read()
            st.profile(False)
        else:
            timers, st.timers = st.timers, None
            dur = {1.0: [], 0.5: []}
            for w, e0, e1 in timers:
                dur[w].append(e0.elapsed_time(e1))
            avg1 = sum(dur[1.0]) / max(1, len(dur[1.0]))
            avg2 = sum(dur[0.5]) / max(1, len(dur[0.5]))
            nl1, nl2, bulk_rows = len(dur[1.0]), len(dur[0.5]), st.n0 - 2 * st.edge_rows
        cells_launch = bulk_rows * n
        ms = block_ms[0]
        value = n * n / ms / 1e3
        timing = "HIP events riding on the launches, %d extra steps after the timed region and %d un-instrumented ones that lead into them" % (nprof, nlead)
        if native and nl1 == 0 and nl2 > 0 and (world > 1 or args.loopback_slabs):
            timing = "HIP events around the interior launch of each of %d extra steps after the timed region and %d un-instrumented ones that lead into them" % (nprof, nlead)
        elif native and nl1 == 0 and nl2 > 0:
            # (events riding on each launch of the fused kernel read 3 % long - longer than the timed steps themselves)
            timing = ("one pair of HIP events around the %d launches of %d extra steps (the gaps between the launches included), after the timed region and %d "
                      "un-instrumented steps that lead into them" % (nl2, nprof, nlead))

        counters = bench_report.Counters(csrc_fingerprint())
        whole_grid = world == 1 and not args.loopback_slabs and n == 4096 and workload == "blast"     # what the PMC passes were recorded on

        def stage_roofline(nbytes, avg, nl, name, traffic_key):
            """a stage kernel of the two-launch step: SURVEY.md 8d's algorithmic bytes of that stage over its launch duration against 8 TB/s"""
            traffic = counters.get(traffic_key) if (whole_grid and counters.current) else None
            r = bench_report.hbm_roofline(name, avg, nl, cells_launch, nbytes, traffic=traffic, timing=timing)
            rec = counters.get(traffic_key.replace("_bytes_per_launch", "_fp64"))
            if rec and avg > 0:
                tf = rec["fp64_flops_per_launch"] / bench_report.REF_CELLS * cells_launch / (avg * 1e-3) / 1e12
                r["fp64"] = {"achieved": tf, "peak": FP64_VECTOR_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tf / FP64_VECTOR_PEAK_TFLOPS, "valu_busy": rec["valu_busy"]}
            if traffic:
                r["hbm_frac_measured"] = traffic / (avg * 1e-3) / 1e9 / HBM_PEAK_GBS
            return r

        fused = native and nl1 == 0 and nl2 > 0          # the stepper took the fused step: its one launch is reported in the second-stage slot
        res = {"value": value, "ms_per_step": ms, "status_word": st.status(), "launches_per_step": 1 if fused else 2, "planar_kernel": took_planar and fused}
        if fused:
            # The one-launch RK2 step is fp64-issue-bound (VALU-busy 0.84 - 0.92, profiles/r0x/kernels_headline.md) and moves 72 (planar) or 80 B per
            # cell - the first-stage field never leaves LDS - so its roofline is the fp64 vector peak: FLOP per cell from the recorded
            # SQ_INSTS_VALU_*_F64 counters x the cells of this launch over this run's launch duration. SURVEY.md 8d's 200 B per zone-update is a
            # throughput unit for such a step (roofline_step below), not bytes it moves.
            nominal = 72 if took_planar else 80          # the planar kernel does not read the third momentum (and writes it as zero)
            tag = "fused%s_%s_%s" % ("_planar" if took_planar else "", arith, riemann)
            rec = counters.get(tag + "_fp64")
            name = "euler2d_fused_rk2_kernel<%s%s> (both RK2 stages, one launch per step)" % (riemann, ", planar" if took_planar else "")
            if rec:
                per_cell_bytes = counters.get(tag + "_bytes_per_launch") / bench_report.REF_CELLS if (counters.current and counters.get(tag + "_bytes_per_launch")) else None
                res["roofline"] = bench_report.fp64_roofline(name, avg2, nl2, cells_launch, rec["fp64_flops_per_launch"] / bench_report.REF_CELLS, rec["valu_busy"],
                                                             per_cell_bytes if whole_grid else None, nominal, counters.provenance(), timing=timing)
                if not whole_grid:
                    res["roofline"]["counters"] += "; FLOP per cell and VALU-busy as recorded on the 4096^2 blast"
            else:
                # no counter record for this variant at all: the bytes the launch moves against 8 TB/s (a hardware fraction, below 1)
                res["roofline"] = bench_report.hbm_roofline(name, avg2, nl2, cells_launch, nominal, timing=timing,
                                                            extra={"bytes_moved_per_cell": nominal, "note": "no FLOP record for this kernel variant: bytes the launch moves (reads + writes) over 8 TB/s"})
            res["roofline_stage1"] = None
        else:
            res["roofline"] = stage_roofline(BYTES_STAGE2, avg2, nl2, "euler2d_stage_kernel<%s,%s,PLM,COMBINE> (second RK2 stage)" % (arith, riemann),
                                             "stage2_%s_%s_bytes_per_launch" % (arith, riemann))
            res["roofline_stage1"] = stage_roofline(BYTES_STAGE1, avg1, nl1, "euler2d_stage_kernel<%s,%s,PLM> (first RK2 stage)" % (arith, riemann),
                                                    "stage1_%s_%s_bytes_per_launch" % (arith, riemann))
        res["roofline_step"] = bench_report.step_equivalents(value / world)
        if nblocks > 1:
            rest = sorted(block_ms[1:])
            res["repeat_blocks"] = {"ms_per_step": block_ms[1:], "median_ms_per_step": rest[len(rest) // 2], "min": rest[0], "max": rest[-1],
                                    "note": "%d further timed blocks of %d steps of the same leg, same bracketing" % (nblocks - 1, args.steps)}
        final = None
        if keep_state:
            final = torch.from_numpy(st.slab_host()) if native else st.u[2:2 + st.n0].permute(0, 2, 1).contiguous().cpu()
        if native:
            st.close()
        res["preconditioning"] = ("%d steps of a scratch grid of the same size right before the warm-up steps, released after the timed blocks (clock settling after "
                                  "the idle upload phase; not part of the workload)" % args.precondition if args.precondition > 0 else "none")
        return res, final, args.warmup + nblocks * args.steps + nlead + nprof

    def partition_check(arith, riemann, u_mine, nsteps_total):
        """N > 1 (or --loopback-slabs): the union of the ranks' slabs against the SAME run on one GPU (rank 0 repeats it alone, outside the
        timed region): per-arithmetic results do not depend on the partition, so the slab fingerprints must match bit for bit."""
        if args.loopback_slabs:
            every = [slab_checksum(u_mine[slice(*partition_rows(n, nslabs, r))]) for r in range(nslabs)]
        else:
            mine = torch.tensor(slab_checksum(u_mine), dtype=torch.int64, device="cuda")
            gathered = [torch.zeros_like(mine) for _ in range(world)]
            dist.all_gather(gathered, mine)
            every = [[int(x) for x in g.tolist()] for g in gathered]
        ok = None
        watch.phase("partition check (rank 0 repeats the run on one GPU, the others wait at a barrier)")
        if rank == 0:
            try:                                   # nothing here may keep rank 0 from the barrier below
                one = NativeSlabStepper((n, n), dl, gamma, args.theta, riemann, 2, "outflow", rank=0, world=1, device=local_rank,
                                        chunk_rows=args.chunk_rows, arith=arith)
                one.load_slab(setups.blast_ic((n, n), gamma))
                one.step(dt, nsteps_total)
                one.synchronize()
                whole = torch.from_numpy(one.slab_host())
                one.close()
                ok = True
                for r in range(nslabs):
                    a, b = partition_rows(n, nslabs, r)
                    ok = ok and slab_checksum(whole[a:b]) == every[r]
            except Exception as e:
                print("bench.py: partition check not completed: %r" % (e,), file=sys.stderr)
                ok = None
        if world > 1:
            dist.barrier()
        return ok

    primary = args.arith
    other = "strict" if primary == "fast" else "fast"
    decomposed = world > 1 or bool(args.loopback_slabs)
    res, u_primary, nsteps_primary = run_leg(primary, args.riemann, "blast", nblocks=max(1, args.blocks), keep_state=True, fuse=False if args.no_fuse else None,
                                             planar=False if args.no_planar else None)
    partition_ok = partition_check(primary, args.riemann, u_primary, nsteps_primary) if decomposed and state["stepper"] == "native" else None

    arith_note = {"strict": "strict: bit-identical to the reference CPU path (tests/test_gpu_parity.py, golden vectors from reference headers)",
                  "fast": "fast: FMA + shared reciprocals, conserved-variable L1 <= 1e-12 vs the reference CPU path (tests/test_gpu_parity.py::test_fast_*)"}
    pin_note = {"hlle": "HLLE: pinned to the reference (golden vectors from its own headers)",
                "hllc": "HLLC: parity unpinned - no Euler HLLC exists upstream; GPU == the repo's C restatement bit for bit + exact-Riemann-solver tests"}
    legs, l1 = {}, None
    if not args.single_arith:
        # same workload, other arithmetic (and the L1 distance between the two after the same number of steps)
        r2, u_other, nsteps_other = run_leg(other, args.riemann, "blast", nblocks=max(1, args.blocks), keep_state=True)
        legs["%s_%s_blast" % (other, args.riemann)] = r2
        if nsteps_other == nsteps_primary:
            s = (u_primary - u_other).abs().sum().to("cuda")
            if world > 1:
                dist.all_reduce(s)
            l1 = float(s.item()) / (n * n * 5)
        del u_other
        other_riemann = "hlle" if args.riemann == "hllc" else "hllc"
        # the further variants are one-GPU measurements: a multi-GPU run stays short (each leg would set up its own communicator)
        more = []
        if not decomposed:
            more = [("strict", other_riemann, "blast"), ("fast", other_riemann, "blast"), (primary, args.riemann, "smooth_wave"), ("strict", "hlle", "smooth_wave")]
        for (a, r, w) in more:
            legs["%s_%s_%s" % (a, r, w)] = run_leg(a, r, w)[0]
        if not decomposed and res["launches_per_step"] == 1:
            # the headline took the fused step: the same workload as the two launches it replaces (bit-identical results), with their per-stage rooflines
            legs["%s_%s_blast_two_launches" % (primary, args.riemann)] = run_leg(primary, args.riemann, "blast", fuse=False)[0]
        if not decomposed and res.get("planar_kernel"):
            # ... and took the planar kernel: the same workload on the GENERAL fused kernel (all five components computed; same bits in the other four)
            legs["%s_%s_blast_general_kernel" % (primary, args.riemann)] = run_leg(primary, args.riemann, "blast", planar=False)[0]
        for key, leg in legs.items():
            a, r, w = key.split("_", 2)
            two = w.endswith("_two_launches")
            general = w.endswith("_general_kernel")
            w = w.replace("_two_launches", "").replace("_general_kernel", "")
            leg["note"] = "%s; %s; workload: %s%s" % (arith_note[a], pin_note[r],
                                                      "Sedov-type blast, outflow" if w == "blast" else
                                                      "smooth periodic wave rho = 1 + 0.2 sin(2 pi x) sin(2 pi y), p = rho^gamma, v = (0.5, 0.25) (SURVEY.md §8d)",
                                                      "; the two launches per step that the fused launch replaces" if two else
                                                      ("; the general fused kernel (planar = -1: the third momentum, identically zero in this field, is read, exchanged and computed)" if general else ""))
    del u_primary

    if rank == 0:
        out = {
            "metric": "zone-updates/sec (Mcells/s) whole node, 2D Euler %d^2 PLM+%s RK2" % (n, args.riemann.upper()),
            "value": res["value"], "unit": "Mcells/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": res["ms_per_step"], "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": "2D Euler Sedov-type blast, %dx%d uniform grid, PLM(theta=%g)+%s, RK2, fp64, fixed dt=0.3*dx/6, outflow BC"
                                   % (n, n, args.theta, args.riemann.upper()),
                       "decomposition": (("one GPU: the whole grid as one slab of the native stepper, no cuts, no exchange" if world == 1 else
                                          "axis-0 slabs x%d (nd::partition_shape formula), %s, %s stepper" % (world, "4-row RCCL halo once per step (fused step across the cuts)" if res["launches_per_step"] == 1 else "2-row RCCL halo per stage", state["stepper"])))
                                        if not args.loopback_slabs else
                                        ("REHEARSAL on one GPU: %d slab objects of the native stepper exchanging through its loopback backend" % nslabs),
                       "arith": arith_note[primary], "riemann": pin_note[args.riemann], "status_word": res["status_word"],
                       "planar_kernel": bool(res.get("planar_kernel")), "launches_per_step": res["launches_per_step"],
                       "planar_note": ("the library verified at upload that this 2-D field carries no third momentum and the fused launch skips that component "
                                       "(mh_euler_cart_desc.planar; bit-identical in the other four components, tests/test_gpu_planar.py); the same leg on the general "
                                       "kernel is legs.%s_%s_blast_general_kernel" % (primary, args.riemann)) if res.get("planar_kernel") else "general kernel (all five components computed)",
                       "timed_region": ("one launch per step, issued as it is (a one-node graph replay costs 29 us more per step); two-launch legs: HIP-graph replay of the step" if not decomposed else ("eager two-stream issue, one exchange per step" if res["launches_per_step"] == 1 else "eager two-stream issue, one exchange per stage"))
                                       + (" (ONE fused launch per RK2 step, mara3_amd/csrc/euler2d_fused.hip: results bit-identical to the two launches, tests/test_gpu_fused_rk2.py)"
                                          if res["launches_per_step"] == 1 else " (two launches per RK2 step)"),
                       "preconditioning": res["preconditioning"]},
            "roofline": res["roofline"], "roofline_stage1": res["roofline_stage1"], "roofline_step": res["roofline_step"],
        }
        if out["roofline_stage1"] is None:
            del out["roofline_stage1"]
        if "repeat_blocks" in res:
            out["repeat_blocks"] = res["repeat_blocks"]
        out["stepper"] = state["stepper"]          # native: the library's slab stepper (RCCL from C++); torch: the torch.distributed fallback
        if partition_ok is not None:
            out["slabs_bit_identical_to_one_gpu_run"] = bool(partition_ok)
        if legs:
            out["legs"] = legs
            out["legs_note"] = ("each leg: the same measurement as the headline's (fused legs: plain launches, two-launch legs: graph replay; scratch-grid preconditioning, HIP events on 5 further steps); "
                                "one-launch legs: roofline bound fp64 (FLOP per cell of the recorded PMC passes over this run's launch time against 78.6 TFLOP/s); two-launch legs: "
                                "roofline = second RK2 stage (120 B per cell), roofline_stage1 = first (80 B), both against 8000 GB/s")
        if l1 is not None:
            out["l1_fast_vs_strict_after_%d_steps" % nsteps_primary] = l1
        if world == 1 and not args.loopback_slabs:
            torch.cuda.empty_cache()
            if not args.single_arith and not args.no_extra_configs and n == 4096:
                out["extra_configs"] = extra_configs()
            if not args.no_cpu_baseline:
                out["cpu_baseline"] = cpu_baseline(n, gamma, args.theta, args.riemann)
                ref = cpu_reference(gamma, args.theta)
                if ref:
                    out["cpu_reference"] = ref
        # Everything goes to bench_details.json next to this file (and to stderr); the LAST stdout line is the compact object the driver parses:
        # at most 6000 characters (round 4's 21.7 KB line came back unparsed), the summary once.
        out["summary"] = bench_report.build_summary(out)
        details = bench_report.round_floats(out, 8)
        text = json.dumps(details)
        for path in (os.path.join(ROOT, "bench_details.json"), os.path.join(ROOT, "gpurun_out", "bench_details.json")):
            try:
                if os.path.isdir(os.path.dirname(path)):
                    with open(path, "w") as f:
                        f.write(text + "\n")
            except OSError as e:
                print("bench.py: could not write %s: %r" % (path, e), file=sys.stderr)
        print("bench.py details: " + text, file=sys.stderr, flush=True)
        print(bench_report.final_line(details), flush=True)
    ranks.close()


if __name__ == "__main__":
    main()
