#!/usr/bin/env python3
"""bench.py — zone-updates/s of the 2-D Euler PLM+HLLC RK2 sweep at 4096^2 (BASELINE.json configs[1]).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--riemann hllc|hlle] [--grid 4096]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
           bench.py --gpus N --steps K --warmup W

One process per GPU. The global 4096^2 grid is cut into axis-0 slabs with the reference's partition formula
(strong scaling: total work fixed), ghost rows travel as RCCL send/recv. A "step" is one full RK2 time step
of the whole grid; inputs are resident in HBM before the timed region. Rank 0 prints ONE JSON line with the
contract keys plus `roofline` (dominant kernel: the second, combining RK2 stage) and, at N=1, `cpu_baseline`
(the parity-pinned plain-C restatement of the reference's thread-slab CPU path, timed on the host cores).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s peak
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


def cpu_baseline(n, gamma, theta, riemann, budget_s=15.0):
    """Time the oracle (test infrastructure, used here ONLY as the reported CPU baseline) on the same workload."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import numpy as np
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
    out = {"value": n * n * steps / t / 1e6, "unit": "Mcells/s", "cores": cores, "kind": "port",
           "sample": "%d RK2 steps of the same %dx%d PLM+%s workload, oracle/mara_oracle.c with %d slab threads"
                     % (steps, n, n, riemann.upper(), cores)}
    return out


def cpu_reference(gamma, theta):
    """If the reference-composed driver was prebuilt (oracle/_ref), time Mara3's own lazy-array path (1 thread)."""
    import subprocess
    import tempfile
    import numpy as np
    from mara3_amd import setups
    exe = os.path.join(ROOT, "oracle", "_ref", "euler_cart_ref")
    if not os.path.exists(exe):
        return None
    n, steps = 512, 4
    u = setups.blast_ic((n, n), gamma)
    with tempfile.TemporaryDirectory() as d:
        fin, fout = os.path.join(d, "in"), os.path.join(d, "out")
        u.tofile(fin)
        hx = lambda x: float(x).hex()
        args = [exe, "2", str(n), str(n), "1", hx(gamma), hx(theta), "2", "0", hx(setups.baseline_dt(n)),
                hx(1.0 / n), hx(1.0 / n), hx(1.0)]
        try:
            t0 = time.perf_counter()
            subprocess.check_call(args + ["0", fin, fout])
            tz = time.perf_counter() - t0
            t0 = time.perf_counter()
            subprocess.check_call(args + [str(steps), fin, fout])
            t = time.perf_counter() - t0 - tz
        except Exception:
            return None
    return {"value": n * n * steps / t / 1e6, "unit": "Mcells/s", "cores": 1, "kind": "reference",
            "sample": "%d RK2 steps at %dx%d PLM+HLLE, reference headers composed as in oracle/ref_drivers/euler_cart_ref.cpp" % (steps, n, n)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--grid", dest="n", type=int, default=4096, help="cells per axis of the global grid")
    ap.add_argument("--riemann", default="hllc", choices=["hllc", "hlle"])
    ap.add_argument("--theta", type=float, default=1.5)
    ap.add_argument("--chunk-rows", type=int, default=0)
    ap.add_argument("--arith", default="fast", choices=["strict", "fast"], help="arithmetic contract of the headline value")
    ap.add_argument("--stepper", default="native", choices=["native", "torch"],
                    help="native: C++ slab stepper of libmara_hip.so (RCCL called from the library); torch: Python stepper over torch.distributed")
    ap.add_argument("--single-arith", action="store_true", help="do not also time the other arithmetic mode")
    ap.add_argument("--no-overlap", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    import mara3_amd
    from mara3_amd import setups
    from mara3_amd.slab import SlabEulerStepper

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if "MARA_BENCH_FORCE_DEVICE" in os.environ:      # rehearsal of the N>1 code path on a one-GPU box
        local_rank = int(os.environ["MARA_BENCH_FORCE_DEVICE"])
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with: python -m torch.distributed.run --nnodes=1 --nproc-per-node %d ... bench.py --gpus %d" % (args.gpus, args.gpus))
        raise SystemExit("--gpus %d does not match WORLD_SIZE %d" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X; there is no CPU path")
    mara3_amd.load_library()           # fails loudly if the HIP library is missing
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    n, gamma = args.n, 5.0 / 3
    dl = (1.0 / n, 1.0 / n)
    dt = setups.baseline_dt(n)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    from mara3_amd.slab import NativeSlabStepper, native_comm_id

    state = {"stepper": args.stepper}

    def make_stepper(arith):
        if state["stepper"] == "native":
            st, err = None, None
            try:
                # an RCCL unique id is good for one communicator: a fresh one per stepper, broadcast from rank 0
                comm_id = native_comm_id(rank, world, device="cuda") if world > 1 else None
                st = NativeSlabStepper((n, n), dl, gamma, args.theta, args.riemann, 2, "outflow", rank=rank, world=world,
                                       comm_id=comm_id, device=local_rank, chunk_rows=args.chunk_rows, arith=arith)
            except mara3_amd.MaraHipError as e:
                err = e
            ok = torch.tensor([0 if st is None else 1], device="cuda")
            if world > 1:
                dist.all_reduce(ok, op=dist.ReduceOp.MIN)          # every rank must take the same path
            if int(ok.item()) == 1:
                return st
            if world == 1:
                raise err
            if rank == 0:
                print("bench.py: native slab stepper unavailable (%s); using the torch.distributed stepper" % err, file=sys.stderr)
            if st is not None:
                st.close()
            state["stepper"] = "torch"
        return SlabEulerStepper((n, n), dl, gamma, args.theta, args.riemann, 2, "outflow", rank=rank, world=world,
                                device="cuda", overlap=not args.no_overlap, chunk_rows=args.chunk_rows, arith=arith)

    def prime(arith):
        """One step of a 128 x 128 throw-away problem: first-use costs of the library (code-object load, stream and event creation) are
        initialisation, not part of a step of the workload - they must not land in the timed region when the caller asks for W = 0."""
        tiny = NativeSlabStepper((128, 128), (1.0 / 128, 1.0 / 128), gamma, args.theta, args.riemann, 2, "outflow", device=local_rank, arith=arith)
        tiny.load_slab(setups.blast_ic((128, 128), gamma))
        tiny.step(setups.baseline_dt(128), 1)
        tiny.synchronize()
        tiny.close()

    def run_mode(arith):
        """W untimed + K timed steps of the whole (slab-decomposed) grid in one arithmetic mode."""
        if state["stepper"] == "native" and args.warmup == 0:      # with W >= 1 the warm-up steps do this (and a profile of the run shows the workload's launches only)
            prime(arith)
        st = make_stepper(arith)
        st.load_slab(setups.blast_ic((n, n), gamma, row_range=(st.row0, st.row1)))
        native = isinstance(st, NativeSlabStepper)
        st.step(dt, args.warmup)
        if native:
            st.synchronize()
        fence()
        # HIP events around every bulk stage launch, on the stream it is launched on. At N = 1 they are recorded in
        # the timed region itself; at N > 1 in a short extra pass, so that the timed region carries no event traffic.
        live = world == 1
        if live:
            if native:
                st.profile(True)
            else:
                st.timers = []
        t0 = time.perf_counter()
        st.step(dt, args.steps)
        if native:
            st.synchronize()
        fence()
        elapsed = time.perf_counter() - t0
        if not live:
            if native:
                st.profile(True)
            else:
                st.timers = []
            st.step(dt, 3)
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
        if world > 1:
            t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        cells_launch = bulk_rows * n
        value = n * n * args.steps / elapsed / 1e6
        res = {
            "value": value, "ms_per_step": elapsed / args.steps * 1e3, "status_word": st.status(),
            "roofline": {"bound": "hbm", "achieved": cells_launch * BYTES_STAGE2 / (avg2 * 1e-3) / 1e9, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": cells_launch * BYTES_STAGE2 / (avg2 * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": None,
                         "kernel": "euler2d_stage_kernel<%s,%s,PLM,COMBINE> (second RK2 stage)" % (arith, args.riemann),
                         "algorithmic_bytes_per_launch": cells_launch * BYTES_STAGE2, "avg_launch_ms": avg2, "launches": nl2,
                         "timing": "HIP events on the launch stream, %s" % ("inside the timed region" if live else "3 extra steps after the timed region")},
            "roofline_stage1": {"achieved": cells_launch * BYTES_STAGE1 / (avg1 * 1e-3) / 1e9,
                                "frac": cells_launch * BYTES_STAGE1 / (avg1 * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                "algorithmic_bytes_per_launch": cells_launch * BYTES_STAGE1, "avg_launch_ms": avg1, "launches": nl1},
            "roofline_step": {"achieved": value * 1e6 * BYTES_STEP / 1e9 / world, "frac": value * 1e6 * BYTES_STEP / 1e9 / world / HBM_PEAK_GBS,
                              "note": "per GPU, 200 B per zone-update over the whole timed step (launch gaps and halo exchange included)"},
        }
        final = torch.from_numpy(st.slab_host()) if native else st.u[2:2 + st.n0].permute(0, 2, 1).contiguous().cpu()
        if native:
            st.close()
        return res, final

    from mara3_amd.slab import slab_fingerprint as slab_checksum

    def partition_check(arith, u_slab):
        """N > 1: the union of the ranks' slabs against the SAME run on one GPU (rank 0 repeats it alone, outside the timed region):
        per-arithmetic results do not depend on the partition, so the slab fingerprints must match bit for bit."""
        nsteps_total = args.warmup + args.steps + 3
        mine = torch.tensor(slab_checksum(u_slab), dtype=torch.int64, device="cuda")
        every = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)
        ok = None
        if rank == 0:
            try:                                   # nothing here may keep rank 0 from the barrier below
                from mara3_amd.slab import partition_rows
                one = NativeSlabStepper((n, n), dl, gamma, args.theta, args.riemann, 2, "outflow", rank=0, world=1, device=local_rank,
                                        chunk_rows=args.chunk_rows, arith=arith)
                one.load_slab(setups.blast_ic((n, n), gamma))
                one.step(dt, nsteps_total)
                one.synchronize()
                whole = torch.from_numpy(one.slab_host())
                one.close()
                ok = True
                for r in range(world):
                    a, b = partition_rows(n, world, r)
                    ok = ok and slab_checksum(whole[a:b]) == [int(x) for x in every[r].tolist()]
            except Exception as e:
                print("bench.py: partition check not completed: %r" % (e,), file=sys.stderr)
                ok = None
        dist.barrier()
        return ok

    primary = args.arith
    other = "strict" if primary == "fast" else "fast"
    res, u_primary = run_mode(primary)
    partition_ok = partition_check(primary, u_primary) if world > 1 and state["stepper"] == "native" else None
    res_other, u_other = (None, None) if args.single_arith else run_mode(other)
    l1 = None
    if u_other is not None:
        s = (u_primary - u_other).abs().sum().to("cuda")
        if world > 1:
            dist.all_reduce(s)
        l1 = float(s.item()) / (n * n * 5)
    del u_primary, u_other

    if rank == 0:
        pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(pmc) and world == 1 and n == 4096:
            try:
                table = json.load(open(pmc))
                res["roofline"]["traffic"] = table.get("stage2_%s_%s_bytes_per_launch" % (primary, args.riemann))
                if res_other:
                    res_other["roofline"]["traffic"] = table.get("stage2_%s_%s_bytes_per_launch" % (other, args.riemann))
            except Exception:
                pass
        arith_note = {"strict": "strict: bit-identical to the reference CPU path (tests/test_gpu_parity.py, golden vectors from reference headers)",
                      "fast": "fast: FMA + shared reciprocals, conserved-variable L1 <= 1e-12 vs the reference CPU path (tests/test_gpu_parity.py::test_fast_*)"}
        out = {
            "metric": "zone-updates/sec (Mcells/s) whole node, 2D Euler %d^2 PLM+%s RK2" % (n, args.riemann.upper()),
            "value": res["value"], "unit": "Mcells/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": res["ms_per_step"], "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": "2D Euler Sedov-type blast, %dx%d uniform grid, PLM(theta=%g)+%s, RK2, fp64, fixed dt=0.3*dx/6, outflow BC"
                                   % (n, n, args.theta, args.riemann.upper()),
                       "decomposition": "axis-0 slabs x%d (nd::partition_shape formula), 2-row RCCL halo per stage, %s stepper" % (world, state["stepper"]),
                       "arith": arith_note[primary], "status_word": res["status_word"]},
            "roofline": res["roofline"], "roofline_stage1": res["roofline_stage1"], "roofline_step": res["roofline_step"],
        }
        if partition_ok is not None:
            out["slabs_bit_identical_to_one_gpu_run"] = bool(partition_ok)
        if res_other:
            out["arith_" + other] = {"note": arith_note[other], "value": res_other["value"], "ms_per_step": res_other["ms_per_step"],
                                     "roofline": res_other["roofline"], "roofline_step": res_other["roofline_step"],
                                     "status_word": res_other["status_word"]}
            out["l1_fast_vs_strict_after_%d_steps" % (args.steps + args.warmup)] = l1
        if world == 1 and not args.no_cpu_baseline:
            torch.cuda.empty_cache()
            out["cpu_baseline"] = cpu_baseline(n, gamma, args.theta, args.riemann)
            ref = cpu_reference(gamma, args.theta)
            if ref:
                out["cpu_reference"] = ref
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
