#!/usr/bin/env python3
"""Measurements of the BASELINE configs OTHER than the headline one, on one MI355X. `bench.py` keeps
the driver's contract (config C2); this script prints one JSON line per requested config in the same schema:

    python bench_configs.py --config c3        # `binary`  depth=5 block_size=64 fixed_dt=1 rk_order=2  (2048^2)
    python bench_configs.py --config c4        # `cloud`   nr=4096 num_decades=1 rk_order=2 PLM theta=1.2 (compiled host)
    python bench_configs.py --config c5        # 3-D Euler blast, PLM+HLLE RK2, --grid^3 on one GPU (512^3 = one rank's share of 1024^3 / 8)

value = zone-updates/s with the state resident in HBM; roofline = algorithmic bytes of the stage kernel / its average
duration from HIP events on the launch stream; cpu_baseline = the oracle (test infrastructure) timed on a bounded sample.
"""
import argparse
import json
import os
import re
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import bench_report
from bench_report import HBM_PEAK_GBS


def oracle():
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import mara_oracle
    mara_oracle.lib()
    return mara_oracle


def run_c3_bands(args):
    """BASELINE config 3 over N GPUs: the 2048^2 mesh as N bands of whole rows of tree blocks (strong scaling), one process per GPU under
    torch.distributed.run (RCCL), or --loopback-bands N: the bands as objects of this process on one GPU (rehearsal)."""
    import numpy as np
    import torch
    import torch.distributed as dist
    from mara3_amd import binary
    import bench_launch
    ranks = bench_launch.Ranks(args.gpus, args.deadline)
    world, rank, local_rank = ranks.world, ranks.rank, ranks.local_rank
    cfg = binary.config(depth=5, block_size=64, fixed_dt=1, rk_order=2, plm_theta=1.8)
    n = binary.grid_size(cfg)
    nbands = args.loopback_bands or world
    res = {}
    for arith in ("fast", "strict"):
        ranks.watch.phase("c3 %s" % arith)
        if args.loopback_bands:
            s = binary.BinaryBandGroup(cfg, world=nbands, device=local_rank, arith=arith)
        else:
            # the band without a communicator first, then - once every rank holds its band - the process's one communicator
            s, err = None, None
            try:
                s = binary.BinaryBand(cfg, rank, world, None, device=local_rank, arith=arith, comm=None, defer=True)
            except Exception as e:
                err = e
            if not ranks.agree(s is not None):
                raise SystemExit("c3: a rank could not create its band (%r) %s" % (err, bench_launch.NO_RETRY))
            s.attach(ranks.process_comm())
        s.next(args.warmup)
        ranks.fence()
        t0 = time.perf_counter()
        safe = s.next(args.steps)
        ranks.fence()
        elapsed = ranks.max_over_ranks(time.perf_counter() - t0)
        res[arith] = {"value": n * n * args.steps / elapsed / 1e6, "ms_per_step": elapsed / args.steps * 1e3, "safe_mode_steps": safe}
        s.close()
    out = {
        "metric": "zone-updates/sec (Mcells/s), subprog_binary 2048^2 (depth=5 block_size=64), PLM+HLLE+viscosity RK2, %d bands" % nbands,
        "value": res["fast"]["value"], "unit": "Mcells/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": res["fast"]["ms_per_step"], "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": "binary depth=5 block_size=64 focus_factor=1e9 fixed_dt=1 rk_order=2 plm_theta=1.8, other run_config defaults",
                   "decomposition": ("REHEARSAL on one GPU: %d band objects exchanging through the loopback backend" % nbands) if args.loopback_bands
                                    else "one band of whole rows of tree blocks per GPU, 2-row RCCL halo per stage, all-reduce of 2 x 18 totals per step"},
        "roofline": {"bound": "hbm", "achieved": None, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": None, "traffic": None},
        "arith_strict": res["strict"],
    }
    ranks.close()
    return out if rank == 0 else None


def run_c3(args):
    import numpy as np
    from mara3_amd import binary
    if args.loopback_bands or int(os.environ.get("WORLD_SIZE", "1")) > 1:
        return run_c3_bands(args)
    cfg = binary.config(depth=5, block_size=64, fixed_dt=1, rk_order=2, plm_theta=1.8)
    n = binary.grid_size(cfg)
    fast = c3_run(args, binary, cfg, n, "fast")
    fast["arith_strict"] = {k: v for k, v in c3_run(args, binary, cfg, n, "strict").items() if k in ("value", "ms_per_step", "roofline")}
    if not args.no_cpu_baseline:
        fast["cpu_baseline"] = c3_cpu_baseline(binary, cfg)
    return fast


def c3_run(args, binary, cfg, n, arith):
    import numpy as np
    s = binary.BinarySolver(cfg, arith=arith)
    s.next(args.warmup)
    t0 = time.perf_counter()
    safe = s.next(args.steps)
    elapsed = time.perf_counter() - t0
    s.profile(True)
    s.next(5)
    avg_ms, nl = s.profile(False)
    u = s.solution()
    ok = bool(np.isfinite(u).all() and (u[..., 0] > 0).all())
    st = binary.state_as_dict(s.state())
    s.close()
    # 120 B per zone-update, 168 B when `initial_conserved_u` is streamed for the buffer term (it is: 24 B per stage) - the figure
    # `frac` is quoted on. The kernel also streams the per-cell buffer rate (8 B per stage): 184 B, quoted beside it.
    bytes_stage = n * n * 168 / 2                 # mean of the two stage kinds of RK2 (first: 72 B, second: 96 B)
    bytes_stage_all = n * n * (80 + 104) / 2      # ... with the buffer-rate array: 80 B and 104 B
    out = {
        "metric": "zone-updates/sec (Mcells/s), subprog_binary 2048^2 (depth=5 block_size=64), PLM+HLLE+viscosity RK2, 1 GPU",
        "value": n * n * args.steps / elapsed / 1e6, "unit": "Mcells/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64",
        "data": "synthetic",
        "config": {"workload": "binary depth=5 block_size=64 focus_factor=1e9 fixed_dt=1 rk_order=2 plm_theta=1.8, other run_config defaults; arith=" + arith,
                   "safe_mode_steps": safe, "finite_and_positive": ok, "iteration": st["iteration"],
                   "note": "one host synchronisation per step (2 x 18 totals + status word)"},
        "roofline": {"bound": "hbm", "achieved": bytes_stage / (avg_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": bytes_stage / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": None,
                     "kernel": "binary_stage_kernel + binary_sink_kernel + binary_reduce_kernel (one stage)",
                     "algorithmic_bytes_per_launch": bytes_stage, "avg_launch_ms": avg_ms, "launches": nl,
                     "bytes_per_zone_update": 168, "frac_at_184_bytes": bytes_stage_all / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                     "bytes_note": "168 B per zone-update = SURVEY.md 8d's figure with initial_conserved_u streamed; 184 B adds the 8 B per stage of the buffer-rate array the kernel also reads",
                     "timing": "one pair of HIP events on the launch stream around the 10 stage launches of 5 extra steps after the timed region (gaps between the stages included, the end-of-call fetch not)"},
    }
    return out


def c3_cpu_baseline(binary, cfg):
    """the oracle's port of advance_u at the config's FULL size (2048^2, depth=5 block_size=64) on all usable host cores - OpenMP over rows and
    blocks, the role of the reference's tree.map(fn, pool) (core_tree.hpp:615-625); results independent of the thread count
    (tests/test_oracle_binary_golden.py)"""
    from bench import host_cores
    mo = oracle()
    cores = host_cores()
    ocfg = mo.binary_config(depth=5, block_size=64, fixed_dt=1)
    xv = mo.binary_vertices(ocfg)
    u0, br, dt = mo.binary_solver_data(ocfg, xv, xv)
    bodies = binary.two_body_state(binary.initial_elements(cfg), 0.0)
    t0 = time.perf_counter()
    u1, _, _ = mo.binary_advance_u(ocfg, xv, xv, u0, u0, br, bodies, dt, nthreads=cores)
    mo.binary_advance_u(ocfg, xv, xv, u1, u0, br, bodies, dt, nthreads=cores)
    t = time.perf_counter() - t0
    m = len(xv) - 1
    return {"value": m * m / t / 1e6, "unit": "Mcells/s", "cores": cores, "kind": "port",
            "sample": "one RK2 step (two advance_u stages) at the full %dx%d (depth=5 block_size=64), oracle/mara_oracle_binary.c, %d OpenMP threads" % (m, m, cores)}


def run_c4(args):
    exe = os.path.join(ROOT, "mara3_amd", "host", "mara_hip")
    nr = args.grid or 4096
    total = args.warmup + args.steps
    import tempfile
    identical = None
    with tempfile.TemporaryDirectory() as d:
        runs = {}
        for arith in ("fast", "strict"):
            cmd = [exe, "cloud", "nr=%d" % nr, "num_decades=1", "rk_order=2", "reconstruct_method=2", "plm_theta=1.2", "max_steps=%d" % total, "profile=1",
                   "outdir=out_" + arith, "arith=" + arith]
            if args.gpus > 1:
                cmd.append("gpus=%d" % args.gpus)      # radial slabs, one process driving N devices (the reference's evaluate_on<N> thread slabs)
            runs[arith] = subprocess.run(cmd, cwd=d, capture_output=True, text=True, timeout=1200)
            if runs[arith].returncode != 0:
                raise SystemExit(runs[arith].stdout[-2000:] + runs[arith].stderr[-2000:])
        if args.gpus > 1:
            # the decomposition's own check, as bench.py's for the headline: the slabs' final state against the SAME run on one device (FAST:
            # two launches per stage and slab against the one-launch step of the whole field) - bit for bit
            one = subprocess.run([exe, "cloud", "nr=%d" % nr, "num_decades=1", "rk_order=2", "reconstruct_method=2", "plm_theta=1.2", "max_steps=%d" % total,
                                  "outdir=out_one", "arith=fast"], cwd=d, capture_output=True, text=True, timeout=1200)
            try:
                identical = one.returncode == 0 and open(os.path.join(d, "out_one", "final.bin"), "rb").read() == open(os.path.join(d, "out_fast", "final.bin"), "rb").read()
            except OSError:
                identical = None
    res = {}
    for arith, p in runs.items():
        res[arith] = c4_parse(args, p.stdout, nr, arith)
    out = res["fast"]
    out["arith_strict"] = {k: res["strict"][k] for k in ("value", "ms_per_step", "roofline", "roofline_step")}
    if identical is not None:
        out["slabs_bit_identical_to_one_gpu_run"] = bool(identical)
        visible = mara_device_count()
        out["config"]["decomposition"] = ("%d radial slabs (nd::partition_shape), %s, one process driving the devices%s"
                                          % (args.gpus, "four-row halo, one exchange per step (the one-launch RK2 step across the cuts)" if out["config"].get("launches_per_step") == 1 else "two-row halo per stage",
                                             "" if visible >= args.gpus else "; REHEARSAL: %d visible device(s), the slabs share them round-robin (peer copies become device-to-device copies)" % visible))
    out["cpu_baseline"] = None if args.no_cpu_baseline else cloud_cpu_baseline()
    return out


def mara_device_count():
    import mara3_amd
    return mara3_amd.load_library().mh_device_count()


def c4_parse(args, stdout, nr, arith):
    class p:
        pass
    p.stdout = stdout
    kz = [float(x) for x in re.findall(r"kzps=([0-9.]+)", p.stdout)]
    shape = re.search(r"profile: stage kernel avg ([0-9.]+) ms over (\d+) launches \((\d+) per step", p.stdout)
    avg_ms, nl, lps = (float(shape.group(1)), int(shape.group(2)), int(shape.group(3))) if shape else (0.0, 0, 2)
    slab_lps = re.search(r"slab launches per step:((?: \d+)+)", p.stdout)          # gpus > 1: what each radial slab takes (1 = the fused step across its cuts)
    if slab_lps and not shape:
        per_slab = [int(x) for x in slab_lps.group(1).split()]
        lps = max(per_slab)
    planar = bool(re.search(r"step kernels: planar", p.stdout))
    m = re.search(r"write out_%s/final.bin" % arith, p.stdout)
    nq = nr                                          # num_decades=1: nr radial x nr polar zones (subprog_cloud.cpp:233-258)
    vertices = (nr + 1) * (nq + 1)
    ms = [vertices / k for k in kz[args.warmup:]]   # the host prints vertices per ms, like the reference (:858)
    per_step = sum(ms) / len(ms)
    cells = nr * nq
    fused = lps == 1               # the RK2 step as ONE launch (csrc/cloud_fused.hip)
    timing = "one pair of HIP events on the launch stream around the launches of 5 further steps after the run (the gaps between the launches included)"
    if not avg_ms:
        # (gpus > 1: the compiled host takes no per-kernel events on the slabs; the line then carries the step figures only)
        roof = {"bound": "fp64" if fused else "hbm", "achieved": None, "peak": 78.6 if fused else HBM_PEAK_GBS, "unit": "TFLOP/s" if fused else "GB/s", "frac": None, "traffic": None,
                "kernel": ("cloud_fused_rk2_kernel across radial cuts (one launch per step and slab)" if fused else "cloud_stage_kernel (two launches per step and slab)"),
                "launches_per_step": lps}
    elif fused:
        # fp64-issue-bound (VALU-busy 0.85, profiles/r04/cloud_fused.md) and moves 104 (planar) / 120 B per cell: until attach_traffic() finds the
        # recorded FLOP count the roofline is the bytes the launch moves against 8 TB/s - a hardware fraction either way, never the 200 B convention
        moved = 104 if planar else 120
        roof = bench_report.hbm_roofline("cloud_fused_rk2_kernel<%s> (both RK2 stages in one launch per step)" % ("planar" if planar else "general"),
                                         avg_ms, nl, cells, moved, timing=timing,
                                         extra={"bytes_moved_per_cell": moved, "launches_per_step": 1, "note": "bytes the launch moves (reads + writes) over 8 TB/s; no FLOP record applied"})
    else:
        roof = bench_report.hbm_roofline("cloud_stage_kernel<%s%s,PLM> (mean of both RK2 stages)" % (arith, ", planar" if planar else ""), avg_ms, nl, cells, (80 + 120) / 2,
                                         timing=timing, extra={"bytes_moved_per_cell": 88 if planar else 100, "launches_per_step": lps})
    return {
        "metric": "zone-updates/sec (Mcells/s), subprog_cloud %dx%d SRHD PLM+HLLE RK2, %d GPU" % (nr, nq, args.gpus),
        "value": cells / per_step / 1e3, "unit": "Mcells/s", "n_gpus": args.gpus, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": per_step, "higher_is_better": True, "scaling": "weak" if args.gpus == 1 else "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": "mara_hip cloud nr=%d num_decades=1 rk_order=2 reconstruct_method=2 plm_theta=1.2 arith=%s (compiled host; per-step host nozzle evaluation and its 160 KB upload are inside the timed step)" % (nr, arith),
                   "final_state_written": bool(m), "launches_per_step": lps, "planar_kernel": planar,
                   "planar_note": ("the library verified at upload / set_inflow that field and nozzle row carry no azimuthal momentum (upstream's problem never has any) and "
                                   "the kernels skip that component (mh_cloud_desc.planar; FAST: same bits in the other four, tests/test_gpu_cloud_fused.py; STRICT: "
                                   "taken on the bit pattern of +0.0, the reference's bits in all five, tests/test_gpu_cloud_planar.py and the golden steps); "
                                   "`mara_hip cloud ... planar=-1` runs the general kernels") if planar else "general kernels"},
        "roofline": roof,
        "roofline_step": bench_report.step_equivalents(cells / per_step / 1e3 / max(1, args.gpus), target=None),
    }


def cloud_cpu_baseline():
    """the reference's own composition of `cloud` (oracle/_ref/cloud_ref: its headers, lazy arrays, one thread) at nr=256 num_decades=1 - the same
    sub-program options as the timed config at a size one core finishes in seconds; where that binary is absent, the C port on its golden case"""
    import numpy as np
    import tempfile
    exe = os.path.join(ROOT, "oracle", "_ref", "cloud_ref")
    if os.path.exists(exe):
        try:
            from bench import host_cores, upstream_threads
            threads = upstream_threads(host_cores())
            if threads >= 12 and threads < 16:
                threads = 12                         # upstream's own MARA_PREFERRED_THREAD_COUNT
            nr, few, many = 256, 2, 62
            with tempfile.TemporaryDirectory() as d:
                def run(nsteps, th):
                    t0 = time.perf_counter()
                    subprocess.run([exe, str(nr), "1", "2", "2", "1.2", str(nsteps), os.path.join(d, "c"), str(th)], check=True, capture_output=True, timeout=600)
                    return time.perf_counter() - t0
                t = run(many, threads) - run(few, threads)             # set-up, diagnostics and file output cancel
                t1 = run(many, 1) - run(few, 1) if threads > 1 else t
            return {"value": nr * nr * (many - few) / t / 1e6, "unit": "Mcells/s", "cores": threads, "kind": "reference", "one_thread": nr * nr * (many - few) / t1 / 1e6,
                    "sample": "%d RK2 steps of cloud nr=%d num_decades=1 rk_order=2 reconstruct_method=2 plm_theta=1.2 (%d x %d zones), the reference's headers composed as in "
                              "oracle/ref_drivers/cloud_ref.cpp through ITS threaded evaluator mara::evaluate_on<%d>() where upstream pipes `| evaluate` "
                              "(src/subprog_cloud.cpp:525-533, :582); bit-identical to the one-thread run" % (many - few, nr, nr, nr, threads)}
        except Exception:
            pass
    mo = oracle()
    g = np.load(os.path.join(ROOT, "tests", "golden", "cloud_nr32_plm_rk2.npz"))
    rv, qv, u0 = g["rv"], g["qv"], g["u0"]
    dt = float(g["dt"])
    reps = 200
    inflow = np.repeat(g["inflow"][:1], reps, axis=0)      # the nozzle row of the first step, held fixed
    t0 = time.perf_counter()
    mo.cloud_run(u0.copy(), rv, qv, inflow, dt, reps, rk=2, theta=1.2)
    t = time.perf_counter() - t0
    cells = (len(rv) - 1) * (len(qv) - 1)
    return {"value": cells * reps / t / 1e6, "unit": "Mcells/s", "cores": 1, "kind": "port",
            "sample": "%d RK2 steps of the %dx%d golden case (tests/golden/cloud_nr32_plm_rk2.npz), oracle/mara_oracle_srhd.c, 1 thread" % (reps, len(rv) - 1, len(qv) - 1)}


def blast_block(global_shape, start, count, gamma, radius=0.1, p_in=10.0, p_out=0.1):
    """setups.blast_ic restricted to the cells [start, start + count) of the global grid (a block's own initial condition)"""
    import numpy as np
    axes = [(np.arange(s, s + k) + 0.5) / N for s, k, N in zip(start, count, global_shape)]
    X = np.meshgrid(*axes, indexing="ij", sparse=True)
    r2 = sum((x - 0.5) ** 2 for x in X)
    u = np.zeros(tuple(count) + (5,))
    u[..., 0] = 1.0
    u[..., 4] = np.where(r2 < radius * radius, p_in, p_out) / (gamma - 1.0)
    return u


def run_c5_blocks(args):
    """BASELINE config 5 as configured: the 3-D blast under the (B0, B1, B2) block decomposition of propose_block_decomposition<3>(N),
    n^3 cells PER RANK (weak scaling: 1024^3 on 8 GPUs at n = 512), ghost exchange on every cut side. One process per GPU under
    torch.distributed.run (RCCL), or --loopback-blocks N: the N blocks as objects of this process on one GPU (rehearsal)."""
    import numpy as np
    import torch
    import torch.distributed as dist
    from mara3_amd import setups
    from mara3_amd.block import NativeBlock, NativeBlockGroup, block_layout
    import bench_launch
    ranks = bench_launch.Ranks(args.gpus, args.deadline)
    world, rank, local_rank = ranks.world, ranks.rank, ranks.local_rank
    nblocks = args.loopback_blocks or world
    n = args.grid or 512
    gamma = 5.0 / 3
    B = block_layout((n, n, n), nblocks, 0)[0]
    shape = tuple(n * b for b in B)
    dl = tuple(1.0 / max(shape) for _ in shape)          # cubic cells; the domain is [0, N_a / max N] per axis
    dt = setups.baseline_dt(max(shape))
    fence = ranks.fence
    res = {}
    for arith in ("fast", "strict"):
        ranks.watch.phase("c5 %s" % arith)
        if args.loopback_blocks:
            st = NativeBlockGroup(shape, dl, gamma, 1.5, args.riemann, 2, "outflow", world=nblocks, device=local_rank, arith=arith)
            st.upload(blast_block(shape, (0, 0, 0), shape, gamma))
            probe = st.members[nblocks // 2]
        else:
            st, err = None, None
            try:
                st = NativeBlock(shape, dl, gamma, 1.5, args.riemann, 2, "outflow", rank=rank, world=world, comm_id=None, device=local_rank, arith=arith)
            except Exception as e:      # e.g. hipMalloc of a 512^3 block: said on every rank before any of them enters a collective
                err = e
            if not ranks.agree(st is not None):
                if st is not None:
                    st.close()
                raise SystemExit("c5: a rank could not create its block (%r) %s" % (err, bench_launch.NO_RETRY))
            if world > 1:
                st.use_comm(ranks.process_comm())
            st.upload(blast_block(shape, st.start, st.count, gamma))
            probe = st
        st.step(dt, args.warmup)
        st.synchronize()
        fence()
        t0 = time.perf_counter()
        st.step(dt, args.steps)
        st.synchronize()
        fence()
        elapsed = ranks.max_over_ranks(time.perf_counter() - t0)
        probe.profile(True)
        st.step(dt, 3)
        st.synchronize()
        (ms1, ms2), (nl1, nl2), cells = probe.profile(False)
        status = st.status()[0]
        ncell = shape[0] * shape[1] * shape[2]
        avg_ms = 0.5 * (ms1 + ms2)
        bytes_stage = cells * (80 + 120) / 2
        res[arith] = {"value": ncell * args.steps / elapsed / 1e6, "ms_per_step": elapsed / args.steps * 1e3, "status_word": status,
                      "roofline": {"bound": "hbm", "achieved": bytes_stage / (avg_ms * 1e-3) / 1e9 if avg_ms else None, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                   "frac": bytes_stage / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if avg_ms else None, "traffic": None,
                                   "kernel": "euler3d_stage_kernel<%s,%s,PLM> interior launch of one block (mean of both RK2 stages)" % (arith, args.riemann),
                                   "algorithmic_bytes_per_launch": bytes_stage, "avg_launch_ms": avg_ms, "launches": nl1 + nl2,
                                   "timing": "HIP events on the launch stream, 3 extra steps after the timed region"},
                      "messages": {"neighbours": probe.neighbours, "doubles_per_axis": probe.message_doubles}}
        st.close()
    identical = None
    if args.loopback_blocks and shape[0] * shape[1] * shape[2] <= 384 ** 3:
        # the decomposition's own check (as bench.py's for the headline): the union of the blocks against the undivided grid after the same steps
        from mara3_amd.engine import EulerCartSolver
        nsteps = 3
        grp = NativeBlockGroup(shape, dl, gamma, 1.5, args.riemann, 2, "outflow", world=nblocks, device=local_rank, arith="fast")
        u0 = blast_block(shape, (0, 0, 0), shape, gamma)
        grp.upload(u0)
        grp.step(dt, nsteps)
        grp.synchronize()
        got = grp.download()
        grp.close()
        one = EulerCartSolver(shape, dl, gamma, 1.5, args.riemann, 2, "outflow", arith="fast")
        one.upload(u0)
        one.step(dt, nsteps)
        identical = bool(np.array_equal(got.view(np.uint64), one.download().view(np.uint64)))
        one.close()
    out = {
        "metric": "zone-updates/sec (Mcells/s), 3D Euler blast %dx%dx%d PLM+%s RK2, (%d,%d,%d) blocks" % (shape + (args.riemann.upper(),) + tuple(B)),
        "value": res["fast"]["value"], "unit": "Mcells/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": res["fast"]["ms_per_step"], "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": "3D Euler blast (radius 0.1), %d^3 cells per block, (%d,%d,%d) blocks of propose_block_decomposition<3>(%d), PLM(theta=1.5)+%s, RK2, fixed dt=0.3*dx/6"
                               % ((n,) + tuple(B) + (nblocks, args.riemann.upper())),
                   "decomposition": ("REHEARSAL on one GPU: %d block objects exchanging through the loopback backend" % nblocks) if args.loopback_blocks
                                    else "one block per GPU, ghost exchange on every cut side as one RCCL group per stage",
                   "arith": "fast (headline of this line); strict beside it", "status_word": res["fast"]["status_word"]},
        "roofline": res["fast"]["roofline"], "messages": res["fast"]["messages"], "arith_strict": res["strict"],
    }
    if identical is not None:
        out["blocks_bit_identical_to_one_gpu_run"] = identical
    ranks.close()
    return out if rank == 0 else None


def run_c5(args):
    import numpy as np
    import mara3_amd
    from mara3_amd import setups
    from mara3_amd.engine import EulerCartSolver
    if args.loopback_blocks or int(os.environ.get("WORLD_SIZE", "1")) > 1:
        return run_c5_blocks(args)
    n = args.grid or 512
    gamma = 5.0 / 3
    res = {}
    for arith in ("fast", "strict"):
        s = EulerCartSolver((n, n, n), (1.0 / n,) * 3, gamma, 1.5, args.riemann, 2, "outflow", arith=arith)
        s.upload(setups.blast_ic((n, n, n), gamma))
        dt = setups.baseline_dt(n)
        s.step(dt, args.warmup)
        s.synchronize()
        s.profile(True)
        t0 = time.perf_counter()
        s.step(dt, args.steps)
        s.synchronize()
        elapsed = time.perf_counter() - t0
        avg_ms, nl = s.profile_read()
        s.profile(False)
        status = s.status()
        s.close()
        bytes_stage = n ** 3 * (80 + 120) / 2
        res[arith] = {"value": n ** 3 * args.steps / elapsed / 1e6, "ms_per_step": elapsed / args.steps * 1e3, "status_word": status,
                      "roofline": {"bound": "hbm", "achieved": bytes_stage / (avg_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                   "frac": bytes_stage / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": None,
                                   "kernel": "euler3d_stage_kernel<%s,%s,PLM> (mean of both RK2 stages)" % (arith, args.riemann),
                                   "algorithmic_bytes_per_launch": bytes_stage, "avg_launch_ms": avg_ms, "launches": nl,
                                   "timing": "one pair of HIP events on the launch stream around the stage launches of the timed call, inside the timed region"}}
    out = {
        "metric": "zone-updates/sec (Mcells/s), 3D Euler blast %d^3 PLM+%s RK2, 1 GPU" % (n, args.riemann.upper()),
        "value": res["fast"]["value"], "unit": "Mcells/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": res["fast"]["ms_per_step"], "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": "3D Euler blast (radius 0.1), %d^3 uniform grid = one rank's share of the 1024^3 / 8-GPU case when 512, PLM(theta=1.5)+%s, RK2, fixed dt=0.3*dx/6" % (n, args.riemann.upper()),
                   "arith": "fast (headline of this line); strict beside it", "status_word": res["fast"]["status_word"]},
        "roofline": res["fast"]["roofline"], "arith_strict": res["strict"],
    }
    if not args.no_cpu_baseline:
        mo = oracle()
        m, cores = 96, min(16, os.cpu_count() or 1)
        u = setups.blast_ic((m, m, m), gamma)
        kind = mo.RIEMANN_HLLC if args.riemann == "hllc" else mo.RIEMANN_HLLE
        t0 = time.perf_counter()
        mo.euler_cart_run(u, (1.0 / m,) * 3, setups.baseline_dt(m), 4, gamma, 1.5, 2, kind, mo.BC_OUTFLOW, nthreads=cores)
        t = time.perf_counter() - t0
        out["cpu_baseline"] = {"value": m ** 3 * 4 / t / 1e6, "unit": "Mcells/s", "cores": cores, "kind": "port",
                               "sample": "4 RK2 steps at %d^3, oracle/mara_oracle.c with %d slab threads" % (m, cores)}
        ref = c5_cpu_reference(m, gamma)
        if ref:
            out["cpu_reference"] = ref
    return out


def c5_cpu_reference(m, gamma):
    """the reference's own lazy-array composition of the 3-D step through its threaded evaluator (oracle/_ref/euler_cart_ref, as bench.py's
    cpu_reference does for the 2-D headline)"""
    import tempfile
    from bench import host_cores, upstream_threads
    from mara3_amd import setups
    exe = os.path.join(ROOT, "oracle", "_ref", "euler_cart_ref")
    if not os.path.exists(exe):
        return None
    threads = upstream_threads(host_cores())
    steps = 2
    try:
        with tempfile.TemporaryDirectory() as d:
            fin, fout = os.path.join(d, "in"), os.path.join(d, "out")
            setups.blast_ic((m, m, m), gamma).tofile(fin)
            hx = lambda x: float(x).hex()
            args = [exe, "3", str(m), str(m), str(m), hx(gamma), hx(1.5), "2", "0", hx(setups.baseline_dt(m)), hx(1.0 / m), hx(1.0 / m), hx(1.0 / m)]

            def seconds(nsteps):
                t0 = time.perf_counter()
                subprocess.check_call(args + [str(nsteps), fin, fout, str(threads)])
                return time.perf_counter() - t0
            t = seconds(steps) - seconds(0)
    except Exception:
        return None
    return {"value": m ** 3 * steps / t / 1e6, "unit": "Mcells/s", "cores": threads, "kind": "reference",
            "sample": "%d RK2 steps at %d^3 PLM+HLLE, the reference's headers composed and evaluated as its own `advance` is (mara::evaluate_on<%d>()), "
                      "oracle/ref_drivers/euler_cart_ref.cpp" % (steps, m, threads)}


def attach_traffic(out, config):
    """From the committed rocprofv3 PMC passes (profiles/pmc_traffic.json, stamped with the hash of the kernel sources): HBM bytes per launch of the
    stage kernels (mean of the two RK2 stage kinds, like `achieved`; only for the build they were measured on) and the fp64 side - FLOP per launch
    from SQ_INSTS_VALU_*_F64 over this run's average launch duration against the 78.6 TF vector peak, with the recorded VALU-busy fraction. The
    one-launch `cloud` step is fp64-issue-bound and gets that as its roofline (bound "fp64"); the two-launch configs keep bound "hbm" at their
    algorithmic stage bytes with the fp64 figure beside it."""
    try:
        from bench import csrc_fingerprint
        counters = bench_report.Counters(csrc_fingerprint())
        t = counters.table
        names = {"c3": ("c3", "binary_stage_kernel<Bin%s, false, false>", "binary_stage_kernel<Bin%s, true, false>"),          # (Fast: BinFastT<true>, the default disk's instantiation)
                 "c4": ("c4", "cloud_stage_kernel<Srhd%sT<@>, true, false>", "cloud_stage_kernel<Srhd%sT<@>, true, true>"),
                 "c5": ("c5", "euler3d_stage_kernel<%sArithT<false>, 0, true, false>", "euler3d_stage_kernel<%sArithT<false>, 0, true, true>")}[config]
        for mode, roof in (("Fast", out.get("roofline")), ("Strict", (out.get("arith_strict") or {}).get("roofline"))):
            if not roof or not roof.get("avg_launch_ms"):
                continue
            tag = "c4s" if (config == "c4" and mode == "Strict") else names[0]
            keys = ["%s:%s" % (tag, (n % ("FastT<true>" if (config == "c3" and mode == "Fast") else mode)).replace("@", "true" if "planar" in roof.get("kernel", "") else "false")) for n in names[1:]]
            if config == "c4" and mode == "Fast" and roof.get("launches_per_step") == 1:
                keys = ["c4:cloud_fused_rk2_kernel<%s>" % ("true" if "planar" in roof.get("kernel", "") else "false")]           # the RK2 step's one launch
            if counters.current and all(k in t for k in keys):
                roof["traffic"] = sum(t[k] for k in keys) / len(keys)
            fk = [k + ":fp64" for k in keys]
            if not all(k in t for k in fk):
                continue
            flops = sum(t[k]["fp64_flops_per_launch"] for k in fk) / len(fk)
            busy = sum(t[k]["valu_busy"] for k in fk) / len(fk)
            if roof.get("launches_per_step") == 1 and config == "c4":
                cells = 4096 * 4096
                fresh = bench_report.fp64_roofline(roof["kernel"], roof["avg_launch_ms"], roof["launches"], cells, flops / cells, busy,
                                                   roof["traffic"] / cells if roof.get("traffic") else None, roof["bytes_moved_per_cell"],
                                                   counters.provenance(), timing=roof.get("timing"))
                fresh["launches_per_step"] = 1
                roof.clear()
                roof.update(fresh)
                continue
            tf = flops / (roof["avg_launch_ms"] * 1e-3) / 1e12
            roof["fp64"] = {"achieved": tf, "peak": 78.6, "unit": "TFLOP/s", "frac": tf / 78.6, "valu_busy": busy, "counters": counters.provenance()}
            if roof.get("traffic"):
                roof["hbm_frac_measured"] = roof["traffic"] / (roof["avg_launch_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS
    except Exception as e:
        print("bench_configs.py: counters not attached: %r" % (e,), file=sys.stderr)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", required=True, choices=["c3", "c4", "c5"])
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--grid", type=int, default=0)
    ap.add_argument("--riemann", default="hlle", choices=["hlle", "hllc"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--loopback-bands", type=int, default=0, help="c3: the mesh as this many bands (objects of one process on one GPU)")
    ap.add_argument("--loopback-blocks", type=int, default=0, help="c5: this many blocks (grid^3 cells each) as objects of one process on one GPU")
    ap.add_argument("--gpus", type=int, default=1,
                    help="c3 / c5: one process per GPU (this process starts them, bench_launch.py); c4: the compiled host's gpus=N (one process, N devices)")
    ap.add_argument("--dry-launch", action="store_true", help="c3 / c5 with --gpus N > 1: print the multi-rank child's command line and stop")
    ap.add_argument("--launch-timeout", type=int, default=900)
    ap.add_argument("--deadline", type=int, default=600, help="N > 1: seconds after which a rank that is still waiting leaves with status 3")
    args = ap.parse_args()
    if args.gpus > 1 and args.config in ("c3", "c5") and "WORLD_SIZE" not in os.environ:
        import bench_launch
        bench_launch.supervise(__file__, sys.argv[1:], args.gpus, timeout_s=args.launch_timeout, dry=args.dry_launch)
    if args.dry_launch:
        raise SystemExit("--dry-launch is for --config c3|c5 --gpus N > 1 without torch.distributed.run in front")
    if args.config in ("c3", "c5") and int(os.environ.get("WORLD_SIZE", "1")) != args.gpus:
        raise SystemExit("--gpus %d does not match WORLD_SIZE %s" % (args.gpus, os.environ.get("WORLD_SIZE", "1")))
    import mara3_amd
    lib = mara3_amd.load_library()
    if lib.mh_device_count() < 1:
        raise SystemExit("bench_configs.py needs an MI355X; there is no CPU path")
    out = {"c3": run_c3, "c4": run_c4, "c5": run_c5}[args.config](args)
    if out is None:          # not rank 0 of a multi-process run
        return
    if not (args.loopback_bands or args.loopback_blocks or args.grid):
        out = attach_traffic(out, args.config)
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
