#!/usr/bin/env python3
"""How much of bench.py's ms_per_step is not the kernel: the fused 4096^2 step timed over K = 1 .. 400 steps per call (host clock around
step + synchronize, as bench.py does). A straight line through (K, elapsed) gives the per-step cost and the fixed cost of one timed region.
usage: python scripts/ab_step_overhead.py [--grid 4096] [--reps 5]"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mara3_amd import setups
from mara3_amd.slab import NativeSlabStepper

ap = argparse.ArgumentParser()
ap.add_argument("--grid", type=int, default=4096)
ap.add_argument("--reps", type=int, default=5)
ap.add_argument("--arith", default="fast")
ap.add_argument("--riemann", default="hllc")
ap.add_argument("--no-fuse", action="store_true")
ap.add_argument("--only-ab", action="store_true")
args = ap.parse_args()
n, gamma = args.grid, 5.0 / 3
dt = setups.baseline_dt(n)
st = NativeSlabStepper((n, n), (1.0 / n, 1.0 / n), gamma, 1.5, args.riemann, 2, "outflow", arith=args.arith, fuse=False if args.no_fuse else None)
st.load_slab(setups.blast_ic((n, n), gamma))
st.step(dt, 60); st.synchronize()
pts = []
for K in (() if args.only_ab else (1, 2, 5, 10, 20, 50, 100, 400)):
    best = []
    for r in range(args.reps):
        st.step(dt, 5)                       # keeps the queue warm, as bench.py's warm-up does
        st.synchronize()
        t0 = time.perf_counter()
        st.step(dt, K); st.synchronize()
        best.append((time.perf_counter() - t0) * 1e3)
    best.sort()
    med = best[len(best) // 2]
    pts.append((K, med))
    print(json.dumps({"K": K, "ms_total_median": round(med, 4), "ms_per_step": round(med / K, 4), "all": [round(b, 4) for b in best]}), flush=True)
# the same steps as plain launches instead of graph replays, alternating with the replays
for rnd in range(4):
    line = {"round": rnd}
    for name, graph in (("graph_replay", True), ("plain_launch", False)):
        st.step(dt, 5, graph=graph); st.synchronize()
        t0 = time.perf_counter()
        st.step(dt, 100, graph=graph); st.synchronize()
        line[name + "_ms_per_step"] = round((time.perf_counter() - t0) * 10, 4)
    print(json.dumps(line), flush=True)
if args.only_ab:
    sys.exit(0)
sx = sum(k for k, _ in pts); sy = sum(m for _, m in pts); sxx = sum(k * k for k, _ in pts); sxy = sum(k * m for k, m in pts); m = len(pts)
slope = (m * sxy - sx * sy) / (m * sxx - sx * sx)
print(json.dumps({"per_step_ms": round(slope, 4), "fixed_ms_per_timed_region": round((sy - slope * sx) / m, 4)}))
