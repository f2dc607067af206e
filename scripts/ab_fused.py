#!/usr/bin/env python3
"""A/B on one GPU, alternating in one process: the 4096^2 RK2 step (FAST) as two launches and as the fused launch (euler2d_fused.hip),
chunk lengths of the fused launch. One JSON line per round: ms per step of every variant.
usage: python scripts/ab_fused.py [--grid 4096] [--riemann hllc] [--workload blast|smooth_wave] [--rounds 4] [--chunks 32,48,64,96,128]"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mara3_amd import setups
from mara3_amd.slab import NativeSlabStepper

ap = argparse.ArgumentParser()
ap.add_argument("--grid", type=int, default=4096)
ap.add_argument("--riemann", default="hllc")
ap.add_argument("--workload", default="blast")
ap.add_argument("--rounds", type=int, default=4)
ap.add_argument("--steps", type=int, default=100)
ap.add_argument("--chunks", default="32,48,64,96,128")
args = ap.parse_args()
n, gamma = args.grid, 5.0 / 3
dl = (1.0 / n, 1.0 / n)
dt = setups.baseline_dt(n)
bc = "outflow" if args.workload == "blast" else "periodic"
u0 = setups.blast_ic((n, n), gamma) if args.workload == "blast" else setups.smooth_wave_ic((n, n), gamma)
variants = [("two_launches", dict(fuse=False))] + [("fused_c%s" % c, dict(fuse=True, chunk_rows=int(c))) for c in args.chunks.split(",")]
steppers = {}
for name, kw in variants:
    st = NativeSlabStepper((n, n), dl, gamma, 1.5, args.riemann, 2, bc, arith="fast", **kw)
    st.load_slab(u0)
    st.step(dt, 30); st.synchronize()
    steppers[name] = st
for r in range(args.rounds):
    line = {"round": r, "grid": n, "riemann": args.riemann, "workload": args.workload}
    for name, _ in variants:
        st = steppers[name]
        st.step(dt, 10); st.synchronize()
        t0 = time.perf_counter()
        st.step(dt, args.steps); st.synchronize()
        line[name] = round((time.perf_counter() - t0) / args.steps * 1e3, 4)
    print(json.dumps(line), flush=True)
import numpy as np
a = steppers["two_launches"].slab_host()
for name, _ in variants[1:]:
    b = steppers[name].slab_host()
    print(json.dumps({"variant": name, "bit_identical_to_two_launches": bool(np.array_equal(a.view(np.uint64), b.view(np.uint64))), "status": steppers[name].status()}), flush=True)
