#!/usr/bin/env python3
"""BASELINE config 5 at its STATED global size on one MI355X: the 1024^3 3-D Euler blast as the (2,2,2) blocks of 512^3 of
propose_block_decomposition<3>(8), all eight as objects of one process exchanging ghost cells through the loopback backend (178 GB of the
288 GB of HBM). Not a bench line (eight GPUs' work on one); it answers "was the stated grid ever run": steps it, times it, and checks what
can be checked at this size - status word, positivity, conservation of mass / momentum / energy against the initial sums (nothing has
reached the outflow boundary), mirror symmetry of the density about the three mid-planes. One JSON line.
usage: python scripts/c5_fullsize_one_gpu.py [--n 512] [--steps 4] [--arith fast]"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mara3_amd import setups
from mara3_amd.block import NativeBlockGroup, block_layout
from bench_configs import blast_block

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=512)
ap.add_argument("--steps", type=int, default=4)
ap.add_argument("--arith", default="fast")
ap.add_argument("--riemann", default="hllc")
args = ap.parse_args()
n, gamma = args.n, 5.0 / 3
B = block_layout((n, n, n), 8, 0)[0]
shape = tuple(n * b for b in B)
dl = tuple(1.0 / max(shape) for _ in shape)
dt = setups.baseline_dt(max(shape))
t0 = time.perf_counter()
u0 = blast_block(shape, (0, 0, 0), shape, gamma)
sums0 = [float(u0[..., q].sum(dtype=np.float64)) for q in range(5)]
print("initial condition %.0f s" % (time.perf_counter() - t0), file=sys.stderr, flush=True)
st = NativeBlockGroup(shape, dl, gamma, 1.5, args.riemann, 2, "outflow", world=8, arith=args.arith)
t0 = time.perf_counter()
st.upload(u0)
del u0
print("upload %.0f s" % (time.perf_counter() - t0), file=sys.stderr, flush=True)
st.step(dt, 2); st.synchronize()
t0 = time.perf_counter()
st.step(dt, args.steps); st.synchronize()
elapsed = time.perf_counter() - t0
status = st.status()[0]
t0 = time.perf_counter()
u = st.download()
print("download %.0f s" % (time.perf_counter() - t0), file=sys.stderr, flush=True)
st.close()
sums = [float(u[..., q].sum(dtype=np.float64)) for q in range(5)]
rho = u[..., 0]
ncell = shape[0] * shape[1] * shape[2]
out = {
    "what": "BASELINE config 5 at its stated size on ONE GPU: %dx%dx%d as (%d,%d,%d) loopback blocks of %d^3, PLM+%s RK2, arith=%s" % (shape + tuple(B) + (n, args.riemann.upper(), args.arith)),
    "steps_timed": args.steps, "ms_per_step": elapsed / args.steps * 1e3, "Mcells_per_s": ncell * args.steps / elapsed / 1e6, "status_word": int(status),
    "all_finite": bool(np.isfinite(rho).all()), "min_density": float(rho.min()), "min_energy": float(u[..., 4].min()),
    "conservation_relative": {"mass": abs(sums[0] - sums0[0]) / sums0[0], "energy": abs(sums[4] - sums0[4]) / sums0[4]},
    "net_momentum_over_total_energy": [abs(sums[q]) / sums0[4] for q in (1, 2, 3)],
    "density_mirror_asymmetry_max": [float(np.abs(rho - np.flip(rho, axis=a)).max()) for a in range(3)],
    "steps_total": args.steps + 2,
}
print(json.dumps(out), flush=True)
