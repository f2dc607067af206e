#!/usr/bin/env python3
"""One-off stress of the fused RK2 launch against the two launches, bit for bit: every column count from 8 to 260 (all positions of the
domain's edge within a workgroup of two pairs: 116 output columns) x a few row counts, both boundary kinds, both Riemann solvers, the
library's chunking and short chunks. Prints the first mismatch, or a summary line. usage: python scripts/stress_fused_shapes.py"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mara3_amd import setups
from mara3_amd.engine import EulerCartSolver
gamma, bad, cases = 1.4, 0, 0
rng = np.random.default_rng(7)
for n1 in list(range(8, 261)) + [347, 348, 349, 463, 464, 465, 579, 580, 581]:
    n0 = int(rng.choice([8, 9, 13, 31, 64, 101]))
    bc = ("outflow", "periodic")[n1 % 2]
    riemann = ("hllc", "hlle")[(n1 // 2) % 2]
    chunk = int(rng.choice([0, 0, 2, 5, 11]))
    shape, dl = (n0, n1), (1.0 / n0, 1.0 / n1)
    u0 = setups.wave_ic(shape, gamma, seed=n1)
    res = []
    for fuse in (False, True):
        s = EulerCartSolver(shape, dl, gamma, 1.5, riemann, 2, bc, arith="fast", fuse=fuse, chunk_rows=chunk)
        s.upload(u0); s.step(0.2 * min(dl) / 2.0, 3); res.append(s.download()); st = s.status(); s.close()
    cases += 1
    if not np.array_equal(res[0].view(np.uint64), res[1].view(np.uint64)) or st != 0:
        bad += 1
        w = np.argwhere(res[0] != res[1])
        print(json.dumps({"mismatch": [n0, n1, bc, riemann, chunk], "first": w[:3].tolist(), "status": int(st)}), flush=True)
        if bad > 5:
            break
print(json.dumps({"cases": cases, "mismatches": bad}))
