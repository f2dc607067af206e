#!/usr/bin/env python3
"""Dev measurement: the 4096^2 step on ONE GPU as 1, 2, 3, 4, 8 bands (native slab group, loopback exchange), with and without band skew."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from mara3_amd import setups
from mara3_amd.slab import NativeSlabGroup, NativeSlabStepper
n, gamma = 4096, 5.0 / 3
dl, dt = (1.0 / n, 1.0 / n), setups.baseline_dt(n)
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 100
for workload in ("blast", "smooth_wave"):
    u0 = setups.blast_ic((n, n), gamma) if workload == "blast" else setups.smooth_wave_ic((n, n), gamma)
    bc = "outflow" if workload == "blast" else "periodic"
    for arith, riemann in (("fast", "hllc"), ("strict", "hllc"), ("strict", "hlle")):
        one = NativeSlabStepper((n, n), dl, gamma, 1.5, riemann, 2, bc, arith=arith)
        one.load_slab(u0); one.step(dt, 5); one.synchronize()
        t0 = time.perf_counter(); one.step(dt, steps); one.synchronize(); t1 = (time.perf_counter() - t0) / steps * 1e3
        one.close()
        row = {"workload": workload, "arith": arith, "riemann": riemann, "graph_1band_ms": round(t1, 4)}
        for skew in (0, 1):
            os.environ["MH_SLAB_GROUP_SKEW"] = str(skew)
            for world in (2, 3, 4, 8):
                g = NativeSlabGroup((n, n), dl, gamma, 1.5, riemann, 2, bc, world=world, arith=arith)
                g.upload(u0); g.step(dt, 5); g.synchronize()
                best = 1e9
                for rep in range(3):
                    t0 = time.perf_counter(); g.step(dt, steps); g.synchronize(); best = min(best, (time.perf_counter() - t0) / steps * 1e3)
                g.close()
                row["bands%d_skew%d_ms" % (world, skew)] = round(best, 4)
        print(json.dumps(row), flush=True)
