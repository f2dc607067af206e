"""Dev helper: time the ctx-API step at 4096^2 for a few chunk_rows / solvers (not part of the bench contract)."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
import mara3_amd
from mara3_amd.engine import EulerCartSolver

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
gamma = 5.0 / 3
u0 = mara3_amd.setups.blast_ic((n, n), gamma)
dl = (1.0 / n, 1.0 / n)
dt = mara3_amd.setups.baseline_dt(n)
import os
ARITH = os.environ.get("ARITH", "strict")
for riemann in ("hlle", "hllc"):
    for chunk in [int(c) for c in (sys.argv[2].split(",") if len(sys.argv) > 2 else ["32", "64", "128"])]:
        s = EulerCartSolver((n, n), dl, gamma, 1.5, riemann, 2, "outflow", chunk_rows=chunk, arith=ARITH)
        s.upload(u0)
        s.step(dt, 3)
        s.synchronize()
        s.profile(True)
        t0 = time.perf_counter()
        s.step(dt, 20)
        s.synchronize()
        t1 = time.perf_counter()
        ms, nl = s.profile_read()
        step_ms = (t1 - t0) / 20 * 1e3
        print(ARITH + " %s chunk=%d: %.3f ms/step  %.1f Mzones/s  avg stage kernel %.3f ms (%d launches)  roofline(200B/zone)=%.1f%% of 8TB/s"
              % (riemann, chunk, step_ms, n * n / step_ms / 1e3, ms, nl, n * n * 200 / (step_ms * 1e-3) / 8e12 * 100), flush=True)
        s.close()
