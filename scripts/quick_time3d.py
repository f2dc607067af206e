"""Dev helper: time the 3-D Euler step at n^3."""
import sys, time, os
sys.path.insert(0, ".")
import mara3_amd
from mara3_amd.engine import EulerCartSolver
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
gamma = 5.0 / 3
u0 = mara3_amd.setups.blast_ic((n, n, n), gamma)
dl = (1.0 / n,) * 3
dt = mara3_amd.setups.baseline_dt(n)
for arith in ("strict", "fast"):
    for riemann in ("hlle", "hllc"):
        for chunk in [int(c) for c in (sys.argv[2].split(",") if len(sys.argv) > 2 else ["16", "32"])]:
            s = EulerCartSolver((n, n, n), dl, gamma, 1.5, riemann, 2, "outflow", chunk_rows=chunk, arith=arith)
            s.upload(u0)
            s.step(dt, 2); s.synchronize()
            t0 = time.perf_counter(); s.step(dt, 5); s.synchronize(); t1 = time.perf_counter()
            ms = (t1 - t0) / 5 * 1e3
            print("%s %s n=%d chunk=%d: %.3f ms/step %.1f Mzones/s roofline(200B)=%.1f%%" % (arith, riemann, n, chunk, ms, n**3 / ms / 1e3, n**3 * 200 / (ms * 1e-3) / 8e12 * 100), flush=True)
            s.close()
