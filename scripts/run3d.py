"""Dev helper: run the 3-D Euler step (for profiling). usage: run3d.py n chunk arith riemann steps"""
import sys, time
sys.path.insert(0, ".")
import mara3_amd
from mara3_amd.engine import EulerCartSolver
n, chunk, arith, riemann, steps = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4], int(sys.argv[5])
gamma = 5.0 / 3
s = EulerCartSolver((n, n, n), (1.0 / n,) * 3, gamma, 1.5, riemann, 2, "outflow", chunk_rows=chunk, arith=arith)
s.upload(mara3_amd.setups.blast_ic((n, n, n), gamma))
dt = mara3_amd.setups.baseline_dt(n)
s.step(dt, 2); s.synchronize()
t0 = time.perf_counter(); s.step(dt, steps); s.synchronize(); t1 = time.perf_counter()
ms = (t1 - t0) / steps * 1e3
print("%s %s n=%d chunk=%d: %.3f ms/step %.1f Mzones/s roofline(200B)=%.1f%%" % (arith, riemann, n, chunk, ms, n**3 / ms / 1e3, n**3 * 200 / (ms * 1e-3) / 8e12 * 100), flush=True)
