"""Dev helper: the boundary's host-buffer round trip (upload + 1 RK2 step + download) at 4096^2, for DESIGN.md §6."""
import sys, time
sys.path.insert(0, ".")
import numpy as np
import mara3_amd
from mara3_amd.engine import EulerCartSolver
n, gamma = 4096, 5.0 / 3
s = EulerCartSolver((n, n), (1.0 / n, 1.0 / n), gamma, 1.5, "hllc", 2, "outflow", arith="fast")
u = mara3_amd.setups.blast_ic((n, n), gamma)
dt = mara3_amd.setups.baseline_dt(n)
s.upload(u); s.step(dt, 2); s.download()
for label, reps in (("upload", 5), ("download", 5), ("upload + 1 step + download", 5)):
    t0 = time.perf_counter()
    for _ in range(reps):
        if "upload" in label: s.upload(u)
        if "step" in label: s.step(dt, 1)
        if "download" in label: u2 = s.download()
    s.synchronize()
    t = (time.perf_counter() - t0) / reps
    print("%-28s %.2f ms  (%.1f GB/s of %d MB)%s" % (label, t * 1e3, u.nbytes / t / 1e9 * (2 if "+" in label else 1), u.nbytes >> 20,
          "  -> %.0f Mcells/s PCIe-inclusive" % (n * n / t / 1e6) if "+" in label else ""))
