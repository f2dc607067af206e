"""Dev helper: time the `binary` path (BASELINE config 3: depth=5 block_size=64 -> 2048^2) on one GPU."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mara3_amd import binary

depth = int(sys.argv[1]) if len(sys.argv) > 1 else 5
bs = int(sys.argv[2]) if len(sys.argv) > 2 else 64
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 50
for fixed in (0, 1):
    for chunk in (0, 16, 32, 64):
        cfg = binary.config(depth=depth, block_size=bs, fixed_dt=fixed)
        s = binary.BinarySolver(cfg, chunk_rows=chunk)
        s.next(3)
        t0 = time.perf_counter()
        s.next(steps)
        dt = time.perf_counter() - t0
        s.profile(True)
        s.next(10)
        ms, nl = s.profile(False)
        n = binary.grid_size(cfg)
        print("n=%d fixed_dt=%d chunk=%d: %.3f ms/step  %.1f Mzones/s   stage (3 kernels) %.3f ms x %d" % (n, fixed, chunk, 1e3 * dt / steps, n * n * steps / dt / 1e6, ms, nl), flush=True)
        s.close()
