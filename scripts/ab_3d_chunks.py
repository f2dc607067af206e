#!/usr/bin/env python3
"""Dev measurement: the C5 kernel (3-D Euler PLM+HLLE RK2, MH_AB_GRID^3, default 512) over the chunk length along axis 0 (planes per work item)."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import json, sys, time
sys.path.insert(0, %r)
from mara3_amd import setups
from mara3_amd.engine import EulerCartSolver
n, gamma = int(%r), 5.0 / 3
chunks = [int(x) for x in sys.argv[1:]]
u0 = setups.blast_ic((n, n, n), gamma); dt = setups.baseline_dt(n)
for arith in ("fast", "strict"):
    for rnd in range(2):
        for ch in chunks:
            s = EulerCartSolver((n, n, n), (1.0 / n,) * 3, gamma, 1.5, "hlle", 2, "outflow", arith=arith, chunk_rows=ch)
            s.upload(u0)
            s.step(dt, 6); s.synchronize()
            t0 = time.perf_counter(); s.step(dt, 10); s.synchronize()
            print(json.dumps({"arith": arith, "chunk": ch, "round": rnd, "ms_per_step": round((time.perf_counter() - t0) / 10 * 1e3, 4), "status": s.status()}), flush=True)
            s.close()
''' % (ROOT, os.environ.get('MH_AB_GRID', '512'))
p = subprocess.run([sys.executable, "-c", CHILD] + (sys.argv[1:] or ["32", "64", "128"]), text=True, timeout=1100)
sys.exit(p.returncode)
