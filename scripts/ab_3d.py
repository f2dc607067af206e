#!/usr/bin/env python3
"""Dev measurement: A/B of library builds on the C5 kernel (3-D Euler 384^3, PLM+HLLE RK2), each in its own child process."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import json, sys, time
sys.path.insert(0, %r)
from mara3_amd import setups
from mara3_amd.engine import EulerCartSolver
n, gamma = int(%r), 5.0 / 3
out = {}
for arith in ("fast", "strict"):
    s = EulerCartSolver((n, n, n), (1.0 / n,) * 3, gamma, 1.5, "hlle", 2, "outflow", arith=arith)
    s.upload(setups.blast_ic((n, n, n), gamma)); dt = setups.baseline_dt(n)
    s.step(dt, 8); s.synchronize()
    t0 = time.perf_counter(); s.step(dt, 10); s.synchronize(); out[arith] = round((time.perf_counter() - t0) / 10 * 1e3, 4)
    s.close()
print(json.dumps(out))
''' % (ROOT, os.environ.get('MH_AB_GRID', '384'))
for rnd in range(2):
    for name in sys.argv[1:]:
        env = dict(os.environ); env["MARA_HIP_LIBRARY"] = os.path.join(ROOT, "build", "variants", name, "libmara_hip.so")
        p = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True, timeout=600)
        line = [l for l in p.stdout.splitlines() if l.startswith("{")]
        print(json.dumps({"variant": name, "round": rnd, **(json.loads(line[-1]) if line else {"error": p.stderr[-300:]})}), flush=True)
