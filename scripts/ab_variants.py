#!/usr/bin/env python3
"""Dev measurement: A/B of library builds (scripts/build_variant.sh) on one box. usage: ab_variants.py name1 name2 ... ; each variant runs in
its own child process (MARA_HIP_LIBRARY), twice in alternation; prints ms per 4096^2 RK2 step (graph replay, best of 3 x 100 steps)."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import json, sys, time
sys.path.insert(0, %r)
from mara3_amd import setups
from mara3_amd.slab import NativeSlabStepper
n, gamma = 4096, 5.0 / 3
dl, dt = (1.0 / n, 1.0 / n), setups.baseline_dt(n)
out = {}
for workload in ("blast", "smooth_wave"):
    u0 = setups.blast_ic((n, n), gamma) if workload == "blast" else setups.smooth_wave_ic((n, n), gamma)
    bc = "outflow" if workload == "blast" else "periodic"
    for arith, riemann in (("fast", "hllc"), ("fast", "hlle"), ("strict", "hllc")):
        s = NativeSlabStepper((n, n), dl, gamma, 1.5, riemann, 2, bc, arith=arith)
        s.load_slab(u0); s.step(dt, 10); s.synchronize()
        best = 1e9
        for rep in range(3):
            t0 = time.perf_counter(); s.step(dt, 100); s.synchronize(); best = min(best, (time.perf_counter() - t0) / 100 * 1e3)
        s.profile(True); s.step(dt, 5); s.synchronize()
        (a1, a2), _, _ = s.profile_read()
        s.close()
        out["%%s_%%s_%%s" %% (workload, arith, riemann)] = [round(best, 4), round(a1, 4), round(a2, 4)]
print(json.dumps(out))
''' % ROOT
for rnd in range(2):
    for name in sys.argv[1:]:
        env = dict(os.environ)
        if name != "product":
            env["MARA_HIP_LIBRARY"] = os.path.join(ROOT, "build", "variants", name, "libmara_hip.so")
        p = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True, timeout=600)
        line = [l for l in p.stdout.splitlines() if l.startswith("{")]
        print(json.dumps({"variant": name, "round": rnd, **(json.loads(line[-1]) if line else {"error": p.stderr[-300:]})}), flush=True)
