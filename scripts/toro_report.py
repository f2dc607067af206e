#!/usr/bin/env python3
"""Dev helper: error norms and front positions of Toro's tests 1-5 on the HIP path (the numbers behind tests/test_gpu_toro.py)."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from mara3_amd import engine
from toro_helpers import metrics
for test in (1, 2, 3, 4, 5):
    for riemann in ("hllc", "hlle"):
        for arith in ("strict", "fast"):
            for n in (200, 400, 800):
                try:
                    m = metrics(engine, test, n, riemann=riemann, arith=arith)
                except Exception as e:
                    m = {"error": repr(e)}
                print(json.dumps({"test": test, "riemann": riemann, "arith": arith, "n": n, **m}), flush=True)
