#!/usr/bin/env python3
"""Dev measurement: A/B of library builds on the C3 workload (binary 2048^2 uniform, fixed_dt, fast and strict), each in its own child process;
prints the program's own kzps (median of the last chunks)."""
import json, os, re, statistics, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for rnd in range(2):
    for name in sys.argv[1:]:
        env = dict(os.environ)
        env["LD_LIBRARY_PATH"] = os.path.join(ROOT, "build", "variants", name) + ":" + env.get("LD_LIBRARY_PATH", "")
        out = {}
        for arith in ("fast", "strict"):
            exe = os.path.join(ROOT, "mara3_amd", "host", "mara_hip")
            p = subprocess.run([exe, "binary", "depth=5", "block_size=64", "focus_factor=1e9", "fixed_dt=1", "max_iterations=400", "steps_per_call=50", "cpi=0", "dfi=0", "tsi=0",
                                "outdir=/tmp/ab_binary_out", "arith=" + arith], env=env, capture_output=True, text=True, timeout=600)
            k = [float(x) for x in re.findall(r"kzps=([0-9.]+)", p.stdout)]
            out[arith] = round(statistics.median(k[2:-1]) / 1e3, 1) if len(k) > 4 else p.stderr[-200:]
        print(json.dumps({"variant": name, "round": rnd, "Mzones_per_s": out}), flush=True)
