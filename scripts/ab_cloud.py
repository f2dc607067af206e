#!/usr/bin/env python3
"""Dev measurement: A/B of library builds on the C4 workload (cloud 4096^2, fast and strict), each in its own child process."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for rnd in range(2):
    for name in sys.argv[1:]:
        env = dict(os.environ)
        env["MARA_HIP_LIBRARY"] = os.path.join(ROOT, "build", "variants", name, "libmara_hip.so")
        env["LD_LIBRARY_PATH"] = os.path.join(ROOT, "build", "variants", name) + ":" + env.get("LD_LIBRARY_PATH", "")
        out = {}
        for arith in ("fast", "strict"):
            exe = os.path.join(ROOT, "mara3_amd", "host", "mara_hip")
            p = subprocess.run([exe, "cloud", "nr=4096", "num_decades=1", "rk_order=2", "reconstruct_method=2", "plm_theta=1.2", "max_steps=12", "profile=1", "cpi=0",
                                "outdir=/tmp/ab_cloud_out", "arith=" + arith], env=env, capture_output=True, text=True, timeout=600)
            import re
            m = re.search(r"profile: stage kernel avg ([0-9.]+) ms", p.stdout)
            out[arith] = float(m.group(1)) if m else p.stdout[-200:]
        print(json.dumps({"variant": name, "round": rnd, **out}), flush=True)
