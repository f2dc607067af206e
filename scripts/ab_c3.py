#!/usr/bin/env python3
"""Dev measurement: A/B of library builds on C3 (`binary` 2048^2), each in its own child process: bench_configs.py --config c3, two rounds."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for rnd in range(2):
    for name in sys.argv[1:]:
        env = dict(os.environ)
        if name != "product":
            env["MARA_HIP_LIBRARY"] = os.path.join(ROOT, "build", "variants", name, "libmara_hip.so")
        p = subprocess.run([sys.executable, os.path.join(ROOT, "bench_configs.py"), "--config", "c3", "--steps", "40", "--warmup", "20", "--no-cpu-baseline"],
                           env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
        line = [l for l in p.stdout.splitlines() if l.startswith("{")]
        if line:
            d = json.loads(line[-1])
            print(json.dumps({"variant": name, "round": rnd, "fast": round(d["value"]), "fast_ms": round(d["ms_per_step"], 4), "stage_ms": round(d["roofline"]["avg_launch_ms"], 4),
                              "strict": round(d["arith_strict"]["value"]), "finite": d["config"]["finite_and_positive"]}), flush=True)
        else:
            print(json.dumps({"variant": name, "round": rnd, "error": p.stderr[-400:]}), flush=True)
