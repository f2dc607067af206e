#!/usr/bin/env python3
"""Dev measurement: the C4 workload (cloud 4096^2, arith=fast) as two launches per RK2 step and as the fused launch (cloud_fused.hip), alternating,
each in its own child process of the compiled host; per-step times from the host's own kzps lines (vertices per ms, like the reference).
usage: ab_cloud_fused.py [variant-name ...]   (build/variants/<name>/libmara_hip.so; none = the product library)
       MH_AB_CHUNKS="0 64 98 196" sets the chunk lengths tried for the fused launch"""
import json, os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
exe = os.path.join(ROOT, "mara3_amd", "host", "mara_hip")
nr, steps, warm = 4096, 40, 15
chunks = [int(x) for x in os.environ.get("MH_AB_CHUNKS", "0").split()]
names = sys.argv[1:] or [""]
for rnd in range(2):
    for name in names:
        env = dict(os.environ)
        if name:
            env["MARA_HIP_LIBRARY"] = os.path.join(ROOT, "build", "variants", name, "libmara_hip.so")
            env["LD_LIBRARY_PATH"] = os.path.join(ROOT, "build", "variants", name) + ":" + env.get("LD_LIBRARY_PATH", "")
        for fuse, chunk in [(-1, 0)] + [(1, c) for c in chunks]:
            p = subprocess.run([exe, "cloud", "nr=%d" % nr, "num_decades=1", "rk_order=2", "reconstruct_method=2", "plm_theta=1.2", "max_steps=%d" % steps, "profile=1", "cpi=0",
                                "outdir=/tmp/ab_cloud_out", "arith=fast", "fuse=%d" % fuse, "chunk_rows=%d" % chunk], env=env, capture_output=True, text=True, timeout=600)
            kz = [float(x) for x in re.findall(r"kzps=([0-9.]+)", p.stdout)]
            ms = sorted((nr + 1) * (nr + 1) / k for k in kz[warm:])
            m = re.search(r"profile: stage kernel avg ([0-9.]+) ms over (\d+)", p.stdout)
            print(json.dumps({"variant": name or "product", "round": rnd, "fuse": fuse, "chunk_rows": chunk, "rc": p.returncode,
                              "ms_per_step_median": ms[len(ms) // 2] if ms else None, "ms_per_step_min": ms[0] if ms else None,
                              "Mzones_per_s_median": nr * nr / ms[len(ms) // 2] / 1e3 if ms else None,
                              "event_ms_per_launch": float(m.group(1)) if m else None, "launches": int(m.group(2)) if m else None,
                              "tail": "" if p.returncode == 0 else (p.stdout + p.stderr)[-300:]}), flush=True)
