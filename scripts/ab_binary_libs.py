"""Dev measurement: the C3 stage kernel (2048^2, fixed_dt, FAST) under several library builds, alternating child processes on one GPU.
usage: python scripts/ab_binary_libs.py product build/variants/x/libmara_hip.so ..."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = ("import sys, time, json; sys.path.insert(0, %r)\n"
        "from mara3_amd import binary\n"
        "cfg = binary.config(depth=5, block_size=64, fixed_dt=1, rk_order=2, plm_theta=1.8)\n"
        "s = binary.BinarySolver(cfg, arith='fast'); s.next(60)\n"
        "t0 = time.perf_counter(); s.next(100); dt = time.perf_counter() - t0\n"
        "s.profile(True); s.next(10); ms, nl = s.profile(False)\n"
        "print(json.dumps([round(1e3 * dt / 100, 4), round(ms, 4)]))\n" % ROOT)
for rnd in range(3):
    line = {"round": rnd}
    for lib in sys.argv[1:]:
        env = dict(os.environ)
        if lib != "product":
            env["MARA_HIP_LIBRARY"] = os.path.join(ROOT, lib)
        p = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
        line[os.path.basename(os.path.dirname(lib)) if lib != "product" else "product"] = json.loads(p.stdout.strip().splitlines()[-1]) if p.returncode == 0 else p.stderr[-200:]
    print(json.dumps(line), flush=True)
