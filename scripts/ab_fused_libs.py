"""Dev measurement: the fused 4096^2 RK2 step (FAST, HLLC) under several library builds, blast and smooth wave, alternating child processes on one GPU.
usage: python scripts/ab_fused_libs.py product build/variants/x/libmara_hip.so ..."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = ("import sys, time, json; sys.path.insert(0, %r)\n"
        "from mara3_amd import setups\n"
        "from mara3_amd.slab import NativeSlabStepper\n"
        "n, gamma = 4096, 5.0 / 3; out = {}\n"
        "for w in ('blast', 'smooth_wave'):\n"
        "    u0 = setups.blast_ic((n, n), gamma) if w == 'blast' else setups.smooth_wave_ic((n, n), gamma)\n"
        "    st = NativeSlabStepper((n, n), (1.0 / n, 1.0 / n), gamma, 1.5, 'hllc', 2, 'outflow' if w == 'blast' else 'periodic', arith='fast')\n"
        "    st.load_slab(u0); st.step(setups.baseline_dt(n), 60); st.synchronize()\n"
        "    t0 = time.perf_counter(); st.step(setups.baseline_dt(n), 100); st.synchronize(); out[w] = round((time.perf_counter() - t0) * 10, 4)\n"
        "    st.close()\n"
        "print(json.dumps(out))\n" % ROOT)
for rnd in range(3):
    line = {"round": rnd}
    for lib in sys.argv[1:]:
        env = dict(os.environ)
        if lib != "product":
            env["MARA_HIP_LIBRARY"] = os.path.join(ROOT, lib)
        p = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
        line[os.path.basename(os.path.dirname(lib)) if lib != "product" else "product"] = json.loads(p.stdout.strip().splitlines()[-1]) if p.returncode == 0 else p.stderr[-200:]
    print(json.dumps(line), flush=True)
