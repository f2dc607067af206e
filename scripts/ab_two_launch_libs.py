import json, os, subprocess, sys
ROOT = "/root/repo" if os.path.isdir("/root/repo") else os.getcwd()
code = ("import sys, time, json; sys.path.insert(0, %r)\n"
        "from mara3_amd import setups\n"
        "from mara3_amd.slab import NativeSlabStepper\n"
        "n, gamma = 4096, 5.0 / 3; out = {}\n"
        "u0 = setups.blast_ic((n, n), gamma)\n"
        "for arith, riemann, fuse, planar in (('strict', 'hlle', None, None), ('strict', 'hlle', None, False), ('strict', 'hllc', None, None), ('fast', 'hllc', False, None), ('fast', 'hllc', False, False)):\n"
        "    st = NativeSlabStepper((n, n), (1.0 / n, 1.0 / n), gamma, 1.5, riemann, 2, 'outflow', arith=arith, fuse=fuse, planar=planar)\n"
        "    st.load_slab(u0); st.step(setups.baseline_dt(n), 40); st.synchronize()\n"
        "    t0 = time.perf_counter(); st.step(setups.baseline_dt(n), 60); st.synchronize(); out[arith + '_' + riemann + ('_general' if planar is False else '')] = round((time.perf_counter() - t0) / 60 * 1e3, 4)\n"
        "    st.close()\n"
        "print(json.dumps(out))\n" % ROOT)
for rnd in range(2):
    line = {"round": rnd}
    for lib in sys.argv[1:]:
        env = dict(os.environ); env["MARA_HIP_LIBRARY"] = os.path.join(ROOT, "build", "variants", lib, "libmara_hip.so")
        p = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
        line[lib] = json.loads(p.stdout.strip().splitlines()[-1]) if p.returncode == 0 else p.stderr[-200:]
    print(json.dumps(line), flush=True)
