"""Dev measurement (round 5): the fused planar RK2 step of slabs of 4096 / N rows x 4096 columns under several library builds / chunk settings,
alternating child processes on one GPU. Smooth periodic wave rows (every face in a shock / rarefaction branch), FAST + HLLC, no neighbours.
usage: python scripts/thin_slab_ab.py [--rows 512,1024,4096] [--chunks 0,25,37] [--self-exchange] product build/variants/x/libmara_hip.so ..."""
import argparse, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ap = argparse.ArgumentParser()
ap.add_argument("--rows", default="512,1024,4096")
ap.add_argument("--chunks", default="0")
ap.add_argument("--rounds", type=int, default=3)
ap.add_argument("--self-exchange", action="store_true", help="the slab exchanges four ghost rows with itself through RCCL (fused step across its cuts)")
ap.add_argument("libs", nargs="+")
args = ap.parse_args()
code = ("import sys, time, json; sys.path.insert(0, %r)\n"
        "import numpy as np\n"
        "from mara3_amd import setups\n"
        "from mara3_amd.slab import NativeSlabStepper, native_comm_id\n"
        "n, gamma = 4096, 5.0 / 3; out = {}\n"
        "full = setups.smooth_wave_ic((n, n), gamma)\n"
        "for rows in %r:\n"
        "  for chunk in %r:\n"
        "    kw = dict(arith='fast', planar=True, chunk_rows=chunk)\n"
        "    if %r and rows < n:\n"
        "        st = NativeSlabStepper((rows, n), (1.0 / n, 1.0 / n), gamma, 1.5, 'hllc', 2, 'periodic', rank=0, world=1, comm_id=native_comm_id(0, 1, device='cuda'), self_exchange=True, **kw)\n"
        "    else:\n"
        "        st = NativeSlabStepper((rows, n), (1.0 / n, 1.0 / n), gamma, 1.5, 'hllc', 2, 'periodic', **kw)\n"
        "    st.load_slab(np.ascontiguousarray(full[:rows])); st.step(setups.baseline_dt(n), max(80, int(3e5 / rows))); st.synchronize()          # (>= ~0.1 s: the first configuration of a process ran during the clock ramp after RCCL's start-up)\n"
        "    best = 1e9\n"
        "    for rep in range(3):\n"
        "        t0 = time.perf_counter(); st.step(setups.baseline_dt(n), 200); st.synchronize(); best = min(best, (time.perf_counter() - t0) / 200 * 1e6)\n"
        "    out['%%d/%%d' %% (rows, chunk)] = round(best, 1)\n"
        "    st.close()\n"
        "print(json.dumps(out))\n" % (ROOT, [int(r) for r in args.rows.split(",")], [int(c) for c in args.chunks.split(",")], bool(args.self_exchange)))
for rnd in range(args.rounds):
    line = {"round": rnd}
    for lib in args.libs:
        env = dict(os.environ)
        if lib != "product":
            env["MARA_HIP_LIBRARY"] = os.path.join(ROOT, lib)
        p = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
        name = os.path.basename(os.path.dirname(lib)) if lib != "product" else "product"
        try:
            line[name] = json.loads([l for l in p.stdout.strip().splitlines() if l.startswith("{")][-1])
        except Exception:
            line[name] = (p.stderr or p.stdout)[-300:]
    print(json.dumps(line), flush=True)
