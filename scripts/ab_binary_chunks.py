"""Dev measurement: stage-kernel time of the C3 workload (2048^2, fixed_dt, FAST) over chunk lengths and library builds (MARA_HIP_LIBRARY),
alternating in one process per library. usage: python scripts/ab_binary_chunks.py [chunks comma list]"""
import sys, time, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mara3_amd import binary
chunks = [int(c) for c in (sys.argv[1] if len(sys.argv) > 1 else "0,18,24,30,36,40,48,64").split(",")]
cfg = binary.config(depth=5, block_size=64, fixed_dt=1, rk_order=2, plm_theta=1.8)
n = binary.grid_size(cfg)
solvers = {c: binary.BinarySolver(cfg, chunk_rows=c, arith="fast") for c in chunks}
for s in solvers.values():
    s.next(30)
for rnd in range(3):
    line = {"round": rnd, "lib": os.environ.get("MARA_HIP_LIBRARY", "product")}
    for c, s in solvers.items():
        s.next(5)
        t0 = time.perf_counter(); s.next(60); dt = time.perf_counter() - t0
        s.profile(True); s.next(5); ms, nl = s.profile(False)
        line["c%d" % c] = [round(1e3 * dt / 60, 4), round(ms, 4)]
    print(json.dumps(line), flush=True)
