#!/usr/bin/env python3
"""Dev measurement: does dealing first-stage and second-stage work items of two row bands into ONE grid (mh_euler_cart_stage_mixed)
beat the two plain launches? Timing only (the two parts work on separate copies of the fields)."""
import ctypes as C, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from mara3_amd import setups, _lib as L
from mara3_amd.slab import euler_cart_desc
lib = L.load_library()
n, gamma = 4096, 5.0 / 3
dt = setups.baseline_dt(n)
for workload in ("blast", "smooth_wave"):
    bc = "outflow" if workload == "blast" else "periodic"
    u0 = torch.from_numpy(setups.blast_ic((n, n), gamma) if workload == "blast" else setups.smooth_wave_ic((n, n), gamma)).cuda()
    for arith, riemann in (("fast", "hllc"), ("fast", "hlle"), ("strict", "hllc"), ("strict", "hlle")):
        d = euler_cart_desc((n, n), (1.0 / n, 1.0 / n), gamma, 1.5, riemann, bc, 0, arith)
        def field():
            f = torch.zeros((n + 4, 5, n), dtype=torch.float64, device="cuda")
            f[2:2 + n] = u0.permute(0, 2, 1)
            L.check(lib.mh_euler_cart_fill_ghosts(C.byref(d), C.c_void_p(f.data_ptr()), None))
            return f
        P, S, Q = field(), field(), field()
        P2, S2, Q2 = field(), field(), field()
        st = torch.zeros(2, dtype=torch.int32, device="cuda")
        ptr = lambda t: C.c_void_p(t.data_ptr())
        stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        def plain(a, b):
            L.check(lib.mh_euler_cart_stage(C.byref(d), ptr(P), None, ptr(S), dt, 1.0, a, b, ptr(st), stream))
        def plain2(a, b):
            L.check(lib.mh_euler_cart_stage(C.byref(d), ptr(S2), ptr(P2), ptr(Q2), dt, 0.5, a, b, ptr(st), stream))
        def mixed(a0, a1, b0, b1):
            L.check(lib.mh_euler_cart_stage_mixed(C.byref(d), dt, ptr(P), ptr(S), a0, a1, ptr(S2), ptr(P2), ptr(Q2), 0.5, b0, b1, ptr(st), stream))
        # S2 = a first-stage result
        L.check(lib.mh_euler_cart_stage(C.byref(d), ptr(P2), None, ptr(S2), dt, 1.0, 0, n, ptr(st), stream))
        def timeit(fn, reps=30):
            for _ in range(3): fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps): fn()
            e1.record(); torch.cuda.synchronize()
            return e0.elapsed_time(e1) / reps
        h = n // 2
        row = {"workload": workload, "arith": arith, "riemann": riemann,
               "plain_s1_plus_s2_ms": round(timeit(lambda: (plain(0, n), plain2(0, n))), 4),
               "mixed_halves_ms": round(timeit(lambda: (mixed(0, h, h, n), mixed(h, n, 0, h))), 4)}
        for frac in (0.4, 0.6):
            c = int(n * frac) // 4 * 4
            row["mixed_cut%.1f_ms" % frac] = round(timeit(lambda: (mixed(0, c, c, n), mixed(c, n, 0, c))), 4)
        print(json.dumps(row), flush=True)
