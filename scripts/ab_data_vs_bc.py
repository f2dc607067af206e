#!/usr/bin/env python3
"""Does the fused 4096^2 step's time follow the DATA or the boundary kind? blast / smooth wave initial condition x outflow / periodic sides,
the same kernel and instruction count in all four; ms per step, alternating. usage: python scripts/ab_data_vs_bc.py"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mara3_amd import setups
from mara3_amd.slab import NativeSlabStepper
n, gamma = 4096, 5.0 / 3
dt = setups.baseline_dt(n)
ics = {"blast": setups.blast_ic((n, n), gamma), "smooth_wave": setups.smooth_wave_ic((n, n), gamma)}
st = {}
for ic in ics:
    for bc in ("outflow", "periodic"):
        s = NativeSlabStepper((n, n), (1.0 / n, 1.0 / n), gamma, 1.5, "hllc", 2, bc, arith="fast")
        s.load_slab(ics[ic]); s.step(dt, 40); s.synchronize()
        st[(ic, bc)] = s
for rnd in range(3):
    line = {"round": rnd}
    for key, s in st.items():
        s.step(dt, 10); s.synchronize()
        t0 = time.perf_counter(); s.step(dt, 100); s.synchronize()
        line["%s/%s" % key] = round((time.perf_counter() - t0) * 10, 4)
    print(json.dumps(line), flush=True)
