"""Dev check: long runs of the 2-D Euler stepper, FAST against STRICT arithmetic on the same initial data - the blast until it has left through
the outflow boundaries, the smooth wave through several periods; conserved-variable L1, max-norm, status words and (periodic) conservation."""
import json, sys
sys.path.insert(0, ".")
import numpy as np
from mara3_amd import setups
from mara3_amd.slab import NativeSlabStepper
gamma = 5.0 / 3
for workload, n, nsteps, riemann in (("blast", 1024, 6000, "hllc"), ("blast", 1024, 6000, "hlle"), ("smooth_wave", 1024, 8000, "hllc"), ("smooth_wave", 512, 8000, "hlle")):
    u0 = setups.blast_ic((n, n), gamma) if workload == "blast" else setups.smooth_wave_ic((n, n), gamma)
    bc = "outflow" if workload == "blast" else "periodic"
    dl, dt = (1.0 / n, 1.0 / n), setups.baseline_dt(n)
    res = {}
    for arith in ("strict", "fast"):
        s = NativeSlabStepper((n, n), dl, gamma, 1.5, riemann, 2, bc, arith=arith)
        s.load_slab(u0)
        s.step(dt, nsteps)
        s.synchronize()
        res[arith] = (s.slab_host(), s.status_result())
        s.close()
    a, b = res["strict"][0], res["fast"][0]
    scale = np.abs(a).reshape(-1, 5).max(axis=0)
    out = {"workload": workload, "n": n, "steps": nsteps, "riemann": riemann, "t_final": nsteps * dt,
           "status": [int(res[k][1][0]) for k in ("strict", "fast")],
           "l1_rel": float(np.abs(a - b).mean() / np.abs(a).mean()), "max_rel_per_var": [float(x) for x in np.abs(a - b).reshape(-1, 5).max(axis=0) / np.where(scale > 0, scale, 1)],
           "min_density": float(a[..., 0].min()), "min_pressure_proxy": float((a[..., 4] - 0.5 * (a[..., 1] ** 2 + a[..., 2] ** 2) / a[..., 0]).min())}
    if bc == "periodic":
        out["conservation_rel"] = [float(abs(x[..., q].sum() - u0[..., q].sum()) / max(abs(u0[..., q].sum()), 1e-300)) for x in (a, b) for q in (0, 4)]
    print(json.dumps(out), flush=True)
