"""Dev measurement: what one GPU of an N-GPU strong-scaling run of the 4096^2 grid has to do - a slab of 4096 / N rows - against 1 / N of the whole
grid's step, on one box (no neighbours, and exchanging with itself through RCCL): the efficiency N GPUs can reach if the xGMI exchange stays hidden."""
import json, sys, time
sys.path.insert(0, ".")
import numpy as np
from mara3_amd import setups
from mara3_amd.slab import NativeSlabStepper, native_comm_id
n, gamma = 4096, 5.0 / 3
dl, dt = (1.0 / n, 1.0 / n), setups.baseline_dt(n)
out = {}
for workload in ("blast", "smooth_wave"):
    full = setups.blast_ic((n, n), gamma) if workload == "blast" else setups.smooth_wave_ic((n, n), gamma)
    bc = "outflow" if workload == "blast" else "periodic"
    res = {}
    for N in (1, 2, 4, 8):
        rows = n // N
        lo = (n - rows) // 2 if workload == "blast" else 0          # the slab through the middle of the blast: the one with the most active cells
        u0 = np.ascontiguousarray(full[lo:lo + rows])
        for mode in ("alone", "self_exchange"):
            if mode == "self_exchange" and N == 1:
                continue
            kw = dict(arith="fast", planar=True)          # (as bench.py: ranks in different processes take the planar kernel on the caller's word)
            if mode == "self_exchange":
                st = NativeSlabStepper((rows, n), dl, gamma, 1.5, "hllc", 2, "periodic", rank=0, world=1, comm_id=native_comm_id(0, 1, device="cuda"), self_exchange=True, **kw)
            else:
                st = NativeSlabStepper((rows, n), dl, gamma, 1.5, "hllc", 2, "periodic" if N > 1 else bc, **kw)
            st.load_slab(u0)
            st.step(dt, 60); st.synchronize()
            best = 1e9
            for rep in range(3):
                t0 = time.perf_counter(); st.step(dt, 200); st.synchronize(); best = min(best, (time.perf_counter() - t0) / 200 * 1e6)
            st.close()
            res["%d_%s" % (N, mode)] = round(best, 1)
    t1 = res["1_alone"]
    res["efficiency"] = {k: round(t1 / (int(k.split("_")[0]) * v), 3) for k, v in res.items() if k != "1_alone"}
    out[workload] = res
    print(json.dumps({workload: res}), flush=True)
