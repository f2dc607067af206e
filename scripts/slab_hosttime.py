"""Dev helper: is the thin-slab stepper host-bound? Time for mh_slab_step to RETURN (all launches issued) against the time to completion."""
import sys, time
sys.path.insert(0, ".")
from mara3_amd import setups
from mara3_amd.slab import NativeSlabStepper, native_comm_id
n1, gamma = 4096, 5.0 / 3
for n0 in (512, 1024):
    import numpy as np
    u0 = np.ascontiguousarray(setups.smooth_wave_ic((4096, n1), gamma)[:n0])          # (planar: no third momentum)
    for mode in ("eager", "self-exchange"):
        kw = dict(comm_id=native_comm_id(0, 1, device="cuda"), self_exchange=True) if mode == "self-exchange" else {}
        st = NativeSlabStepper((n0, n1), (1.0 / 4096, 1.0 / 4096), gamma, 1.5, "hllc", 2, "periodic", arith="fast", planar=True, **kw)
        st.load_slab(u0)
        st.step(1e-5, 20, graph=False); st.synchronize()
        t0 = time.perf_counter(); st.step(1e-5, 200, graph=False); t1 = time.perf_counter(); st.synchronize(); t2 = time.perf_counter()
        print("rows=%d %-14s issue %.1f us/step, complete %.1f us/step" % (n0, mode, (t1 - t0) / 200 * 1e6, (t2 - t0) / 200 * 1e6), flush=True)
        st.close()
