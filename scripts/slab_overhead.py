"""Dev helper: per-step time of ONE rank's share at N-GPU scale, on one GPU, with the RCCL exchange going to self.
Compares against the same rows without any exchange (pure kernel time)."""
import sys, time
sys.path.insert(0, ".")
import numpy as np
from mara3_amd import setups
from mara3_amd.slab import NativeSlabStepper, native_comm_id
n1 = 4096
gamma = 5.0 / 3
for arith in ("fast", "strict"):
    for n0 in (4096, 2048, 1024, 512):
        dl = (1.0 / 4096, 1.0 / 4096)
        u0 = setups.wave_ic((n0, n1), gamma, seed=1)
        out = []
        for ex in (False, True):
            kw = dict(comm_id=native_comm_id(0, 1), self_exchange=True) if ex else {}
            st = NativeSlabStepper((n0, n1), dl, gamma, 1.5, "hllc", 2, "periodic", arith=arith, **kw)
            st.load_slab(u0)
            st.step(1e-5, 5); st.synchronize()
            t0 = time.perf_counter(); st.step(1e-5, 50); st.synchronize(); t = (time.perf_counter() - t0) / 50 * 1e3
            out.append(t)
            st.close()
        print("%s rows=%4d (N=%d): no-exchange %.3f ms/step, with RCCL self-exchange %.3f ms/step -> parallel efficiency vs N=1 kernel time: %.0f%%"
              % (arith, n0, 4096 // n0, out[0], out[1], 100 * (out[0] if n0 == 4096 else None or 0) / out[1] if n0 == 4096 else 0), flush=True)
