"""Dev helper: step time of a 512/1024/2048-row slab (N = 8/4/2 share of 4096^2) vs chunk_rows, with RCCL self-exchange."""
import sys, time
sys.path.insert(0, ".")
from mara3_amd import setups
from mara3_amd.slab import NativeSlabStepper, native_comm_id
n1, gamma = 4096, 5.0 / 3
dl = (1.0 / 4096, 1.0 / 4096)
for arith in ("fast",):
    for n0 in (512, 1024, 2048):
        u0 = setups.wave_ic((n0, n1), gamma, seed=1)
        for chunk in (6, 8, 10, 12, 16, 24, 32):
            st = NativeSlabStepper((n0, n1), dl, gamma, 1.5, "hllc", 2, "periodic", arith=arith, chunk_rows=chunk,
                                   comm_id=native_comm_id(0, 1), self_exchange=True)
            st.load_slab(u0)
            st.step(1e-5, 5); st.synchronize()
            t0 = time.perf_counter(); st.step(1e-5, 50); st.synchronize(); t = (time.perf_counter() - t0) / 50 * 1e3
            st.close()
            print("%s rows=%4d chunk=%2d: %.3f ms/step" % (arith, n0, chunk, t), flush=True)
