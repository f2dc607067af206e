"""Dev helper: does a launch that fits exactly one residency round (<= 2048 waves) beat many short chunks?"""
import sys, time
sys.path.insert(0, ".")
from mara3_amd import setups
from mara3_amd.slab import NativeSlabStepper
n1, gamma = 4096, 5.0 / 3
dl = (1.0 / 4096, 1.0 / 4096)
for arith in ("fast", "strict"):
    for n0, chunks in ((4096, (32, 137, 142, 147, 160)), (2048, (24, 69, 71, 74)), (1024, (12, 35, 36, 38)), (512, (8, 17, 18, 19, 20))):
        u0 = setups.wave_ic((n0, n1), gamma, seed=1)
        for chunk in chunks:
            st = NativeSlabStepper((n0, n1), dl, gamma, 1.5, "hllc", 2, "periodic", arith=arith, chunk_rows=chunk)
            st.load_slab(u0)
            st.step(1e-5, 5); st.synchronize()
            t0 = time.perf_counter(); st.step(1e-5, 40); st.synchronize(); t = (time.perf_counter() - t0) / 40 * 1e3
            st.close()
            print("%s rows=%4d chunk=%3d: %.3f ms/step" % (arith, n0, chunk, t), flush=True)
