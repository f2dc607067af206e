"""Dev helper: one thin slab without neighbours, step time against rows per chunk (how many residency rounds the launch takes)."""
import sys, time
sys.path.insert(0, ".")
from mara3_amd import setups
from mara3_amd.slab import NativeSlabStepper
n1, gamma = 4096, 5.0 / 3
for n0 in (512, 1024):
    u0 = setups.wave_ic((n0, n1), gamma, seed=1)
    for chunk in (0, 5, 6, 8, 9, 10, 12, 14, 16, 18, 20, 24, 36):
        st = NativeSlabStepper((n0, n1), (1.0 / 4096, 1.0 / 4096), gamma, 1.5, "hllc", 2, "periodic", arith="fast", chunk_rows=chunk)
        st.load_slab(u0)
        st.step(1e-5, 20, graph=False); st.synchronize()
        t0 = time.perf_counter(); st.step(1e-5, 300, graph=False); st.synchronize(); t2 = time.perf_counter()
        nch = -(-n0 // chunk) if chunk else 0
        print("rows=%d chunk=%d (%d chunks, %d waves): %.1f us/step" % (n0, chunk, nch, nch * 69, (t2 - t0) / 300 * 1e6), flush=True)
        st.close()
