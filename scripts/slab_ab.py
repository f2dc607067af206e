"""Dev helper: A/B of the slab stepper's synchronisation variants on ONE box (env switches read at mh_slab_create)."""
import os, sys, time
sys.path.insert(0, ".")
from mara3_amd import setups
from mara3_amd.slab import NativeSlabStepper, native_comm_id
n1, gamma = 4096, 5.0 / 3
for n0 in (512, 1024, 2048):
    u0 = setups.wave_ic((n0, n1), gamma, seed=1)
    for rep in range(2):
        for stagger, onlaunch, chunk in ((0, 0, 0), (0, 1, 0), (2, 0, 0), (2, 1, 0), (4, 1, 0), (None, None, 0)):
            if stagger is None:
                kw = {}
            else:
                os.environ["MH_SLAB_STAGGER"] = str(stagger); os.environ["MH_SLAB_EVENT_ON_LAUNCH"] = str(onlaunch)
                kw = dict(comm_id=native_comm_id(0, 1), self_exchange=True)
            st = NativeSlabStepper((n0, n1), (1.0 / 4096, 1.0 / 4096), gamma, 1.5, "hllc", 2, "periodic", arith="fast", chunk_rows=chunk, **kw)
            st.load_slab(u0)
            st.step(1e-5, 20, graph=False); st.synchronize()
            t0 = time.perf_counter(); st.step(1e-5, 300, graph=False); st.synchronize(); t2 = time.perf_counter()
            print("rows=%d stagger=%s event_on_launch=%s chunk=%d: %.1f us/step" % (n0, stagger, onlaunch, chunk, (t2 - t0) / 300 * 1e6), flush=True)
            st.close()
