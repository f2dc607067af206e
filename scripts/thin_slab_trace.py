"""Dev measurement (round 5): ONE lone slab of `rows` x 4096 (fused planar RK2 step, FAST + HLLC, smooth periodic wave), `steps` steps; prints us per step.
Run under `rocprofv3 --kernel-trace --stats` to split the step period into kernel duration and launch gap.
usage: python scripts/thin_slab_trace.py [rows] [steps] [chunk_rows]"""
import json, sys, time
sys.path.insert(0, ".")
import numpy as np
from mara3_amd import setups
from mara3_amd.slab import NativeSlabStepper
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 512
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 400
chunk = int(sys.argv[3]) if len(sys.argv) > 3 else 0
n, gamma = 4096, 5.0 / 3
full = setups.smooth_wave_ic((n, n), gamma)
st = NativeSlabStepper((rows, n), (1.0 / n, 1.0 / n), gamma, 1.5, "hllc", 2, "periodic", arith="fast", planar=True, chunk_rows=chunk)
st.load_slab(np.ascontiguousarray(full[:rows]))
st.step(setups.baseline_dt(n), 100); st.synchronize()
t0 = time.perf_counter(); st.step(setups.baseline_dt(n), steps); st.synchronize()
print(json.dumps({"rows": rows, "chunk_rows": chunk, "steps": steps, "us_per_step": round((time.perf_counter() - t0) / steps * 1e6, 2)}))
st.close()
