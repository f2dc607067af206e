"""Dev helper: one thin slab (rows x 4096, fast HLLC RK2) with the RCCL exchange going to self, for a rocprofv3 kernel trace:
  rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_slab -- python3 scripts/slab_trace.py 512
then scripts/slab_trace_report.py gpurun_out/prof_slab prints the timeline of a few steps."""
import sys
sys.path.insert(0, ".")
from mara3_amd import setups
from mara3_amd.slab import NativeSlabStepper, native_comm_id
n0 = int(sys.argv[1]) if len(sys.argv) > 1 else 512
mode = sys.argv[2] if len(sys.argv) > 2 else "self-exchange"
n1, gamma = 4096, 5.0 / 3
import numpy as np
u0 = np.ascontiguousarray(setups.smooth_wave_ic((4096, n1), gamma)[:n0])          # planar rows (no third momentum), as bench.py's slabs
kw = dict(comm_id=native_comm_id(0, 1, device="cuda"), self_exchange=True) if mode == "self-exchange" else {}
st = NativeSlabStepper((n0, n1), (1.0 / 4096, 1.0 / 4096), gamma, 1.5, "hllc", 2, "periodic", arith="fast", planar=True, **kw)
st.load_slab(u0)
st.step(setups.baseline_dt(4096), 400, graph=False)
st.synchronize()
st.close()
