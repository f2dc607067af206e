import sys, os
sys.path.insert(0, os.getcwd())
from mara3_amd import setups
from mara3_amd.slab import NativeSlabStepper, native_comm_id
n0 = int(sys.argv[1]) if len(sys.argv) > 1 else 512
u0 = setups.wave_ic((n0, 4096), 5.0 / 3, seed=1)
st = NativeSlabStepper((n0, 4096), (1.0 / 4096, 1.0 / 4096), 5.0 / 3, 1.5, "hllc", 2, "periodic", arith="fast", comm_id=native_comm_id(0, 1), self_exchange=True)
st.load_slab(u0)
st.step(1e-5, 12); st.synchronize()
st.close()
