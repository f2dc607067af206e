"""Dev probe of the native slab stepper on one GPU: modes = none | self_eager | self_graph"""
import sys, faulthandler
faulthandler.enable()
sys.path.insert(0, ".")
import numpy as np
from mara3_amd import setups
from mara3_amd.slab import NativeSlabStepper, native_comm_id
from mara3_amd.engine import EulerCartSolver
mode = sys.argv[1]
shape, gamma = (256, 300), 1.4
dl = (1.0 / shape[0], 1.0 / shape[1])
u0 = setups.wave_ic(shape, gamma, seed=5)
bc = "periodic"
ref = EulerCartSolver(shape, dl, gamma, 1.5, "hlle", 2, bc)
ref.upload(u0); ref.step(1e-3, 6); want = ref.download()
print("ref done", flush=True)
kw = {}
if mode != "none":
    kw = dict(comm_id=native_comm_id(0, 1), self_exchange=True)
print("creating", flush=True)
st = NativeSlabStepper(shape, dl, gamma, 1.5, "hlle", 2, bc, **kw)
print("created rows", st.row0, st.row1, flush=True)
st.load_slab(u0)
print("loaded", flush=True)
st.step(1e-3, 6, graph=(mode.endswith("graph")))
st.synchronize()
print("stepped", flush=True)
got = st.slab_host()
print(mode, "bit-identical:", np.array_equal(got.view(np.uint64), want.view(np.uint64)), flush=True)
