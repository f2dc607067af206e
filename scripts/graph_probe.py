"""Dev probe: which part of the step can be captured into a HIP graph on this stack?"""
import os, sys, faulthandler
faulthandler.enable()
sys.path.insert(0, ".")
import torch, torch.distributed as dist
from mara3_amd import setups
from mara3_amd.slab import SlabEulerStepper, TorchDistExchange
mode = sys.argv[1]
shape, gamma = (192, 250), 1.4
dl = (1.0 / shape[0], 1.0 / shape[1])
u0 = setups.wave_ic(shape, gamma, seed=8)
torch.cuda.set_device(0)
if mode != "local":
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29544")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
kw = {}
if mode == "p2p_nooverlap":
    kw = dict(exchange=TorchDistExchange(0, 1, True, self_exchange=True), overlap=False)
if mode == "p2p_overlap":
    kw = dict(exchange=TorchDistExchange(0, 1, True, self_exchange=True), overlap=True)
st = SlabEulerStepper(shape, dl, gamma, 1.5, "hllc", 2, "periodic", arith="fast", **kw)
st.load_slab(u0)
st.step(1e-3, 2)
torch.cuda.synchronize()
print("capturing", mode, flush=True)
st.capture(1e-3)
print("captured", flush=True)
st.step(1e-3, 3)
torch.cuda.synchronize()
print("replayed OK", mode, flush=True)
