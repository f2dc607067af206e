"""Rehearsal of the RCCL send/recv + two-stream overlap path on ONE GPU: a periodic domain whose wrap-around
is done by the rank sending its edge rows to itself. Must equal the kernel's local periodic handling bit for bit."""
import os, sys
sys.path.insert(0, ".")
import numpy as np, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
from mara3_amd import setups
from mara3_amd.slab import SlabEulerStepper, TorchDistExchange
shape, gamma = (256, 300), 1.4
dl = (1.0 / shape[0], 1.0 / shape[1])
u0 = setups.wave_ic(shape, gamma, seed=3)
ref = SlabEulerStepper(shape, dl, gamma, 1.5, "hlle", 2, "periodic")
ref.load_slab(u0); ref.step(1e-3, 5)
for overlap in (False, True):
    st = SlabEulerStepper(shape, dl, gamma, 1.5, "hlle", 2, "periodic", exchange=TorchDistExchange(0, 1, True, self_exchange=True), overlap=overlap)
    assert st.has_neighbours and st.desc.bc_lo0 == 2 and st.desc.bc_hi0 == 2
    st.load_slab(u0); st.step(1e-3, 5)
    torch.cuda.synchronize()
    same = torch.equal(st.slab(), ref.slab())
    print("overlap=%s bit-identical=%s" % (overlap, same), flush=True)
    assert same
dist.barrier(); dist.destroy_process_group()
print("OK")
