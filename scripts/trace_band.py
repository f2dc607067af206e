import sys, os
sys.path.insert(0, os.getcwd())
from mara3_amd import binary
from mara3_amd.slab import native_comm_id
cfg = binary.config(depth=5, block_size=64, fixed_dt=1, rk_order=2, plm_theta=1.8)
edge = None if sys.argv[1] == "default" else int(sys.argv[1])
s = binary.BinaryBand(cfg, 0, 1, native_comm_id(0, 1), arith="fast", self_exchange=True, edge_rows=edge)
s.next(12)
s.close()
