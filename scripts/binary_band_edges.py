#!/usr/bin/env python3
"""Dev measurement on ONE GPU: BASELINE config 3 (binary 2048^2, FAST) as one band whose ghost rows travel through RCCL to itself
(self_exchange) - one launch per stage with the exchange behind it (edge_rows = 0, rounds 1-2) against edge rows first with the
exchange on a second stream beside the interior launch (round 3) - and as 2 / 4 / 8 loopback bands with and without the cut.
One JSON line per variant. usage: python scripts/binary_band_edges.py [steps=60]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mara3_amd import binary
from mara3_amd.slab import native_comm_id
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 60
cfg = binary.config(depth=5, block_size=64, fixed_dt=1, rk_order=2, plm_theta=1.8)
n = binary.grid_size(cfg)


def timed(s):
    s.next(10)
    best = 1e9
    for rep in range(3):
        t0 = time.perf_counter(); s.next(steps); best = min(best, (time.perf_counter() - t0) / steps * 1e3)
    s.close()
    return round(best, 4)


one = binary.BinarySolver(cfg, arith="fast")
print(json.dumps({"variant": "single domain (eager first stage)", "ms_per_step": timed(one)}), flush=True)
for edge in (0, -1, 4, 8):
    s = binary.BinaryBand(cfg, 0, 1, native_comm_id(0, 1), arith="fast", self_exchange=True, edge_rows=edge)
    print(json.dumps({"variant": "one band, RCCL exchange to self", "edge_rows": "recommended" if edge == -1 else edge, "ms_per_step": timed(s)}), flush=True)
for world in (2, 4, 8):
    for edge in (0, -1):
        g = binary.BinaryBandGroup(cfg, world=world, arith="fast", edge_rows=edge)
        print(json.dumps({"variant": "%d loopback bands" % world, "edge_rows": "recommended" if edge == -1 else edge, "ms_per_step": timed(g)}), flush=True)
