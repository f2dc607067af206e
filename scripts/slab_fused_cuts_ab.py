#!/usr/bin/env python3
"""Dev measurement on ONE GPU: one rank's slab of the 4096-wide grid (512 / 1024 / 2048 rows, FAST HLLC RK2) with its ghost rows travelling
through RCCL to itself - the two-launch schedule (an exchange of two rows per stage, edges first on a second stream;
MH_SLAB_FUSED_CUTS=0) against the fused step across cuts (four rows, one exchange per step, round 3) - and the same slab without
neighbours (no exchange at all) for scale. Also 2 / 4 / 8 loopback slabs of the whole 4096^2. us per step, one JSON line per case.
usage: python scripts/slab_fused_cuts_ab.py"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mara3_amd import setups
from mara3_amd.slab import NativeSlabStepper, NativeSlabGroup, native_comm_id
n1, gamma = 4096, 5.0 / 3
dl = (1.0 / 4096, 1.0 / 4096)


def timed(st, steps=300):
    st.step(1e-5, 30); st.synchronize()
    best = 1e9
    for rep in range(3):
        t0 = time.perf_counter(); st.step(1e-5, steps); st.synchronize(); best = min(best, (time.perf_counter() - t0) / steps * 1e6)
    st.close()
    return round(best, 1)


for n0 in (512, 1024, 2048):
    u0 = setups.wave_ic((n0, n1), gamma, seed=1)
    line = {"rows": n0}
    for name, env, kw in (("no_neighbours_fused", None, {}), ("two_launch_exchange_per_stage", "0", dict(comm_id=native_comm_id(0, 1), self_exchange=True)),
                          ("fused_across_cuts_one_exchange_per_step", "1", dict(comm_id=native_comm_id(0, 1), self_exchange=True))):
        if env is not None:
            os.environ["MH_SLAB_FUSED_CUTS"] = env
        st = NativeSlabStepper((n0, n1), dl, gamma, 1.5, "hllc", 2, "periodic", arith="fast", **kw)
        st.load_slab(u0)
        line[name + "_us_per_step"] = timed(st)
    print(json.dumps(line), flush=True)
u0 = setups.blast_ic((4096, 4096), gamma)
for world in (2, 4, 8):
    line = {"loopback_slabs_of_4096x4096": world}
    for name, env in (("two_launch", "0"), ("fused_across_cuts", "1")):
        os.environ["MH_SLAB_FUSED_CUTS"] = env
        g = NativeSlabGroup((4096, 4096), dl, gamma, 1.5, "hllc", 2, "outflow", world=world, arith="fast")
        g.upload(u0)
        line[name + "_us_per_step"] = timed(g, 100)
    print(json.dumps(line), flush=True)
