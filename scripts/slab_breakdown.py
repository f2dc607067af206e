"""Dev helper: where a thin slab's step time goes. rows x 4096, fast HLLC RK2: (a) no neighbours, graph replay; (b) no neighbours,
eager; (c) RCCL self-exchange (edge/interior split + send/recv to self)."""
import sys, time
sys.path.insert(0, ".")
from mara3_amd import setups
from mara3_amd.slab import NativeSlabStepper, native_comm_id
n1, gamma = 4096, 5.0 / 3
dl = (1.0 / 4096, 1.0 / 4096)
for n0 in (512, 1024, 2048, 4096):
    u0 = setups.wave_ic((n0, n1), gamma, seed=1)
    res = {}
    for mode in ("graph", "eager", "self-exchange"):
        kw = dict(comm_id=native_comm_id(0, 1), self_exchange=True) if mode == "self-exchange" else {}
        st = NativeSlabStepper((n0, n1), dl, gamma, 1.5, "hllc", 2, "periodic", arith="fast", **kw)
        st.load_slab(u0)
        st.step(1e-5, 5, graph=(mode == "graph")); st.synchronize()
        t0 = time.perf_counter(); st.step(1e-5, 100, graph=(mode == "graph")); st.synchronize(); res[mode] = (time.perf_counter() - t0) / 100 * 1e3
        st.close()
    ideal = 0.772 * n0 / 4096
    print("rows=%4d  ideal(1/N of 0.772)=%.3f  graph=%.3f  eager=%.3f  self-exchange=%.3f ms/step  -> efficiency %.0f%% / %.0f%% / %.0f%%" %
          (n0, ideal, res["graph"], res["eager"], res["self-exchange"], 100 * ideal / res["graph"], 100 * ideal / res["eager"], 100 * ideal / res["self-exchange"]), flush=True)
