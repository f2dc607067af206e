"""Dev check: `mara_hip cloud` run to its default end time with STRICT and with FAST arithmetic (RK2, PLM), final states compared: the FAST
Newton iteration, limiter and pole handling over thousands of steps of the jet-cloud problem."""
import json, os, struct, subprocess, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
EXE = os.path.join(ROOT, "mara3_amd", "host", "mara_hip")


def read_dump(path):
    raw = open(path, "rb").read()
    off = 0
    (rank,) = struct.unpack_from("q", raw, off); off += 8
    shape = struct.unpack_from("%dq" % rank, raw, off); off += 8 * rank
    (nq,) = struct.unpack_from("q", raw, off); off += 8
    (time,) = struct.unpack_from("d", raw, off); off += 8
    (iteration,) = struct.unpack_from("q", raw, off); off += 8
    (nv,) = struct.unpack_from("q", raw, off); off += 8
    off += 8 * nv
    return time, iteration, np.frombuffer(raw, dtype=np.float64, offset=off).reshape(tuple(shape) + (nq,))


for nr, extra in ((256, []), (384, []), (512, [])):
    res, threw = {}, {}
    with tempfile.TemporaryDirectory() as tmp:
        for arith in ("strict", "fast"):
            p = subprocess.run([EXE, "cloud", "nr=%d" % nr, "rk_order=2", "reconstruct_method=2", "cpi=0", "outdir=" + arith, "arith=" + arith] + extra,
                               cwd=tmp, capture_output=True, text=True, timeout=900)
            steps = p.stdout.count("kzps=")
            if p.returncode != 0:
                # where the reference throws (physics_srhd.hpp:430-449) the host ends the run with the same message; STRICT is bit-identical to
                # the reference's arithmetic, so its failing step and cell are the reference's
                threw[arith] = {"after_steps": steps, "message": (p.stdout + p.stderr).strip().splitlines()[-1][:160]}
            else:
                res[arith] = read_dump(os.path.join(tmp, arith, "final.bin"))
    out = {"nr": nr, "extra": extra}
    if threw:
        out["threw"] = threw
    if len(res) == 2:
        (ts, its, a), (tf, itf, b) = res["strict"], res["fast"]
        out.update({"shape": list(a.shape), "iterations": [int(its), int(itf)], "time": [ts, tf],
                    "l1_rel_D_Sr_Sq_tau": [float(np.abs(a[..., q] - b[..., q]).mean() / np.abs(a[..., q]).mean()) for q in (0, 1, 2, 4)]})
    print(json.dumps(out), flush=True)
