"""Dev check (GPU; needs oracle/_ref/long/*.f64 from scripts/make_long_reference_states.py): the Euler path through thousands of steps against
final states of the reference's own composition - STRICT + HLLE must be bit-identical; FAST's distance is reported."""
import hashlib, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from mara3_amd import setups
from mara3_amd.engine import EulerCartSolver
LONG = os.path.join(ROOT, "oracle", "_ref", "long")
meta = json.load(open(os.path.join(LONG, "euler_long_cases.json")))
for name, c in meta.items():
    shape = tuple(c["shape"])
    ref = np.fromfile(os.path.join(LONG, name + ".f64"), dtype=np.float64).reshape(shape + (5,))
    assert hashlib.sha256(ref.tobytes()).hexdigest() == c["sha256"]
    u0 = setups.blast_ic(shape, c["gamma"]) if c["ic"] == "blast" else setups.smooth_wave_ic(shape, c["gamma"])
    out = {"case": name, "steps": c["nsteps"], "cells": int(np.prod(shape)), "reference_sha256": c["sha256"], "reference_cpu_seconds": c["reference_cpu_seconds"]}
    for arith in ("strict", "fast"):
        s = EulerCartSolver(shape, tuple(1.0 / n for n in shape), c["gamma"], c["theta"], "hlle", 2, "periodic" if c["bc"] else "outflow", arith=arith)
        s.upload(u0)
        s.step(c["dt"], c["nsteps"])
        u = s.download()
        same = (u.view(np.uint64) == ref.view(np.uint64)) | ((u == 0) & (ref == 0))
        out[arith] = {"status": int(s.status()), "sha256": hashlib.sha256(u.tobytes()).hexdigest(), "bit_identical_cells": int(same.all(axis=-1).sum()),
                      "l1_rel": float(np.abs(u - ref).mean() / np.abs(ref).mean()), "max_rel": float(np.abs(u - ref).max() / np.abs(ref).max())}
    print(json.dumps(out), flush=True)
