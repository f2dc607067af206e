"""Runs HERE (CPU, needs oracle/_ref built from /root/reference): long runs of the reference-composed Euler driver (oracle/ref_drivers/
euler_cart_ref.cpp - the reference's own headers) whose final states go to oracle/_ref/long/ (travels to the GPU box, never committed: 5-10 MB
each). scripts/long_euler_vs_reference.py compares the device against them; only hashes and distances are recorded under profiles/."""
import hashlib, json, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from mara3_amd import setups
OUT = os.path.join(ROOT, "oracle", "_ref", "long")
os.makedirs(OUT, exist_ok=True)
REF = os.path.join(ROOT, "oracle", "_ref", "euler_cart_ref")
gamma = 5.0 / 3
CASES = {
    "euler2d_blast512_plm15_rk2_1500steps": dict(shape=(512, 512), ic="blast", bc=0, theta=1.5, nsteps=1500),
    "euler2d_wave384_plm15_rk2_periodic_2000steps": dict(shape=(384, 384), ic="wave", bc=1, theta=1.5, nsteps=2000),
    "euler3d_blast96_plm15_rk2_250steps": dict(shape=(96, 96, 96), ic="blast", bc=0, theta=1.5, nsteps=250),
}
meta = {}
for name, c in CASES.items():
    shape = c["shape"]
    n = shape[0]
    u0 = setups.blast_ic(shape, gamma) if c["ic"] == "blast" else setups.smooth_wave_ic(shape, gamma)
    dl = [1.0 / s for s in shape] + [1.0] * (3 - len(shape))
    dt = setups.baseline_dt(n)
    fin, fout = os.path.join(OUT, name + ".in.f64"), os.path.join(OUT, name + ".f64")
    np.ascontiguousarray(u0, dtype=np.float64).tofile(fin)
    dims = list(shape) + [1] * (3 - len(shape))
    t0 = time.time()
    subprocess.check_call([REF, str(len(shape))] + [str(d) for d in dims] + [gamma.hex(), float(c["theta"]).hex(), "2", str(c["bc"]), dt.hex()] + [x.hex() for x in dl] + [str(c["nsteps"]), fin, fout])
    u = np.fromfile(fout, dtype=np.float64)
    meta[name] = dict(c, shape=list(shape), gamma=gamma, dt=dt, sha256=hashlib.sha256(u.tobytes()).hexdigest(), reference_cpu_seconds=round(time.time() - t0, 1))
    os.remove(fin)
    print(name, meta[name], flush=True)
json.dump(meta, open(os.path.join(OUT, "euler_long_cases.json"), "w"), indent=1)
