"""Runs HERE (CPU, minutes): the reference's own composition at the BASELINE sizes - config 2 (4096^2, 20 RK2 steps), config 4 (`cloud`
nr=4096, one decade: 4096 x 4096, 3 RK2 steps) and, memory permitting, config 5's per-GPU share - and prints the SHA-256 of the final states
(0.7 - 5 GB each; nothing is kept). usage: make_fullsize_reference_hashes.py c2|c4|c5|c5full [...]"""
import hashlib, json, os, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from mara3_amd import setups
gamma = 5.0 / 3


def sha_of_file(path):
    h = hashlib.sha256()
    with open(path, "rb") as f:
        for chunk in iter(lambda: f.read(1 << 26), b""):
            h.update(chunk)
    return h.hexdigest()


def euler(shape, nsteps, tmp):
    n = shape[0]
    fin, fout = os.path.join(tmp, "in.f64"), os.path.join(tmp, "out.f64")
    np.ascontiguousarray(setups.blast_ic(shape, gamma), dtype=np.float64).tofile(fin)
    dl = [1.0 / s for s in shape] + [1.0] * (3 - len(shape))
    dims = list(shape) + [1] * (3 - len(shape))
    dt = setups.baseline_dt(n)
    subprocess.check_call([os.path.join(ROOT, "oracle", "_ref", "euler_cart_ref"), str(len(shape))] + [str(d) for d in dims]
                          + [gamma.hex(), (1.5).hex(), "2", "0", dt.hex()] + [x.hex() for x in dl] + [str(nsteps), fin, fout])
    return dict(shape=list(shape), nsteps=nsteps, gamma=gamma, theta=1.5, dt=dt, sha256=sha_of_file(fout))


out = {}
for what in sys.argv[1:]:
    t0 = time.time()
    with tempfile.TemporaryDirectory(dir=os.path.join(ROOT, "gpurun_out")) as tmp:
        if what == "c2":
            out[what] = euler((4096, 4096), 20, tmp)
        elif what == "c5":
            out[what] = euler((384, 384, 384), 3, tmp)
        elif what == "c5full":          # config 5's per-GPU share: 134 M cells, ~30 GB of lazy-array temporaries here
            out[what] = euler((512, 512, 512), 1, tmp)
        elif what == "c4":
            subprocess.check_call([os.path.join(ROOT, "oracle", "_ref", "cloud_ref"), "4096", "1", "2", "2", "1.2", "3", os.path.join(tmp, "c")])
            out[what] = dict(args=["nr=4096", "num_decades=1", "rk_order=2", "reconstruct_method=2", "plm_theta=1.2", "max_steps=3"], shape=[4096, 4096], sha256=sha_of_file(os.path.join(tmp, "c.un.f64")))
    out[what]["reference_cpu_seconds"] = round(time.time() - t0, 1)
    print(what, json.dumps(out[what]), flush=True)
