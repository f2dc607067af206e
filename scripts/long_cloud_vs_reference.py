"""Dev check (needs oracle/_ref/long/cloud_nr512_rk2_plm12_388steps.un.f64, made here from the reference's own headers by
`oracle/_ref/cloud_ref 512 2 2 2 1.2 388 <prefix>`, 7 CPU-minutes; the file travels to the GPU box with oracle/_ref but is not committed):
`mara_hip cloud nr=512 rk_order=2 reconstruct_method=2 max_steps=388 arith=strict` - the 1024 x 512 jet-cloud problem through the 388 RK2
steps that precede the reference's `recover_primitive failure: negative density` - against that state, bit for bit."""
import hashlib, json, os, struct, subprocess, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "mara3_amd", "host", "mara_hip")
REF = os.path.join(ROOT, "oracle", "_ref", "long", "cloud_nr512_rk2_plm12_388steps.un.f64")
ref = np.fromfile(REF, dtype=np.float64).reshape(1024, 512, 5)
out = {"reference_state_sha256": hashlib.sha256(ref.tobytes()).hexdigest(), "cells_with_nonpositive_D_in_reference_state": int((ref[..., 0] <= 0).sum())}
with tempfile.TemporaryDirectory() as tmp:
    for arith in ("strict", "fast"):
        p = subprocess.run([EXE, "cloud", "nr=512", "rk_order=2", "reconstruct_method=2", "max_steps=388", "cpi=0", "outdir=o", "arith=" + arith], cwd=tmp, capture_output=True, text=True, timeout=900)
        assert p.returncode == 0, (p.stdout + p.stderr)[-400:]
        raw = open(os.path.join(tmp, "o", "final.bin"), "rb").read()
        off = 8; rank = struct.unpack_from("q", raw, 0)[0]
        shape = struct.unpack_from("%dq" % rank, raw, off); off += 8 * rank + 8 + 8 + 8
        nv = struct.unpack_from("q", raw, off)[0]; off += 8 + 8 * nv
        u = np.frombuffer(raw, dtype=np.float64, offset=off).reshape(tuple(shape) + (5,))
        same = (u.view(np.uint64) == ref.view(np.uint64)) | ((u == 0) & (ref == 0))
        out[arith] = {"state_sha256": hashlib.sha256(u.tobytes()).hexdigest(), "bit_identical_cells": int(same.all(axis=-1).sum()), "cells": int(u.shape[0] * u.shape[1]),
                      "l1_rel": [float(np.abs(u[..., q] - ref[..., q]).mean() / np.abs(ref[..., q]).mean()) for q in (0, 1, 2, 4)]}
print(json.dumps(out))
