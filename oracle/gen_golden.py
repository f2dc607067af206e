#!/usr/bin/env python3
"""TEST INFRASTRUCTURE — generate tests/golden/*.npz from the REFERENCE.

Runs the reference-composed drivers in oracle/_ref/ (built by oracle/Makefile
from the reference's own headers under /root/reference/src) on seeded
synthetic inputs and stores {inputs, parameters, expected outputs} as small
.npz fixtures. Only data is committed; no reference source.

Run in the build container only (needs /root/reference):  python oracle/gen_golden.py
"""
import os
import subprocess
import sys
import tempfile
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.path.join(HERE, "_ref")
OUT = os.environ.get("MARA_GOLDEN_OUT") or os.path.join(os.path.dirname(HERE), "tests", "golden")      # MARA_GOLDEN_OUT: regenerate elsewhere, to compare


def hexf(x):
    return float(x).hex()


def run_ref(exe, args):
    subprocess.check_call([os.path.join(REF, exe)] + [str(a) for a in args])


def ref_func(mode, n, a0, a1, data, out_dtype=np.float64):
    with tempfile.TemporaryDirectory() as d:
        fin, fout = os.path.join(d, "in"), os.path.join(d, "out")
        if data is None:
            fin = "-"
        else:
            np.ascontiguousarray(data, dtype=np.float64).tofile(fin)
        run_ref("funcs_ref", [mode, n, hexf(a0), hexf(a1), fin, fout])
        return np.fromfile(fout, dtype=out_dtype)


def ref_euler_cart(u0, dl, dt, nsteps, gamma, theta, rk, bc):
    rank = u0.ndim - 1
    shape = list(u0.shape[:-1]) + [1] * (3 - rank)
    dl3 = list(dl) + [1.0] * (3 - rank)
    with tempfile.TemporaryDirectory() as d:
        fin, fout = os.path.join(d, "in"), os.path.join(d, "out")
        np.ascontiguousarray(u0, dtype=np.float64).tofile(fin)
        run_ref("euler_cart_ref", [rank, *shape, hexf(gamma), hexf(theta), rk, bc, hexf(dt), *map(hexf, dl3), nsteps, fin, fout])
        return np.fromfile(fout).reshape(u0.shape)


# ---- synthetic initial conditions (also used by tests and bench; keep in sync with mara3_amd/setups.py) ----
def blast_ic(shape, gamma, radius=0.1, p_in=10.0, p_out=0.1):
    """SURVEY.md §8(d) primary IC: rho=1, v=0, p = p_in inside r<radius of the centre else p_out, on [0,1]^rank."""
    rank = len(shape)
    axes = [(np.arange(n) + 0.5) / n for n in shape]
    X = np.meshgrid(*axes, indexing="ij")
    r2 = sum((x - 0.5) ** 2 for x in X)
    p = np.where(r2 < radius * radius, p_in, p_out)
    u = np.zeros(tuple(shape) + (5,))
    u[..., 0] = 1.0
    u[..., 4] = p / (gamma - 1.0)
    return u


def wave_ic(shape, gamma, seed=0):
    """Smooth isentropic wave with all three velocity components non-zero plus seeded noise (periodic tests)."""
    rng = np.random.default_rng(seed)
    rank = len(shape)
    axes = [(np.arange(n) + 0.5) / n for n in shape]
    X = np.meshgrid(*axes, indexing="ij")
    s = np.ones(shape)
    for x in X:
        s = s * np.sin(2 * np.pi * x)
    d = 1.0 + 0.2 * s + 0.01 * rng.standard_normal(shape)
    p = d ** gamma
    v = [0.5 + 0.1 * s, -0.25 + 0.05 * np.cos(2 * np.pi * X[0]), 0.125 * np.ones(shape) + 0.01 * rng.standard_normal(shape)]
    u = np.zeros(tuple(shape) + (5,))
    u[..., 0] = d
    for k in range(3):
        u[..., 1 + k] = d * v[k]
    u[..., 4] = 0.5 * d * (v[0] ** 2 + v[1] ** 2 + v[2] ** 2) + p / (gamma - 1.0)
    return u


def random_prims(rng, n):
    P = np.empty((n, 5))
    P[:, 0] = 10.0 ** rng.uniform(-3, 3, n)
    P[:, 1:4] = rng.standard_normal((n, 3)) * 10.0 ** rng.uniform(-2, 2, (n, 1))
    P[:, 4] = 10.0 ** rng.uniform(-4, 3, n)
    return P


def ref_srhd(mode, n, a0, a1, data):
    with tempfile.TemporaryDirectory() as d:
        fin, fout = os.path.join(d, "in"), os.path.join(d, "out")
        np.ascontiguousarray(data, dtype=np.float64).tofile(fin)
        run_ref("funcs_srhd_ref", [mode, n, hexf(a0), hexf(a1), fin, fout])
        return np.fromfile(fout)


def ref_cloud(nr, num_decades, rk, method, theta, nsteps):
    with tempfile.TemporaryDirectory() as d:
        prefix = os.path.join(d, "c")
        run_ref("cloud_ref", [nr, hexf(num_decades), rk, method, hexf(theta), nsteps, prefix])
        rv, qv = np.fromfile(prefix + ".rv.f64"), np.fromfile(prefix + ".qv.f64")
        n0, n1 = rv.size - 1, qv.size - 1
        meta = np.fromfile(prefix + ".meta.f64")
        return dict(rv=rv, qv=qv, u0=np.fromfile(prefix + ".u0.f64").reshape(n0, n1, 5),
                    un=np.fromfile(prefix + ".un.f64").reshape(n0, n1, 5),
                    inflow=np.fromfile(prefix + ".inflow.f64").reshape(nsteps, n1, 5), dt=meta[0], tfloor=meta[1],
                    rk=rk, method=method, theta=theta, nsteps=nsteps)


def gen_cloud_diag():
    """CloudProblem::make_diagnostic_fields of evolved states (the jet has entered the grid in the longer case)."""
    for name, (nr, num_decades, rk, method, theta, nsteps) in (("clouddiag_nr48_150steps", (48, 1.0, 2, 2, 1.2, 150)),
                                                             ("clouddiag_nr40_pcm_60steps", (40, 1.0, 1, 1, 1.2, 60)),
                                                             ("clouddiag_nr32_4steps", (32, 1.0, 1, 2, 1.2, 4))):
        with tempfile.TemporaryDirectory() as d:
            prefix = os.path.join(d, "c")
            run_ref("cloud_ref", [nr, hexf(num_decades), rk, method, hexf(theta), nsteps, prefix])
            rv, qv = np.fromfile(prefix + ".rv.f64"), np.fromfile(prefix + ".qv.f64")
            n0, n1 = rv.size - 1, qv.size - 1
            meta = np.fromfile(prefix + ".meta.f64")
            out = dict(rv=rv, qv=qv, un=np.fromfile(prefix + ".un.f64").reshape(n0, n1, 5), tfloor=meta[1], nsteps=nsteps, dt=meta[0],
                       diag_fields=np.fromfile(prefix + ".diag_fields.f64").reshape(5, n0, n1),
                       diag_columns=np.fromfile(prefix + ".diag_columns.f64").reshape(15, n1),
                       diag_meta=np.fromfile(prefix + ".diag_meta.f64"))
            np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
            print(name, "ok", n0, n1, "shock radii (cm)", out["diag_columns"][2, ::max(1, n1 // 4)])
    return 0


def random_srhd_prims(rng, n):
    P = np.empty((n, 5))
    P[:, 0] = 10.0 ** rng.uniform(-3, 2, n)
    u = 10.0 ** rng.uniform(-3, 1.3, n)                       # |gamma-beta| up to ~20
    dirs = rng.standard_normal((n, 3))
    dirs /= np.linalg.norm(dirs, axis=1, keepdims=True)
    P[:, 1:4] = dirs * u[:, None]
    P[:, 4] = P[:, 0] * 10.0 ** rng.uniform(-6, 1, n)         # cold to relativistically hot
    return P


def gen_srhd(rng):
    n = 4096
    g = 4.0 / 3
    P, Pr = random_srhd_prims(rng, n), random_srhd_prims(rng, n)
    Pr[:512] = P[:512] * (1.0 + 1e-3 * rng.standard_normal((512, 5)))
    P[512:520, 1:4] = 0.0
    out = {"Pl": P, "Pr": Pr}
    U = ref_srhd("p2c", n, g, 0, P).reshape(n, 5)
    out["U"] = U
    r = ref_srhd("c2p", n, g, 0.0, U).reshape(n, 6)
    out["c2p"], out["c2p_threw"] = r[:, :5], r[:, 5]
    # floor-active and failing cases: scale tau down so that the recovered pressure is tiny / negative
    Ubad = U.copy()
    Ubad[:, 4] *= rng.uniform(0.0, 1.0, n) ** 4
    out["Ubad"] = Ubad
    for name, floor in (("floor", 1e-8), ("nofloor", 0.0)):
        r = ref_srhd("c2p", n, g, floor, Ubad).reshape(n, 6)
        out["c2p_bad_" + name], out["c2p_bad_" + name + "_threw"] = r[:, :5], r[:, 5]
    for axis in range(3):
        out["hlle_%d" % axis] = ref_srhd("hlle", n, g, axis, np.hstack([P, Pr])).reshape(n, 5)
        out["lam_%d" % axis] = ref_srhd("lam", n, g, axis, P).reshape(n, 2)
    rr = 10.0 ** rng.uniform(0, 2, n)
    qq = rng.uniform(1e-3, np.pi - 1e-3, n)
    out["src_r"], out["src_q"] = rr, qq
    out["src"] = ref_srhd("src", n, g, 0, np.hstack([P, rr[:, None], qq[:, None]])).reshape(n, 5)
    np.savez_compressed(os.path.join(OUT, "srhd_functions.npz"), **out)
    print("srhd functions ok; c2p throws:", int(out["c2p_threw"].sum()), int(out["c2p_bad_floor_threw"].sum()), int(out["c2p_bad_nofloor_threw"].sum()))

    for name, args in (("cloud_nr32_plm_rk2", (32, 1.0, 2, 2, 1.2, 3)), ("cloud_nr32_plm_rk1", (32, 1.0, 1, 2, 1.2, 4)),
                       ("cloud_nr24_pcm_rk1", (24, 1.0, 1, 1, 1.2, 4)), ("cloud_nr20x2dec_plm_rk2", (20, 2.0, 2, 2, 1.5, 2)),
                       ("cloud_nr70_plm_rk2", (70, 1.0, 2, 2, 1.2, 2))):
        np.savez_compressed(os.path.join(OUT, name + ".npz"), **ref_cloud(*args))
        print(name, "ok")


def ref_iso2d(mode, n, axis, data):
    with tempfile.TemporaryDirectory() as d:
        fin, fout = os.path.join(d, "in"), os.path.join(d, "out")
        np.ascontiguousarray(data, dtype=np.float64).tofile(fin)
        run_ref("funcs_iso2d_ref", [mode, n, axis, fin, fout])
        return np.fromfile(fout)


def gen_iso2d(rng):
    n = 4096
    def prims(n):
        P = np.empty((n, 3))
        P[:, 0] = 10.0 ** rng.uniform(-4, 2, n)
        P[:, 1:] = rng.standard_normal((n, 2)) * 10.0 ** rng.uniform(-2, 1.5, (n, 1))
        return P
    Pl, Pr = prims(n), prims(n)
    Pr[:512] = Pl[:512] * (1.0 + 1e-3 * rng.standard_normal((512, 3)))
    # the reference's own known-answer test, src/physics_test.cpp:143-153: contact speed exactly 0
    Pl[0], Pr[0] = (1.0, 0.0, 0.0), (2.0, 0.0, 0.0)
    cs2l, cs2r = 10.0 ** rng.uniform(-3, 1, n), 10.0 ** rng.uniform(-3, 1, n)
    cs2r[:512] = cs2l[:512] * (1.0 + 1e-3 * rng.standard_normal(512))
    cs2l[0], cs2r[0] = 1.0, 0.5
    x = rng.standard_normal((n, 2)) * 10.0 ** rng.uniform(-2, 1, (n, 1))
    x[1], x[2] = (1.0, 2.0), (1e-8, 1e-8)                       # src/physics_test.cpp:115-141
    out = {"Pl": Pl, "Pr": Pr, "cs2l": cs2l, "cs2r": cs2r, "x": x}
    out["U"] = ref_iso2d("p2c", n, 0, Pl).reshape(n, 3)
    r = ref_iso2d("c2p", n, 0, out["U"]).reshape(n, 4)
    out["c2p"], out["c2p_threw"] = r[:, :3], r[:, 3]
    Uneg = out["U"].copy()
    Uneg[::7, 0] *= -1.0
    out["Uneg"] = Uneg
    r = ref_iso2d("c2p", n, 0, Uneg).reshape(n, 4)
    out["c2p_neg"], out["c2p_neg_threw"] = r[:, :3], r[:, 3]
    out["Q"] = ref_iso2d("p2q", n, 0, np.hstack([Pl, x])).reshape(n, 3)
    r = ref_iso2d("q2p", n, 0, np.hstack([out["Q"], x])).reshape(n, 4)
    out["q2p"], out["q2p_threw"] = r[:, :3], r[:, 3]
    both = np.hstack([Pl, Pr, cs2l[:, None], cs2r[:, None]])
    for axis in range(2):
        out["flux_%d" % axis] = ref_iso2d("flux", n, axis, np.hstack([Pl, cs2l[:, None]])).reshape(n, 3)
        out["lam_%d" % axis] = ref_iso2d("lam", n, axis, np.hstack([Pl, cs2l[:, None]])).reshape(n, 3)
        out["hlle_%d" % axis] = ref_iso2d("hlle", n, axis, both).reshape(n, 3)
        r = ref_iso2d("hllc", n, axis, both).reshape(n, 5)
        out["hllc_%d" % axis], out["hllc_contact_%d" % axis], out["hllc_threw_%d" % axis] = r[:, :3], r[:, 3], r[:, 4]
    assert out["hllc_contact_0"][0] == 0.0
    np.savez_compressed(os.path.join(OUT, "iso2d_functions.npz"), **out)
    print("iso2d ok; c2p_neg throws", int(out["c2p_neg_threw"].sum()), "hllc throws", int(out["hllc_threw_0"].sum()))


BINARY_CASES = {
    # name: overrides of the sub-program's configuration (src/subprog_binary.cpp:55-99) + nsteps / safe_mode
    "binary_d2_b16": dict(depth=2, block_size=16, domain_radius=4.0, nsteps=3),
    "binary_d1_b24_nu": dict(depth=1, block_size=24, domain_radius=3.0, fixed_dt=1, nu=1e-3, alpha_cutoff_radius=1.5, mass_ratio=0.5,
                             eccentricity=0.3, density_floor=0.05, rk_order=1, nsteps=4, sink_radius=0.2, softening_radius=0.1),
    "binary_d3_b8_axisym": dict(depth=3, block_size=8, domain_radius=6.0, axisymmetric_cs2=1, counter_rotate=1, mdot=1e-5, nsteps=2,
                                no_accretion_force=1, plm_theta=1.2),
    "binary_d2_b16_safe": dict(depth=2, block_size=16, domain_radius=4.0, nsteps=1, safe_mode=1),
    "binary_d2_b32": dict(depth=2, block_size=32, nsteps=2),
    # the binary is live from the second stage on: orbital elements evolve and feed back into the next stage's body positions
    # the angular-momentum conserving scheme, advance_q (conserve_linear_p = 0)
    "binary_d2_b16_q": dict(depth=2, block_size=16, domain_radius=4.0, conserve_linear_p=0, nsteps=3),
    "binary_d1_b24_q_nu": dict(depth=1, block_size=24, domain_radius=3.0, conserve_linear_p=0, fixed_dt=1, nu=1e-3, mass_ratio=0.5, eccentricity=0.3,
                               rk_order=1, nsteps=4, sink_radius=0.2, softening_radius=0.1, source_term_softening=4.0, axisymmetric_cs2=1),
    "binary_d2_b16_live": dict(depth=2, block_size=16, domain_radius=4.0, begin_live_binary=0.0, mass_ratio=0.7, eccentricity=0.2,
                               disk_mass=1e-2, nsteps=3),
}


def gen_binary():
    import json
    for name, cfg in BINARY_CASES.items():
        with tempfile.TemporaryDirectory() as d:
            prefix = os.path.join(d, "b")
            run_ref("binary_ref", [prefix] + ["%s=%s" % (k, repr(float(v))) for k, v in cfg.items()])
            n = int(cfg["block_size"]) << int(cfg["depth"])
            out = {"config": np.array(json.dumps(cfg))}
            for key, shape in (("xv", (n + 1,)), ("yv", (n + 1,)), ("u_init", (n, n, 3)), ("br", (n, n)), ("u_stage", (n, n, 3)),
                               ("stage_scalars", (-1,)), ("u_final", (n, n, 3)), ("scalars", (-1,))):
                out[key] = np.fromfile(prefix + "." + key + ".f64").reshape(shape)
            np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
            print(name, "ok", n, "dt", out["stage_scalars"][0])
    return 0


BINARY_TREE_CASES = {
    # graded block trees (refinement towards the origin: prolonged / restricted guard zones, flux correction)
    "binary_tree_d3_b8": dict(depth=3, block_size=8, domain_radius=4.0, nsteps=2),
    "binary_tree_d4_b8_default_focus": dict(depth=4, block_size=8, nsteps=1),
    "binary_tree_d3_b12_nu": dict(depth=3, block_size=12, domain_radius=5.0, focus_factor=1.5, nu=1e-3, mass_ratio=0.5, eccentricity=0.3, rk_order=1,
                                  fixed_dt=1, nsteps=3, sink_radius=0.2, softening_radius=0.1, density_floor=0.05),
    "binary_tree_d3_b8_q": dict(depth=3, block_size=8, domain_radius=4.0, conserve_linear_p=0, nsteps=2),
    "binary_tree_d2_b16_q_uniform": dict(depth=2, block_size=16, domain_radius=4.0, focus_factor=1e9, conserve_linear_p=0, nsteps=3),   # == binary_d2_b16_q
    "binary_tree_d2_b16_uniform": dict(depth=2, block_size=16, domain_radius=4.0, focus_factor=1e9, nsteps=3),     # == binary_d2_b16, through the tree machinery
}


def gen_binary_tree():
    import json
    for name, cfg in BINARY_TREE_CASES.items():
        with tempfile.TemporaryDirectory() as d:
            prefix = os.path.join(d, "b")
            run_ref("binary_tree_ref", [prefix] + ["%s=%s" % (k, repr(float(v))) for k, v in cfg.items()])
            bs = int(cfg["block_size"])
            blocks = np.fromfile(prefix + ".blocks.i32", dtype=np.int32).reshape(-1, 3)
            nb = len(blocks)
            out = {"config": np.array(json.dumps(cfg)), "blocks": blocks}
            for key, shape in (("xv", (nb, 2, bs + 1)), ("u_init", (nb, bs, bs, 3)), ("br", (nb, bs, bs)), ("u_stage", (nb, bs, bs, 3)),
                               ("stage_scalars", (-1,)), ("u_final", (nb, bs, bs, 3)), ("scalars", (-1,)),
                               # diagnostics of u_final (subprog_binary_diagnostics.cpp): [disk_mass, disk_angular_momentum, x1, y1, x2, y2]; sigma, v_r, v_phi
                               ("diag_scalars", (6,)), ("diag_fields", (nb, 3, bs, bs))):
                out[key] = np.fromfile(prefix + "." + key + ".f64").reshape(shape)
            np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
            print(name, "ok", nb, "blocks; levels", np.bincount(blocks[:, 0]), "dt", out["stage_scalars"][0])
    return 0


def gen_binary_long():
    """Mode `binary_long`: the three long / full-size `binary` fixtures (35 + 23 + 47 CPU-seconds of the reference-composed drivers).
    binary_d4_b16_r8_200steps      256^2 uniform tree, 200 CFL-limited steps: final state and scalars           (tests/test_gpu_binary.py)
    binary_tree_default_200steps   the sub-program's default graded tree, 200 steps: mesh, solver data, final state (tests/test_gpu_binary_tree.py)
    binary_c3_fullsize_2steps_digest   BASELINE config 3 (2048^2), 2 steps; the 100 MB state as a digest: block means, one cell per block,
                                   the 48 x 48 cells around the binary, the field scale                           (tests/test_gpu_binary.py)"""
    import json

    def args_of(cfg):
        return ["%s=%s" % (k, repr(float(v))) for k, v in cfg.items()]

    cfg = dict(depth=4, block_size=16, domain_radius=8.0, nsteps=200)
    with tempfile.TemporaryDirectory() as d:
        prefix = os.path.join(d, "b")
        run_ref("binary_ref", [prefix] + args_of(cfg))
        n = cfg["block_size"] << cfg["depth"]
        np.savez_compressed(os.path.join(OUT, "binary_d4_b16_r8_200steps.npz"), config=np.array(json.dumps(cfg)),
                            stage_scalars=np.fromfile(prefix + ".stage_scalars.f64"), u_final=np.fromfile(prefix + ".u_final.f64").reshape(n, n, 3),
                            scalars=np.fromfile(prefix + ".scalars.f64"))
        print("binary_d4_b16_r8_200steps ok")

    cfg = dict(nsteps=200)
    with tempfile.TemporaryDirectory() as d:
        prefix = os.path.join(d, "b")
        run_ref("binary_tree_ref", [prefix] + args_of(cfg))
        bs = 24                                                        # the sub-program's default block_size (src/subprog_binary.cpp:57-99)
        blocks = np.fromfile(prefix + ".blocks.i32", dtype=np.int32).reshape(-1, 3)
        nb = len(blocks)
        out = {"config": np.array(json.dumps(cfg)), "blocks": blocks}
        for key, shape in (("xv", (nb, 2, bs + 1)), ("u_init", (nb, bs, bs, 3)), ("br", (nb, bs, bs)), ("stage_scalars", (-1,)),
                           ("u_final", (nb, bs, bs, 3)), ("scalars", (-1,))):
            out[key] = np.fromfile(prefix + "." + key + ".f64").reshape(shape)
        np.savez_compressed(os.path.join(OUT, "binary_tree_default_200steps.npz"), **out)
        print("binary_tree_default_200steps ok", nb, "blocks")

    cfg = dict(depth=5, block_size=64, nsteps=2)
    with tempfile.TemporaryDirectory() as d:
        prefix = os.path.join(d, "b")
        run_ref("binary_ref", [prefix] + args_of(cfg))
        n, bs = cfg["block_size"] << cfg["depth"], cfg["block_size"]
        u = np.fromfile(prefix + ".u_final.f64").reshape(n, n, 3)
        np.savez_compressed(os.path.join(OUT, "binary_c3_fullsize_2steps_digest.npz"), config=np.array(json.dumps(cfg)),
                            block_means=u.reshape(n // bs, bs, n // bs, bs, 3).mean(axis=(1, 3)), samples=u[31::64, 17::64].copy(),
                            centre=u[n // 2 - 24:n // 2 + 24, n // 2 - 24:n // 2 + 24].copy(), scale=np.abs(u).reshape(-1, 3).max(axis=0),
                            scalars=np.fromfile(prefix + ".scalars.f64"), stage_scalars=np.fromfile(prefix + ".stage_scalars.f64"))
        print("binary_c3_fullsize_2steps_digest ok")
    return 0


def main():
    os.makedirs(OUT, exist_ok=True)
    if len(sys.argv) > 1 and sys.argv[1] == "binary_long":
        return gen_binary_long()
    if len(sys.argv) > 1 and sys.argv[1] == "sedov_srhd":
        with tempfile.TemporaryDirectory() as d:
            sed = {}
            for ns in (1, 10, 100):
                fv, f0, fn = (os.path.join(d, x) for x in ("v", "u0", "un"))
                run_ref("sedov_ref", [256, hexf(100.0), ns, fv, f0, fn, "srhd"])
                sed["vertices"] = np.fromfile(fv)
                sed["u0"] = np.fromfile(f0).reshape(-1, 5)
                sed["u_%d" % ns] = np.fromfile(fn).reshape(-1, 5)
            np.savez_compressed(os.path.join(OUT, "sedov_srhd_nr256.npz"), **sed)
            print("sedov srhd ok", sed["u0"].shape)
        return 0
    if len(sys.argv) > 1 and sys.argv[1] == "cloud_diag":
        return gen_cloud_diag()
    if len(sys.argv) > 1 and sys.argv[1] == "sedov_diag":
        # make_diagnostic_fields / compute_time_series_data of evolved sedov states, both hydro systems (new files: the step vectors stay as they are)
        for name, extra in (("sedovdiag_newtonian_nr256", []), ("sedovdiag_srhd_nr256", ["srhd"])):
            with tempfile.TemporaryDirectory() as d:
                out = {}
                for ns in (100, 400):
                    fv, f0, fn = (os.path.join(d, x) for x in ("v", "u0", "un"))
                    run_ref("sedov_ref", [256, hexf(100.0), ns, fv, f0, fn] + extra)
                    out["vertices"] = np.fromfile(fv)
                    u = np.fromfile(fn).reshape(-1, 5)
                    diag = np.fromfile(fn + ".diag")
                    nz = u.shape[0]
                    out["u_%d" % ns] = u
                    out["fields_%d" % ns] = diag[:4 * nz].reshape(4, nz)
                    out["indices_%d" % ns] = diag[4 * nz:4 * nz + 3].astype(np.int32)
                    out["series_%d" % ns] = diag[4 * nz + 3:]
                    print(name, ns, "indices", out["indices_%d" % ns], "series", out["series_%d" % ns])
                np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
        return 0
    if len(sys.argv) > 1 and sys.argv[1] == "two_body":
        rng = np.random.default_rng(20260404)
        n = 2048
        P = np.empty((n, 11))
        P[:, 0] = rng.uniform(-np.pi, np.pi, n)          # pomega
        P[:, 1] = rng.uniform(-3, 3, n)                   # tau
        P[:, 2:6] = rng.standard_normal((n, 4)) * 0.1      # centre-of-mass position / velocity
        P[:, 6] = 10.0 ** rng.uniform(-0.5, 0.5, n)        # separation
        P[:, 7] = 10.0 ** rng.uniform(-0.3, 0.3, n)        # total mass
        P[:, 8] = rng.uniform(0.05, 1.0, n)                # mass ratio
        P[:, 9] = rng.uniform(0.0, 0.9, n)                 # eccentricity
        P[:256, 9] = 0.0                                   # circular orbits take the other branch
        P[:, 10] = rng.uniform(-5, 20, n)                  # time
        P[0] = (0, 0, 0, 0, 0, 0, 1.0, 1.0, 1.0, 0.0, 0.0)   # the sub-program's default binary at t = 0
        with tempfile.TemporaryDirectory() as d:
            fin, fout = os.path.join(d, "in"), os.path.join(d, "out")
            P.tofile(fin)
            run_ref("two_body_ref", ["state", n, fin, fout])
            S = np.fromfile(fout).reshape(n, 10)
            X = np.hstack([S, P[:, 10:11]])
            X[-64:, 3:5] *= 3.0                             # unbound: the reference throws
            X.tofile(fin)
            run_ref("two_body_ref", ["elements", n, fin, fout])
            E = np.fromfile(fout).reshape(n, 11)
        np.savez_compressed(os.path.join(OUT, "two_body.npz"), elements_in=P, state=S, state_in=X, elements=E[:, :10], threw=E[:, 10])
        print("two_body ok; throws:", int(E[:, 10].sum()))
        return 0
    if len(sys.argv) > 1 and sys.argv[1] == "binary_tree":
        return gen_binary_tree()
    if len(sys.argv) > 1 and sys.argv[1] == "binary":
        return gen_binary()
    if len(sys.argv) > 1 and sys.argv[1] == "iso2d":
        return gen_iso2d(np.random.default_rng(20260303))
    if len(sys.argv) > 1 and sys.argv[1] == "srhd":
        return gen_srhd(np.random.default_rng(20260202))
    rng = np.random.default_rng(20260101)

    # ---- a1: plm_gradient, incl. sign changes, zeros, equal neighbours, signed zeros ----
    n = 4096
    y = rng.standard_normal((n, 3))
    y[:256] = np.round(y[:256])                       # many exact ties / zeros
    y[256:320, 1] = y[256:320, 0]                     # y0 == yl
    y[320:384, 2] = y[320:384, 1]                     # yr == y0
    y[384:400] = 0.0
    y[400:416] = -0.0
    y[416:432, 0] = -0.0
    y[432:448] *= 1e-300
    y[448:464] *= 1e300
    plm = {}
    for theta in (1.0, 1.2, 1.5, 1.8, 2.0, 0.0):
        plm["g_%g" % theta] = ref_func("plm", n, theta, 0.0, y)
    np.savez_compressed(os.path.join(OUT, "plm_gradient.npz"), y=y, **plm)

    # ---- a2-a4: Euler per-cell / per-face functions ----
    n = 4096
    P = random_prims(rng, n)
    Pr = random_prims(rng, n)
    Pr[:512] = P[:512] * (1.0 + 1e-3 * rng.standard_normal((512, 5)))   # near-equal pairs
    Pr[512:520] = P[512:520]                                             # identical pairs
    P[520:530, 1:4] = 0.0                                                # static
    P[530:534, 1:4] = -0.0
    out = {"Pl": P, "Pr": Pr}
    for gname, gamma in (("53", 5.0 / 3), ("43", 4.0 / 3), ("14", 1.4)):
        U = ref_func("euler_p2c", n, gamma, 0.0, P).reshape(n, 5)
        out["U_" + gname] = U
        out["c2p_" + gname] = ref_func("euler_c2p", n, gamma, 0.0, U).reshape(n, 5)
        for axis in range(3):
            out["hlle_%s_%d" % (gname, axis)] = ref_func("euler_hlle", n, gamma, axis, np.hstack([P, Pr])).reshape(n, 5)
            out["flux_%s_%d" % (gname, axis)] = ref_func("euler_flux", n, gamma, axis, P).reshape(n, 5)
            out["lam_%s_%d" % (gname, axis)] = ref_func("euler_lam", n, gamma, axis, P).reshape(n, 2)
    # temperature floor active: negative-pressure conserved states
    Uneg = out["U_53"].copy()
    Uneg[:, 4] *= rng.uniform(0.0, 1.2, n)
    out["Uneg"] = Uneg
    out["c2p_floor_53"] = ref_func("euler_c2p", n, 5.0 / 3, 1e-3, Uneg).reshape(n, 5)
    out["c2p_nofloor_53"] = ref_func("euler_c2p", n, 5.0 / 3, 0.0, Uneg).reshape(n, 5)
    np.savez_compressed(os.path.join(OUT, "euler_functions.npz"), **out)

    # ---- integer work: a18 / a19 ----
    ints = {}
    for rank in (1, 2, 3):
        ints["decomp_rank%d" % rank] = ref_func("decomp", 32, rank, 0, None, np.int64).reshape(32, rank)
    for count in (4096, 4097, 1000, 640, 10, 7):
        for nparts in range(1, 17):
            ints["partition_%d_%d" % (count, nparts)] = ref_func("partition", nparts, count, 0, None, np.int64).reshape(nparts, 2)
            if nparts <= count:
                ints["blocks_%d_%d" % (count, nparts)] = ref_func("blocks", nparts, count, 0, None, np.int64).reshape(nparts, 2)
    np.savez_compressed(os.path.join(OUT, "decomposition.npz"), **ints)

    # ---- per-step: uniform cartesian Euler (configs 2 and 5 at fixture size) ----
    def step_case(name, u0, dl, dt, nsteps_list, gamma, theta, rk, bc):
        d = {"u0": u0, "dl": np.array(dl), "dt": dt, "gamma": gamma, "theta": theta, "rk": rk, "bc": bc,
             "nsteps": np.array(nsteps_list)}
        for ns in nsteps_list:
            d["u_%d" % ns] = ref_euler_cart(u0, dl, dt, ns, gamma, theta, rk, bc)
        np.savez_compressed(os.path.join(OUT, name + ".npz"), **d)
        print(name, "ok")

    g = 5.0 / 3
    N = 64
    step_case("euler2d_blast64_plm15_rk2", blast_ic((N, N), g), (1 / N, 1 / N), 0.3 / N / 6, [1, 2, 10], g, 1.5, 2, 0)
    step_case("euler2d_blast64_plm20_rk1", blast_ic((N, N), g), (1 / N, 1 / N), 0.3 / N / 6, [1, 10], g, 2.0, 1, 0)
    step_case("euler2d_blast64_pcm_rk1", blast_ic((N, N), g), (1 / N, 1 / N), 0.3 / N / 6, [1, 10], g, -1.0, 1, 0)
    N = 128
    step_case("euler2d_blast128_plm15_rk2", blast_ic((N, N), g, radius=0.25), (1 / N, 1 / N), 0.3 / N / 6, [10], g, 1.5, 2, 0)
    step_case("euler2d_wave48x40_plm15_rk2_periodic", wave_ic((48, 40), 1.4, 1), (1 / 48, 1 / 40), 0.002, [1, 5], 1.4, 1.5, 2, 1)
    step_case("euler2d_wave33x70_plm12_rk2_outflow", wave_ic((33, 70), 1.4, 2), (1 / 33, 1 / 70), 0.002, [3], 1.4, 1.2, 2, 0)
    step_case("euler1d_wave200_plm15_rk2_periodic", wave_ic((200,), 1.4, 3), (1 / 200,), 0.001, [1, 10], 1.4, 1.5, 2, 1)
    step_case("euler1d_blast128_plm15_rk2", blast_ic((128,), g), (1 / 128,), 0.3 / 128 / 6, [10], g, 1.5, 2, 0)
    step_case("euler3d_blast24_plm15_rk2", blast_ic((24, 24, 24), g, radius=0.2), (1 / 24,) * 3, 0.3 / 24 / 6, [1, 4], g, 1.5, 2, 0)
    step_case("euler3d_wave20x12x16_plm15_rk2_periodic", wave_ic((20, 12, 16), 1.4, 4), (1 / 20, 1 / 12, 1 / 16), 0.004, [2], 1.4, 1.5, 2, 1)

    # ---- per-step: sedov newtonian=1 (config 1) ----
    with tempfile.TemporaryDirectory() as d:
        sed = {}
        for ns in (1, 10, 100):
            fv, f0, fn = (os.path.join(d, x) for x in ("v", "u0", "un"))
            run_ref("sedov_ref", [256, hexf(100.0), ns, fv, f0, fn])
            sed["vertices"] = np.fromfile(fv)
            sed["u0"] = np.fromfile(f0).reshape(-1, 5)
            sed["u_%d" % ns] = np.fromfile(fn).reshape(-1, 5)
        np.savez_compressed(os.path.join(OUT, "sedov_newtonian_nr256.npz"), **sed)
        print("sedov ok", sed["u0"].shape)


if __name__ == "__main__":
    sys.exit(main())
