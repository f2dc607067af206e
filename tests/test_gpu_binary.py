"""GPU parity tests of the `binary` path (BASELINE config 3) through the C ABI against
the vectors of the reference-composed driver (tests/golden/binary_*.npz) and the C oracle.

Tolerance, stated here once: the device evaluates the reference's pow(x, 1/2), pow(x, 3/2), exp and tanh with sqrt,
x sqrt(x) and the device math library, so results differ from the glibc-built reference in the last bits. Everything
else follows the reference's operation order with IEEE division / sqrt and no FMA contraction. Fields must agree to
REL = 1e-12 of the field's largest magnitude per component (the north star's L1 <= 1e-12 is an absolute bound on O(1)
data; surface densities here are O(1e-5), so the relative form is the stricter, meaningful one); scalars that are sums
over all cells (totals, accumulators) to 1e-10 of their scale (different summation order)."""
import ctypes as C
import json
import numpy as np
import pytest
from conftest import golden, bits_equal

pytestmark = pytest.mark.gpu
REL = 1e-12
CASES = ["binary_d2_b16", "binary_d1_b24_nu", "binary_d3_b8_axisym", "binary_d2_b16_safe", "binary_d2_b32", "binary_d2_b16_live", "binary_d2_b16_q", "binary_d1_b24_q_nu"]


@pytest.fixture(scope="module")
def mods():
    import mara3_amd
    from mara3_amd import binary, engine, _lib
    assert mara3_amd.load_library().mh_device_count() >= 1
    return mara3_amd.load_library(), binary, engine, _lib


def cfg_of(binary, g):
    over = json.loads(str(g["config"]))
    return binary.config(**{k: v for k, v in over.items() if k not in ("nsteps", "safe_mode")}), over


def to_field(u):
    """host AoS [n][n][3] -> device layout [(n + 4)][3][n] with the two periodic ghost rows per side"""
    n = u.shape[0]
    rows = np.arange(-2, n + 2) % n
    return np.ascontiguousarray(u[rows].transpose(0, 2, 1))


def from_field(f, n):
    return f.reshape(n + 4, 3, n)[2:n + 2].transpose(0, 2, 1)


def field_close(a, b, rel=REL):
    scale = np.abs(b).reshape(-1, 3).max(axis=0)
    err = np.abs(a - b).reshape(-1, 3).max(axis=0)
    return bool(np.all(err <= rel * scale)), err / scale


def run_stage(mods, cfg, g, u0, bodies, dt, safe=False, base=None, weight=1.0, chunk_rows=0, arith="strict"):
    lib, binary, engine, L = mods
    n = binary.grid_size(cfg)
    d = binary.make_desc(cfg, safe_mode=safe, chunk_rows=chunk_rows, xv=g["xv"], yv=g["yv"], arith=arith)
    D = engine.DeviceArray
    xv, yv = D(g["xv"]), D(g["yv"])
    u_in, u_init, br = D(to_field(u0)), D(to_field(g["u_init"])), D(g["br"])
    u_base = D(to_field(base)) if base is not None else None
    u_out = D.empty(((n + 4) * 3 * n,))
    totals = D.empty((L.BINARY_NTOTALS,))
    scratch = D.empty((lib.mh_binary_scratch_doubles(C.byref(d)),))
    status = D.empty((1,))
    b = np.ascontiguousarray(bodies, dtype=np.float64)
    L.check(lib.mh_binary_stage(C.byref(d), xv.ptr, yv.ptr, u_in.ptr, u_base.ptr if u_base else None, u_out.ptr, u_init.ptr, br.ptr,
                                b.ctypes.data_as(C.c_void_p), float(dt), float(weight), totals.ptr, scratch.ptr, status.ptr, None))
    st = status.get().view(np.int32)[0]
    return from_field(u_out.get(), n), totals.get(), st, u_out.get().reshape(n + 4, 3, n)


@pytest.mark.parametrize("name", CASES)
def test_one_stage_against_reference_vectors(mods, name):
    lib, binary, engine, L = mods
    g = golden(name)
    cfg, over = cfg_of(binary, g)
    ss = g["stage_scalars"]
    u1, tot, status, raw = run_stage(mods, cfg, g, g["u_init"], ss[61:71], ss[0], safe=bool(over.get("safe_mode", 0)))
    assert status == 0
    ok, rel = field_close(u1, g["u_stage"])
    assert ok, rel
    ref = ss[43:61]
    # sums over all cells: 1e-10 of the total's own scale, or - for totals that are pure cancellation (the torque on a
    # symmetric disk is rounding noise of either summation order) - 1e-12 of the largest total
    gross = np.abs(ref).max()
    for k in range(0, 18, 2):
        pair = slice(k, k + 2)
        assert np.all(np.abs(tot[pair] - ref[pair]) <= 1e-10 * np.abs(ref[pair]).max() + 1e-12 * gross), (k, tot[pair], ref[pair])
    # periodic ghost rows of the output are current
    n = u1.shape[0]
    assert np.array_equal(raw[0], raw[n]) and np.array_equal(raw[1], raw[n + 1]) and np.array_equal(raw[n + 2], raw[2]) and np.array_equal(raw[n + 3], raw[3])


@pytest.mark.parametrize("arith, name", [("strict", "binary_d2_b32"), ("fast", "binary_d2_b32"), ("fast", "binary_d2_b16_q"), ("fast", "binary_d3_b8_axisym")])
def test_stage_is_independent_of_the_chunking(mods, arith, name):
    """bit for bit, in both arithmetic modes. (FAST is compiled with FMA contraction: values carried from row to row are settled where
    they are formed, binary_device.hpp, or a chunk's first rows would differ in the last bit from the same rows inside a chunk.)"""
    lib, binary, engine, L = mods
    g = golden(name)
    cfg, _ = cfg_of(binary, g)
    ss = g["stage_scalars"]
    a, ta, _, _ = run_stage(mods, cfg, g, g["u_init"], ss[61:71], ss[0], chunk_rows=0, arith=arith)
    for chunk in (5, 7, 32, 128):
        b, tb, _, _ = run_stage(mods, cfg, g, g["u_init"], ss[61:71], ss[0], chunk_rows=chunk, arith=arith)
        assert np.array_equal(a, b), chunk
        assert np.allclose(ta, tb, rtol=1e-12, atol=1e-12 * np.abs(ta).max())     # partial sums are grouped by chunk


def test_rk_combine_is_fused_exactly(mods):
    lib, binary, engine, L = mods
    g = golden("binary_d2_b16")
    cfg, _ = cfg_of(binary, g)
    ss = g["stage_scalars"]
    u1, _, _, _ = run_stage(mods, cfg, g, g["u_stage"], ss[61:71], ss[0])
    uc, _, _, _ = run_stage(mods, cfg, g, g["u_stage"], ss[61:71], ss[0], base=g["u_init"], weight=0.5)
    assert np.array_equal(uc, g["u_init"] * 0.5 + u1 * 0.5)


def test_maximum_wavespeed(mods):
    lib, binary, engine, L = mods
    for name in CASES:
        g = golden(name)
        cfg, _ = cfg_of(binary, g)
        ss = g["stage_scalars"]
        d = binary.make_desc(cfg, xv=g["xv"], yv=g["yv"])
        D = engine.DeviceArray
        xv, yv, u = D(g["xv"]), D(g["yv"]), D(to_field(g["u_init"]))
        out = D.empty((1,))
        b = np.ascontiguousarray(ss[61:71])
        L.check(lib.mh_binary_max_wavespeed(C.byref(d), xv.ptr, yv.ptr, u.ptr, b.ctypes.data_as(C.c_void_p), out.ptr, None))
        n = binary.grid_size(cfg)
        h = 2.0 * cfg["domain_radius"] / n
        assert abs(h / out.get()[0] - ss[2]) <= 1e-14 * ss[2]


@pytest.mark.parametrize("name", CASES)
def test_next_solution_against_reference_vectors(mods, name):
    lib, binary, engine, L = mods
    g = golden(name)
    cfg, over = cfg_of(binary, g)
    if over.get("safe_mode", 0):
        pytest.skip("safe mode is entered by the solver itself; see test_safe_mode_retry")
    nsteps = int(over["nsteps"])
    s = binary.BinarySolver(cfg, xv=g["xv"], yv=g["yv"], u_init=g["u_init"], buffer_rate=g["br"], recommended_time_step=g["stage_scalars"][1])
    dts = []
    for _ in range(nsteps):
        assert s.next(1) == 0
        dts.append(s.last_dt)
    sc = g["scalars"]
    assert np.allclose(dts, sc[42:42 + nsteps], rtol=1e-13, atol=0)
    ok, rel = field_close(s.solution(), g["u_final"])
    assert ok, rel
    st = binary.state_as_dict(s.state())
    assert abs(st["time"] - sc[0]) <= 1e-13 * abs(sc[0]) and st["iteration"] == int(sc[1])
    acc = st["mass_accreted_on"] + st["angular_momentum_accreted_on"] + st["integrated_torque_on"] + st["work_done_on"] + [st["mass_ejected"], st["angular_momentum_ejected"]]
    ref = sc[2:12]
    gross = np.abs(ref).max()
    for k in range(0, 10, 2):
        assert np.all(np.abs(np.array(acc[k:k + 2]) - ref[k:k + 2]) <= 1e-9 * np.abs(ref[k:k + 2]).max() + 1e-11 * gross), (k, acc[k:k + 2], ref[k:k + 2])
    # the orbital elements themselves are untouched before begin_live_binary; once live they integrate the perturbations,
    # which follow the totals (and feed back into the body positions of the following stages)
    if cfg["begin_live_binary"] >= 1e6:
        assert np.array_equal(st["orbital_elements"], sc[32:42])
    else:
        assert not np.array_equal(sc[32:42], binary.initial_elements(cfg).as_array()), "the case must actually evolve the binary"
        assert np.allclose(st["orbital_elements"], sc[32:42], rtol=1e-11, atol=1e-13), (st["orbital_elements"], sc[32:42])
    for got, ref in ((st["orbital_elements_acc"], sc[12:22]), (st["orbital_elements_grav"], sc[22:32])):
        assert np.allclose(got, ref, rtol=1e-6, atol=1e-14), (got, ref)
    s.close()


@pytest.mark.parametrize("name", ["binary_d2_b16", "binary_d1_b24_nu", "binary_d3_b8_axisym", "binary_d2_b32", "binary_d2_b16_q", "binary_d1_b24_q_nu"])
def test_fast_arithmetic_within_tolerance_of_reference(mods, name):
    """MH_ARITH_FAST (reciprocal / rsq arithmetic, FMAs) for the binary stage: same tolerance as the strict kernel."""
    lib, binary, engine, L = mods
    g = golden(name)
    cfg, over = cfg_of(binary, g)
    s = binary.BinarySolver(cfg, xv=g["xv"], yv=g["yv"], u_init=g["u_init"], buffer_rate=g["br"], recommended_time_step=g["stage_scalars"][1], arith="fast")
    nsteps = int(over["nsteps"])
    assert s.next(nsteps) == 0
    ok, rel = field_close(s.solution(), g["u_final"])
    assert ok, rel
    sc = g["scalars"]
    assert abs(s.state().time - sc[0]) <= 1e-12 * abs(sc[0])
    s.close()


@pytest.mark.parametrize("name", ["binary_d2_b16", "binary_d1_b24_nu", "binary_d2_b16_q"])
def test_fast_stage_on_random_extreme_states_lands_on_the_strict_result(mods, name):
    """Uncorrelated random cells on the meshes of three reference cases (alpha and constant-nu viscosity, both conserved-variable forms):
    surface densities over three decades, velocities to ten times the local sound speed in any direction. One forward-Euler stage, FAST
    against STRICT, compared through the increments u1 - u0 (which carry the whole of the stage's arithmetic)."""
    lib, binary, engine, L = mods
    g = golden(name)
    cfg, over = cfg_of(binary, g)
    cfg["rk_order"], cfg["fixed_dt"] = 1, 1
    n = g["u_init"].shape[0]
    rng = np.random.default_rng(n)
    sigma = 10.0 ** rng.uniform(-2.0, 1.0, (n, n))
    xc = 0.5 * (g["xv"][1:] + g["xv"][:-1])
    cs = np.sqrt(1.0 / np.sqrt(xc[:, None] ** 2 + xc[None, :] ** 2 + 0.01)) / float(cfg["mach_number"])
    u0 = np.zeros((n, n, 3))
    u0[..., 0] = sigma
    u0[..., 1] = sigma * rng.uniform(-10.0, 10.0, (n, n)) * cs
    u0[..., 2] = sigma * rng.uniform(-10.0, 10.0, (n, n)) * cs
    if not int(cfg.get("conserve_linear_p", 1)):
        # (Sigma, Sigma s_r, Sigma l_z) at the cell centres
        px, py = u0[..., 1].copy(), u0[..., 2].copy()
        X, Y = np.meshgrid(xc, xc, indexing="ij")
        r = np.sqrt(X * X + Y * Y)
        u0[..., 1] = (px * X + py * Y) / r
        u0[..., 2] = X * py - Y * px
    dt = 1e-3 * float(g["stage_scalars"][1])
    out = {}
    for arith in ("strict", "fast"):
        s = binary.BinarySolver(cfg, xv=g["xv"], yv=g["yv"], u_init=u0, buffer_rate=g["br"], recommended_time_step=dt, arith=arith)
        assert s.next(1) == 0
        out[arith] = s.solution()
        s.close()
    da, db = out["strict"] - u0, out["fast"] - u0
    scale = np.abs(da).reshape(-1, 3).mean(axis=0)
    scale[1:] = scale[1:].max()
    err = np.abs(da - db).reshape(-1, 3).mean(axis=0) / scale
    assert np.all(err <= 1e-10), err


def test_next_in_one_call_equals_step_by_step(mods):
    """The look-ahead wavespeed reduction must not change a single bit."""
    lib, binary, engine, L = mods
    g = golden("binary_d2_b16")
    cfg, _ = cfg_of(binary, g)
    kw = dict(xv=g["xv"], yv=g["yv"], u_init=g["u_init"], buffer_rate=g["br"], recommended_time_step=g["stage_scalars"][1])
    a, b = binary.BinarySolver(cfg, **kw), binary.BinarySolver(cfg, **kw)
    a.next(4)
    for _ in range(4):
        b.next(1)
    assert np.array_equal(a.solution(), b.solution())
    assert a.last_dt == b.last_dt and a.state().time == b.state().time
    a.close(); b.close()


def test_safe_mode_retry(mods, oracle):
    """A state that drives a cell negative: the reference catches the exception and retries the whole step with dt / 10 and
    theta = 0 from the OLD solution (src/subprog_binary.cpp:285-292). Checked against the C oracle run the same way."""
    lib, binary, engine, L = mods
    g = golden("binary_d2_b16")
    cfg, _ = cfg_of(binary, g)
    cfg["fixed_dt"] = 1
    u0 = g["u_init"].copy()
    u0[20, 20, 0] *= 1e-2                  # a hole next to a fast stream: the ordinary step overshoots below zero
    u0[20, 21, 1:] *= 10.0
    u0[21, 20, 1:] *= 10.0
    rec = g["stage_scalars"][1] * 2
    s = binary.BinarySolver(cfg, xv=g["xv"], yv=g["yv"], u_init=g["u_init"], buffer_rate=g["br"], recommended_time_step=rec)
    st = s.state()
    s.set_solution(u0, st)
    E = binary.initial_elements(cfg)
    b1 = binary.two_body_state(E, 0.0)
    _, _, neg = oracle.binary_advance_u(cfg, g["xv"], g["yv"], u0, g["u_init"], g["br"], b1, rec)
    assert neg, "the test state must fail the ordinary step"
    assert s.next(1) == 1
    dt = rec * 0.1
    assert s.last_dt == dt
    ua, _, nega = oracle.binary_advance_u(cfg, g["xv"], g["yv"], u0, g["u_init"], g["br"], b1, dt, safe_mode=True)
    ub, _, negb = oracle.binary_advance_u(cfg, g["xv"], g["yv"], ua, g["u_init"], g["br"], binary.two_body_state(E, dt), dt, safe_mode=True)
    assert not nega and not negb
    ok, rel = field_close(s.solution(), u0 * 0.5 + ub * 0.5)
    assert ok, rel
    s.close()


@pytest.mark.parametrize("arith", ["strict", "fast"])
def test_200_adaptive_steps_against_the_reference_composition(mods, arith):
    """A uniform-depth tree fine enough to run without the safe mode (256^2: depth 4, block_size 16, domain_radius 8), 200 CFL-limited
    steps in one call (look-ahead time-step bound, one host synchronisation per step) against binary_ref's state - 35 CPU-seconds of the
    reference's leaf physics under the restated scheme glue (row a16 stays "partially pinned"). Host-side set-up from the configuration."""
    lib, binary, engine, L = mods
    g = golden("binary_d4_b16_r8_200steps")
    s = binary.BinarySolver(binary.config(depth=4, block_size=16, domain_radius=8.0), arith=arith)
    assert s.next(200) == 0
    u, ref, sc = s.solution(), g["u_final"], g["scalars"]
    scale = np.abs(ref).reshape(-1, 3).max(axis=0)
    err = np.abs(u - ref).reshape(-1, 3).max(axis=0) / scale
    assert np.all(err <= 1e-12), err
    assert abs(s.state().time - sc[0]) <= 1e-14 * sc[0] and s.state().iteration == int(sc[1]) == 200
    s.close()


@pytest.mark.parametrize("arith", ["strict", "fast"])
def test_config3_at_full_size_against_a_digest_of_the_reference_composition(mods, arith):
    """BASELINE config 3 as stated (2048^2 = 32 x 32 blocks of 64^2, defaults otherwise, CFL-limited dt), two RK2 steps, against binary_ref's
    state (47 CPU-seconds). The state itself is 100 MB, so the fixture is a digest of it: the mean of every tree block, one cell of every
    block, and the 48 x 48 cells around the binary (sinks, softened potentials). Tolerance as everywhere on this path: 1e-12 of the field
    scale (device libm for exp / tanh / pow)."""
    lib, binary, engine, L = mods
    g = golden("binary_c3_fullsize_2steps_digest")
    s = binary.BinarySolver(binary.config(depth=5, block_size=64), arith=arith)
    assert s.next(2) == 0
    u, scale, sc = s.solution(), g["scale"], g["scalars"]
    n, bs = 2048, 64
    assert np.all(np.abs(u.reshape(n // bs, bs, n // bs, bs, 3).mean(axis=(1, 3)) - g["block_means"]) <= 1e-12 * scale)
    assert np.all(np.abs(u[31::64, 17::64] - g["samples"]) <= 1e-12 * scale)
    assert np.all(np.abs(u[n // 2 - 24:n // 2 + 24, n // 2 - 24:n // 2 + 24] - g["centre"]) <= 1e-12 * scale)
    assert abs(s.state().time - sc[0]) <= 1e-14 * sc[0] and s.state().iteration == int(sc[1]) == 2
    s.close()


def test_full_size_point_symmetry_and_positivity(mods):
    """BASELINE config 3 at full size (2048^2 = 32 x 32 blocks of 64^2). An equal-mass circular binary and the disk model
    are symmetric under (x, y) -> (-x, -y) with (px, py) -> (-px, -py); the scheme preserves that up to rounding."""
    lib, binary, engine, L = mods
    cfg = binary.config(depth=5, block_size=64)
    s = binary.BinarySolver(cfg)
    assert s.next(3) == 0
    u = s.solution()
    assert np.isfinite(u).all() and (u[..., 0] > 0).all()
    r = u[::-1, ::-1]
    scale = np.abs(u).reshape(-1, 3).max(axis=0)
    assert np.abs(u[..., 0] - r[..., 0]).max() <= 1e-11 * scale[0]
    assert np.abs(u[..., 1] + r[..., 1]).max() <= 1e-11 * scale[1]
    assert np.abs(u[..., 2] + r[..., 2]).max() <= 1e-11 * scale[2]
    st = binary.state_as_dict(s.state())
    assert st["iteration"] == 3 and st["mass_accreted_on"][0] > 0
    assert abs(st["mass_accreted_on"][0] - st["mass_accreted_on"][1]) <= 1e-9 * st["mass_accreted_on"][0]
    s.close()


@pytest.mark.parametrize("arith", ["strict", "fast"])
def test_eager_first_stage_of_the_next_step_changes_no_bit(mods, arith):
    """With a fixed time step and the binary not live, a call for several steps issues the first stage of step n + 1 before the totals of
    step n have come back (mara3_amd/csrc/binary_api.hip: binary_attempt; the reference's data flow needs them only for the accumulators,
    src/subprog_binary.cpp:258-293). One step per call never does. Same field, same scalars, bit for bit - also when a step in the
    middle of the call is rejected and retried in safe mode, which discards the eager stage."""
    lib, binary, engine, L = mods
    cfg = binary.config(depth=2, block_size=16, domain_radius=4.0, fixed_dt=1)
    a, b = binary.BinarySolver(cfg, arith=arith), binary.BinarySolver(cfg, arith=arith)
    assert a.next(7) == 0
    for _ in range(7):
        assert b.next(1) == 0
    assert bits_equal(a.solution(), b.solution())
    assert binary.state_as_dict(a.state()) == binary.state_as_dict(b.state())
    # a state that makes the NEXT ordinary step fail: the call's first step is retried in safe mode, the others run eagerly again
    base, s = a.solution(), a.state()
    for boost in (10.0, 30.0, 100.0, 300.0):
        u = base.copy()
        u[20, 20, 0] *= 1e-2
        u[20, 21, 1:] *= boost
        u[21, 20, 1:] *= boost
        b.set_solution(u, s)
        if b.next(1) == 1:
            break
    else:
        pytest.skip("no test state made the ordinary step fail")
    a.set_solution(u, s); b.set_solution(u, s)
    na = a.next(5)
    nb = sum(b.next(1) for _ in range(5))
    assert na == nb >= 1
    assert bits_equal(a.solution(), b.solution())
    assert binary.state_as_dict(a.state()) == binary.state_as_dict(b.state())
    a.close(); b.close()
