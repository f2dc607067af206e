"""ctypes binding of the plain-C oracle (oracle/mara_oracle.c).

TEST INFRASTRUCTURE: only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import this module. The product path (mara3_amd) never
does; it fails loudly when its HIP library is missing.
"""
import ctypes as C
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "_build", "libmara_oracle.so")

BC_OUTFLOW, BC_PERIODIC = 0, 1
RIEMANN_HLLE, RIEMANN_HLLC = 0, 1


def build():
    """Compile the C oracle (and, where /root/reference exists, the reference-composed drivers)."""
    subprocess.check_call(["make", "-s", "-C", _HERE, "all"])


class _CartCfg(C.Structure):
    _fields_ = [
        ("rank", C.c_int),
        ("shape", C.c_size_t * 3),
        ("dl", C.c_double * 3),
        ("gamma", C.c_double),
        ("plm_theta", C.c_double),
        ("riemann", C.c_int),
        ("bc", C.c_int),
        ("rk_order", C.c_int),
        ("nthreads", C.c_int),
    ]


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB):
            build()
        _lib = C.CDLL(_LIB)
        dp = C.POINTER(C.c_double)
        _lib.mo_plm_gradient.restype = C.c_double
        _lib.mo_plm_gradient.argtypes = [C.c_double] * 4
        _lib.mo_plm_gradient_n.argtypes = [C.c_size_t, dp, dp, dp, C.c_double, dp]
        _lib.mo_euler_recover_primitive_n.argtypes = [C.c_size_t, dp, C.c_double, C.c_double, dp]
        _lib.mo_euler_to_conserved_density_n.argtypes = [C.c_size_t, dp, C.c_double, dp]
        _lib.mo_euler_riemann_n.argtypes = [C.c_size_t, dp, dp, C.c_int, C.c_double, C.c_int, dp]
        _lib.mo_euler_cart_advance.argtypes = [C.POINTER(_CartCfg), dp, C.c_double, dp]
        _lib.mo_euler_cart_run.argtypes = [C.POINTER(_CartCfg), dp, C.c_double, C.c_int]
        _lib.mo_sedov_vertices.argtypes = [C.c_int, C.c_double, C.POINTER(C.c_size_t), dp]
        _lib.mo_sedov_initial.argtypes = [C.c_size_t, dp, C.c_double, C.c_double, C.c_double, C.c_double, dp]
        _lib.mo_sedov_timestep.restype = C.c_double
        _lib.mo_sedov_timestep.argtypes = [dp, C.c_double]
        _lib.mo_sedov_advance.argtypes = [C.c_size_t, dp, C.c_double, C.c_double, dp, dp]
        _lib.mo_sedov_advance_srhd.argtypes = [C.c_size_t, dp, C.c_double, C.c_double, dp, dp]
        _lib.mo_sedov_initial_system.argtypes = [C.c_int, C.c_size_t, dp, C.c_double, C.c_double, C.c_double, C.c_double, dp]
        ip = C.POINTER(C.c_int)
        _lib.mo_srhd_recover_primitive_n.argtypes = [C.c_size_t, dp, C.c_double, C.c_double, dp, ip]
        _lib.mo_srhd_to_conserved_density_n.argtypes = [C.c_size_t, dp, C.c_double, dp]
        _lib.mo_srhd_riemann_hlle_n.argtypes = [C.c_size_t, dp, dp, C.c_int, C.c_double, dp]
        _lib.mo_srhd_source_terms_n.argtypes = [C.c_size_t, dp, dp, dp, C.c_double, dp]
        _lib.mo_cloud_geometry.argtypes = [C.c_size_t, C.c_size_t, dp, dp, dp, dp, dp]
        _lib.mo_cloud_advance.argtypes = [C.c_size_t, C.c_size_t, dp, dp, dp, C.c_double, C.c_double, C.c_double, C.c_double, dp, dp]
        _lib.mo_cloud_run.argtypes = [C.c_size_t, C.c_size_t, dp, dp, dp, C.c_double, C.c_double, C.c_double, C.c_int, C.c_double, C.c_int, dp]
        _lib.mo_iso2d_to_conserved_n.argtypes = [C.c_size_t, dp, dp]
        _lib.mo_iso2d_recover_primitive_n.argtypes = [C.c_size_t, dp, dp, ip]
        _lib.mo_iso2d_to_conserved_angmom_n.argtypes = [C.c_size_t, dp, dp, dp]
        _lib.mo_iso2d_recover_primitive_angmom_n.argtypes = [C.c_size_t, dp, dp, dp, ip]
        _lib.mo_iso2d_flux_n.argtypes = [C.c_size_t, dp, dp, C.c_int, dp]
        _lib.mo_iso2d_wavespeeds_n.argtypes = [C.c_size_t, dp, dp, C.c_int, dp]
        _lib.mo_iso2d_riemann_n.argtypes = [C.c_size_t, dp, dp, dp, dp, C.c_int, C.c_int, dp, dp, ip]
        sp = C.POINTER(C.c_size_t)
        _lib.mo_partition_rows.argtypes = [C.c_size_t, C.c_size_t, C.c_size_t, sp, sp]
        _lib.mo_block_extent.argtypes = [C.c_size_t, C.c_size_t, C.c_size_t, sp, sp]
        _lib.mo_prime_factors.argtypes = [C.c_ulong, C.POINTER(C.c_ulong)]
        _lib.mo_propose_block_decomposition.argtypes = [C.c_int, C.c_ulong, C.POINTER(C.c_ulong)]
    return _lib


def _dp(a):
    assert a.dtype == np.float64 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def plm_gradient(yl, y0, yr, theta):
    yl, y0, yr = _f64(yl), _f64(y0), _f64(yr)
    g = np.empty_like(y0)
    lib().mo_plm_gradient_n(y0.size, _dp(yl), _dp(y0), _dp(yr), theta, _dp(g))
    return g


def euler_recover_primitive(U, gamma, tfloor=0.0):
    U = _f64(U)
    P = np.empty_like(U)
    lib().mo_euler_recover_primitive_n(U.size // 5, _dp(U), gamma, tfloor, _dp(P))
    return P


def euler_to_conserved_density(P, gamma):
    P = _f64(P)
    U = np.empty_like(P)
    lib().mo_euler_to_conserved_density_n(P.size // 5, _dp(P), gamma, _dp(U))
    return U


def euler_riemann(Pl, Pr, axis, gamma, solver=RIEMANN_HLLE):
    Pl, Pr = _f64(Pl), _f64(Pr)
    F = np.empty_like(Pl)
    lib().mo_euler_riemann_n(Pl.size // 5, _dp(Pl), _dp(Pr), axis, gamma, solver, _dp(F))
    return F


def _cart_cfg(shape, dl, gamma, theta, riemann, bc, rk, nthreads):
    rank = len(shape)
    cfg = _CartCfg()
    cfg.rank = rank
    for a in range(3):
        cfg.shape[a] = shape[a] if a < rank else 1
        cfg.dl[a] = dl[a] if a < rank else 1.0
    cfg.gamma, cfg.plm_theta = gamma, theta
    cfg.riemann, cfg.bc, cfg.rk_order, cfg.nthreads = riemann, bc, rk, nthreads
    return cfg


def euler_cart_advance(u0, dl, dt, gamma, theta, riemann=RIEMANN_HLLE, bc=BC_OUTFLOW, nthreads=1):
    """One stage on an AoS field u0[n0][n1][n2][5] (rank = u0.ndim - 1)."""
    u0 = _f64(u0)
    cfg = _cart_cfg(u0.shape[:-1], dl, gamma, theta, riemann, bc, 1, nthreads)
    u1 = np.empty_like(u0)
    rc = lib().mo_euler_cart_advance(C.byref(cfg), _dp(u0), dt, _dp(u1))
    assert rc == 0
    return u1


def euler_cart_run(u, dl, dt, nsteps, gamma, theta, rk=2, riemann=RIEMANN_HLLE, bc=BC_OUTFLOW, nthreads=1):
    """nsteps full steps; returns a new array."""
    u = _f64(u).copy()
    cfg = _cart_cfg(u.shape[:-1], dl, gamma, theta, riemann, bc, rk, nthreads)
    rc = lib().mo_euler_cart_run(C.byref(cfg), _dp(u), dt, nsteps)
    assert rc == 0
    return u


def sedov_vertices(nr=256, outer_radius=100.0):
    nz = C.c_size_t()
    lib().mo_sedov_vertices(nr, outer_radius, C.byref(nz), None)
    v = np.empty(nz.value + 1)
    lib().mo_sedov_vertices(nr, outer_radius, C.byref(nz), _dp(v))
    return v


def sedov_initial(vertices, gamma=4.0 / 3, explosion_density=1.0, explosion_pressure=1.0, density_index=0.0):
    v = _f64(vertices)
    u = np.empty((v.size - 1, 5))
    lib().mo_sedov_initial(v.size - 1, _dp(v), gamma, explosion_density, explosion_pressure, density_index, _dp(u))
    return u


def sedov_initial_srhd(vertices, gamma=4.0 / 3, explosion_density=1.0, explosion_pressure=1.0, density_index=0.0):
    v = _f64(vertices)
    u = np.empty((v.size - 1, 5))
    lib().mo_sedov_initial_system(1, v.size - 1, _dp(v), gamma, explosion_density, explosion_pressure, density_index, _dp(u))
    return u


def sedov_advance_srhd(vertices, u0, dt, gamma=4.0 / 3):
    v, u0 = _f64(vertices), _f64(u0)
    u1 = np.empty_like(u0)
    st = lib().mo_sedov_advance_srhd(u0.shape[0], _dp(v), gamma, dt, _dp(u0), _dp(u1))
    return u1, st


def sedov_diagnostics(vertices, u, time, srhd=False, gamma=4.0 / 3):
    """-> (fields [4][nz], indices [3] int32, series [6], status); subprog_sedov.cpp:252-308"""
    v, u = _f64(vertices), _f64(u)
    nz = u.shape[0]
    fields, indices, series = np.zeros((4, nz)), np.zeros(3, dtype=np.int32), np.zeros(6)
    L = lib()
    L.mo_sedov_diagnostics.restype = C.c_int
    L.mo_sedov_diagnostics.argtypes = [C.c_int, C.c_size_t, C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_double, C.c_double,
                                       C.POINTER(C.c_double), C.POINTER(C.c_int), C.POINTER(C.c_double)]
    st = L.mo_sedov_diagnostics(int(bool(srhd)), nz, _dp(v), _dp(u), gamma, float(time), _dp(fields), indices.ctypes.data_as(C.POINTER(C.c_int)), _dp(series))
    return fields, indices, series, st


def sedov_timestep(vertices, cfl=0.4):
    return lib().mo_sedov_timestep(_dp(_f64(vertices)), cfl)


def sedov_advance(vertices, u0, dt, gamma=4.0 / 3):
    v, u0 = _f64(vertices), _f64(u0)
    u1 = np.empty_like(u0)
    lib().mo_sedov_advance(u0.shape[0], _dp(v), gamma, dt, _dp(u0), _dp(u1))
    return u1


def partition_rows(count, nparts, part):
    a, b = C.c_size_t(), C.c_size_t()
    lib().mo_partition_rows(count, nparts, part, C.byref(a), C.byref(b))
    return a.value, b.value


def propose_block_decomposition(rank, nblocks):
    out = (C.c_ulong * 3)()
    lib().mo_propose_block_decomposition(rank, nblocks, out)
    return tuple(out[i] for i in range(rank))


# ---- mara::srhd and the cloud stage ------------------------------------------------------------------------
def srhd_recover_primitive(U, gamma=4.0 / 3, tfloor=0.0):
    """Returns (P, status) with status bits C2P_* where the reference throws."""
    U = _f64(U)
    P = np.empty_like(U)
    st = np.zeros(U.size // 5, dtype=np.int32)
    lib().mo_srhd_recover_primitive_n(U.size // 5, _dp(U), gamma, tfloor, _dp(P), st.ctypes.data_as(C.POINTER(C.c_int)))
    return P, st


def srhd_to_conserved_density(P, gamma=4.0 / 3):
    P = _f64(P)
    U = np.empty_like(P)
    lib().mo_srhd_to_conserved_density_n(P.size // 5, _dp(P), gamma, _dp(U))
    return U


def srhd_riemann_hlle(Pl, Pr, axis, gamma=4.0 / 3):
    Pl, Pr = _f64(Pl), _f64(Pr)
    F = np.empty_like(Pl)
    lib().mo_srhd_riemann_hlle_n(Pl.size // 5, _dp(Pl), _dp(Pr), axis, gamma, _dp(F))
    return F


def srhd_source_terms(P, r, theta, gamma=4.0 / 3):
    P, r, theta = _f64(P), _f64(r), _f64(theta)
    S = np.empty_like(P)
    lib().mo_srhd_source_terms_n(P.size // 5, _dp(P), _dp(r), _dp(theta), gamma, _dp(S))
    return S


def cloud_geometry(rv, qv):
    rv, qv = _f64(rv), _f64(qv)
    nr, nq = rv.size - 1, qv.size - 1
    dAr, dAq, dv = np.empty((nr + 1, nq)), np.empty((nr, nq + 1)), np.empty((nr, nq))
    lib().mo_cloud_geometry(nr, nq, _dp(rv), _dp(qv), _dp(dAr), _dp(dAq), _dp(dv))
    return dAr, dAq, dv


def cloud_run(u, rv, qv, inflow, dt, nsteps, rk=1, theta=1.2, tfloor=1e-8, gamma=4.0 / 3):
    """u: [nr][nq][5] cell-integrated conserved; inflow: [nsteps][nq][5]. Returns (u_new, status)."""
    u, rv, qv, inflow = _f64(u).copy(), _f64(rv), _f64(qv), _f64(inflow)
    nr, nq = u.shape[0], u.shape[1]
    assert inflow.shape == (nsteps, nq, 5)
    st = lib().mo_cloud_run(nr, nq, _dp(rv), _dp(qv), _dp(inflow), gamma, theta, tfloor, rk, dt, nsteps, _dp(u))
    return u, st


def cloud_diagnostics(u, rv, qv, units, tfloor=1e-8, gamma=4.0 / 3):
    """CloudProblem::make_diagnostic_fields: u [nr][nq][5], units (length, mass, time) -> (fields [5][nr][nq], columns [15][nq], status)"""
    u, rv, qv = _f64(u), _f64(rv), _f64(qv)
    nr, nq = u.shape[0], u.shape[1]
    fields, columns = np.zeros((5, nr, nq)), np.zeros((15, nq))
    L = lib()
    L.mo_cloud_diagnostics.restype = C.c_int
    L.mo_cloud_diagnostics.argtypes = [C.c_size_t, C.c_size_t, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_double,
                                       C.c_double, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double)]
    st = L.mo_cloud_diagnostics(nr, nq, _dp(rv), _dp(qv), _dp(u), gamma, tfloor, _dp(_f64(np.array(units, dtype=np.float64))), _dp(fields), _dp(columns))
    return fields, columns, st


# ---- mara::iso2d ---------------------------------------------------------------------------------------------
def _ip(a):
    return a.ctypes.data_as(C.POINTER(C.c_int))


def iso2d_to_conserved(P):
    P = _f64(P)
    U = np.empty_like(P)
    lib().mo_iso2d_to_conserved_n(P.size // 3, _dp(P), _dp(U))
    return U


def iso2d_recover_primitive(U):
    U = _f64(U)
    P = np.empty_like(U)
    t = np.zeros(U.size // 3, dtype=np.int32)
    lib().mo_iso2d_recover_primitive_n(U.size // 3, _dp(U), _dp(P), _ip(t))
    return P, t


def iso2d_to_conserved_angmom(P, x):
    P, x = _f64(P), _f64(x)
    Q = np.empty_like(P)
    lib().mo_iso2d_to_conserved_angmom_n(P.size // 3, _dp(P), _dp(x), _dp(Q))
    return Q


def iso2d_recover_primitive_angmom(Q, x):
    Q, x = _f64(Q), _f64(x)
    P = np.empty_like(Q)
    t = np.zeros(Q.size // 3, dtype=np.int32)
    lib().mo_iso2d_recover_primitive_angmom_n(Q.size // 3, _dp(Q), _dp(x), _dp(P), _ip(t))
    return P, t


def iso2d_flux(P, cs2, axis):
    P, cs2 = _f64(P), _f64(cs2)
    F = np.empty_like(P)
    lib().mo_iso2d_flux_n(P.size // 3, _dp(P), _dp(cs2), axis, _dp(F))
    return F


def iso2d_wavespeeds(P, cs2, axis):
    P, cs2 = _f64(P), _f64(cs2)
    lam = np.empty_like(P)
    lib().mo_iso2d_wavespeeds_n(P.size // 3, _dp(P), _dp(cs2), axis, _dp(lam))
    return lam


def iso2d_riemann(Pl, Pr, cs2l, cs2r, axis, solver=RIEMANN_HLLE):
    """Returns (F, contact_speed, threw)."""
    Pl, Pr, cs2l, cs2r = _f64(Pl), _f64(Pr), _f64(cs2l), _f64(cs2r)
    n = Pl.size // 3
    F = np.empty_like(Pl)
    contact = np.zeros(n)
    t = np.zeros(n, dtype=np.int32)
    lib().mo_iso2d_riemann_n(n, _dp(Pl), _dp(Pr), _dp(cs2l), _dp(cs2r), axis, solver, _dp(F), _dp(contact), _ip(t))
    return F, contact, t


# ---- circumbinary disk scheme (mara_oracle_binary.c) -----------------------------------------------------------
T_MASS_ACC, T_L_ACC, T_TORQUE, T_PX_ACC, T_PY_ACC, T_FX, T_FY, T_WORK, T_MASS_EJ, T_L_EJ, BINARY_NTOTALS = 0, 2, 4, 6, 8, 10, 12, 14, 16, 17, 18

# the sub-program's run configuration, src/subprog_binary.cpp:55-99 (numeric items only)
BINARY_DEFAULTS = dict(
    cfl_number=0.4, fixed_dt=0, depth=4, begin_live_binary=1e6, block_size=24, rk_order=2, plm_theta=1.8,
    source_term_softening=1.0, softening_radius=0.05, sink_radius=0.05, sink_rate=1.0, buffer_damping_rate=10.0,
    domain_radius=12.0, disk_radius=2.0, disk_mass=1e-3, ambient_density=1e-4, density_floor=0.0, separation=1.0,
    mass_ratio=1.0, eccentricity=0.0, counter_rotate=0, mach_number=10.0, axisymmetric_cs2=0, no_accretion_force=0,
    alpha_cutoff_radius=0.0, alpha=0.1, nu=0.0, mdot=0.0, conserve_linear_p=1)


class _BinaryParams(C.Structure):
    _fields_ = [("n", C.c_int), ("block_size", C.c_int), ("domain_radius", C.c_double), ("mach_number", C.c_double),
                ("alpha", C.c_double), ("nu", C.c_double), ("alpha_cutoff_radius", C.c_double), ("sink_rate", C.c_double),
                ("sink_radius", C.c_double), ("softening_radius", C.c_double), ("density_floor", C.c_double),
                ("plm_theta", C.c_double), ("axisymmetric_cs2", C.c_int), ("angmom_form", C.c_int), ("gst_suppr_radius", C.c_double)]


class _BinaryModel(C.Structure):
    _fields_ = [("softening_radius", C.c_double), ("disk_radius", C.c_double), ("mach_number", C.c_double),
                ("disk_mass", C.c_double), ("ambient_density", C.c_double), ("mdot", C.c_double), ("counter_rotate", C.c_int),
                ("buffer_damping_rate", C.c_double), ("domain_radius", C.c_double), ("cfl_number", C.c_double), ("angmom_form", C.c_int)]


def binary_config(**overrides):
    cfg = dict(BINARY_DEFAULTS)
    for k, v in overrides.items():
        if k not in cfg:
            raise KeyError(k)
        cfg[k] = v
    return cfg


def _binary_params(cfg, safe_mode=False, xv=None, yv=None):
    P = _BinaryParams()
    P.n = int(cfg["block_size"]) << int(cfg["depth"])
    P.block_size = int(cfg["block_size"])
    for k in ("domain_radius", "mach_number", "alpha", "nu", "alpha_cutoff_radius", "sink_rate", "sink_radius", "softening_radius"):
        setattr(P, k, float(cfg[k]))
    P.density_floor = float(cfg["density_floor"]) * float(cfg["disk_mass"])
    P.plm_theta = 0.0 if safe_mode else float(cfg["plm_theta"])
    P.axisymmetric_cs2 = int(cfg["axisymmetric_cs2"])
    P.angmom_form = 0 if int(cfg["conserve_linear_p"]) else 1
    if xv is not None:      # source_term_softening * min(min_dx, min_dy), solver_data.cpp:91
        P.gst_suppr_radius = float(cfg["source_term_softening"]) * min(float(np.diff(_f64(xv)).min()), float(np.diff(_f64(yv)).min()))
    return P


def _binary_model(cfg):
    m = _BinaryModel()
    for k in ("softening_radius", "disk_radius", "mach_number", "disk_mass", "ambient_density", "mdot", "buffer_damping_rate", "domain_radius", "cfl_number"):
        setattr(m, k, float(cfg[k]))
    m.counter_rotate = int(cfg["counter_rotate"])
    m.angmom_form = 0 if int(cfg["conserve_linear_p"]) else 1
    return m


def _binary_lib():
    L = lib()
    if not getattr(L, "_binary_ready", False):
        dp = C.POINTER(C.c_double)
        L.mo_binary_advance_u.argtypes = [C.POINTER(_BinaryParams), dp, dp, dp, dp, dp, dp, C.c_double, dp, dp]
        L.mo_binary_advance_u.restype = C.c_int
        L.mo_binary_maximum_timestep.argtypes = [C.POINTER(_BinaryParams), dp, dp, dp, dp]
        L.mo_binary_maximum_timestep.restype = C.c_double
        L.mo_binary_vertices.argtypes = [C.c_int, C.c_int, C.c_double, dp]
        L.mo_binary_solver_data.argtypes = [C.POINTER(_BinaryModel), C.c_int, dp, dp, dp, dp]
        L.mo_binary_solver_data.restype = C.c_double
        L.mo_binary_diagnostics.argtypes = [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), dp, dp, dp, dp]
        L.mo_binary_diagnostics.restype = None
        L.mo_binary_set_threads.argtypes = [C.c_int]
        L.mo_binary_set_threads.restype = None
        L._binary_ready = True
    return L


def binary_vertices(cfg):
    n = int(cfg["block_size"]) << int(cfg["depth"])
    v = np.zeros(n + 1)
    _binary_lib().mo_binary_vertices(int(cfg["block_size"]), int(cfg["depth"]), float(cfg["domain_radius"]), _dp(v))
    return v


def binary_solver_data(cfg, xv, yv):
    """-> (u_init [n][n][3], buffer_rate [n][n], recommended_time_step)"""
    n = len(xv) - 1
    u = np.zeros((n, n, 3))
    br = np.zeros((n, n))
    m = _binary_model(cfg)
    dt = _binary_lib().mo_binary_solver_data(C.byref(m), n, _dp(_f64(xv)), _dp(_f64(yv)), _dp(u), _dp(br))
    return u, br, dt


def binary_advance_u(cfg, xv, yv, u0, u_init, br, bodies, dt, safe_mode=False, nthreads=1):
    """-> (u1, totals[18], negative_density); nthreads: OpenMP threads over rows / blocks (the results do not depend on it)"""
    P = _binary_params(cfg, safe_mode, xv, yv)
    _binary_lib().mo_binary_set_threads(int(nthreads))
    u0 = _f64(u0)
    u1 = np.zeros_like(u0)
    tot = np.zeros(BINARY_NTOTALS)
    neg = _binary_lib().mo_binary_advance_u(C.byref(P), _dp(_f64(xv)), _dp(_f64(yv)), _dp(u0), _dp(_f64(u_init)), _dp(_f64(br)),
                                           _dp(_f64(bodies)), float(dt), _dp(u1), _dp(tot))
    return u1, tot, bool(neg)


def binary_maximum_timestep(cfg, xv, yv, u, bodies):
    P = _binary_params(cfg, False, xv, yv)
    return _binary_lib().mo_binary_maximum_timestep(C.byref(P), _dp(_f64(xv)), _dp(_f64(yv)), _dp(_f64(u)), _dp(_f64(bodies)))


def binary_diagnostics(angmom_form, blocks, edges, u):
    """subprog_binary_diagnostics.cpp:21-82 on a block tree: blocks [nb][3] (level, i, j) in tree order, edges [nb][2][bs+1],
    u [nb][bs][bs][3] -> (disk_mass, disk_angular_momentum, fields [nb][3][bs][bs] = sigma, v_r, v_phi)"""
    blocks = np.ascontiguousarray(blocks, dtype=np.int32)
    u = _f64(u)
    nb, bs = u.shape[0], u.shape[1]
    tot = np.zeros(2)
    fields = np.zeros((nb, 3, bs, bs))
    _binary_lib().mo_binary_diagnostics(int(bool(angmom_form)), bs, nb, blocks.ctypes.data_as(C.POINTER(C.c_int)), _dp(_f64(edges)), _dp(u), _dp(tot), _dp(fields))
    return tot[0], tot[1], fields
