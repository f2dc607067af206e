"""Host-side mirror of the `binary` sub-program's interface (src/subprog_binary.cpp) over the C ABI.

The run configuration uses the reference's item names and defaults (create_config_template, :55-99). All arithmetic -
set-up with the host libm, time stepping on the device - happens in libmara_hip.so; this module owns none and there is
no CPU fallback.
"""
import ctypes as C
import numpy as np
from . import _lib as L

# numeric items of binary::create_config_template (src/subprog_binary.cpp:55-99)
DEFAULTS = dict(
    focus_factor=2.0, focus_index=2.0,
    cfl_number=0.4, fixed_dt=0, depth=4, begin_live_binary=1e6, block_size=24, rk_order=2, plm_theta=1.8,
    source_term_softening=1.0, softening_radius=0.05, sink_radius=0.05, sink_rate=1.0, buffer_damping_rate=10.0,
    domain_radius=12.0, disk_radius=2.0, disk_mass=1e-3, ambient_density=1e-4, density_floor=0.0, separation=1.0,
    mass_ratio=1.0, eccentricity=0.0, counter_rotate=0, mach_number=10.0, axisymmetric_cs2=0, no_accretion_force=0,
    alpha_cutoff_radius=0.0, alpha=0.1, nu=0.0, mdot=0.0, conserve_linear_p=1)

ELEMENT_NAMES = ("pomega", "tau", "cm_position_x", "cm_position_y", "cm_velocity_x", "cm_velocity_y",
                 "separation", "total_mass", "mass_ratio", "eccentricity")


def config(**overrides):
    cfg = dict(DEFAULTS)
    for k, v in overrides.items():
        if k not in cfg:
            raise KeyError("binary: no run_config item '%s'" % k)
        cfg[k] = v
    return cfg


def grid_size(cfg):
    return int(cfg["block_size"]) << int(cfg["depth"])


def vertices(cfg):
    """Vertex coordinates of the uniform-depth block tree (the same array for both axes)."""
    lib = L.load_library()
    v = np.zeros(grid_size(cfg) + 1)
    L.check(lib.mh_binary_vertices(int(cfg["block_size"]), int(cfg["depth"]), float(cfg["domain_radius"]), v.ctypes.data_as(C.c_void_p)))
    return v


def _model(cfg):
    m = L.BinaryModel()
    for k in ("softening_radius", "disk_radius", "mach_number", "disk_mass", "ambient_density", "mdot", "buffer_damping_rate", "domain_radius", "cfl_number"):
        setattr(m, k, float(cfg[k]))
    m.counter_rotate = int(cfg["counter_rotate"])
    m.angmom_form = 0 if int(cfg["conserve_linear_p"]) else 1
    return m


def solver_data(cfg, xv=None, yv=None):
    """(u_init [n][n][3], buffer_rate [n][n], recommended_time_step) - binary::create_solver_data."""
    lib = L.load_library()
    xv = vertices(cfg) if xv is None else np.ascontiguousarray(xv, dtype=np.float64)
    yv = xv if yv is None else np.ascontiguousarray(yv, dtype=np.float64)
    n = len(xv) - 1
    u = np.zeros((n, n, 3))
    br = np.zeros((n, n))
    dt = C.c_double()
    m = _model(cfg)
    L.check(lib.mh_binary_solver_data(C.byref(m), n, xv.ctypes.data_as(C.c_void_p), yv.ctypes.data_as(C.c_void_p),
                                      u.ctypes.data_as(C.c_void_p), br.ctypes.data_as(C.c_void_p), C.byref(dt)))
    return u, br, dt.value


def gst_suppr_radius(cfg, xv, yv):
    """source_term_softening * min(min_dx, min_dy) - src/subprog_binary_solver_data.cpp:91."""
    return float(cfg["source_term_softening"]) * min(float(np.diff(xv).min()), float(np.diff(yv).min()))


def make_desc(cfg, safe_mode=False, chunk_rows=0, xv=None, yv=None, arith="strict"):
    """conserve_linear_p = 0 selects the angular-momentum form (advance_q) and needs the vertices for gst_suppr_radius."""
    d = L.BinaryDesc()
    d.n = grid_size(cfg)
    d.block_size = int(cfg["block_size"])
    for k in ("domain_radius", "mach_number", "alpha", "nu", "alpha_cutoff_radius", "sink_rate", "sink_radius", "softening_radius"):
        setattr(d, k, float(cfg[k]))
    d.density_floor = float(cfg["density_floor"]) * float(cfg["disk_mass"])
    d.plm_theta = 0.0 if safe_mode else float(cfg["plm_theta"])
    d.axisymmetric_cs2 = int(cfg["axisymmetric_cs2"])
    d.chunk_rows = int(chunk_rows)
    d.arith = {"strict": L.ARITH_STRICT, "fast": L.ARITH_FAST}[arith]
    d.angmom_form = 0 if int(cfg["conserve_linear_p"]) else 1
    if d.angmom_form:
        xv = vertices(cfg) if xv is None else np.ascontiguousarray(xv, dtype=np.float64)
        yv = xv if yv is None else np.ascontiguousarray(yv, dtype=np.float64)
        d.gst_suppr_radius = gst_suppr_radius(cfg, xv, yv)
    return d


# ---- graded block trees (the sub-program's default mesh) ------------------------------------------------------------------
def tree_blocks(cfg):
    """Leaf blocks (level, i, j) of the tree the run configuration describes, in the reference's traversal order."""
    lib = L.load_library()
    args = (int(cfg["block_size"]), int(cfg["depth"]), float(cfg["focus_factor"]), float(cfg["focus_index"]))
    n = lib.mh_binary_tree_build(*args, None, 0)
    if n < 0:
        L.check(n)
    out = np.zeros((n, 3), dtype=np.int32)
    L.check(min(0, lib.mh_binary_tree_build(*args, out.ctypes.data_as(C.c_void_p), n)))
    return out


def tree_is_uniform(blocks):
    return len(np.unique(blocks[:, 0])) == 1


def tree_vertices(cfg, blocks):
    """[nblocks][2][bs + 1]: x of each block's vertex columns, y of its vertex rows."""
    lib = L.load_library()
    bs = int(cfg["block_size"])
    blocks = np.ascontiguousarray(blocks, dtype=np.int32)
    e = np.zeros((len(blocks), 2, bs + 1))
    L.check(lib.mh_binary_tree_vertices(bs, float(cfg["domain_radius"]), blocks.ctypes.data_as(C.c_void_p), len(blocks), e.ctypes.data_as(C.c_void_p)))
    return e


def tree_solver_data(cfg, blocks, edges):
    lib = L.load_library()
    bs = int(cfg["block_size"])
    blocks = np.ascontiguousarray(blocks, dtype=np.int32)
    edges = np.ascontiguousarray(edges, dtype=np.float64)
    u = np.zeros((len(blocks), bs, bs, 3))
    br = np.zeros((len(blocks), bs, bs))
    dt = C.c_double()
    m = _model(cfg)
    L.check(lib.mh_binary_tree_solver_data(C.byref(m), bs, blocks.ctypes.data_as(C.c_void_p), len(blocks), edges.ctypes.data_as(C.c_void_p),
                                           u.ctypes.data_as(C.c_void_p), br.ctypes.data_as(C.c_void_p), C.byref(dt)))
    return u, br, dt.value


def initial_elements(cfg):
    """make_full_orbital_elements(create_binary_params(run_config)) - src/subprog_binary.cpp:187-195."""
    E = L.FullOrbitalElements()
    E.elements.total_mass = 1.0
    E.elements.separation = float(cfg["separation"])
    E.elements.mass_ratio = float(cfg["mass_ratio"])
    E.elements.eccentricity = float(cfg["eccentricity"])
    return E


def two_body_state(elements, t):
    """bodies = (mass, x, y, vx, vy) x 2 of mara::compute_two_body_state."""
    lib = L.load_library()
    out = (C.c_double * 10)()
    L.check(lib.mh_two_body_state(C.byref(elements), float(t), out))
    return np.array(out[:])


def state_as_dict(s):
    return dict(
        time=s.time, iteration=s.iteration,
        mass_accreted_on=list(s.mass_accreted_on), angular_momentum_accreted_on=list(s.angular_momentum_accreted_on),
        integrated_torque_on=list(s.integrated_torque_on), work_done_on=list(s.work_done_on),
        mass_ejected=s.mass_ejected, angular_momentum_ejected=s.angular_momentum_ejected,
        orbital_elements_acc=s.orbital_elements_acc.as_array(), orbital_elements_grav=s.orbital_elements_grav.as_array(),
        orbital_elements=s.orbital_elements.as_array())


class BinarySolver:
    """binary::state_t's solution + next_solution on one MI355X (uniform-depth tree)."""

    def __init__(self, cfg, device=0, xv=None, yv=None, u_init=None, buffer_rate=None, recommended_time_step=None, chunk_rows=0,
                 arith="strict"):
        self.lib = L.load_library()
        self.cfg = cfg
        self.n = grid_size(cfg)
        self.xv = vertices(cfg) if xv is None else np.ascontiguousarray(xv, dtype=np.float64)
        self.yv = self.xv if yv is None else np.ascontiguousarray(yv, dtype=np.float64)
        if u_init is None or buffer_rate is None or recommended_time_step is None:
            u_init, buffer_rate, recommended_time_step = solver_data(cfg, self.xv, self.yv)
        self.u_init = np.ascontiguousarray(u_init, dtype=np.float64)
        self.buffer_rate = np.ascontiguousarray(buffer_rate, dtype=np.float64)
        assert self.u_init.shape == (self.n, self.n, 3) and self.buffer_rate.shape == (self.n, self.n)
        self.desc = make_desc(cfg, chunk_rows=chunk_rows, xv=self.xv, yv=self.yv, arith=arith)
        run = L.BinaryRun()
        run.rk_order = int(cfg["rk_order"])
        run.fixed_dt = int(cfg["fixed_dt"])
        run.no_accretion_force = int(cfg["no_accretion_force"])
        run.cfl_number = float(cfg["cfl_number"])
        run.recommended_time_step = float(recommended_time_step)
        run.begin_live_binary = float(cfg["begin_live_binary"])
        self.run = run
        self.handle = C.c_void_p()
        L.check(self.lib.mh_binary_create(C.byref(self.handle), device, C.byref(self.desc), C.byref(run),
                                          self.xv.ctypes.data_as(C.c_void_p), self.yv.ctypes.data_as(C.c_void_p),
                                          self.u_init.ctypes.data_as(C.c_void_p), self.buffer_rate.ctypes.data_as(C.c_void_p)))
        # binary::create_solution (src/subprog_binary.cpp:197-229)
        s = L.BinaryState()
        s.orbital_elements = initial_elements(cfg)
        self.set_solution(None, s)

    def set_solution(self, u, state):
        up = None if u is None else np.ascontiguousarray(u, dtype=np.float64).ctypes.data_as(C.c_void_p)
        L.check(self.lib.mh_binary_set_solution(self.handle, up, C.byref(state)))

    def state(self):
        s = L.BinaryState()
        L.check(self.lib.mh_binary_get_solution(self.handle, None, C.byref(s)))
        return s

    def solution(self):
        u = np.empty((self.n, self.n, 3))
        L.check(self.lib.mh_binary_get_solution(self.handle, u.ctypes.data_as(C.c_void_p), None))
        return u

    def next(self, nsteps=1):
        """nsteps x binary::next_solution; returns how many of them fell back to safe mode."""
        safe = C.c_int(0)
        L.check(self.lib.mh_binary_next(self.handle, int(nsteps), C.byref(safe)))
        return safe.value

    @property
    def last_dt(self):
        return self.lib.mh_binary_last_dt(self.handle)

    def last_failure(self):
        """(status bits, whole-mesh flat index of the first failing cell or None) of a failed attempt of the most recent next() call"""
        r = L.StepResult()
        L.check(self.lib.mh_binary_last_failure(self.handle, C.byref(r)))
        return r.status, (None if r.status == 0 else int(r.first_bad_index))

    def disk_totals(self):
        """binary::disk_mass, binary::disk_angular_momentum (subprog_binary_diagnostics.cpp:21-46), reduced on the device."""
        m, l = C.c_double(), C.c_double()
        L.check(self.lib.mh_binary_disk_totals(self.handle, C.byref(m), C.byref(l)))
        return m.value, l.value

    def diagnostic_fields(self):
        """binary::diagnostic_fields (:52-82): (sigma, radial_velocity, phi_velocity), each shaped like one component of `solution()`."""
        shape = self.solution_shape()[:-1]
        out = [np.empty(shape) for _ in range(3)]
        L.check(self.lib.mh_binary_diagnostic_fields(self.handle, *[a.ctypes.data_as(C.c_void_p) for a in out]))
        return tuple(out)

    def solution_shape(self):
        return (self.n, self.n, 3)

    def profile(self, enable=True):
        ms, nl = C.c_double(), C.c_int()
        L.check(self.lib.mh_binary_profile(self.handle, int(enable), C.byref(ms), C.byref(nl)))
        return ms.value, nl.value

    def close(self):
        if getattr(self, "handle", None):
            self.lib.mh_binary_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class BinaryTreeSolver(BinarySolver):
    """The same solution object on a GRADED block tree (the sub-program's default mesh). Arrays are block-major:
    [nblocks][bs][bs][3], blocks in the reference's traversal order (`tree_blocks`)."""

    def __init__(self, cfg, device=0, blocks=None, edges=None, u_init=None, buffer_rate=None, recommended_time_step=None, arith="strict"):
        self.lib = L.load_library()
        self.cfg = cfg
        self.bs = int(cfg["block_size"])
        self.blocks = tree_blocks(cfg) if blocks is None else np.ascontiguousarray(blocks, dtype=np.int32)
        self.edges = tree_vertices(cfg, self.blocks) if edges is None else np.ascontiguousarray(edges, dtype=np.float64)
        if u_init is None or buffer_rate is None or recommended_time_step is None:
            u_init, buffer_rate, recommended_time_step = tree_solver_data(cfg, self.blocks, self.edges)
        self.u_init = np.ascontiguousarray(u_init, dtype=np.float64)
        self.buffer_rate = np.ascontiguousarray(buffer_rate, dtype=np.float64)
        nb = len(self.blocks)
        assert self.u_init.shape == (nb, self.bs, self.bs, 3) and self.buffer_rate.shape == (nb, self.bs, self.bs)
        self.desc = make_desc(config(**{**cfg, "conserve_linear_p": 1}), arith=arith)
        if not int(cfg["conserve_linear_p"]):
            self.desc.angmom_form = 1
            self.desc.gst_suppr_radius = float(cfg["source_term_softening"]) * float(np.diff(self.edges, axis=2).min())     # solver_data.cpp:91
        run = L.BinaryRun()
        run.rk_order = int(cfg["rk_order"])
        run.fixed_dt = int(cfg["fixed_dt"])
        run.no_accretion_force = int(cfg["no_accretion_force"])
        run.cfl_number = float(cfg["cfl_number"])
        run.recommended_time_step = float(recommended_time_step)
        run.begin_live_binary = float(cfg["begin_live_binary"])
        self.run = run
        self.handle = C.c_void_p()
        L.check(self.lib.mh_binary_tree_create(C.byref(self.handle), device, C.byref(self.desc), C.byref(run), self.blocks.ctypes.data_as(C.c_void_p), nb,
                                               self.edges.ctypes.data_as(C.c_void_p), self.u_init.ctypes.data_as(C.c_void_p),
                                               self.buffer_rate.ctypes.data_as(C.c_void_p)))
        s = L.BinaryState()
        s.orbital_elements = initial_elements(cfg)
        self.set_solution(None, s)

    def solution_shape(self):
        return (len(self.blocks), self.bs, self.bs, 3)

    def solution(self):
        u = np.empty((len(self.blocks), self.bs, self.bs, 3))
        L.check(self.lib.mh_binary_get_solution(self.handle, u.ctypes.data_as(C.c_void_p), None))
        return u


def tree_curve_order(blocks):
    """order[k] = the block (row of `blocks`) standing k-th along the Hilbert curve through the leaves (mh_binary_tree_curve_order)"""
    lib = L.load_library()
    blocks = np.ascontiguousarray(blocks, dtype=np.int32)
    order = np.empty(len(blocks), dtype=np.int32)
    L.check(lib.mh_binary_tree_curve_order(blocks.ctypes.data_as(C.c_void_p), len(blocks), order.ctypes.data_as(C.c_void_p)))
    return order


class _TreeSetup:
    """what BinaryTreeSolver's constructor prepares, for the distributed forms"""

    def __init__(self, cfg, blocks=None, arith="strict"):
        self.lib = L.load_library()
        self.cfg = cfg
        self.bs = int(cfg["block_size"])
        self.blocks = tree_blocks(cfg) if blocks is None else np.ascontiguousarray(blocks, dtype=np.int32)
        self.edges = tree_vertices(cfg, self.blocks)
        self.u_init, self.buffer_rate, recommended_time_step = tree_solver_data(cfg, self.blocks, self.edges)
        self.u_init = np.ascontiguousarray(self.u_init, dtype=np.float64)
        self.buffer_rate = np.ascontiguousarray(self.buffer_rate, dtype=np.float64)
        self.desc = make_desc(config(**{**cfg, "conserve_linear_p": 1}), arith=arith)
        if not int(cfg["conserve_linear_p"]):
            self.desc.angmom_form = 1
            self.desc.gst_suppr_radius = float(cfg["source_term_softening"]) * float(np.diff(self.edges, axis=2).min())
        run = L.BinaryRun()
        run.rk_order = int(cfg["rk_order"])
        run.fixed_dt = int(cfg["fixed_dt"])
        run.no_accretion_force = int(cfg["no_accretion_force"])
        run.cfl_number = float(cfg["cfl_number"])
        run.recommended_time_step = float(recommended_time_step)
        run.begin_live_binary = float(cfg["begin_live_binary"])
        self.run = run

    def pointers(self):
        return (self.blocks.ctypes.data_as(C.c_void_p), len(self.blocks), self.edges.ctypes.data_as(C.c_void_p),
                self.u_init.ctypes.data_as(C.c_void_p), self.buffer_rate.ctypes.data_as(C.c_void_p))


class BinaryTreeGroup:
    """A GRADED tree distributed over `world` members - runs of the Hilbert curve through its leaves - as objects of one process on one
    GPU (LOOPBACK; include/mara_hip.h: mh_binary_tree_group_create). Arrays are the whole tree in the caller's block order,
    [nblocks][bs][bs][3], as BinaryTreeSolver's."""

    def __init__(self, cfg, world=2, device=0, blocks=None, arith="strict"):
        t = _TreeSetup(cfg, blocks, arith)
        self.lib, self.cfg, self.world, self.setup = t.lib, cfg, world, t
        self.blocks, self.bs = t.blocks, t.bs
        self.handles = (C.c_void_p * world)()
        L.check(self.lib.mh_binary_tree_group_create(self.handles, world, device, C.byref(t.desc), C.byref(t.run), *t.pointers()))
        self.owned = []
        for r in range(world):
            n = C.c_int()
            ids = np.empty(len(self.blocks), dtype=np.int32)
            L.check(self.lib.mh_binary_tree_owned_blocks(C.c_void_p(self.handles[r]), ids.ctypes.data_as(C.c_void_p), C.byref(n)))
            self.owned.append(ids[:n.value].copy())
        s = L.BinaryState()
        s.orbital_elements = initial_elements(cfg)
        self.set_solution(None, s)

    def set_solution(self, u, state):
        up = None if u is None else np.ascontiguousarray(u, dtype=np.float64).ctypes.data_as(C.c_void_p)
        L.check(self.lib.mh_binary_group_set_solution(self.handles, self.world, up, C.byref(state)))

    def state(self):
        s = L.BinaryState()
        L.check(self.lib.mh_binary_group_get_solution(self.handles, self.world, None, C.byref(s)))
        return s

    def solution(self):
        u = np.empty((len(self.blocks), self.bs, self.bs, 3))
        L.check(self.lib.mh_binary_group_get_solution(self.handles, self.world, u.ctypes.data_as(C.c_void_p), None))
        return u

    def member_solution(self, r):
        """the whole tree as member r holds it (every member holds every block)"""
        u = np.empty((len(self.blocks), self.bs, self.bs, 3))
        L.check(self.lib.mh_binary_get_solution(C.c_void_p(self.handles[r]), u.ctypes.data_as(C.c_void_p), None))
        return u

    def next(self, nsteps=1):
        safe = C.c_int(0)
        L.check(self.lib.mh_binary_group_next(self.handles, self.world, int(nsteps), C.byref(safe)))
        return safe.value

    @property
    def last_dt(self):
        return self.lib.mh_binary_last_dt(C.c_void_p(self.handles[0]))

    def last_failure(self):
        r = L.StepResult()
        L.check(self.lib.mh_binary_last_failure(C.c_void_p(self.handles[0]), C.byref(r)))
        return r.status, (None if r.status == 0 else int(r.first_bad_index))

    def close(self):
        for r in range(self.world):
            if self.handles[r]:
                self.lib.mh_binary_destroy(C.c_void_p(self.handles[r]))
                self.handles[r] = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class BinaryTreeBand(BinaryTreeSolver):
    """ONE member of a distributed graded tree in this process (RCCL: one process per GPU; mh_binary_tree_band_create)."""

    def __init__(self, cfg, rank, world, comm_id, device=0, blocks=None, arith="strict", comm=None):
        t = _TreeSetup(cfg, blocks, arith)
        self.lib, self.cfg, self.setup = t.lib, cfg, t
        self.blocks, self.bs, self.edges, self.u_init, self.buffer_rate, self.desc, self.run = t.blocks, t.bs, t.edges, t.u_init, t.buffer_rate, t.desc, t.run
        self.handle = C.c_void_p()
        idbuf = C.create_string_buffer(bytes(comm_id), 128) if comm_id is not None else None
        L.check(self.lib.mh_binary_tree_band_create(C.byref(self.handle), device, C.byref(t.desc), C.byref(t.run), *t.pointers(), rank, world, idbuf))
        if comm is not None:
            L.check(self.lib.mh_binary_band_use_comm(self.handle, comm.handle))
            self._comm = comm
        s = L.BinaryState()
        s.orbital_elements = initial_elements(cfg)
        self.set_solution(None, s)


class BinaryBandGroup:
    """The uniform-depth mesh cut into `world` BANDS of whole rows of tree blocks, as objects of one process on one GPU exchanging ghost
    rows through the LOOPBACK backend (include/mara_hip.h: mh_binary_group_*; the RCCL form is mh_binary_band_create, one process per
    GPU). Same interface as BinarySolver where it applies; arrays are the WHOLE mesh [n][n][3]."""

    def __init__(self, cfg, world=2, device=0, chunk_rows=0, arith="strict", edge_rows=None):
        """edge_rows: None or 0 = one launch per stage (the library's default); -1 = the recommended cut of every band into edge rows
        (stepped first) and interior; else that many rows per side (mh_binary_band_set_edge_rows)"""
        self.lib = L.load_library()
        self.cfg, self.world = cfg, world
        self.n = grid_size(cfg)
        self.xv = vertices(cfg)
        self.yv = self.xv
        self.u_init, self.buffer_rate, recommended_time_step = solver_data(cfg, self.xv, self.yv)
        self.desc = make_desc(cfg, chunk_rows=chunk_rows, xv=self.xv, yv=self.yv, arith=arith)
        run = L.BinaryRun()
        run.rk_order = int(cfg["rk_order"])
        run.fixed_dt = int(cfg["fixed_dt"])
        run.no_accretion_force = int(cfg["no_accretion_force"])
        run.cfl_number = float(cfg["cfl_number"])
        run.recommended_time_step = float(recommended_time_step)
        run.begin_live_binary = float(cfg["begin_live_binary"])
        self.run = run
        self.handles = (C.c_void_p * world)()
        L.check(self.lib.mh_binary_group_create(self.handles, world, device, C.byref(self.desc), C.byref(run), self.xv.ctypes.data_as(C.c_void_p),
                                                self.yv.ctypes.data_as(C.c_void_p), self.u_init.ctypes.data_as(C.c_void_p),
                                                self.buffer_rate.ctypes.data_as(C.c_void_p)))
        self.rows = []
        for r in range(world):
            a, b = C.c_int(), C.c_int()
            L.check(self.lib.mh_binary_band_rows(C.c_void_p(self.handles[r]), C.byref(a), C.byref(b)))
            self.rows.append((a.value, b.value))
            if edge_rows is not None:
                L.check(self.lib.mh_binary_band_set_edge_rows(C.c_void_p(self.handles[r]), int(edge_rows)))
        s = L.BinaryState()
        s.orbital_elements = initial_elements(cfg)
        self.set_solution(None, s)

    def set_solution(self, u, state):
        up = None if u is None else np.ascontiguousarray(u, dtype=np.float64).ctypes.data_as(C.c_void_p)
        L.check(self.lib.mh_binary_group_set_solution(self.handles, self.world, up, C.byref(state)))

    def state(self):
        s = L.BinaryState()
        L.check(self.lib.mh_binary_group_get_solution(self.handles, self.world, None, C.byref(s)))
        return s

    def solution(self):
        u = np.empty((self.n, self.n, 3))
        L.check(self.lib.mh_binary_group_get_solution(self.handles, self.world, u.ctypes.data_as(C.c_void_p), None))
        return u

    def next(self, nsteps=1):
        safe = C.c_int(0)
        L.check(self.lib.mh_binary_group_next(self.handles, self.world, int(nsteps), C.byref(safe)))
        return safe.value

    @property
    def last_dt(self):
        return self.lib.mh_binary_last_dt(C.c_void_p(self.handles[0]))

    def last_failure(self):
        """as BinarySolver.last_failure; the same record on every member"""
        recs = []
        for r in range(self.world):
            res = L.StepResult()
            L.check(self.lib.mh_binary_last_failure(C.c_void_p(self.handles[r]), C.byref(res)))
            recs.append((res.status, None if res.status == 0 else int(res.first_bad_index)))
        assert all(x == recs[0] for x in recs), recs
        return recs[0]

    def close(self):
        # member 0 owns the stream the others run on: destroy it last
        for r in reversed(range(self.world)):
            if self.handles[r]:
                self.lib.mh_binary_destroy(C.c_void_p(self.handles[r]))
                self.handles[r] = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class BinaryBand(BinarySolver):
    """ONE band of the mesh in this process (RCCL backend: one process per GPU; `comm_id` = the 128-byte unique id every rank shares,
    mara3_amd.slab.native_comm_id). next() is collective over the ranks; solution() returns the whole-mesh array with this band's rows
    filled in (rows [row0, row1))."""

    def __init__(self, cfg, rank, world, comm_id, device=0, chunk_rows=0, arith="strict", self_exchange=False, comm=None, defer=False, edge_rows=None):
        """comm: the process's communicator (mara3_amd.slab.NativeComm) instead of a unique id - no ncclCommInitRank of its own.
        defer=True: neither; the caller attach()es the communicator once every rank holds its band."""
        self.lib = L.load_library()
        self.cfg = cfg
        self.n = grid_size(cfg)
        self.xv = vertices(cfg)
        self.yv = self.xv
        self.u_init, self.buffer_rate, recommended_time_step = solver_data(cfg, self.xv, self.yv)
        self.desc = make_desc(cfg, chunk_rows=chunk_rows, xv=self.xv, yv=self.yv, arith=arith)
        run = L.BinaryRun()
        run.rk_order = int(cfg["rk_order"])
        run.fixed_dt = int(cfg["fixed_dt"])
        run.no_accretion_force = int(cfg["no_accretion_force"])
        run.cfl_number = float(cfg["cfl_number"])
        run.recommended_time_step = float(recommended_time_step)
        run.begin_live_binary = float(cfg["begin_live_binary"])
        self.run = run
        self.handle = C.c_void_p()
        idbuf = C.create_string_buffer(bytes(comm_id), 128) if comm_id is not None else None
        L.check(self.lib.mh_binary_band_create(C.byref(self.handle), device, C.byref(self.desc), C.byref(run), self.xv.ctypes.data_as(C.c_void_p),
                                               self.yv.ctypes.data_as(C.c_void_p), self.u_init.ctypes.data_as(C.c_void_p),
                                               self.buffer_rate.ctypes.data_as(C.c_void_p), rank, world, idbuf, 1 if self_exchange else 0))
        a, b = C.c_int(), C.c_int()
        L.check(self.lib.mh_binary_band_rows(self.handle, C.byref(a), C.byref(b)))
        self.row0, self.row1 = a.value, b.value
        if edge_rows is not None:          # (BinaryBandGroup's docstring)
            L.check(self.lib.mh_binary_band_set_edge_rows(self.handle, int(edge_rows)))
        if comm is not None:
            self.attach(comm)
        elif not defer:
            self._start()

    def attach(self, comm):
        """lend the band the process's communicator (ends with the exchange of the initial ghost rows: collective), then start from the initial state"""
        L.check(self.lib.mh_binary_band_use_comm(self.handle, comm.handle))
        self._comm = comm
        self._start()

    def _start(self):
        s = L.BinaryState()
        s.orbital_elements = initial_elements(self.cfg)
        self.set_solution(None, s)
