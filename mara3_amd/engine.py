"""Python host mirror of the reference's step interface for the uniform-cartesian Euler path.

`EulerCartSolver` plays the role of the reference's (solution_t, advance, next_solution) triple
(src/subprog_cloud.cpp:511-584, :676-697): upload a host AoS field once, call `step(dt, n)`, download
when a task is due. All compute happens in libmara_hip.so through the C ABI; this class owns no arithmetic.
"""
import ctypes as C
import numpy as np
from . import _lib as L


class _StatusMixin:
    """Error contract of the context API (include/mara_hip.h: mh_step_result, mh_status, mh_step_checked)."""

    def status_result(self):
        """(status bits, flat index of the first failing cell in host order, or None); clears the device word"""
        r = L.StepResult()
        L.check(self.lib.mh_status(self.ctx, C.byref(r)), self.ctx)
        return r.status, (None if r.status == 0 else int(r.first_bad_index))

    def step_checked(self, dt):
        """One full time step as a transaction: returns (0, None) and commits, or (bits, first failing cell) and leaves the previous
        solution in place (the reference's callers retry from the OLD solution, src/subprog_binary.cpp:285-292)."""
        r = L.StepResult()
        rc = self.lib.mh_step_checked(self.ctx, dt, C.byref(r))
        if rc == -5:          # MH_E_PHYSICS
            return r.status, int(r.first_bad_index)
        L.check(rc, self.ctx)
        return 0, None


class EulerCartSolver(_StatusMixin):
    def __init__(self, shape, dl, gamma, plm_theta=1.5, riemann="hlle", rk_order=2, bc="outflow",
                 bc_lo0=None, bc_hi0=None, device=0, chunk_rows=0, arith="strict", tail=None, fuse=None, planar=None):
        """tail = (rows, chunk_rows) forces the graded tail of the stage launch (None: the library's default on large grids).
        fuse: the descriptor's fuse_stages - None = where available (one launch per RK2 step: FAST, PLM, physical sides), False = never,
        True = required.
        planar: the descriptor's planar - None = the fused step skips the third momentum where the uploaded 2-D field has none (verified at
        upload), False = never, True = asserted (upload fails on a field that has one)."""
        self.lib = L.load_library()
        self.shape = tuple(int(n) for n in shape)
        rank = len(self.shape)
        d = L.EulerCartDesc()
        d.rank = rank
        for a in range(3):
            d.n[a] = self.shape[a] if a < rank else 1
            d.dl[a] = dl[a] if a < rank else 1.0
        d.gamma = gamma
        d.plm_theta = plm_theta
        d.riemann = {"hlle": L.RIEMANN_HLLE, "hllc": L.RIEMANN_HLLC}[riemann]
        bcs = {"outflow": L.BC_OUTFLOW, "periodic": L.BC_PERIODIC, "external": L.BC_EXTERNAL}
        d.bc_transverse = bcs[bc]
        d.bc_lo0 = bcs[bc_lo0 or bc]
        d.bc_hi0 = bcs[bc_hi0 or bc]
        d.arith = {"strict": L.ARITH_STRICT, "fast": L.ARITH_FAST}[arith]
        d.chunk_rows = chunk_rows
        if tail is not None:
            d.tail_rows, d.tail_chunk_rows = int(tail[0]), int(tail[1])
        d.fuse_stages = 0 if fuse is None else (1 if fuse else -1)
        d.planar = 0 if planar is None else (1 if planar else -1)
        self.desc = d
        self.rk_order = rk_order
        self.ctx = C.c_void_p()
        L.check(self.lib.mh_create(C.byref(self.ctx), device))
        L.check(self.lib.mh_euler_cart_configure(self.ctx, C.byref(d), rk_order), self.ctx)
        self.ncell = int(np.prod(self.shape))

    def close(self):
        if self.ctx:
            self.lib.mh_destroy(self.ctx)
            self.ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def upload(self, u_aos):
        u = np.ascontiguousarray(u_aos, dtype=np.float64)
        assert u.shape == self.shape + (5,), (u.shape, self.shape)
        L.check(self.lib.mh_upload(self.ctx, u.ctypes.data_as(C.c_void_p), self.ncell), self.ctx)

    def download(self):
        u = np.empty(self.shape + (5,), dtype=np.float64)
        L.check(self.lib.mh_download(self.ctx, u.ctypes.data_as(C.c_void_p), self.ncell), self.ctx)
        return u

    def step(self, dt, nsteps=1):
        L.check(self.lib.mh_step(self.ctx, dt, nsteps), self.ctx)

    def synchronize(self):
        L.check(self.lib.mh_synchronize(self.ctx), self.ctx)

    def is_planar(self):
        """True while the fused step takes its planar kernel (the uploaded field had no third momentum)"""
        return bool(self.lib.mh_field_is_planar(self.ctx))

    def status(self):
        s = C.c_int32()
        L.check(self.lib.mh_status_word(self.ctx, C.byref(s)), self.ctx)
        return s.value

    def profile(self, on=True):
        L.check(self.lib.mh_profile_enable(self.ctx, 1 if on else 0), self.ctx)

    def profile_read(self):
        ms, n = C.c_double(), C.c_int()
        L.check(self.lib.mh_profile_read(self.ctx, C.byref(ms), C.byref(n)), self.ctx)
        return ms.value, n.value


class SedovSolver(_StatusMixin):
    """Host mirror of SedovProblem<mara::euler> (src/subprog_sedov.cpp): vertices + volume-integrated conserved state."""

    def __init__(self, vertices, gamma=4.0 / 3, device=0, system="euler"):
        self.lib = L.load_library()
        self.vertices = np.ascontiguousarray(vertices, dtype=np.float64)
        self.nz = self.vertices.size - 1
        d = L.SedovDesc(nz=self.nz, gamma=gamma, system={"euler": 0, "srhd": 2}[system], arith=L.ARITH_STRICT)
        self.ctx = C.c_void_p()
        L.check(self.lib.mh_create(C.byref(self.ctx), device))
        L.check(self.lib.mh_sedov_configure(self.ctx, C.byref(d), self.vertices.ctypes.data_as(C.c_void_p)), self.ctx)

    def timestep(self, cfl=0.4):
        return cfl * (self.vertices[1] - self.vertices[0])      # src/subprog_sedov.cpp:404-405

    def upload(self, u):
        u = np.ascontiguousarray(u, dtype=np.float64)
        assert u.shape == (self.nz, 5)
        L.check(self.lib.mh_upload(self.ctx, u.ctypes.data_as(C.c_void_p), self.nz), self.ctx)

    def download(self):
        u = np.empty((self.nz, 5))
        L.check(self.lib.mh_download(self.ctx, u.ctypes.data_as(C.c_void_p), self.nz), self.ctx)
        return u

    def step(self, dt, nsteps=1):
        L.check(self.lib.mh_step(self.ctx, dt, nsteps), self.ctx)

    def diagnostics(self):
        """SedovProblem::make_diagnostic_fields and the shock locator (subprog_sedov.cpp:252-308) of the resident state ->
        (fields [4][nz]: specific_entropy, gas_pressure, mass_density, radial velocity / gamma-beta; indices: shock, downstream, upstream)"""
        fields, indices = np.empty((4, self.nz)), np.zeros(3, dtype=np.int32)
        L.check(self.lib.mh_sedov_diagnostics(self.ctx, fields.ctypes.data_as(C.c_void_p), indices.ctypes.data_as(C.c_void_p)), self.ctx)
        return fields, indices

    def status(self):
        s = C.c_int32()
        L.check(self.lib.mh_status_word(self.ctx, C.byref(s)), self.ctx)
        return s.value

    def close(self):
        if self.ctx:
            self.lib.mh_destroy(self.ctx)
            self.ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class CloudSolver(_StatusMixin):
    """Host mirror of CloudProblem's (solution_t, advance, next_solution) (src/subprog_cloud.cpp): vertices,
    cell-integrated SRHD conserved state, per-step nozzle-inflow row."""

    def __init__(self, r_vertices, q_vertices, rk_order=1, plm_theta=1.2, temperature_floor=1e-8, gamma=4.0 / 3, device=0, chunk_rows=0,
                 arith="strict", tail=None, fuse=None, planar=None):
        """fuse: the descriptor's fuse_stages - None = where available (one launch per RK2 step: FAST, PLM; csrc/cloud_fused.hip),
        False = never, True = required. planar: the descriptor's planar - None = the fused step skips the azimuthal momentum where field and
        nozzle row have none (verified at upload / set_inflow), False = never, True = asserted."""
        self.lib = L.load_library()
        self.rv = np.ascontiguousarray(r_vertices, dtype=np.float64)
        self.qv = np.ascontiguousarray(q_vertices, dtype=np.float64)
        self.nr, self.nq = self.rv.size - 1, self.qv.size - 1
        d = L.CloudDesc(nr=self.nr, nq=self.nq, nr_global=self.nr, row_offset=0, gamma=gamma, plm_theta=plm_theta,
                        temperature_floor=temperature_floor, bc_lo0=L.BC_INFLOW, bc_hi0=L.BC_OUTFLOW,
                        arith={"strict": L.ARITH_STRICT, "fast": L.ARITH_FAST}[arith], chunk_rows=chunk_rows)
        if tail is not None:
            d.tail_rows, d.tail_chunk_rows = int(tail[0]), int(tail[1])
        d.fuse_stages = 0 if fuse is None else (1 if fuse else -1)
        d.planar = 0 if planar is None else (1 if planar else -1)
        self.ctx = C.c_void_p()
        L.check(self.lib.mh_create(C.byref(self.ctx), device))
        L.check(self.lib.mh_cloud_configure(self.ctx, C.byref(d), self.rv.ctypes.data_as(C.c_void_p),
                                            self.qv.ctypes.data_as(C.c_void_p), rk_order), self.ctx)

    def timestep(self, cfl=0.4):
        return (self.rv[1] - self.rv[0]) / 1.0 * cfl            # src/subprog_cloud.cpp:678-679

    def set_inflow(self, prims):
        p = np.ascontiguousarray(prims, dtype=np.float64)
        assert p.shape == (self.nq, 5)
        L.check(self.lib.mh_cloud_set_inflow(self.ctx, p.ctypes.data_as(C.c_void_p)), self.ctx)

    def upload(self, u):
        u = np.ascontiguousarray(u, dtype=np.float64)
        assert u.shape == (self.nr, self.nq, 5)
        L.check(self.lib.mh_upload(self.ctx, u.ctypes.data_as(C.c_void_p), self.nr * self.nq), self.ctx)

    def download(self):
        u = np.empty((self.nr, self.nq, 5))
        L.check(self.lib.mh_download(self.ctx, u.ctypes.data_as(C.c_void_p), self.nr * self.nq), self.ctx)
        return u

    def diagnostics(self, units):
        """CloudProblem::make_diagnostic_fields (subprog_cloud.cpp:334-433) of the resident state; units = (length, mass, time) in cgs.
        -> (fields [5][nr][nq]: mass_density, gas_pressure, specific_entropy, radial_gamma_beta, radial_energy_flow; columns [15][nq])"""
        un = np.ascontiguousarray(units, dtype=np.float64)
        fields, columns = np.empty((5, self.nr, self.nq)), np.empty((15, self.nq))
        L.check(self.lib.mh_cloud_diagnostics(self.ctx, un.ctypes.data_as(C.c_void_p), fields.ctypes.data_as(C.c_void_p),
                                              columns.ctypes.data_as(C.c_void_p)), self.ctx)
        return fields, columns

    def step(self, dt, nsteps=1):
        L.check(self.lib.mh_step(self.ctx, dt, nsteps), self.ctx)

    def is_planar(self):
        """True while the fused step takes its planar kernel (field and nozzle row without azimuthal momentum)"""
        return bool(self.lib.mh_field_is_planar(self.ctx))

    def status(self):
        s = C.c_int32()
        L.check(self.lib.mh_status_word(self.ctx, C.byref(s)), self.ctx)
        return s.value

    def close(self):
        if self.ctx:
            self.lib.mh_destroy(self.ctx)
            self.ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class DeviceArray:
    """A raw device allocation made through the C ABI (tests of the per-function entry points)."""

    def __init__(self, host):
        self.lib = L.load_library()
        host = np.ascontiguousarray(host, dtype=np.float64)
        self.shape = host.shape
        self.nbytes = host.nbytes
        self.ptr = C.c_void_p()
        L.check(self.lib.mh_malloc(C.byref(self.ptr), max(self.nbytes, 8)))
        if self.nbytes:
            L.check(self.lib.mh_memcpy_h2d(self.ptr, host.ctypes.data_as(C.c_void_p), self.nbytes))

    @classmethod
    def empty(cls, shape):
        return cls(np.zeros(shape))

    def get(self):
        out = np.empty(self.shape, dtype=np.float64)
        L.check(self.lib.mh_device_synchronize())
        if self.nbytes:
            L.check(self.lib.mh_memcpy_d2h(out.ctypes.data_as(C.c_void_p), self.ptr, self.nbytes))
        return out

    def __del__(self):
        try:
            if self.ptr:
                self.lib.mh_free(self.ptr)
        except Exception:
            pass


_ARITH = {"strict": L.ARITH_STRICT, "fast": L.ARITH_FAST}


def plm_gradient(yl, y0, yr, theta, arith="strict"):
    lib = L.load_library()
    a, b, c = DeviceArray(yl), DeviceArray(y0), DeviceArray(yr)
    g = DeviceArray.empty(a.shape)
    L.check(lib.mh_plm_gradient_n(int(np.prod(a.shape)), a.ptr, b.ptr, c.ptr, theta, g.ptr, _ARITH[arith], None))
    return g.get()


def euler_recover_primitive(U, gamma, tfloor=0.0, arith="strict"):
    lib = L.load_library()
    u = DeviceArray(U)
    p = DeviceArray.empty(u.shape)
    L.check(lib.mh_euler_recover_primitive_n(int(np.prod(u.shape)) // 5, u.ptr, gamma, tfloor, p.ptr, _ARITH[arith], None))
    return p.get()


def euler_to_conserved(P, gamma, arith="strict"):
    lib = L.load_library()
    p = DeviceArray(P)
    u = DeviceArray.empty(p.shape)
    L.check(lib.mh_euler_to_conserved_n(int(np.prod(p.shape)) // 5, p.ptr, gamma, u.ptr, _ARITH[arith], None))
    return u.get()


def euler_riemann(Pl, Pr, axis, gamma, solver="hlle", arith="strict"):
    lib = L.load_library()
    a, b = DeviceArray(Pl), DeviceArray(Pr)
    f = DeviceArray.empty(a.shape)
    kind = {"hlle": L.RIEMANN_HLLE, "hllc": L.RIEMANN_HLLC}[solver]
    L.check(lib.mh_euler_riemann_n(int(np.prod(a.shape)) // 5, a.ptr, b.ptr, axis, gamma, kind, f.ptr, _ARITH[arith], None))
    return f.get()


def srhd_recover_primitive(U, gamma=4.0 / 3, tfloor=0.0):
    lib = L.load_library()
    u = DeviceArray(U)
    p = DeviceArray.empty(u.shape)
    n = int(np.prod(u.shape)) // 5
    st = DeviceArray(np.zeros(max(n, 1)))          # int32 words in an 8-byte-per-item buffer
    L.check(lib.mh_srhd_recover_primitive_n(n, u.ptr, gamma, tfloor, p.ptr, st.ptr, None))
    status = st.get().view(np.int32)[:n].copy()
    return p.get(), status


def srhd_to_conserved(P, gamma=4.0 / 3):
    lib = L.load_library()
    p = DeviceArray(P)
    u = DeviceArray.empty(p.shape)
    L.check(lib.mh_srhd_to_conserved_n(int(np.prod(p.shape)) // 5, p.ptr, gamma, u.ptr, None))
    return u.get()


def srhd_riemann_hlle(Pl, Pr, axis, gamma=4.0 / 3):
    lib = L.load_library()
    a, b = DeviceArray(Pl), DeviceArray(Pr)
    f = DeviceArray.empty(a.shape)
    L.check(lib.mh_srhd_riemann_hlle_n(int(np.prod(a.shape)) // 5, a.ptr, b.ptr, axis, gamma, f.ptr, None))
    return f.get()


def srhd_source_terms(P, r, cot_theta, gamma=4.0 / 3):
    lib = L.load_library()
    p, rr, cc = DeviceArray(P), DeviceArray(r), DeviceArray(cot_theta)
    s = DeviceArray.empty(p.shape)
    L.check(lib.mh_srhd_source_terms_n(int(np.prod(p.shape)) // 5, p.ptr, rr.ptr, cc.ptr, gamma, s.ptr, None))
    return s.get()


# ---- mara::iso2d per-function entry points ---------------------------------------------------------------------
def _flags(n):
    return DeviceArray(np.zeros(max(n, 1)))


def _read_flags(arr, n):
    return arr.get().view(np.int32)[:n].copy()


def iso2d_to_conserved(P):
    lib = L.load_library()
    p = DeviceArray(P); u = DeviceArray.empty(p.shape)
    L.check(lib.mh_iso2d_to_conserved_n(int(np.prod(p.shape)) // 3, p.ptr, u.ptr, None))
    return u.get()


def iso2d_recover_primitive(U):
    lib = L.load_library()
    u = DeviceArray(U); p = DeviceArray.empty(u.shape); n = int(np.prod(u.shape)) // 3; f = _flags(n)
    L.check(lib.mh_iso2d_recover_primitive_n(n, u.ptr, p.ptr, f.ptr, None))
    return p.get(), _read_flags(f, n)


def iso2d_to_conserved_angmom(P, x):
    lib = L.load_library()
    p, xx = DeviceArray(P), DeviceArray(x); q = DeviceArray.empty(p.shape)
    L.check(lib.mh_iso2d_to_conserved_angmom_n(int(np.prod(p.shape)) // 3, p.ptr, xx.ptr, q.ptr, None))
    return q.get()


def iso2d_recover_primitive_angmom(Q, x):
    lib = L.load_library()
    q, xx = DeviceArray(Q), DeviceArray(x); p = DeviceArray.empty(q.shape); n = int(np.prod(q.shape)) // 3; f = _flags(n)
    L.check(lib.mh_iso2d_recover_primitive_angmom_n(n, q.ptr, xx.ptr, p.ptr, f.ptr, None))
    return p.get(), _read_flags(f, n)


def iso2d_flux(P, cs2, axis):
    lib = L.load_library()
    p, c = DeviceArray(P), DeviceArray(cs2); f = DeviceArray.empty(p.shape)
    L.check(lib.mh_iso2d_flux_n(int(np.prod(p.shape)) // 3, p.ptr, c.ptr, axis, f.ptr, None))
    return f.get()


def iso2d_wavespeeds(P, cs2, axis):
    lib = L.load_library()
    p, c = DeviceArray(P), DeviceArray(cs2); f = DeviceArray.empty(p.shape)
    L.check(lib.mh_iso2d_wavespeeds_n(int(np.prod(p.shape)) // 3, p.ptr, c.ptr, axis, f.ptr, None))
    return f.get()


def iso2d_riemann(Pl, Pr, cs2l, cs2r, axis, solver="hlle"):
    lib = L.load_library()
    a, b, c, d = DeviceArray(Pl), DeviceArray(Pr), DeviceArray(cs2l), DeviceArray(cs2r)
    n = int(np.prod(a.shape)) // 3
    F = DeviceArray.empty(a.shape); contact = DeviceArray.empty((max(n, 1),)); f = _flags(n)
    kind = {"hlle": L.RIEMANN_HLLE, "hllc": L.RIEMANN_HLLC}[solver]
    L.check(lib.mh_iso2d_riemann_n(n, a.ptr, b.ptr, c.ptr, d.ptr, axis, kind, F.ptr, contact.ptr, f.ptr, None))
    return F.get(), contact.get()[:n], _read_flags(f, n)
