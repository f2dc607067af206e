"""GPU parity tests for mara::srhd and the `cloud` stage (BASELINE config 4) through the C ABI, against golden
vectors produced by the reference's own headers. MH_ARITH_STRICT: bit-exact; north_star's bound (conserved L1
<= 1e-12) is asserted too."""
import glob
import math
import os
import numpy as np
import pytest
from conftest import golden, bits_equal, l1, GOLDEN

pytestmark = pytest.mark.gpu
G = 4.0 / 3


@pytest.fixture(scope="module")
def eng():
    import mara3_amd
    from mara3_amd import engine
    assert mara3_amd.load_library().mh_device_count() >= 1
    return engine


def test_srhd_functions_bit_exact(eng):
    g = golden("srhd_functions")
    assert bits_equal(eng.srhd_to_conserved(g["Pl"], G), g["U"])
    P, st = eng.srhd_recover_primitive(g["U"], G, 0.0)
    assert not st.any()
    assert bits_equal(P, g["c2p"])
    for axis in range(3):
        assert bits_equal(eng.srhd_riemann_hlle(g["Pl"], g["Pr"], axis, G), g["hlle_%d" % axis]), axis
    cot = np.array([math.tan(math.pi / 2 - q) for q in g["src_q"]])       # libm tan, as the host geometry code
    assert bits_equal(eng.srhd_source_terms(g["Pl"], g["src_r"], cot, G), g["src"])


@pytest.mark.parametrize("name,floor", [("floor", 1e-8), ("nofloor", 0.0)])
def test_srhd_recover_primitive_status_matches_reference_exceptions(eng, name, floor):
    g = golden("srhd_functions")
    P, st = eng.srhd_recover_primitive(g["Ubad"], G, floor)
    threw = g["c2p_bad_%s_threw" % name] != 0
    assert np.array_equal(st != 0, threw)
    assert bits_equal(P[~threw], g["c2p_bad_" + name][~threw])


CLOUD_CASES = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, "cloud_*.npz")))


@pytest.mark.parametrize("planar", [None, False])
@pytest.mark.parametrize("chunk", [0, 5])
@pytest.mark.parametrize("case", CLOUD_CASES)
def test_cloud_steps_vs_reference_golden(eng, case, chunk, planar):
    """the reference's own steps, bit for bit in all five components - by the general kernels (planar=False) and by the planar ones, which
    the stepper takes by itself: upstream's cloud has no azimuthal momentum (+0.0 in every cell and in the nozzle row)"""
    g = golden(case)
    theta = float(g["theta"]) if int(g["method"]) == 2 else -1.0
    s = eng.CloudSolver(g["rv"], g["qv"], int(g["rk"]), theta, float(g["tfloor"]), chunk_rows=chunk, planar=planar)
    s.upload(g["u0"])
    assert abs(s.timestep() - float(g["dt"])) == 0.0
    for n in range(int(g["nsteps"])):
        s.set_inflow(g["inflow"][n])          # nozzle row at the step-start time, used by both RK stages
        assert s.is_planar() == (planar is None and theta >= 0.0)
        s.step(float(g["dt"]), 1)
    got = s.download()
    assert s.status() == 0
    assert l1(got, g["un"]) <= 1e-12
    assert bits_equal(got, g["un"]), np.abs(got - g["un"]).max()


@pytest.mark.parametrize("case", CLOUD_CASES)
def test_fast_cloud_steps_within_tolerance_of_reference(eng, case):
    """MH_ARITH_FAST for mara::srhd (rcp / rsq arithmetic, FMAs): not bit-exact. Tolerance = the north star's: conserved-variable
    L1 <= 1e-12, taken per variable RELATIVE to that variable's mean magnitude - the momentum vector shares one scale - because
    cell-integrated conserved quantities of the cloud problem span many decades: an absolute bound would be vacuous for some and
    unattainable for others."""
    g = golden(case)
    theta = float(g["theta"]) if int(g["method"]) == 2 else -1.0
    s = eng.CloudSolver(g["rv"], g["qv"], int(g["rk"]), theta, float(g["tfloor"]), arith="fast")
    s.upload(g["u0"])
    for n in range(int(g["nsteps"])):
        s.set_inflow(g["inflow"][n])
        s.step(float(g["dt"]), 1)
    got = s.download()
    assert s.status() == 0
    scale = np.abs(g["un"]).reshape(-1, 5).mean(axis=0)
    scale[1:4] = scale[1:4].max()          # one scale for the momentum vector (a component can vanish by symmetry)
    err = np.abs(got - g["un"]).reshape(-1, 5).mean(axis=0)
    assert np.all(err <= 1e-12 * scale), err / scale


@pytest.mark.parametrize("theta,rk", [(1.2, 1), (2.0, 1), (-1.0, 1)])      # (one stage: random data do not survive a second one)
def test_fast_cloud_stage_on_random_extreme_states_lands_on_the_strict_result(eng, theta, rk):
    """Uncorrelated random cells - Lorentz factors up to 30 in either radial direction, polar motion, pressures from 1e-6 rho to 30 rho,
    densities over four decades - so that every face is an extreme Riemann problem, the limiter sees every sign pattern and the Newton
    iteration of recover_primitive starts far from its root: the FAST kernel (leaner Newton step, unscaled limiter, zero pole constants,
    conserved rows in LDS) must land on the STRICT kernel's update in all of them."""
    rng = np.random.default_rng(int(1000 * abs(theta)) + rk)
    nr, nq = 48, 130                                    # three strips of 60 columns, ragged; both poles
    rv = np.logspace(0.0, 0.5, nr + 1)
    qv = np.linspace(0.0, np.pi, nq + 1)
    rho = 10.0 ** rng.uniform(-3.0, 1.0, (nr, nq))
    P = np.zeros((nr, nq, 5))
    P[..., 0] = rho
    P[..., 1] = rng.uniform(-1.0, 1.0, (nr, nq)) * 10.0 ** rng.uniform(-2.0, 1.5, (nr, nq))      # radial gamma-beta
    P[..., 2] = rng.uniform(-0.5, 0.5, (nr, nq))
    P[..., 4] = rho * 10.0 ** rng.uniform(-6.0, 1.5, (nr, nq))
    U = eng.srhd_to_conserved(P.reshape(-1, 5)).reshape(nr, nq, 5)
    dmu = -np.cos(qv[1:]) - -np.cos(qv[:-1])
    dv = ((rv[1:] ** 3 - rv[:-1] ** 3)[:, None] * dmu[None, :] * 2 * np.pi) / 3
    u0 = U * dv[..., None]
    inflow = P[0].copy()
    dt = 0.05 * (rv[1] - rv[0])
    out = {}
    for arith in ("strict", "fast"):
        s = eng.CloudSolver(rv, qv, rk, theta, 0.0, arith=arith)
        s.upload(u0)
        s.set_inflow(inflow)
        s.step(dt, 1)
        out[arith] = s.download()
        st = s.status()
        assert st == 0, st
    a, b = out["strict"], out["fast"]
    ok = np.isfinite(a).all(axis=-1) & np.isfinite(b).all(axis=-1)
    assert ok.mean() > 0.99
    scale = np.abs(a[ok]).mean(axis=0)
    scale[1:4] = scale[1:4].max()
    assert np.all(np.abs(a[ok] - b[ok]).mean(axis=0) <= 1e-12 * scale), np.abs(a[ok] - b[ok]).mean(axis=0) / scale
    # no cell is off by more than the reach of one Newton step taken or not taken at the |f| < 1e-10 threshold
    cell = np.abs(a[ok]).copy()
    cell[:, 1:4] = np.abs(a[ok][:, 1:4]).max(axis=1, keepdims=True)
    assert np.all(np.abs(a[ok] - b[ok]) <= 1e-8 * np.maximum(cell, scale[None, :] * 1e-6))


def test_cloud_reports_c2p_failure_in_status_word(eng):
    g = golden("cloud_nr32_plm_rk1")
    u = g["u0"].copy()
    u[10, 12, 4] = -abs(u[10, 12, 4])          # negative tau: the reference's recover_primitive throws
    s = eng.CloudSolver(g["rv"], g["qv"], 1, 1.2, 0.0)
    s.upload(u)
    s.set_inflow(g["inflow"][0])
    s.step(float(g["dt"]), 1)
    assert s.status() != 0


def test_sedov_srhd_bit_exact_vs_reference(eng):
    """`mara sedov` with its default system mara::srhd (512 zones, PCM + HLLE + forward Euler) after 1/10/100 steps."""
    g = golden("sedov_srhd_nr256")
    s = eng.SedovSolver(g["vertices"], system="srhd")
    s.upload(g["u0"])
    dt = s.timestep()
    done = 0
    for n in (1, 10, 100):
        s.step(dt, n - done)
        done = n
        assert bits_equal(s.download(), g["u_%d" % n]), n
    assert s.status() == 0


def test_cloud_slabs_on_one_gpu_bit_identical_to_single_domain(eng):
    """BASELINE config 4 is a radial slab decomposition: run the `cloud` stage on two and three radial slabs in
    one process (MH_BC_EXTERNAL sides, row_offset, ghost rows copied between the slabs as the RCCL exchange
    would) and require the union to equal the single-domain reference-generated result bit for bit."""
    import ctypes as C
    from mara3_amd import _lib as L
    from mara3_amd.engine import DeviceArray
    lib = L.load_library()
    g = golden("cloud_nr70_plm_rk2")
    rv, qv, u0 = g["rv"], g["qv"], g["u0"]
    nr, nq = u0.shape[0], u0.shape[1]
    dt, nsteps = float(g["dt"]), int(g["nsteps"])
    for nslabs in (2, 3):
        cuts = [(k * nr) // nslabs for k in range(nslabs + 1)]
        slabs = []
        for k in range(nslabs):
            a, b = cuts[k], cuts[k + 1]
            d = L.CloudDesc(nr=b - a, nq=nq, nr_global=nr, row_offset=a, gamma=4.0 / 3, plm_theta=float(g["theta"]),
                            temperature_floor=float(g["tfloor"]), bc_lo0=L.BC_INFLOW if k == 0 else L.BC_EXTERNAL,
                            bc_hi0=L.BC_OUTFLOW if k == nslabs - 1 else L.BC_EXTERNAL, arith=L.ARITH_STRICT, chunk_rows=7)
            geom = np.zeros(lib.mh_cloud_geometry_doubles(C.byref(d)))
            L.check(lib.mh_cloud_pack_geometry(C.byref(d), rv.ctypes.data_as(C.c_void_p), qv.ctypes.data_as(C.c_void_p), geom.ctypes.data_as(C.c_void_p)))
            # host copies of the two fields in the device layout [n0+4][5][nq]
            f = np.zeros((b - a + 4, 5, nq))
            f[2:-2] = u0[a:b].transpose(0, 2, 1)
            slabs.append(dict(d=d, n0=b - a, geom=DeviceArray(geom), u=DeviceArray(f), s=DeviceArray(np.zeros_like(f)),
                              status=DeviceArray(np.zeros(1))))

        def exchange(key):
            host = [sl[key].get() for sl in slabs]
            for k in range(nslabs - 1):
                lo, hi = host[k], host[k + 1]
                lo[slabs[k]["n0"] + 2:] = hi[2:4]
                hi[0:2] = lo[slabs[k]["n0"]:slabs[k]["n0"] + 2]
            for sl, h in zip(slabs, host):
                sl[key] = DeviceArray(h)

        def stage(src, dst, base, w, inflow):
            for sl in slabs:
                L.check(lib.mh_cloud_stage(C.byref(sl["d"]), sl["geom"].ptr, inflow.ptr, sl[src].ptr,
                                           sl[base].ptr if base else None, sl[dst].ptr, dt, w, 0, sl["n0"], sl["status"].ptr, None))
            L.check(lib.mh_device_synchronize())

        exchange("u")
        for n in range(nsteps):
            inflow = DeviceArray(np.ascontiguousarray(g["inflow"][n].T))      # [5][nq]
            stage("u", "s", None, 1.0, inflow)
            exchange("s")
            stage("s", "u", "u", 0.5, inflow)
            exchange("u")
        got = np.concatenate([sl["u"].get()[2:-2].transpose(0, 2, 1) for sl in slabs], axis=0)
        assert bits_equal(got, g["un"]), (nslabs, np.abs(got - g["un"]).max())


class _NoExchange:
    """Tells the stepper which sides are cut; the test moves the ghost rows itself (one process, one GPU)."""
    def __init__(self, lo, hi):
        self.lo, self.hi = lo, hi

    def start(self, field, n0):
        return []

    @staticmethod
    def finish(reqs):
        pass


@pytest.mark.parametrize("nslabs", [2, 3])
def test_slab_cloud_stepper_on_one_gpu_matches_reference(eng, nslabs):
    """mara3_amd.slab.SlabCloudStepper (the product's radial-slab host path of BASELINE config 4) with the real HIP stage,
    edge / interior launch split included; the exchange is done by the test between the slabs of one process."""
    import torch
    from mara3_amd.slab import SlabCloudStepper, HALO
    g = golden("cloud_nr70_plm_rk2")
    sts = []
    for r in range(nslabs):
        st = SlabCloudStepper(g["rv"], g["qv"], rk_order=2, plm_theta=float(g["theta"]), temperature_floor=float(g["tfloor"]), rank=r, world=nslabs,
                              device="cuda", exchange=_NoExchange(r - 1 if r > 0 else None, r + 1 if r < nslabs - 1 else None),
                              overlap=False, chunk_rows=9, edge_chunk_rows=4)
        st.load_slab(g["u0"][st.row0:st.row1])
        sts.append(st)

    def exchange(field_of):
        for k in range(nslabs - 1):
            lo, hi = sts[k], sts[k + 1]
            field_of(lo)[lo.n0 + HALO:lo.n0 + 2 * HALO] = field_of(hi)[HALO:2 * HALO]
            field_of(hi)[0:HALO] = field_of(lo)[lo.n0:lo.n0 + HALO]

    dt = float(g["dt"])
    exchange(lambda s: s.u)
    for n in range(int(g["nsteps"])):
        for st in sts:
            st.set_inflow(g["inflow"][n])
            st.compute_stage(st.u, None, st.scratch, dt, 1.0)
        exchange(lambda s: s.scratch)
        for st in sts:
            st.compute_stage(st.scratch, st.u, st.u, dt, 0.5)
        exchange(lambda s: s.u)
    torch.cuda.synchronize()
    assert all(st.status() == 0 for st in sts)
    got = np.concatenate([st.slab().cpu().numpy() for st in sts], axis=0)
    assert bits_equal(got, g["un"]), np.abs(got - g["un"]).max()


@pytest.mark.parametrize("tail", [(12, 4), (9, 2)])
@pytest.mark.parametrize("case", CLOUD_CASES)
def test_cloud_graded_tail_is_bit_exact(eng, case, tail):
    """The graded tail of the cloud stage launch (cloud.hip: the last rows go to short waves at the end of every XCD's share) forced on
    the small golden grids: bit-identical to the reference."""
    g = golden(case)
    theta = float(g["theta"]) if int(g["method"]) == 2 else -1.0
    s = eng.CloudSolver(g["rv"], g["qv"], int(g["rk"]), theta, float(g["tfloor"]), tail=tail)
    s.upload(g["u0"])
    for n in range(int(g["nsteps"])):
        s.set_inflow(g["inflow"][n])
        s.step(float(g["dt"]), 1)
    assert s.status() == 0
    assert bits_equal(s.download(), g["un"])
