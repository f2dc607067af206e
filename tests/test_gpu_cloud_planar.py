"""The planar forms of the `cloud` STAGE kernels (mh_cloud_desc.planar; csrc/cloud.hip, csrc/srhd_device.hpp): a field and a nozzle row
without azimuthal momentum - upstream's problem, src/subprog_cloud.cpp:626-660 and :466-493 - are advanced without that component.
MH_ARITH_STRICT takes them only on the bit pattern of +0.0 and must return what the general kernel returns, bit for bit, in all FIVE
components (the general kernel is the one pinned to the reference's golden steps and long runs); MH_ARITH_FAST keeps the bits of the other four.
The one-launch FAST step has its own tests (test_gpu_cloud_fused.py)."""
import numpy as np
import pytest
from conftest import bits_equal
from test_gpu_cloud_fused import smooth_cloud_state

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(600)]


@pytest.fixture(scope="module")
def eng():
    import mara3_amd
    from mara3_amd import engine
    assert mara3_amd.load_library().mh_device_count() >= 1
    return engine


def run(eng, rv, qv, u0, inflow, dt, nsteps, planar, arith, rk=2, theta=1.2, chunk=0, tfloor=0.0):
    s = eng.CloudSolver(rv, qv, rk, theta, tfloor, arith=arith, fuse=False, planar=planar, chunk_rows=chunk)
    s.upload(u0)
    took = []
    for n in range(nsteps):
        s.set_inflow(inflow[n % len(inflow)])
        took.append(s.is_planar())
        s.step(dt, 1)
    out = s.download()
    st = s.status_result()
    s.close()
    return out, st, took


def extreme_cloud_state(eng, nr, nq, seed):
    """uncorrelated random cells (every sign pattern in the limiter, inward and outward motion, both signs of u_r u_q and of cot(theta)):
    one stage only - such data do not survive a second one"""
    rng = np.random.default_rng(seed)
    rv = np.logspace(0.0, 0.5, nr + 1)
    qv = np.linspace(0.0, np.pi, nq + 1)
    rho = 10.0 ** rng.uniform(-3.0, 1.0, (nr, nq))
    P = np.zeros((nr, nq, 5))
    P[..., 0] = rho
    P[..., 1] = rng.uniform(-1.0, 1.0, (nr, nq)) * 10.0 ** rng.uniform(-2.0, 1.0, (nr, nq))
    P[..., 2] = rng.uniform(-0.5, 0.5, (nr, nq))
    P[..., 4] = rho * 10.0 ** rng.uniform(-6.0, 1.5, (nr, nq))
    # cells at rest, cells that move along one axis only: products like u_r u_q are then zeros of either sign
    P[::5, ::3, 1] = 0.0
    P[::7, 1::4, 2] = 0.0
    P[3::11, 2::5, 1:3] = 0.0
    P[1::6, ::2, 2] = -0.0
    U = eng.srhd_to_conserved(P.reshape(-1, 5)).reshape(nr, nq, 5)
    dmu = -np.cos(qv[1:]) - -np.cos(qv[:-1])
    dv = ((rv[1:] ** 3 - rv[:-1] ** 3)[:, None] * dmu[None, :] * 2 * np.pi) / 3
    u0 = U * dv[..., None]
    u0[..., 3] = 0.0
    return rv, qv, u0, P[:1].copy(), 0.05 * (rv[1] - rv[0])


@pytest.mark.parametrize("rk", [1, 2])
@pytest.mark.parametrize("nr,nq,chunk", [(130, 250, 0), (64, 117, 9), (33, 57, 2), (96, 1000, 0), (12, 3, 0), (5, 40, 4), (41, 300, 7)])
def test_strict_planar_stages_return_the_general_kernels_bits_in_all_five_components(eng, nr, nq, chunk, rk):
    rv, qv, u0, inflow, dt = smooth_cloud_state(eng, nr, nq, seed=nr + nq)
    general, st_g, took_g = run(eng, rv, qv, u0, inflow, dt, 4, False, "strict", rk, chunk=chunk)
    planar, st_p, took_p = run(eng, rv, qv, u0, inflow, dt, 4, None, "strict", rk, chunk=chunk)
    assert took_g == [False] * 4 and took_p == [True] * 4 and st_g == (0, None) and st_p == (0, None)
    assert bits_equal(planar, general), [np.abs(planar[..., q] - general[..., q]).max() for q in range(5)]
    assert not np.signbit(planar[..., 3]).any() and np.all(planar[..., 3] == 0.0)          # +0.0, as the reference leaves it


@pytest.mark.parametrize("theta", [1.2, 2.0])
@pytest.mark.parametrize("seed", [1, 2, 3])
def test_strict_planar_stage_on_random_extreme_states(eng, seed, theta):
    """signs of zero: faces at rest, one-axis motion, -0.0 velocities, pole cells - wherever the azimuthal zero could reach another component
    through a sum or a product, the planar kernel keeps the general kernel's operation"""
    rv, qv, u0, inflow, dt = extreme_cloud_state(eng, 48, 130, seed)
    general, st_g, took_g = run(eng, rv, qv, u0, inflow, dt, 1, False, "strict", 1, theta)
    planar, st_p, took_p = run(eng, rv, qv, u0, inflow, dt, 1, None, "strict", 1, theta)
    assert took_g == [False] and took_p == [True] and st_g == st_p
    ok = np.isfinite(general).all(axis=-1)
    assert ok.mean() > 0.99
    assert bits_equal(planar[ok], general[ok])
    assert np.array_equal(np.isfinite(planar).all(axis=-1), ok)


@pytest.mark.parametrize("nr,nq", [(130, 250), (33, 57), (96, 1000)])
def test_fast_planar_stages_keep_the_bits_of_the_other_four_components_and_match_the_one_launch_step(eng, nr, nq):
    rv, qv, u0, inflow, dt = smooth_cloud_state(eng, nr, nq, seed=nr + nq)
    general, st_g, took_g = run(eng, rv, qv, u0, inflow, dt, 4, False, "fast")
    planar, st_p, took_p = run(eng, rv, qv, u0, inflow, dt, 4, None, "fast")
    assert took_g == [False] * 4 and took_p == [True] * 4 and st_g == (0, None) and st_p == (0, None)
    for q in (0, 1, 2, 4):
        assert bits_equal(planar[..., q], general[..., q]), q
    assert np.all(planar[..., 3] == 0.0)
    s = eng.CloudSolver(rv, qv, 2, 1.2, 0.0, arith="fast", fuse=True)
    s.upload(u0)
    for n in range(4):
        s.set_inflow(inflow[n])
        s.step(dt, 1)
    assert bits_equal(s.download(), planar)
    s.close()


def test_strict_takes_the_planar_stages_only_on_the_bit_pattern_of_plus_zero(eng):
    import mara3_amd
    nr, nq = 64, 120
    rv, qv, u0, inflow, dt = smooth_cloud_state(eng, nr, nq, seed=5)
    minus = u0.copy()
    minus[10, 7, 3] = -0.0
    # a -0.0 in the field: STRICT stays with the general kernels (the reference's operations need not return +0.0 from it), FAST does not mind
    out_m, _, took = run(eng, rv, qv, minus, inflow, dt, 2, None, "strict")
    ref_m, _, _ = run(eng, rv, qv, minus, inflow, dt, 2, False, "strict")
    assert took == [False, False] and bits_equal(out_m, ref_m)
    _, _, took_fast = run(eng, rv, qv, minus, inflow, dt, 2, None, "fast")
    assert took_fast == [True, True]
    # a -0.0 in the nozzle row, from the second step on
    rows = inflow[:3].copy()
    rows[1:, 5, 3] = -0.0
    auto, _, took = run(eng, rv, qv, u0, rows, dt, 3, None, "strict")
    never, _, _ = run(eng, rv, qv, u0, rows, dt, 3, False, "strict")
    assert took == [True, False, False] and bits_equal(auto, never)
    # a rotating nozzle: general kernels from that step on, as if they had run throughout
    rows = inflow[:4].copy()
    rows[2:, :, 3] = 0.01
    auto, _, took = run(eng, rv, qv, u0, rows, dt, 4, None, "strict")
    never, _, _ = run(eng, rv, qv, u0, rows, dt, 4, False, "strict")
    assert took == [True, True, False, False] and bits_equal(auto, never) and np.abs(auto[..., 3]).max() > 0.0
    # asserted planarity refuses them
    s = eng.CloudSolver(rv, qv, 2, 1.2, 0.0, arith="strict", planar=True)
    with pytest.raises(mara3_amd.MaraHipError, match="azimuthal"):
        s.upload(minus)
    s.upload(u0)
    with pytest.raises(mara3_amd.MaraHipError, match="azimuthal"):
        s.set_inflow(rows[3])
    s.close()


def test_piecewise_constant_steps_have_no_planar_form(eng):
    rv, qv, u0, inflow, dt = smooth_cloud_state(eng, 40, 70, seed=9)
    _, st, took = run(eng, rv, qv, u0, inflow, dt, 2, None, "strict", 1, theta=-1.0)
    assert took == [False, False] and st == (0, None)


@pytest.mark.parametrize("arith", ["strict", "fast"])
@pytest.mark.parametrize("world", [2, 3])
def test_slab_group_takes_the_planar_stages_and_matches_the_whole_field_run(eng, world, arith):
    from mara3_amd import slab
    nr, nq = 96, 130
    rv, qv, u0, inflow, dt = smooth_cloud_state(eng, nr, nq, seed=world)
    whole, st, took = run(eng, rv, qv, u0, inflow, dt, 4, None, arith)
    assert took == [True] * 4 and st == (0, None)
    g = slab.NativeSlabGroup(world=world, rk_order=2, plm_theta=1.2, gamma=4.0 / 3, arith=arith, r_vertices=rv, q_vertices=qv, temperature_floor=0.0)
    g.upload(u0)
    assert not g.is_planar()                                  # (round 5: a cloud slab knows no nozzle row until it is handed one)
    for n in range(4):
        g.set_inflow(inflow[n])
        assert g.is_planar()
        g.step(dt, 1)
    g.synchronize()
    assert g.status() == (0, None)
    assert bits_equal(g.download(), whole)
    # a rotating nozzle reaches every member in the same call
    rows = inflow[:2].copy()
    rows[1, :, 3] = 0.02
    g.upload(u0)
    g.set_inflow(rows[0])
    assert g.is_planar()
    g.step(dt, 1)
    g.set_inflow(rows[1])
    assert not g.is_planar()
    g.step(dt, 1)
    g.synchronize()
    ref, _, took = run(eng, rv, qv, u0, rows, dt, 2, None, arith)
    assert took == [True, False] and bits_equal(g.download(), ref)
    # advisor finding, round 4: a host that hands the rotating row to the nozzle-side member ONLY (the per-slab call) must not leave the other
    # members on their planar kernels: they all go general with it, and the result is the whole-field run's
    import ctypes as C
    from mara3_amd import _lib as L
    g.upload(u0)
    g.set_inflow(rows[0])
    assert g.is_planar()
    g.step(dt, 1)
    p = np.ascontiguousarray(rows[1])
    L.check(g.lib.mh_slab_set_inflow(C.c_void_p(g.handles[0]), p.ctypes.data_as(C.c_void_p)))
    assert not any(bool(g.lib.mh_slab_is_planar(C.c_void_p(h))) for h in g.handles)
    g.step(dt, 1)
    g.synchronize()
    assert bits_equal(g.download(), ref)
    # ... and through the per-slab call with a planar row no member GAINS the planar kernels (the group call resolves them together)
    g.upload(u0)
    L.check(g.lib.mh_slab_set_inflow(C.c_void_p(g.handles[0]), np.ascontiguousarray(rows[0]).ctypes.data_as(C.c_void_p)))
    assert not g.is_planar()
    g.close()
