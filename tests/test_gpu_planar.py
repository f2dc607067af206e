"""The planar kernel of the fused 2-D Euler step (mh_euler_cart_desc.planar, euler2d_fused_rk2_kernel<RIEMANN, PLANAR = true>).

mara::euler's state always has three momenta (src/physics_euler.hpp:46); in a 2-D run the third one is identically zero and the reference
carries it as zeros through every operator. Where a stepper has VERIFIED that of the field it was given (one pass at upload), the fused
launch neither reads nor exchanges nor computes that component and writes it as zero. Required here: the other four components keep
their BITS against the general kernel (`x + 0` and `fma(0, 0, x)` are x), the third stays zero; a field that does carry a third momentum
takes the general kernel by itself, with the results of `planar=False`; an asserted `planar=True` refuses such a field; a later upload of
the other kind switches. Covered through the context API, the lone slab of the native stepper (incl. graph replay) and loopback groups
whose slabs take the fused step across their cuts."""
import numpy as np
import pytest
from conftest import bits_equal

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(600)]


@pytest.fixture(scope="module")
def eng():
    import mara3_amd
    from mara3_amd import engine
    assert mara3_amd.load_library().mh_device_count() >= 1
    return engine


def planar_state(shape, seed, third=0.0):
    from mara3_amd import setups
    u = setups.wave_ic(shape, 1.4, seed=seed)
    u[..., 3] = 0.0
    if third:
        rng = np.random.default_rng(seed)
        u[..., 3] = third * u[..., 0] * rng.uniform(-1.0, 1.0, shape)
        u[..., 4] += 0.5 * u[..., 3] ** 2 / u[..., 0]          # keep the pressure
    return u


def run(eng, shape, riemann, bc, planar, u0, nsteps=(1, 2, 3), chunk=0, fuse=True):
    s = eng.EulerCartSolver(shape, (1.0 / shape[0], 1.0 / shape[1]), 1.4, 1.5, riemann, 2, bc, arith="fast", fuse=fuse, planar=planar, chunk_rows=chunk)
    s.upload(u0)
    took = s.is_planar()
    out = []
    for n in nsteps:
        s.step(4e-4, n)
        out.append(s.download())
    st = s.status()
    s.close()
    return out, st, took


@pytest.mark.parametrize("riemann", ["hllc", "hlle"])
@pytest.mark.parametrize("bc", ["outflow", "periodic"])
@pytest.mark.parametrize("shape,chunk", [((250, 300), 0), ((64, 56), 0), ((129, 113), 16), ((67, 200), 7), ((40, 500), 2)])
def test_planar_kernel_keeps_the_bits_of_the_other_four_components(eng, shape, chunk, bc, riemann):
    u0 = planar_state(shape, seed=11)
    general, st_g, took_g = run(eng, shape, riemann, bc, False, u0, chunk=chunk)
    planar, st_p, took_p = run(eng, shape, riemann, bc, None, u0, chunk=chunk)
    assert (took_g, took_p) == (False, True) and st_g == 0 and st_p == 0
    for a, b in zip(planar, general):
        for q in (0, 1, 2, 4):
            assert bits_equal(a[..., q], b[..., q]), (shape, chunk, bc, riemann, q, np.abs(a[..., q] - b[..., q]).max())
        assert np.all(a[..., 3] == 0.0) and np.all(b[..., 3] == 0.0)
    assert not bits_equal(planar[0], u0)


def test_a_field_with_a_third_momentum_takes_the_general_kernel_by_itself(eng):
    shape = (150, 170)
    u0 = planar_state(shape, seed=5, third=0.3)
    auto, _, took = run(eng, shape, "hllc", "outflow", None, u0)
    never, _, _ = run(eng, shape, "hllc", "outflow", False, u0)
    two, _, _ = run(eng, shape, "hllc", "outflow", None, u0, fuse=False)
    assert took is False
    for a, b, c in zip(auto, never, two):
        assert bits_equal(a, b) and bits_equal(a, c)
        assert np.abs(a[..., 3]).max() > 0.0
    # one single cell with a tiny third momentum is enough
    u1 = planar_state(shape, seed=5)
    u1[77, 3, 3] = 1e-300
    assert run(eng, shape, "hlle", "periodic", None, u1, nsteps=(1,))[2] is False
    u1[77, 3, 3] = -0.0          # a negative zero is a zero
    assert run(eng, shape, "hlle", "periodic", None, u1, nsteps=(1,))[2] is True


def test_asserted_planarity_refuses_a_field_that_is_not(eng):
    import mara3_amd
    shape = (64, 80)
    s = eng.EulerCartSolver(shape, (1.0 / 64, 1.0 / 80), 1.4, 1.5, "hllc", 2, "outflow", arith="fast", planar=True)
    with pytest.raises(mara3_amd.MaraHipError, match="third momentum"):
        s.upload(planar_state(shape, seed=2, third=0.1))
    s.upload(planar_state(shape, seed=2))
    assert s.is_planar()
    s.step(4e-4, 2)
    assert s.status() == 0
    s.close()


def test_a_later_upload_of_the_other_kind_switches_kernels(eng):
    shape = (96, 120)
    s = eng.EulerCartSolver(shape, (1.0 / 96, 1.0 / 120), 1.4, 1.5, "hllc", 2, "outflow", arith="fast")
    flat, tilted = planar_state(shape, seed=8), planar_state(shape, seed=8, third=0.2)
    want = {}
    for name, u0 in (("flat", flat), ("tilted", tilted)):
        want[name] = run(eng, shape, "hllc", "outflow", False, u0, nsteps=(3,))[0][0]
    for name, u0, planar in (("flat", flat, True), ("tilted", tilted, False), ("flat", flat, True)):
        s.upload(u0)
        assert s.is_planar() is planar
        s.step(4e-4, 3)
        got = s.download()
        for q in range(5):
            assert np.array_equal(got[..., q], want[name][..., q]), (name, q)          # (== : the third component may be -0 against +0)
    s.close()


@pytest.mark.parametrize("graph", [False, True])
def test_lone_slab_of_the_native_stepper_takes_the_planar_kernel(eng, graph, monkeypatch):
    from mara3_amd.slab import NativeSlabStepper
    if graph:
        monkeypatch.setenv("MH_SLAB_FUSED_GRAPH", "1")
    shape = (200, 260)
    dl = (1.0 / 200, 1.0 / 260)
    flat, tilted = planar_state(shape, seed=4), planar_state(shape, seed=4, third=0.2)
    res = {}
    for planar in (None, False):
        st = NativeSlabStepper(shape, dl, 1.4, 1.5, "hllc", 2, "outflow", arith="fast", planar=planar)
        for name, u0 in (("flat", flat), ("tilted", tilted), ("flat2", flat)):
            st.load_slab(u0)
            assert st.is_planar() is (planar is None and name != "tilted")
            st.step(4e-4, 5, graph=graph)
            st.synchronize()
            res[(planar, name)] = st.slab_host()
        st.close()
    for name in ("flat", "tilted", "flat2"):
        for q in range(5):
            assert np.array_equal(res[(None, name)][..., q], res[(False, name)][..., q]), (name, q)
    assert bits_equal(res[(None, "flat")][..., 4], res[(None, "flat2")][..., 4])


@pytest.mark.parametrize("world,shape", [(3, (48, 200)), (2, (60, 130))])
def test_loopback_group_across_fused_cuts_is_planar_only_if_every_member_is(eng, world, shape, monkeypatch):
    from mara3_amd.slab import NativeSlabGroup
    monkeypatch.setenv("MH_SLAB_FUSED_CUTS", "1")
    dl = (1.0 / shape[0], 1.0 / shape[1])
    flat = planar_state(shape, seed=6)
    one_row = flat.copy()
    one_row[shape[0] - 2, 10, 3] = 0.05          # a third momentum in the LAST member's rows only
    for u0, want_planar in ((flat, True), (one_row, False)):
        g = NativeSlabGroup(shape, dl, 1.4, 1.5, "hllc", 2, "periodic", world=world, arith="fast")
        g.upload(u0)
        assert [bool(g.lib.mh_slab_is_planar(h)) for h in g.handles] == [want_planar] * world
        g.step(4e-4, 4)
        g.synchronize()
        got = g.download()
        assert g.status()[0] == 0
        g.close()
        ref = run(eng, shape, "hllc", "periodic", False, u0, nsteps=(4,))[0][0]
        for q in range(5):
            assert np.array_equal(got[..., q], ref[..., q]), (world, q, want_planar)


# ---- the two-launch kernels (euler2d.hip): STRICT must keep the REFERENCE's bits, third momentum included

@pytest.mark.parametrize("arith", ["strict", "fast"])
@pytest.mark.parametrize("riemann", ["hlle", "hllc"])
@pytest.mark.parametrize("bc", ["outflow", "periodic"])
def test_two_launch_planar_kernels_are_bit_identical_to_the_general_ones(eng, arith, riemann, bc):
    """a field whose third momentum is +0.0 in every cell: STRICT every bit of all five components (the reference's operations return +0.0 for
    the third direction, euler_device.hpp), FAST the bits of the other four"""
    for shape, chunk in (((130, 250), 0), ((67, 200), 7), ((40, 64), 2)):
        u0 = planar_state(shape, seed=13)
        res = {}
        for planar in (None, False):
            s = eng.EulerCartSolver(shape, (1.0 / shape[0], 1.0 / shape[1]), 1.4, 1.5, riemann, 2, bc, arith=arith, fuse=False, planar=planar, chunk_rows=chunk)
            s.upload(u0)
            assert s.is_planar() is (planar is None)
            s.step(4e-4, 3)
            res[planar] = s.download()
            assert s.status() == 0
            s.close()
        if arith == "strict":
            assert bits_equal(res[None], res[False]), (shape, riemann, bc)
        else:
            for q in (0, 1, 2, 4):
                assert bits_equal(res[None][..., q], res[False][..., q]), (shape, riemann, bc, q)
            assert np.all(res[None][..., 3] == 0.0)


def test_strict_planarity_is_about_the_bit_pattern(eng):
    """STRICT keeps a -0.0 of the reference a -0.0: a field with a negative zero in its third momentum takes the general kernel"""
    shape = (64, 80)
    u0 = planar_state(shape, seed=3)
    u0[10, 20, 3] = -0.0
    s = eng.EulerCartSolver(shape, (1.0 / 64, 1.0 / 80), 1.4, 1.5, "hlle", 2, "outflow", arith="strict")
    s.upload(u0)
    assert s.is_planar() is False
    s.close()
    f = eng.EulerCartSolver(shape, (1.0 / 64, 1.0 / 80), 1.4, 1.5, "hlle", 2, "outflow", arith="fast")
    f.upload(u0)
    assert f.is_planar() is True
    f.close()


@pytest.mark.parametrize("case", ["euler2d_blast64_plm15_rk2", "euler2d_blast128_plm15_rk2", "euler2d_wave33x70_plm12_rk2_outflow"])
def test_strict_planar_kernel_reproduces_the_reference_golden_steps(eng, case):
    """the reference's own golden steps (its third momentum: identically zero) on the STRICT planar kernel, bit for bit incl. that component"""
    from conftest import golden
    g = golden(case)
    u0 = g["u0"]
    nsteps = sorted(int(n) for n in g["nsteps"])
    s = eng.EulerCartSolver(u0.shape[:-1], tuple(g["dl"]), float(g["gamma"]), float(g["theta"]), "hlle", int(g["rk"]), "outflow" if int(g["bc"]) == 0 else "periodic", arith="strict")
    s.upload(u0)
    planar = s.is_planar()
    assert planar is bool(np.all(u0[..., 3].view(np.uint64) == 0))
    done = 0
    for n in nsteps:
        s.step(float(g["dt"]), n - done)
        done = n
        assert bits_equal(s.download(), g["u_%d" % n]), (case, n, planar)
    s.close()


@pytest.mark.parametrize("arith", ["fast", "strict"])
def test_context_with_an_external_side_does_not_go_planar_on_its_own_rows(eng, arith):
    """advisor finding, round 4: a context with an MH_BC_EXTERNAL side sees only its own rows at mh_upload - its ghost rows are the caller's
    (mh_field_ptr) and may carry a third momentum. planar = 0 must then mean the GENERAL kernels (as for slabs that exchange with other
    processes); planar > 0 is the caller's word for the whole grid. UPPER half of a grid whose lower half has a third momentum (the flow
    of wave_ic runs towards higher rows, so that momentum crosses the cut), RK1, one step: bit-identical to the same rows of the whole-grid
    run - which the planar kernel (it never reads component 3 of the ghost rows) is not."""
    import ctypes as C
    from mara3_amd import _lib as L
    n0, n1 = 64, 96
    whole = planar_state((2 * n0, n1), seed=21)
    tilted = planar_state((2 * n0, n1), seed=21, third=0.4)
    whole[:n0] = tilted[:n0]                                           # rows [0, n0) with a third momentum, rows [n0, 2 n0) planar
    dl = (1.0 / (2 * n0), 1.0 / n1)
    ref = eng.EulerCartSolver((2 * n0, n1), dl, 1.4, 1.5, "hllc", 1, "outflow", arith=arith, planar=False)
    ref.upload(whole)
    ref.step(4e-4, 1)
    want = ref.download()[n0:]
    ref.close()
    assert np.abs(want[:2, :, 3]).max() > 0.0                              # the lower half's third momentum does cross the cut within the step

    def upper_half(planar):
        s = eng.EulerCartSolver((n0, n1), dl, 1.4, 1.5, "hllc", 1, "outflow", bc_lo0="external", arith=arith, planar=planar)
        s.upload(whole[n0:])
        took = s.is_planar()
        # the two ghost rows below row 0: rows n0 - 2, n0 - 1 of the whole grid, in the device layout ((i + 2) * 5 + q) * pitch + t, i = -2, -1
        ghost = np.ascontiguousarray(np.transpose(whole[n0 - 2:n0], (0, 2, 1)))        # [row][q][t]
        lib = L.load_library()
        lib.mh_field_ptr.restype = C.c_void_p
        base = lib.mh_field_ptr(s.ctx, 0)
        L.check(lib.mh_memcpy_h2d(C.c_void_p(base), ghost.ctypes.data_as(C.c_void_p), ghost.nbytes))
        s.step(4e-4, 1)
        got = s.download()
        s.close()
        return took, got

    took, got = upper_half(None)
    assert took is False and bits_equal(got, want)
    took_asserted, got_asserted = upper_half(True)                        # the caller's (here: wrong) word is taken
    assert took_asserted is True and not bits_equal(got_asserted, want)
