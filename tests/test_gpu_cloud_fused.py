"""The fused RK2 step of `cloud` (mara3_amd/csrc/cloud_fused.hip: both stages of `s0 * 0.5 + advance(advance(s0)) * 0.5`,
src/subprog_cloud.cpp:676-697 with `advance` :511-584, in ONE launch - the first-stage field lives in an LDS ring between a producer and
a consumer wave) against the two launches of cloud_stage_kernel it replaces. Same SrhdFast functions on the same values in the same
order, so the requirement is BIT-IDENTITY with the two-launch FAST path, which tests/test_gpu_srhd_cloud.py::test_fast_* hold to the
reference (conserved L1 <= 1e-12 per variable); the reference's RK2 + PLM golden steps are asserted here on the fused launch directly
too. Covered: the reference's cases (incl. two decades of radius), ragged shapes (strips of 116 columns and chunks that do not divide
the grid, one-row last chunks, a grid narrower than a pair), chunk lengths down to 2, odd and even step counts (the two fields swap),
a changing nozzle row, the status contract and the transactional step."""
import glob
import os
import numpy as np
import pytest
from conftest import golden, bits_equal, GOLDEN

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(600)]

RK2_PLM_CASES = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, "cloud_*_plm_rk2.npz")))


@pytest.fixture(scope="module")
def eng():
    import mara3_amd
    from mara3_amd import engine
    assert mara3_amd.load_library().mh_device_count() >= 1
    return engine


def smooth_cloud_state(eng, nr, nq, seed):
    """a smooth relativistic flow on a logarithmic radial grid from pole to pole: cell-integrated conserved state, nozzle rows, dt"""
    rng = np.random.default_rng(seed)
    rv = np.logspace(0.0, 0.6, nr + 1)
    qv = np.linspace(0.0, np.pi, nq + 1)
    rc = 0.5 * (rv[1:] + rv[:-1])[:, None]
    qc = 0.5 * (qv[1:] + qv[:-1])[None, :]
    k1, k2, k3 = rng.uniform(1.0, 3.0, 3)
    P = np.zeros((nr, nq, 5))
    P[..., 0] = (1.0 + 0.3 * np.sin(k1 * qc) * np.cos(k2 * np.log(rc))) / rc ** 2
    P[..., 1] = 0.8 + 0.5 * np.cos(k3 * qc) * np.sin(2.0 * np.log(rc))
    P[..., 2] = 0.1 * np.sin(2.0 * qc) * np.cos(k1 * np.log(rc))
    P[..., 4] = 0.05 * P[..., 0] * (1.0 + 0.5 * np.sin(k2 * qc))
    U = eng.srhd_to_conserved(P.reshape(-1, 5)).reshape(nr, nq, 5)
    dmu = -np.cos(qv[1:]) - -np.cos(qv[:-1])
    dv = ((rv[1:] ** 3 - rv[:-1] ** 3)[:, None] * dmu[None, :] * 2 * np.pi) / 3
    inflow = np.zeros((8, nq, 5))
    for n in range(8):                                  # a nozzle that changes from step to step
        inflow[n, :, 0] = P[0, :, 0] * (1.0 + 0.02 * n)
        inflow[n, :, 1] = P[0, :, 1] + 0.05 * n * np.exp(-(qc[0] / 0.3) ** 2) + 0.05 * n * np.exp(-((np.pi - qc[0]) / 0.3) ** 2)
        inflow[n, :, 4] = P[0, :, 4]
    return rv, qv, U * dv[..., None], inflow, 0.3 * (rv[1] - rv[0])


def run(eng, rv, qv, u0, inflow, dt, pieces, fuse, chunk=0, theta=1.2, tfloor=0.0):
    s = eng.CloudSolver(rv, qv, 2, theta, tfloor, arith="fast", fuse=fuse, chunk_rows=chunk)
    s.upload(u0)
    out, k = [], 0
    for n in pieces:
        for _ in range(n):
            s.set_inflow(inflow[k % len(inflow)])
            s.step(dt, 1)
            k += 1
        out.append(s.download())
    st = s.status_result()
    s.close()
    return out, st


@pytest.mark.parametrize("chunk", [0, 2, 5, 23])
@pytest.mark.parametrize("case", RK2_PLM_CASES)
def test_fused_cloud_step_on_the_reference_cases(eng, case, chunk):
    """bit-identical to the two FAST launches, and within north_star's tolerance of the reference's own steps (per variable, relative to its mean)"""
    g = golden(case)
    n = int(g["nsteps"])
    two, st2 = run(eng, g["rv"], g["qv"], g["u0"], g["inflow"], float(g["dt"]), (n,), False, chunk, float(g["theta"]), float(g["tfloor"]))
    one, st1 = run(eng, g["rv"], g["qv"], g["u0"], g["inflow"], float(g["dt"]), (n,), True, chunk, float(g["theta"]), float(g["tfloor"]))
    assert st1 == (0, None) and st2 == (0, None)
    assert bits_equal(one[0], two[0]), np.abs(one[0] - two[0]).max()
    scale = np.abs(g["un"]).reshape(-1, 5).mean(axis=0)
    scale[1:4] = scale[1:4].max()
    err = np.abs(one[0] - g["un"]).reshape(-1, 5).mean(axis=0)
    assert np.all(err <= 1e-12 * scale), err / scale


@pytest.mark.parametrize("nr,nq,chunk", [(130, 250, 0), (97, 116, 32), (64, 117, 9), (33, 57, 2), (41, 300, 7), (200, 64, 3), (96, 1000, 0), (12, 3, 0), (5, 40, 4)])
def test_fused_cloud_step_is_bit_identical_to_the_two_launches(eng, nr, nq, chunk):
    rv, qv, u0, inflow, dt = smooth_cloud_state(eng, nr, nq, seed=nr * 1000 + nq)
    pieces = (1, 2, 3)                       # odd and even step counts: the result lands in either field
    two, st2 = run(eng, rv, qv, u0, inflow, dt, pieces, False, chunk)
    one, st1 = run(eng, rv, qv, u0, inflow, dt, pieces, True, chunk)
    assert st1 == (0, None) and st2 == (0, None), (st1, st2)
    for a, b, n in zip(one, two, pieces):
        assert np.isfinite(a).all()
        assert bits_equal(a, b), (nr, nq, chunk, n, np.abs(a - b).max())
    assert not bits_equal(one[0], u0)


def test_default_is_the_fused_step_where_it_exists_and_never_means_never(eng, monkeypatch):
    """fuse=None takes the one-launch step (one launch per step between the profile events), fuse=False two"""
    rv, qv, u0, inflow, dt = smooth_cloud_state(eng, 64, 120, seed=5)
    counts = {}
    for fuse in (None, False):
        s = eng.CloudSolver(rv, qv, 2, 1.2, 0.0, arith="fast", fuse=fuse)
        s.upload(u0)
        s.set_inflow(inflow[0])
        s.lib.mh_profile_enable(s.ctx, 1)
        s.step(dt, 3)
        import ctypes as C
        ms, n = C.c_double(), C.c_int()
        assert s.lib.mh_profile_read(s.ctx, C.byref(ms), C.byref(n)) == 0
        counts[fuse] = n.value
        s.close()
    assert counts == {None: 3, False: 6}


def test_required_fusion_is_refused_where_it_does_not_exist(eng):
    import mara3_amd
    rv, qv, u0, inflow, dt = smooth_cloud_state(eng, 32, 40, seed=1)
    with pytest.raises(mara3_amd.MaraHipError, match="fuse_stages"):
        eng.CloudSolver(rv, qv, 2, 1.2, 0.0, arith="strict", fuse=True)
    with pytest.raises(mara3_amd.MaraHipError, match="fuse_stages"):
        eng.CloudSolver(rv, qv, 1, 1.2, 0.0, arith="fast", fuse=True)
    with pytest.raises(mara3_amd.MaraHipError, match="fuse_stages"):
        eng.CloudSolver(rv, qv, 2, -1.0, 0.0, arith="fast", fuse=True)          # piecewise constant
    # STRICT with fuse=None: two launches, bit-identical to the reference as before
    s = eng.CloudSolver(rv, qv, 2, 1.2, 0.0, arith="strict")
    s.upload(u0); s.set_inflow(inflow[0]); s.step(dt, 1)
    assert s.status() == 0
    s.close()


@pytest.mark.parametrize("where", [(0, 0), (17, 59), (17, 60), (40, 119), (63, 5)])
def test_status_contract_of_the_fused_cloud_step(eng, where):
    """a cell the reference's recover_primitive throws on (negative tau): the same status bits and the same first failing cell as the two
    launches, in the first row, at the seam between the two pairs of a workgroup, at a pole and in the last row. (The first stage's NaNs
    reach the second stage's recover_primitive in the neighbouring rows, so the FIRST failing cell of the step may lie up to two rows before
    the poisoned one - in both forms alike.)"""
    rv, qv, u0, inflow, dt = smooth_cloud_state(eng, 64, 120, seed=9)
    u = u0.copy()
    u[where[0], where[1], 4] = -abs(u[where[0], where[1], 4]) * 10.0
    _, st2 = run(eng, rv, qv, u, inflow, dt, (1,), False, 16)
    _, st1 = run(eng, rv, qv, u, inflow, dt, (1,), True, 16)
    assert st2[0] != 0 and st1 == st2, (st1, st2)
    assert max(where[0] - 2, 0) * 120 <= st1[1] <= where[0] * 120 + where[1]


def test_transactional_step_with_the_fused_cloud_launch(eng):
    rv, qv, u0, inflow, dt = smooth_cloud_state(eng, 48, 130, seed=3)
    s = eng.CloudSolver(rv, qv, 2, 1.2, 0.0, arith="fast", fuse=True)
    s.upload(u0)
    s.set_inflow(inflow[0])
    assert s.step_checked(dt) == (0, None)
    good = s.download()
    two, _ = run(eng, rv, qv, u0, inflow, dt, (1,), False)
    assert bits_equal(good, two[0])
    bad = good.copy()
    bad[20, 70, 4] = -1.0
    s.upload(bad)
    bits, cell = s.step_checked(dt)
    assert bits != 0 and 18 * 130 <= cell <= 20 * 130 + 70
    assert bits_equal(s.download(), bad)                 # the previous solution is still in place
    s.close()


# ---- the planar kernel (mh_cloud_desc.planar): field and nozzle row without azimuthal momentum - the `cloud` problem as upstream sets it up

def run_planar(eng, rv, qv, u0, inflow, dt, nsteps, planar, fuse=True):
    s = eng.CloudSolver(rv, qv, 2, 1.2, 0.0, arith="fast", fuse=fuse, planar=planar)
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


@pytest.mark.parametrize("nr,nq", [(130, 250), (64, 117), (33, 57), (96, 1000)])
def test_planar_cloud_kernel_keeps_the_bits_of_the_other_four_components(eng, nr, nq):
    rv, qv, u0, inflow, dt = smooth_cloud_state(eng, nr, nq, seed=nr + nq)          # (no azimuthal motion anywhere, like upstream's problem)
    general, st_g, took_g = run_planar(eng, rv, qv, u0, inflow, dt, 4, False)
    planar, st_p, took_p = run_planar(eng, rv, qv, u0, inflow, dt, 4, None)
    assert took_g == [False] * 4 and took_p == [True] * 4 and st_g == (0, None) and st_p == (0, None)
    for q in (0, 1, 2, 4):
        assert bits_equal(planar[..., q], general[..., q]), (nr, nq, q, np.abs(planar[..., q] - general[..., q]).max())
    assert np.all(planar[..., 3] == 0.0) and np.all(general[..., 3] == 0.0)


def test_azimuthal_motion_in_field_or_nozzle_takes_the_general_cloud_kernel(eng):
    import mara3_amd
    nr, nq = 64, 120
    rv, qv, u0, inflow, dt = smooth_cloud_state(eng, nr, nq, seed=21)
    # (i) a field with azimuthal momentum: general kernel from the start, equal to the two launches
    spun = u0.copy()
    spun[..., 3] = 0.05 * np.abs(u0[..., 1])
    one, _, took = run_planar(eng, rv, qv, spun, inflow, dt, 3, None)
    two, _, _ = run_planar(eng, rv, qv, spun, inflow, dt, 3, None, fuse=False)
    assert took == [False] * 3 and bits_equal(one, two) and np.abs(one[..., 3]).max() > 0.0
    # (ii) a nozzle row that starts to rotate at the third step: planar until then, general from then on - as if it had been general throughout
    rows = inflow[:4].copy()
    rows[2:, :, 3] = 0.01
    auto, _, took = run_planar(eng, rv, qv, u0, rows, dt, 4, None)
    never, _, _ = run_planar(eng, rv, qv, u0, rows, dt, 4, False)
    assert took == [True, True, False, False]
    for q in range(5):
        assert np.array_equal(auto[..., q], never[..., q]), q
    assert np.abs(auto[..., 3]).max() > 0.0
    # (iii) asserted planarity refuses both
    s = eng.CloudSolver(rv, qv, 2, 1.2, 0.0, arith="fast", planar=True)
    with pytest.raises(mara3_amd.MaraHipError, match="azimuthal"):
        s.upload(spun)
    s.upload(u0)
    with pytest.raises(mara3_amd.MaraHipError, match="azimuthal"):
        s.set_inflow(rows[3])
    s.close()


# ---- round 5: the one-launch step across RADIAL CUTS (BASELINE config 4 is a 4-slab run) -------------------------------------------------
# Both RK stages use the step-start nozzle row (src/subprog_cloud.cpp:466-493, :524), so a radial slab takes the fused launch as the whole
# field does: four rows of each neighbour, ONE exchange per step, the neighbours' first-stage rows recomputed inside the launch; the nozzle
# rows apply on the slab that owns row 0, the zero-gradient copy and the zeroed edge slope (:563) on the one that owns the last row.

def run_slabs(rv, qv, u0, inflow, dt, pieces, world, fuse, chunk=0, planar=None):
    from mara3_amd.slab import NativeSlabGroup
    g = NativeSlabGroup(world=world, rk_order=2, plm_theta=1.2, gamma=4.0 / 3, arith="fast", r_vertices=rv, q_vertices=qv, temperature_floor=0.0,
                        fuse=fuse, chunk_rows=chunk, planar=planar)
    lps, rows = g.launches_per_step(), list(g.rows)
    g.upload(u0)
    out, k = [], 0
    for n in pieces:
        for _ in range(n):
            g.set_inflow(inflow[k % len(inflow)])
            g.step(dt, 1)
            k += 1
        g.synchronize()
        out.append(g.download())
    st = g.status()
    g.close()
    return out, st, lps, rows


@pytest.mark.parametrize("nr,nq,world,chunk", [(96, 130, 2, 0), (97, 250, 3, 0), (130, 117, 4, 0), (61, 64, 4, 5), (50, 300, 3, 2), (200, 57, 4, 23), (48, 40, 4, 0)])
def test_fused_cloud_step_across_radial_cuts_is_the_whole_field_run(eng, nr, nq, world, chunk, monkeypatch):
    """2 - 4 radial slabs (uneven cuts where world does not divide nr: nd::partition_shape), slabs down to twelve rows, chunk lengths that do not
    divide the segments, a nozzle row that changes every step, odd and even step counts - bit-identical to the ONE-launch step of the whole
    field, and (therefore) to the two FAST launches per slab that MH_SLAB_FUSED_CUTS=0 keeps"""
    monkeypatch.setenv("MH_SLAB_FUSED_CUTS", "1")          # (default: from 384 rows per slab on)
    rv, qv, u0, inflow, dt = smooth_cloud_state(eng, nr, nq, seed=nr + 7 * nq + world)
    pieces = (1, 2, 3)
    whole, st_w = run(eng, rv, qv, u0, inflow, dt, pieces, True, chunk)
    cut, st_c, lps, rows = run_slabs(rv, qv, u0, inflow, dt, pieces, world, None, chunk)
    assert lps == [1] * world, lps                        # EVERY slab takes the fused launch
    assert rows[0][0] == 0 and rows[-1][1] == nr and all(a[1] == b[0] for a, b in zip(rows, rows[1:]))
    assert st_w == (0, None) and st_c == (0, None)
    for a, b, n in zip(cut, whole, pieces):
        assert bits_equal(a, b), (nr, nq, world, chunk, n, np.abs(a - b).max())
    monkeypatch.setenv("MH_SLAB_FUSED_CUTS", "0")
    two, st_t, lps2, _ = run_slabs(rv, qv, u0, inflow, dt, pieces, world, None, chunk)
    assert lps2 == [2] * world and st_t == (0, None)
    for a, b in zip(two, whole):
        assert bits_equal(a, b)


def test_fused_cuts_are_the_default_from_384_rows_per_slab_and_can_be_refused_or_required(eng, monkeypatch):
    import mara3_amd
    monkeypatch.delenv("MH_SLAB_FUSED_CUTS", raising=False)
    rv, qv, u0, inflow, dt = smooth_cloud_state(eng, 800, 64, seed=3)
    cut, st, lps, _ = run_slabs(rv, qv, u0, inflow, dt, (2,), 2, None)
    whole, _ = run(eng, rv, qv, u0, inflow, dt, (2,), True)
    assert lps == [1, 1] and st == (0, None) and bits_equal(cut[0], whole[0])
    assert run_slabs(rv, qv, u0, inflow, dt, (1,), 4, None)[2] == [2] * 4          # 200 rows per slab: the two launches
    assert run_slabs(rv, qv, u0, inflow, dt, (1,), 2, False)[2] == [2, 2]           # fuse_stages < 0: never
    rv, qv, u0, inflow, dt = smooth_cloud_state(eng, 40, 64, seed=3)
    with pytest.raises(mara3_amd.MaraHipError, match="fuse_stages is required"):
        run_slabs(rv, qv, u0, inflow, dt, (1,), 4, True)                            # ten rows per slab: cannot


def test_fused_cuts_general_kernel_with_azimuthal_motion_and_the_status_contract(eng, monkeypatch):
    """a field WITH azimuthal momentum takes the general fused kernel on every slab (same bits as the whole field); a cell that fails
    recover_primitive is reported once, with its GLOBAL flat index, by the slab that owns it - not by the neighbour that recomputes its row"""
    monkeypatch.setenv("MH_SLAB_FUSED_CUTS", "1")
    nr, nq, world = 90, 70, 3
    rv, qv, u0, inflow, dt = smooth_cloud_state(eng, nr, nq, seed=12)
    rng = np.random.default_rng(4)
    spin = u0.copy()
    spin[..., 3] = 0.05 * spin[..., 0] * rng.uniform(-1.0, 1.0, (nr, nq))
    whole, st_w = run(eng, rv, qv, spin, inflow, dt, (3,), True)
    cut, st_c, lps, rows = run_slabs(rv, qv, spin, inflow, dt, (3,), world, None)
    assert lps == [1] * world and st_w == (0, None) and st_c == (0, None)
    assert bits_equal(cut[0], whole[0]) and np.abs(cut[0][..., 3]).max() > 0.0
    bad = u0.copy()
    i, j = rows[1][0], 33                               # the first row of the middle slab: slab 0 recomputes it as a first-stage ghost row
    bad[i, j, 4] = -abs(bad[i, j, 4])                   # negative energy: recover_primitive fails there
    _, st_w = run(eng, rv, qv, bad, inflow, dt, (1,), True)
    _, st_c, _, _ = run_slabs(rv, qv, bad, inflow, dt, (1,), world, None)
    # (the failure may spread to the cells around it within the step: what is pinned is that slabs and whole field report the SAME bits and first cell)
    assert st_w[0] != 0 and st_c == st_w and st_w[1] <= i * nq + j and st_w[1] >= (i - 3) * nq, (st_w, st_c)


def test_fused_cuts_stay_bit_identical_through_a_long_run(eng, monkeypatch):
    """400 steps of a 512 x 256 grid over four slabs of 128 rows (the exchange's four-row blocks, the recomputed first-stage rows and the nozzle row
    that changes every step, 400 times over): the slabs' final state is the whole-field run's, bit for bit, and so are the status words"""
    monkeypatch.setenv("MH_SLAB_FUSED_CUTS", "1")
    rv, qv, u0, inflow, dt = smooth_cloud_state(eng, 512, 256, seed=77)
    whole, st_w = run(eng, rv, qv, u0, inflow, 0.5 * dt, (400,), True)
    cut, st_c, lps, _ = run_slabs(rv, qv, u0, inflow, 0.5 * dt, (400,), 4, None)
    assert lps == [1] * 4 and st_w == st_c
    assert bits_equal(cut[0], whole[0]) and np.isfinite(whole[0]).all()


def test_tapered_interior_launch_of_cloud_slabs_is_the_same_step(eng, monkeypatch):
    """cloud_fused.hip carries euler2d_fused.hip's TAPER (the interior launch of a radial slab with neighbours ends in shorter chunks, launched
    last): on request on a small grid - by default it applies from 24-row short chunks on, i.e. to config 4's 1024-row slabs - the launch does cut
    its rows that way and the slabs' state is the whole-field run's, bit for bit"""
    import ctypes as C
    import mara3_amd
    lib = mara3_amd.load_library()
    monkeypatch.setenv("MH_SLAB_FUSED_CUTS", "1")
    rv, qv, u0, inflow, dt = smooth_cloud_state(eng, 594, 1400, seed=41)          # 13 strips: 59 chunks per strip and round
    whole, st_w = run(eng, rv, qv, u0, inflow, dt, (3,), True)
    for taper, least, cut in (("3", "2", (5, 2)), ("0", "24", None)):
        monkeypatch.setenv("MH_FUSED_TAPER_ROWS", taper)
        monkeypatch.setenv("MH_FUSED_TAPER_MIN", least)
        got_cut, st_c, lps, _ = run_slabs(rv, qv, u0, inflow, dt, (3,), 2, None)
        out = (C.c_int32 * 4)()
        assert lib.mh_debug_last_fused_cut(3, out) == 0
        if cut is not None:
            assert tuple(out)[:2] == cut and out[3] > out[2] > 0, tuple(out)
        else:
            assert out[0] == out[1], tuple(out)
        assert lps == [1, 1] and st_c == st_w == (0, None)
        assert bits_equal(got_cut[0], whole[0]), (taper, least)
