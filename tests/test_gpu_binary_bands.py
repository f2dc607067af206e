"""Multi-GPU decomposition of the `binary` sub-program's uniform-depth mesh (BASELINE config 3) into BANDS of whole rows of tree blocks
(mara3_amd/csrc/binary_api.hip), executed on ONE GPU through the loopback backend: 2, 3 and 4 band objects of one process, per stage a
two-row ghost exchange with the periodic neighbours, per synchronisation a sum of the 2 x 18 totals over the bands. The reference hands
whole blocks to its thread pool (tree.map(fn, pool), src/core_tree.hpp:615-625) and sums the per-block totals in tree order.

While the binary is not live the FIELD cannot depend on the partition: bit-identical to the single-domain run. The totals and everything
derived from them (accumulators, orbital elements) agree to the order of summation."""
import numpy as np
import pytest
from conftest import golden, bits_equal

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(600)]


@pytest.fixture(scope="module")
def binary():
    import mara3_amd
    from mara3_amd import binary
    assert mara3_amd.load_library().mh_device_count() >= 1
    return binary


def scalars_close(a, b, rtol=1e-9):
    """scalars() of two runs: time / iteration exact; the accumulators (sums over all cells, with cancellation) to rtol of their own size
    + 1e-11 of the largest of them, as tests/test_gpu_binary.py does against the reference; orbital elements to rtol."""
    if not (a[0] == b[0] and a[1] == b[1]):
        return False
    acc_a, acc_b = a[2:12], b[2:12]
    gross = np.abs(acc_a).max()
    ok = np.all(np.abs(acc_a - acc_b) <= rtol * np.abs(acc_a) + 1e-11 * gross)
    return bool(ok and np.allclose(a[12:], b[12:], rtol=rtol, atol=1e-13))


def scalars(binary, s):
    d = binary.state_as_dict(s)
    out = [d["time"], float(d["iteration"])]
    for k in ("mass_accreted_on", "angular_momentum_accreted_on", "integrated_torque_on", "work_done_on"):
        out += list(d[k])
    out += [d["mass_ejected"], d["angular_momentum_ejected"]]
    for k in ("orbital_elements_acc", "orbital_elements_grav", "orbital_elements"):
        out += list(d[k])
    return np.array(out)


@pytest.mark.parametrize("world", [2, 3, 4])
@pytest.mark.parametrize("overrides", [
    dict(depth=2, block_size=16),                                             # 64^2, CFL time step (max wavespeed reduced over the bands)
    dict(depth=2, block_size=16, fixed_dt=1, rk_order=1),
    dict(depth=2, block_size=16, conserve_linear_p=0),                        # advance_q
    dict(depth=3, block_size=8, axisymmetric_cs2=1, nu=1e-3, alpha=0.0, alpha_cutoff_radius=0.5, density_floor=1e-6),
    dict(depth=2, block_size=32, mass_ratio=0.5, eccentricity=0.3),
])
@pytest.mark.parametrize("arith", ["strict", "fast"])
def test_bands_equal_single_domain(binary, overrides, world, arith):
    cfg = binary.config(**overrides)
    one = binary.BinarySolver(cfg, arith=arith)
    grp = binary.BinaryBandGroup(cfg, world=world, arith=arith)
    nblock_rows = binary.grid_size(cfg) // int(cfg["block_size"])
    assert [b - a for a, b in grp.rows] == [((r + 1) * nblock_rows // world - r * nblock_rows // world) * int(cfg["block_size"]) for r in range(world)]
    for nsteps in (1, 3):
        assert one.next(nsteps) == 0
        assert grp.next(nsteps) == 0
        assert grp.last_dt == one.last_dt
        assert bits_equal(grp.solution(), one.solution()), (overrides, world, nsteps)
        a, b = scalars(binary, one.state()), scalars(binary, grp.state())
        assert scalars_close(a, b), np.abs(a - b).max()
    one.close(); grp.close()


@pytest.mark.parametrize("arith", ["strict", "fast"])
@pytest.mark.parametrize("overrides, world", [(dict(depth=2, block_size=16), 2), (dict(depth=2, block_size=16, conserve_linear_p=0), 3),
                                              (dict(depth=3, block_size=8, fixed_dt=1), 4), (dict(depth=3, block_size=32, fixed_dt=1), 2)])
def test_edge_rows_first_changes_no_bit_of_the_field(binary, overrides, world, arith):
    """Round 3: a band steps its first and last rows in one small launch, then the interior (so that, between GPUs, the ghost exchange of
    the stage travels beside the interior launch - here, loopback, the copies stand between the two). Whatever the cut - none, the
    recommended one, 2, 3 rows - the field is the single-domain field bit for bit; the scalars agree to the order of summation."""
    cfg = binary.config(**overrides)
    one = binary.BinarySolver(cfg, arith=arith)
    safe = one.next(4)          # (the 8-zone blocks at fixed_dt need a safe-mode step: the bands must take the same one)
    want, ws = one.solution(), scalars(binary, one.state())
    n0_min = min(b - a for a, b in binary.BinaryBandGroup(cfg, world=world).rows)
    for edge in (None, -1, 2, 3):
        if edge and edge > 0 and n0_min - 2 * edge < 1:
            continue
        grp = binary.BinaryBandGroup(cfg, world=world, arith=arith, edge_rows=edge)
        assert grp.next(1) + grp.next(3) == safe
        assert bits_equal(grp.solution(), want), (overrides, world, edge)
        assert scalars_close(scalars(binary, grp.state()), ws)
        grp.close()
    one.close()


def test_edge_rows_that_do_not_fit_are_refused(binary):
    from mara3_amd import _lib as L
    cfg = binary.config(depth=2, block_size=16)
    with pytest.raises(L.MaraHipError, match="edge rows"):
        binary.BinaryBandGroup(cfg, world=4, edge_rows=8)          # bands of 16 rows: 2 x 8 leave no interior
    with pytest.raises(L.MaraHipError, match="edge rows"):
        binary.BinaryBandGroup(cfg, world=4, edge_rows=1)


@pytest.mark.parametrize("name", ["binary_d2_b16", "binary_d3_b8_axisym", "binary_d2_b32", "binary_d2_b16_q"])
def test_bands_reproduce_the_reference_vectors(binary, name):
    """The vectors of the reference-composed driver (tests/golden/binary_*.npz) through 4 bands, at the single-domain test's tolerance
    (1e-12 of the field scale per component; tests/test_gpu_binary.py states why it is not bit-exact)."""
    import json
    g = golden(name)
    over = json.loads(str(g["config"]))
    cfg = binary.config(**{k: v for k, v in over.items() if k not in ("nsteps", "safe_mode")})
    grp = binary.BinaryBandGroup(cfg, world=4)
    assert bits_equal(grp.u_init, g["u_init"]) and bits_equal(grp.buffer_rate, g["br"])          # host set-up: bit-exact
    for _ in range(int(over["nsteps"])):
        assert grp.next(1) == 0
    got, want = grp.solution(), g["u_final"]
    scale = np.abs(want).reshape(-1, 3).max(axis=0)
    err = np.abs(got - want).reshape(-1, 3).max(axis=0)
    assert np.all(err <= 1e-12 * scale), err / scale
    grp.close()


@pytest.mark.parametrize("world", [2, 4])
def test_bands_with_a_live_binary(binary, world):
    """t > begin_live_binary: the second stage's body positions depend on the first stage's totals (two synchronisations per step), so
    the field inherits the summation-order difference of the totals: tolerance instead of bit identity."""
    cfg = binary.config(depth=2, block_size=16, begin_live_binary=0.0, mass_ratio=0.7)
    one = binary.BinarySolver(cfg)
    grp = binary.BinaryBandGroup(cfg, world=world)
    one.next(1); grp.next(1)          # the first step starts at t = 0 (not live yet), the next ones are live
    one.next(3); grp.next(3)
    u, v = one.solution(), grp.solution()
    scale = np.abs(u).reshape(-1, 3).mean(axis=0)
    assert np.all(np.abs(u - v).reshape(-1, 3).mean(axis=0) <= 1e-11 * scale)
    a, b = scalars(binary, one.state()), scalars(binary, grp.state())
    assert scalars_close(a, b, rtol=1e-8), np.abs(a - b).max()
    one.close(); grp.close()


def test_bands_safe_mode_retry_is_collective(binary):
    """A negative density in ONE band makes the whole team retry from the old solution (dt * 0.1, theta = 0, src/subprog_binary.cpp:285-292)
    and land where the single-domain solver lands."""
    cfg = binary.config(depth=2, block_size=16, fixed_dt=1)
    one = binary.BinarySolver(cfg)
    grp = binary.BinaryBandGroup(cfg, world=4)
    u = one.solution()
    u[50, 20, 0] = 1e-14          # nearly empty cell next to full ones: the PLM step drives it negative
    u[50, 20, 1:] = 0.0
    s = one.state()
    one.set_solution(u, s); grp.set_solution(u, s)
    sa, sb = one.next(1), grp.next(1)
    assert sa == sb
    assert bits_equal(grp.solution(), one.solution())
    one.close(); grp.close()
