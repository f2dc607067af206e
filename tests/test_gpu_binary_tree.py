"""GPU parity tests of `binary` on GRADED block trees against vectors produced with the reference's own
tree machinery (oracle/ref_drivers/binary_tree_ref.cpp: create_vertex_quadtree, ensure_valid_quadtree, get_cell_block,
restrict_extrinsic). Tolerances as in tests/test_gpu_binary.py: 1e-12 of each field's largest magnitude (device libm)."""
import json
import numpy as np
import pytest
from conftest import golden

pytestmark = pytest.mark.gpu
CASES = ["binary_tree_d3_b8", "binary_tree_d4_b8_default_focus", "binary_tree_d3_b12_nu", "binary_tree_d2_b16_uniform", "binary_tree_d3_b8_q", "binary_tree_d2_b16_q_uniform"]


@pytest.fixture(scope="module")
def binary():
    import mara3_amd
    from mara3_amd import binary
    assert mara3_amd.load_library().mh_device_count() >= 1
    return binary


def cfg_of(binary, g, **extra):
    over = json.loads(str(g["config"]))
    cfg = binary.config(**{k: v for k, v in over.items() if k not in ("nsteps", "safe_mode")})
    cfg.update(extra)
    return cfg, over


def close(a, b, rel=1e-12):
    scale = np.abs(b).reshape(-1, 3).max(axis=0)
    err = np.abs(a - b).reshape(-1, 3).max(axis=0)
    return bool(np.all(err <= rel * scale)), err / scale


def make(binary, cfg, g, dt=None, arith="strict"):
    return binary.BinaryTreeSolver(cfg, blocks=g["blocks"], edges=g["xv"], u_init=g["u_init"], buffer_rate=g["br"],
                                   recommended_time_step=g["stage_scalars"][1] if dt is None else dt, arith=arith)


@pytest.mark.parametrize("name", CASES)
def test_one_stage_on_a_graded_tree(binary, name):
    """One advance_u (guard zones across refinement jumps, flux correction, per-block totals): a forward-Euler step with the
    reference's dt is exactly one stage."""
    g = golden(name)
    ss = g["stage_scalars"]
    cfg, _ = cfg_of(binary, g, rk_order=1, fixed_dt=1)
    s = make(binary, cfg, g, dt=ss[0])
    assert s.next(1) == 0 and s.last_dt == ss[0]
    ok, rel = close(s.solution(), g["u_stage"])
    assert ok, rel
    st = binary.state_as_dict(s.state())
    acc = np.array(st["mass_accreted_on"] + st["angular_momentum_accreted_on"] + st["integrated_torque_on"] + st["work_done_on"] + [st["mass_ejected"], st["angular_momentum_ejected"]])
    ref = ss[3:13]
    assert np.all(np.abs(acc - ref) <= 1e-9 * np.abs(ref) + 1e-11 * np.abs(ref).max()), (acc, ref)
    s.close()


@pytest.mark.parametrize("name", CASES)
def test_next_solution_on_a_graded_tree(binary, name):
    g = golden(name)
    cfg, over = cfg_of(binary, g)
    s = make(binary, cfg, g)
    nsteps = int(over["nsteps"])
    dts = []
    for _ in range(nsteps):
        assert s.next(1) == 0
        dts.append(s.last_dt)
    sc = g["scalars"]
    assert np.allclose(dts, sc[42:42 + nsteps], rtol=1e-13, atol=0)          # min over blocks of spacing / wavespeed
    ok, rel = close(s.solution(), g["u_final"])
    assert ok, rel
    assert abs(s.state().time - sc[0]) <= 1e-13 * sc[0] and s.state().iteration == int(sc[1])
    s.close()


@pytest.mark.parametrize("name", ["binary_tree_d3_b8", "binary_tree_d4_b8_default_focus", "binary_tree_d3_b8_q"])
@pytest.mark.parametrize("arith", ["strict", "fast"])
def test_time_step_bound_folded_into_the_final_stage_equals_the_stand_alone_one(binary, name, arith):
    """Within one call of several steps the next step's bound (min over blocks of spacing / largest wavespeed) comes out of the final
    stage's own kernels, evaluated with the next step's bodies; step by step it comes from the two stand-alone kernels. max and min are
    order-independent, so time steps and solutions must agree bit for bit."""
    g = golden(name)
    cfg, _ = cfg_of(binary, g)
    a, b = make(binary, cfg, g, arith=arith), make(binary, cfg, g, arith=arith)
    n = 7
    assert a.next(n) == 0
    dts = []
    for _ in range(n):
        assert b.next(1) == 0
        dts.append(b.last_dt)
    assert a.last_dt == dts[-1] and len(set(dts)) > 1                 # the bound moves from step to step, and the last one agrees
    assert a.state().time == b.state().time and a.state().iteration == b.state().iteration == n
    assert np.array_equal(a.solution(), b.solution())
    a.close()
    b.close()


@pytest.mark.parametrize("arith", ["strict", "fast"])
def test_default_tree_through_200_steps_against_the_reference_composition(binary, arith):
    """The sub-program's default mesh (64 blocks of 24^2 on levels 2-4) and options, 200 adaptive steps (0.29 orbits) in ONE call - so with
    the time-step bound of every step but the first coming out of the final stage's own kernels - against binary_tree_ref's state
    (23 CPU-seconds of the reference's tree machinery and leaf physics). Measured: 8e-15 of the field scale, the time to 2.5e-16."""
    g = golden("binary_tree_default_200steps")
    s = binary.BinaryTreeSolver(binary.config(), blocks=g["blocks"], edges=g["xv"], u_init=g["u_init"], buffer_rate=g["br"],
                                recommended_time_step=g["stage_scalars"][1], arith=arith)
    assert s.next(200) == 0
    ok, rel = close(s.solution(), g["u_final"], rel=1e-12)
    assert ok, rel
    sc = g["scalars"]
    assert abs(s.state().time - sc[0]) <= 1e-14 * sc[0] and s.state().iteration == int(sc[1]) == 200
    assert abs(s.last_dt - sc[42 + 199]) <= 1e-13 * sc[42 + 199]
    s.close()


def test_uniform_tree_through_both_kernel_families_is_bit_identical(binary):
    """A uniform-depth tree can run through the block kernels (binary_tree.hip) or as one periodic grid (binary.hip): same
    policy arithmetic, so in STRICT mode the two must agree to the last bit. In FAST mode (round 3) the grid kernel gathers the common
    factors of the scheme's glue into FMAs of its own (binary_kernel.hpp), the block kernels keep the generic form: the two agree
    within the mode's own tolerance, 1e-12 of the field scale."""
    for tname, uname in (("binary_tree_d2_b16_uniform", "binary_d2_b16"), ("binary_tree_d2_b16_q_uniform", "binary_d2_b16_q")):
        _both_families(binary, golden(tname), golden(uname))


def _both_families(binary, g, gu):
    cfg, _ = cfg_of(binary, g)
    for arith in ("strict", "fast"):
        t = make(binary, cfg, g, arith=arith)
        u = binary.BinarySolver(cfg, xv=gu["xv"], yv=gu["yv"], u_init=gu["u_init"], buffer_rate=gu["br"], recommended_time_step=gu["stage_scalars"][1], arith=arith)
        t.next(3)
        u.next(3)
        grid = u.solution()
        blocks = t.solution()
        scale = np.abs(grid).reshape(-1, 3).max(axis=0)
        for k, (l, i, j) in enumerate(g["blocks"]):
            mine = grid[i * 16:(i + 1) * 16, j * 16:(j + 1) * 16]
            if arith == "strict":
                assert np.array_equal(blocks[k], mine), (arith, k)
            else:
                assert np.all(np.abs(blocks[k] - mine).reshape(-1, 3).max(axis=0) <= 1e-12 * scale), (arith, k)
        if arith == "strict":
            assert t.last_dt == u.last_dt and t.state().time == u.state().time
        else:          # the time steps follow the states' largest wavespeeds
            assert abs(t.last_dt - u.last_dt) <= 1e-11 * u.last_dt and abs(t.state().time - u.state().time) <= 1e-11 * u.state().time
        t.close(); u.close()


def test_default_configuration_runs_and_stays_symmetric(binary):
    """The sub-program's default mesh (depth=4 block_size=24 focus_factor=2): 64 blocks over three levels. Equal-mass circular
    binary => point symmetry (x, y) -> (-x, -y), which maps block (l, i, j) to (l, 2^l - 1 - i, 2^l - 1 - j) reversed."""
    cfg = binary.config()
    s = binary.BinaryTreeSolver(cfg)
    assert len(s.blocks) == 64
    assert s.next(5) == 0
    u = s.solution()
    assert np.isfinite(u).all() and (u[..., 0] > 0).all()
    index = {tuple(b): k for k, b in enumerate(s.blocks)}
    scale = np.abs(u).reshape(-1, 3).max(axis=0)
    for k, (l, i, j) in enumerate(s.blocks):
        m = index[(l, (1 << l) - 1 - i, (1 << l) - 1 - j)]
        r = u[m][::-1, ::-1]
        assert np.abs(u[k][..., 0] - r[..., 0]).max() <= 1e-11 * scale[0]
        assert np.abs(u[k][..., 1] + r[..., 1]).max() <= 1e-11 * scale[1]
    s.close()
