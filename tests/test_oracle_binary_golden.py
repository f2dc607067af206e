"""The C restatement of the circumbinary-disk scheme (oracle/mara_oracle_binary.c) against the vectors emitted by the
reference-composed driver oracle/ref_drivers/binary_ref.cpp (tests/golden/binary_*.npz, generator oracle/gen_golden.py binary).
Bit-exact: both sides run the same libm on the same host ISA."""
import json
import numpy as np
import pytest
from conftest import golden, bits_equal

CASES = ["binary_d2_b16", "binary_d1_b24_nu", "binary_d3_b8_axisym", "binary_d2_b16_safe", "binary_d2_b32", "binary_d2_b16_live", "binary_d2_b16_q", "binary_d1_b24_q_nu"]


def load(name, oracle):
    g = golden(name)
    over = json.loads(str(g["config"]))
    cfg = oracle.binary_config(**{k: v for k, v in over.items() if k not in ("nsteps", "safe_mode")})
    return g, cfg, over


@pytest.mark.parametrize("name", CASES)
def test_vertices_and_solver_data(name, oracle):
    g, cfg, _ = load(name, oracle)
    v = oracle.binary_vertices(cfg)
    assert bits_equal(v, g["xv"]) and bits_equal(v, g["yv"])
    u_init, br, dt = oracle.binary_solver_data(cfg, g["xv"], g["yv"])
    assert bits_equal(u_init, g["u_init"])
    assert bits_equal(br, g["br"])
    assert dt == g["stage_scalars"][1]


@pytest.mark.parametrize("name", CASES)
def test_one_stage(name, oracle):
    g, cfg, over = load(name, oracle)
    ss = g["stage_scalars"]
    dt, bodies, totals = ss[0], ss[61:71], ss[43:61]
    assert oracle.binary_maximum_timestep(cfg, g["xv"], g["yv"], g["u_init"], bodies) == ss[2]
    u1, tot, neg = oracle.binary_advance_u(cfg, g["xv"], g["yv"], g["u_init"], g["u_init"], g["br"], bodies, dt, safe_mode=bool(over.get("safe_mode", 0)))
    assert not neg
    assert bits_equal(u1, g["u_stage"])
    assert bits_equal(tot, totals)
    # the accumulators after one stage from zero are the totals themselves (scheme.cpp:893-898)
    acc = ss[3:13]
    assert bits_equal(acc, tot[[0, 1, 2, 3, 4, 5, 14, 15, 16, 17]] + 0.0)


@pytest.mark.parametrize("name", ["binary_tree_d3_b8", "binary_tree_d4_b8_default_focus", "binary_tree_d3_b12_nu", "binary_tree_d3_b8_q",
                                  "binary_tree_d2_b16_q_uniform", "binary_tree_d2_b16_uniform"])
def test_diagnostics_vs_reference_composition(name, oracle):
    """disk_mass, disk_angular_momentum and the diagnostic fields (sigma, v_r, v_phi) of the final state, bit for bit against the
    reference's own array / tree operators composed as subprog_binary_diagnostics.cpp composes them (sum order included)."""
    import json
    g = golden(name)
    q = int(json.loads(str(g["config"])).get("conserve_linear_p", 1)) == 0
    mass, lz, fields = oracle.binary_diagnostics(q, g["blocks"], g["xv"], g["u_final"])
    assert mass == g["diag_scalars"][0] and lz == g["diag_scalars"][1]
    assert bits_equal(fields, g["diag_fields"])


def test_binary_port_threads_do_not_change_results():
    """mo_binary_set_threads: OpenMP over rows and blocks (the role of the reference's tree.map(fn, pool), core_tree.hpp:615-625) - field, totals
    and the negative-density flag are those of the one-thread run, bit for bit (bench_configs.py times the port on all host cores)"""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
    import mara_oracle as mo
    cfg = mo.binary_config(depth=2, block_size=16, fixed_dt=1)
    xv = mo.binary_vertices(cfg)
    u0, br, dt = mo.binary_solver_data(cfg, xv, xv)
    bodies = np.array([0.5, 0.5, 0.0, 0.0, 0.5, 0.5, -0.5, 0.0, 0.0, -0.5])
    one = mo.binary_advance_u(cfg, xv, xv, u0, u0, br, bodies, dt, nthreads=1)
    for th in (2, 5, 8):
        many = mo.binary_advance_u(cfg, xv, xv, u0, u0, br, bodies, dt, nthreads=th)
        assert np.array_equal(one[0].view(np.uint64), many[0].view(np.uint64)) and np.array_equal(one[1].view(np.uint64), many[1].view(np.uint64)) and one[2] == many[2]
