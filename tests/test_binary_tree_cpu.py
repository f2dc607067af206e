"""Host side of `binary` on graded block trees, no GPU work: the tree builder (create_vertex_quadtree + ensure_valid_quadtree
restated), block vertices and solver data, bit-exact against what the REAL tree machinery of the reference produced
(oracle/ref_drivers/binary_tree_ref.cpp -> tests/golden/binary_tree_*.npz)."""
import json
import numpy as np
import pytest
from conftest import golden, bits_equal

CASES = ["binary_tree_d3_b8", "binary_tree_d4_b8_default_focus", "binary_tree_d3_b12_nu", "binary_tree_d2_b16_uniform", "binary_tree_d3_b8_q"]


def cfg_of(g):
    from mara3_amd import binary
    over = json.loads(str(g["config"]))
    return binary.config(**{k: v for k, v in over.items() if k not in ("nsteps", "safe_mode")}), over


@pytest.mark.parametrize("name", CASES)
def test_tree_structure_vertices_and_solver_data(name):
    from mara3_amd import binary
    g = golden(name)
    cfg, _ = cfg_of(g)
    blocks = binary.tree_blocks(cfg)
    assert np.array_equal(blocks, g["blocks"])                 # same leaves, same traversal order
    edges = binary.tree_vertices(cfg, blocks)
    assert bits_equal(edges, g["xv"])
    u_init, br, dt = binary.tree_solver_data(cfg, blocks, edges)
    assert bits_equal(u_init, g["u_init"]) and bits_equal(br, g["br"])
    assert dt == g["stage_scalars"][1]


def test_default_tree_is_two_to_one_balanced():
    from mara3_amd import binary
    blocks = binary.tree_blocks(binary.config())               # depth=4 block_size=24 focus_factor=2 focus_index=2
    assert len(blocks) == 64 and list(np.bincount(blocks[:, 0])) == [0, 0, 4, 44, 16]
    leaves = {tuple(b) for b in blocks}
    for (l, i, j) in leaves:                                    # no neighbour more than one level away
        n = 1 << l
        for (di, dj) in ((1, 0), (-1, 0), (0, 1), (0, -1)):
            ni, nj = (i + di) % n, (j + dj) % n
            same = (l, ni, nj) in leaves
            coarser = (l - 1, ni // 2, nj // 2) in leaves
            finer = all((l + 1, 2 * ni + a, 2 * nj + b) in leaves for a in (0, 1) for b in (0, 1))
            assert same or coarser or finer, (l, i, j, di, dj)
    assert not binary.tree_is_uniform(blocks)
    assert binary.tree_is_uniform(binary.tree_blocks(binary.config(focus_factor=1e9)))
