"""Host-side set-up and bookkeeping of the `binary` path through the C ABI (no GPU work): block-tree vertices, initial
model, buffer-rate field and recommended time step, bit-exact against vectors from the reference-composed driver
(oracle/ref_drivers/binary_ref.cpp, which builds the vertices with mara::create_vertex_quadtree itself)."""
import ctypes as C
import json
import numpy as np
import pytest
from conftest import golden, bits_equal

CASES = ["binary_d2_b16", "binary_d1_b24_nu", "binary_d3_b8_axisym", "binary_d2_b32", "binary_d2_b16_live", "binary_d2_b16_q", "binary_d1_b24_q_nu"]


def cfg_of(g):
    from mara3_amd import binary
    over = json.loads(str(g["config"]))
    return binary.config(**{k: v for k, v in over.items() if k not in ("nsteps", "safe_mode")}), over


@pytest.mark.parametrize("name", CASES)
def test_vertices_and_solver_data_bit_exact(name):
    from mara3_amd import binary
    g = golden(name)
    cfg, _ = cfg_of(g)
    v = binary.vertices(cfg)
    assert bits_equal(v, g["xv"]) and bits_equal(v, g["yv"])
    u_init, br, dt = binary.solver_data(cfg)
    assert bits_equal(u_init, g["u_init"])
    assert bits_equal(br, g["br"])
    assert dt == g["stage_scalars"][1]


@pytest.mark.parametrize("name", CASES)
def test_initial_bodies_bit_exact(name):
    from mara3_amd import binary
    g = golden(name)
    cfg, _ = cfg_of(g)
    bodies = binary.two_body_state(binary.initial_elements(cfg), 0.0)
    assert bits_equal(bodies, g["stage_scalars"][61:71])


def test_elements_diff_wraps_angles():
    import mara3_amd
    from mara3_amd import _lib as L
    lib = mara3_amd.load_library()
    a, b, d = L.FullOrbitalElements(), L.FullOrbitalElements(), L.FullOrbitalElements()
    a.pomega, b.pomega = 3.0, -3.0
    a.elements.separation = b.elements.separation = 1.0
    a.elements.total_mass = b.elements.total_mass = 1.0
    a.tau, b.tau = 0.1, 6.2
    lib.mh_orbital_elements_diff(C.byref(a), C.byref(b), C.byref(d))
    assert d.pomega == (-3.0 - 3.0) + 2 * np.pi        # nearest image
    assert d.tau == (6.2 - 0.1) - 2 * np.pi            # period of a unit binary is 2 pi
    assert d.elements.separation == 0.0


def test_bad_descriptors_are_rejected():
    import mara3_amd
    from mara3_amd import binary
    lib = mara3_amd.load_library()
    d = binary.make_desc(binary.config(depth=2, block_size=16))
    assert lib.mh_binary_field_doubles(C.byref(d)) == 3 * 68 * 64
    d.n = 65
    assert lib.mh_binary_scratch_doubles(C.byref(d)) == 0
    d.n = 48        # 48 / 16 = 3 blocks per side: not a uniform-depth tree
    assert lib.mh_binary_scratch_doubles(C.byref(d)) == 0
    with pytest.raises(KeyError):
        binary.config(no_such_item=1)
