"""GPU tests of the `binary` diagnostics: binary::disk_mass, disk_angular_momentum and diagnostic_fields
of a device-resident solution, against vectors produced by the reference's own array / tree operators composed as
src/subprog_binary_diagnostics.cpp composes them (oracle/ref_drivers/binary_tree_ref.cpp -> diag_scalars, diag_fields).

Tolerances, and why they are not zero:
  * sigma: bit-exact.
  * v_r, v_phi: 4 ulp of |v| per cell - the device takes sqrt(r2) where the reference calls std::pow(r2, 0.5) (glibc's pow is
    not correctly rounded), so rhat can differ in its last bit.
  * disk_mass, disk_angular_momentum: 1e-13 relative - a tree reduction on the device against the reference's sequential sums."""
import json
import numpy as np
import pytest
from conftest import golden, bits_equal

pytestmark = pytest.mark.gpu
CASES = ["binary_tree_d3_b8", "binary_tree_d4_b8_default_focus", "binary_tree_d3_b12_nu", "binary_tree_d3_b8_q",
         "binary_tree_d2_b16_uniform", "binary_tree_d2_b16_q_uniform"]


@pytest.fixture(scope="module")
def binary():
    import mara3_amd
    from mara3_amd import binary
    assert mara3_amd.load_library().mh_device_count() >= 1
    return binary


def cfg_of(binary, g):
    over = json.loads(str(g["config"]))
    return binary.config(**{k: v for k, v in over.items() if k not in ("nsteps", "safe_mode")})


def check_fields(got, ref):
    sigma, vr, vp = got
    assert bits_equal(sigma, ref[:, 0] if ref.ndim == 4 else ref[0])
    rr, rp = (ref[:, 1], ref[:, 2]) if ref.ndim == 4 else (ref[1], ref[2])
    vmag = np.sqrt(rr * rr + rp * rp)
    tol = 4 * np.spacing(vmag)
    assert np.all(np.abs(vr - rr) <= tol) and np.all(np.abs(vp - rp) <= tol), (np.abs(vr - rr).max(), np.abs(vp - rp).max())


@pytest.mark.parametrize("arith", ["strict", "fast"])
@pytest.mark.parametrize("name", CASES)
def test_tree_solver_diagnostics(binary, name, arith):
    g = golden(name)
    cfg = cfg_of(binary, g)
    s = binary.BinaryTreeSolver(cfg, blocks=g["blocks"], edges=g["xv"], u_init=g["u_init"], buffer_rate=g["br"],
                                recommended_time_step=g["stage_scalars"][1], arith=arith)
    s.set_solution(g["u_final"], s.state())
    mass, lz = s.disk_totals()
    assert abs(mass - g["diag_scalars"][0]) <= 1e-13 * abs(g["diag_scalars"][0])
    assert abs(lz - g["diag_scalars"][1]) <= 1e-13 * abs(g["diag_scalars"][1])
    check_fields(s.diagnostic_fields(), g["diag_fields"])
    assert bits_equal(s.solution(), g["u_final"])            # the diagnostics use the stage buffers as scratch, never the solution
    s.close()


@pytest.mark.parametrize("tname,uname", [("binary_tree_d2_b16_uniform", "binary_d2_b16"), ("binary_tree_d2_b16_q_uniform", "binary_d2_b16_q")])
def test_grid_solver_diagnostics(binary, tname, uname):
    """The same quantities from the periodic-grid layout of a uniform-depth tree; the reference's blocks are laid into the grid."""
    gt, gu = golden(tname), golden(uname)
    cfg = cfg_of(binary, gu)
    bs = int(cfg["block_size"])
    n = gu["u_final"].shape[0]
    ref = np.empty((3, n, n))
    for k, (_, i, j) in enumerate(gt["blocks"]):
        ref[:, i * bs:(i + 1) * bs, j * bs:(j + 1) * bs] = gt["diag_fields"][k]
    s = binary.BinarySolver(cfg, xv=gu["xv"], yv=gu["yv"], u_init=gu["u_init"], buffer_rate=gu["br"], recommended_time_step=gu["stage_scalars"][1])
    s.set_solution(gu["u_final"], s.state())
    mass, lz = s.disk_totals()
    assert abs(mass - gt["diag_scalars"][0]) <= 1e-13 * abs(gt["diag_scalars"][0])
    assert abs(lz - gt["diag_scalars"][1]) <= 1e-13 * abs(gt["diag_scalars"][1])
    check_fields(s.diagnostic_fields(), ref)
    s.close()


def test_diagnostics_after_stepping_match_the_oracle(binary, oracle):
    """Step on the device, then compare the device diagnostics with the CPU oracle evaluated on the downloaded solution."""
    mo = oracle
    g = golden("binary_tree_d4_b8_default_focus")
    cfg = cfg_of(binary, g)
    s = binary.BinaryTreeSolver(cfg, blocks=g["blocks"], edges=g["xv"], u_init=g["u_init"], buffer_rate=g["br"], recommended_time_step=g["stage_scalars"][1])
    s.next(5)
    mass, lz = s.disk_totals()
    m0, l0, f0 = mo.binary_diagnostics(False, g["blocks"], g["xv"], s.solution())
    assert abs(mass - m0) <= 1e-13 * abs(m0) and abs(lz - l0) <= 1e-13 * abs(l0)
    check_fields(s.diagnostic_fields(), f0)
    s.close()
