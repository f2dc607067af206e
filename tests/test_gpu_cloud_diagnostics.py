"""GPU tests of the `cloud` diagnostics: CloudProblem::make_diagnostic_fields of a device-resident state
through the C ABI, against vectors produced by the reference's own operators (oracle/ref_drivers/cloud_ref.cpp -> diag_fields,
diag_columns in tests/golden/clouddiag_*.npz).

Tolerances:
  * mass_density, gas_pressure, radial_gamma_beta, radial_energy_flow, solid angle, total energy, Lorentz factor, flow powers:
    bit-exact (STRICT primitive recovery and flux, IEEE division and sqrt, sequential sums).
  * specific_entropy: 4 ulp of max(|log p|, gamma |log rho|) - log and pow are the device library's, not glibc's.
  * shock indices (radii, and the powers / gamma sampled at them): equal in every golden case; the entropy's last-place difference
    could move an index only in an exact tie."""
import numpy as np
import pytest
from conftest import golden, bits_equal

pytestmark = pytest.mark.gpu
CASES = ["clouddiag_nr48_150steps", "clouddiag_nr40_pcm_60steps", "clouddiag_nr32_4steps"]
G = 4.0 / 3


@pytest.fixture(scope="module")
def eng():
    import mara3_amd
    from mara3_amd import engine
    assert mara3_amd.load_library().mh_device_count() >= 1
    return engine


def check(fields, columns, ref_fields, ref_columns):
    for k in (0, 1, 3, 4):
        assert bits_equal(fields[k], ref_fields[k]), k
    scale = np.maximum(np.abs(ref_fields[2]), 1.0)
    assert np.all(np.abs(fields[2] - ref_fields[2]) <= 4 * np.spacing(scale) * 8), np.abs(fields[2] - ref_fields[2]).max()
    assert bits_equal(columns, ref_columns), [k for k in range(15) if not bits_equal(columns[k], ref_columns[k])]


@pytest.mark.parametrize("case", CASES)
def test_cloud_diagnostics_vs_reference_composition(eng, case):
    g = golden(case)
    s = eng.CloudSolver(g["rv"], g["qv"], 1, 1.2, float(g["tfloor"]))
    s.upload(g["un"])
    fields, columns = s.diagnostics(g["diag_meta"][1:4])
    assert s.status() == 0
    check(fields, columns, g["diag_fields"], g["diag_columns"])
    assert bits_equal(s.download(), g["un"])            # the stage scratch is used, never the solution


def test_cloud_diagnostics_after_stepping_match_the_oracle(eng, oracle):
    g = golden("clouddiag_nr48_150steps")
    s = eng.CloudSolver(g["rv"], g["qv"], 2, 1.2, float(g["tfloor"]))
    s.upload(g["un"])
    s.set_inflow(np.zeros((g["un"].shape[1], 5)) + np.array([1e-3, 0.0, 0.0, 0.0, 1e-9]))
    s.step(float(g["dt"]), 5)
    u = s.download()
    fields, columns = s.diagnostics(g["diag_meta"][1:4])
    f0, c0, st = oracle.cloud_diagnostics(u, g["rv"], g["qv"], g["diag_meta"][1:4], float(g["tfloor"]))
    assert st == 0
    check(fields, columns, f0, c0)


@pytest.mark.parametrize("name,system", [("sedovdiag_newtonian_nr256", "euler"), ("sedovdiag_srhd_nr256", "srhd")])
def test_sedov_diagnostics_vs_reference_composition(eng, name, system):
    """mh_sedov_diagnostics: pressure, density, velocity bit-exact; entropy to a few ulp (device log / pow); the three shock-locator
    indices equal to the reference's."""
    g = golden(name)
    s = eng.SedovSolver(g["vertices"], system=system)
    for ns in (100, 400):
        s.upload(g["u_%d" % ns])
        fields, indices = s.diagnostics()
        ref = g["fields_%d" % ns]
        for k in (1, 2, 3):
            assert bits_equal(fields[k], ref[k]), k
        assert np.all(np.abs(fields[0] - ref[0]) <= 32 * np.spacing(np.maximum(np.abs(ref[0]), 1.0)))
        assert np.array_equal(indices, g["indices_%d" % ns])
        assert bits_equal(s.download(), g["u_%d" % ns])
    assert s.status() == 0
