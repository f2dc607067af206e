"""CPU test of the host-side Kepler two-body model of libmara_hip.so through the C ABI:
bit-exact against vectors produced by the reference's src/model_two_body.hpp (oracle/ref_drivers/two_body_ref.cpp),
plus the analytic properties the reference's own tests check (src/physics_test.cpp:156-305)."""
import ctypes as C
import numpy as np
import pytest
from conftest import golden, bits_equal


class Elements(C.Structure):
    _fields_ = [("separation", C.c_double), ("total_mass", C.c_double), ("mass_ratio", C.c_double), ("eccentricity", C.c_double)]


class FullElements(C.Structure):
    _fields_ = [("pomega", C.c_double), ("tau", C.c_double), ("cm_position_x", C.c_double), ("cm_position_y", C.c_double),
                ("cm_velocity_x", C.c_double), ("cm_velocity_y", C.c_double), ("elements", Elements)]


class TwoBody(C.Structure):
    _fields_ = [("body1", C.c_double * 5), ("body2", C.c_double * 5)]


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__
    __graft_entry__.build()
    from mara3_amd import _lib
    return _lib.load_library()


def make_full(row):
    return FullElements(row[0], row[1], row[2], row[3], row[4], row[5], Elements(row[6], row[7], row[8], row[9]))


def state_of(lib, row, t):
    out = TwoBody()
    assert lib.mh_two_body_state(C.byref(make_full(row)), t, C.byref(out)) == 0
    return np.array(list(out.body1) + list(out.body2))


def test_two_body_state_bit_exact(lib):
    g = golden("two_body")
    got = np.array([state_of(lib, row, row[10]) for row in g["elements_in"]])
    assert bits_equal(got, g["state"])


def test_orbital_elements_bit_exact_and_unbound_orbits_rejected(lib):
    g = golden("two_body")
    threw = g["threw"] != 0
    assert threw.any()
    for row, want, t in zip(g["state_in"], g["elements"], threw):
        s = TwoBody((C.c_double * 5)(*row[0:5]), (C.c_double * 5)(*row[5:10]))
        out = FullElements()
        rc = lib.mh_orbital_elements_from_state(C.byref(s), row[10], C.byref(out))
        assert (rc != 0) == bool(t)
        if not t:
            got = np.array([out.pomega, out.tau, out.cm_position_x, out.cm_position_y, out.cm_velocity_x, out.cm_velocity_y,
                            out.elements.separation, out.elements.total_mass, out.elements.mass_ratio, out.elements.eccentricity])
            assert bits_equal(got, want)


def test_analytic_properties(lib):
    """Default binary (a = M = q = 1, e = 0): bodies at +-1/2 on a circle with speed 1/2; energy -1/8 is conserved in
    time; elements -> state -> elements is the identity to round-off for an eccentric orbit."""
    base = [0, 0, 0, 0, 0, 0, 1.0, 1.0, 1.0, 0.0]
    s = state_of(lib, base, 0.0)
    assert np.allclose(s, [0.5, 0.5, 0, 0, 0.5, 0.5, -0.5, 0, 0, -0.5], atol=1e-15)
    def energy(s):
        return 0.5 * s[0] * (s[3] ** 2 + s[4] ** 2) + 0.5 * s[5] * (s[8] ** 2 + s[9] ** 2) - s[0] * s[5] / np.hypot(s[1] - s[6], s[2] - s[7])
    ecc = [0.3, 0.2, 0.01, -0.02, 0.03, 0.04, 1.7, 1.3, 0.4, 0.6]
    e0 = energy(state_of(lib, ecc, 0.0))
    for t in np.linspace(0.1, 30, 17):
        assert abs(energy(state_of(lib, ecc, t)) - e0) < 1e-9 * abs(e0)
    st = state_of(lib, ecc, 4.2)
    out = FullElements()
    s = TwoBody((C.c_double * 5)(*st[0:5]), (C.c_double * 5)(*st[5:10]))
    assert lib.mh_orbital_elements_from_state(C.byref(s), 4.2, C.byref(out)) == 0
    assert abs(out.elements.separation - 1.7) < 1e-9 and abs(out.elements.eccentricity - 0.6) < 1e-9
    assert abs(out.elements.mass_ratio - 0.4) < 1e-12 and abs(out.pomega - 0.3) < 1e-9
