"""An anchor for the Euler HLLC solver (it has no upstream counterpart, so it cannot be pinned): in the limit
gamma -> 1 its wave-speed estimates (pressure-based, q_K = sqrt(1 + (gamma + 1) / (2 gamma) (p* / p_K - 1)) -> sqrt(p* / p_K)), its
contact speed and the mass and momentum components of its flux must turn into those of the reference's ISOTHERMAL HLLC solver
(src/physics_iso2d.hpp:556-583, 610-687), which the oracle restates bit for bit against reference-made vectors
(tests/test_oracle_iso2d_golden.py). Same formulas, same region selection - the energy equation is the only thing the limit drops."""
import numpy as np
import pytest


@pytest.fixture(scope="module")
def oracle():
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
    import mara_oracle
    return mara_oracle


def states(n, seed):
    rng = np.random.default_rng(seed)
    rho = rng.uniform(0.2, 3.0, (2, n))
    v = rng.uniform(-1.5, 1.5, (2, n, 2))
    cs2 = rng.uniform(0.3, 1.2, (2, n))
    return rho, v, cs2


@pytest.mark.parametrize("axis", [0, 1])
def test_gamma_to_one_gives_the_reference_isothermal_hllc(oracle, axis):
    n, eps = 4000, 1e-9
    gamma = 1.0 + eps
    rho, v, cs2 = states(n, 7 + axis)
    P3 = [np.stack([rho[k], v[k, :, 0], v[k, :, 1]], axis=1) for k in range(2)]
    # Euler states with the same density, velocity and sound speed: a^2 = gamma p / rho = cs2
    P5 = [np.stack([rho[k], v[k, :, 0], v[k, :, 1], np.zeros(n), rho[k] * cs2[k] / gamma], axis=1) for k in range(2)]
    Fi, contact, threw = oracle.iso2d_riemann(P3[0], P3[1], cs2[0], cs2[1], axis, oracle.RIEMANN_HLLC)
    Fe = oracle.euler_riemann(P5[0], P5[1], axis, gamma, oracle.RIEMANN_HLLC)
    ok = threw == 0
    assert ok.sum() > 0.99 * n
    scale = np.abs(Fi[ok]).max(axis=0)
    err = np.abs(Fe[ok][:, :3] - Fi[ok]).max(axis=0) / scale
    assert np.all(err <= 1e-6), err                      # O(eps) apart; a different estimate or region choice would show at O(1)
    assert np.all(Fe[ok][:, 3] == 0.0)                   # no flux of the momentum component that is not there
    # every region of the solver is exercised: left / right of the contact, star states and the supersonic branches
    sl = np.minimum(v[0, :, axis] - np.sqrt(cs2[0]), v[1, :, axis] - np.sqrt(cs2[1]))
    sr = np.maximum(v[0, :, axis] + np.sqrt(cs2[0]), v[1, :, axis] + np.sqrt(cs2[1]))
    assert (contact[ok] > 0).any() and (contact[ok] < 0).any() and (sl[ok] > 0).any() and (sr[ok] < 0).any()
