"""The exact Riemann solver the GPU tests of the Euler HLLC path compare against (tests/exact_riemann.py) against Toro's tabulated
star states, and basic consistency of its sampling - so that the independent check is itself checked."""
import numpy as np
from exact_riemann import TORO, TORO_STAR, SOD, SOD_STAR, star_state, sample, cell_averages


def test_star_states_match_toro_tables():
    p, u = star_state(*SOD, 1.4)
    assert abs(p - SOD_STAR[0]) < 1e-5 and abs(u - SOD_STAR[1]) < 1e-5
    for k, (ps, us) in TORO_STAR.items():
        p, u = star_state(*TORO[k][:6], 1.4)
        assert abs(p - ps) <= 2e-5 * max(1.0, ps) and abs(u - us) <= 2e-5 * max(1.0, abs(us)), (k, p, u)


def test_sampling_is_consistent():
    for k, (rl, ul, pl, rr, ur, pr, x0, t) in TORO.items():
        rho, u, p = sample(rl, ul, pl, rr, ur, pr, 1.4, np.array([-1e3, 1e3]))
        assert (rho[0], u[0], p[0]) == (rl, ul, pl) and (rho[1], u[1], p[1]) == (rr, ur, pr)
        ps, us = star_state(rl, ul, pl, rr, ur, pr, 1.4)
        xi = np.array([us - 1e-9, us + 1e-9])
        rho, u, p = sample(rl, ul, pl, rr, ur, pr, 1.4, xi)
        assert np.allclose(p, ps) and np.allclose(u, us)          # pressure and velocity are continuous across the contact
        # Rankine-Hugoniot across a left shock: mass flux continuous in the shock frame
        if ps > pl:
            al = np.sqrt(1.4 * pl / rl)
            s = ul - al * np.sqrt(2.4 / 2.8 * ps / pl + 0.4 / 2.8)
            r2, u2, _ = sample(rl, ul, pl, rr, ur, pr, 1.4, np.array([s + 1e-9]))
            assert abs(rl * (ul - s) - r2[0] * (u2[0] - s)) <= 1e-6 * abs(rl * (ul - s))
        rho, u, p = cell_averages(k, 64)
        assert np.isfinite(rho).all() and (rho > 0).all() and (p > 0).all()
