"""Synthetic initial conditions shared by tests and bench.py. Host-side numpy only."""
import numpy as np


def blast_ic(shape, gamma, radius=0.1, p_in=10.0, p_out=0.1, row_range=None):
    """rho=1, v=0, p = p_in inside r<radius of the centre of [0,1]^rank else p_out (cell-centre sampled).

    Returns conserved densities as host-order AoS [n0][n1](...)[5]. row_range=(a,b) builds only
    rows [a,b) of axis 0 (slab of a larger grid)."""
    a, b = row_range if row_range is not None else (0, shape[0])
    axes = [(np.arange(a, b) + 0.5) / shape[0]] + [(np.arange(n) + 0.5) / n for n in shape[1:]]
    X = np.meshgrid(*axes, indexing="ij", sparse=True)
    r2 = sum((x - 0.5) ** 2 for x in X)
    p = np.where(r2 < radius * radius, p_in, p_out)
    u = np.zeros((b - a,) + tuple(shape[1:]) + (5,))
    u[..., 0] = 1.0
    u[..., 4] = p / (gamma - 1.0)
    return u


def wave_ic(shape, gamma, seed=0):
    """Smooth wave with all velocity components non-zero plus seeded noise (periodic tests)."""
    rng = np.random.default_rng(seed)
    axes = [(np.arange(n) + 0.5) / n for n in shape]
    X = np.meshgrid(*axes, indexing="ij")
    s = np.ones(shape)
    for x in X:
        s = s * np.sin(2 * np.pi * x)
    d = 1.0 + 0.2 * s + 0.01 * rng.standard_normal(shape)
    p = d ** gamma
    v = [0.5 + 0.1 * s, -0.25 + 0.05 * np.cos(2 * np.pi * X[0]), 0.125 * np.ones(shape) + 0.01 * rng.standard_normal(shape)]
    u = np.zeros(tuple(shape) + (5,))
    u[..., 0] = d
    for k in range(3):
        u[..., 1 + k] = d * v[k]
    u[..., 4] = 0.5 * d * (v[0] ** 2 + v[1] ** 2 + v[2] ** 2) + p / (gamma - 1.0)
    return u


def smooth_wave_ic(shape, gamma, row_range=None):
    """SURVEY.md §8d's second initial condition: rho = 1 + 0.2 sin(2 pi x) sin(2 pi y), p = rho^gamma, v = (0.5, 0.25), periodic.
    No quiescent regions: every face sees a genuine Riemann problem. row_range=(a,b) builds only rows [a,b) of axis 0."""
    a, b = row_range if row_range is not None else (0, shape[0])
    x = ((np.arange(a, b) + 0.5) / shape[0])[:, None]
    y = ((np.arange(shape[1]) + 0.5) / shape[1])[None, :]
    d = 1.0 + 0.2 * np.sin(2 * np.pi * x) * np.sin(2 * np.pi * y)
    u = np.zeros((b - a, shape[1], 5))
    u[..., 0] = d
    u[..., 1] = d * 0.5
    u[..., 2] = d * 0.25
    u[..., 4] = 0.5 * d * (0.5 ** 2 + 0.25 ** 2) + d ** gamma / (gamma - 1.0)
    return u


def baseline_dt(n, cfl=0.3, vmax=6.0):
    """Fixed step of the primary config: dt = 0.3*dx/6."""
    return cfl * (1.0 / n) / vmax
