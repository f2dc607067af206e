"""Run one of Toro's shock-tube tests on the HIP path (test infrastructure shared by tests/test_gpu_toro.py and scripts/toro_report.py)."""
import numpy as np
from exact_riemann import TORO, sample, cell_averages, wave_positions

GAMMA = 1.4


def run_tube(engine, test, n, axis=0, rank=2, riemann="hllc", arith="strict", theta=1.5, rk=2, cfl=0.4):
    """-> (rho, velocity along the tube, pressure) as 1-D profiles of n cells at Toro's output time, from a rank-D run whose tube lies
    along `axis` (the other axes carry 8 identical cells; outflow everywhere)"""
    rl, ul, pl, rr, ur, pr, x0, t_end = TORO[test]
    shape = [8] * rank
    shape[axis] = n
    x = (np.arange(n) + 0.5) / n
    sel = [None] * rank
    sel[axis] = slice(None)
    left = (x < x0)[tuple(sel)] & np.ones(shape, dtype=bool)
    d = np.where(left, rl, rr)
    v = np.where(left, ul, ur)
    p = np.where(left, pl, pr)
    u = np.zeros(tuple(shape) + (5,))
    u[..., 0] = d
    u[..., 1 + axis] = d * v
    u[..., 4] = 0.5 * d * v * v + p / (GAMMA - 1)
    # fixed step from the fastest signal of the exact solution
    rho_e, u_e, p_e = sample(rl, ul, pl, rr, ur, pr, GAMMA, np.linspace(-60, 60, 4001))
    smax = float(np.max(np.abs(u_e) + np.sqrt(GAMMA * p_e / rho_e)))
    dx = 1.0 / n
    nsteps = int(np.ceil(t_end / (cfl * dx / smax)))
    dt = t_end / nsteps
    dl = [1.0 / 8] * rank
    dl[axis] = dx
    s = engine.EulerCartSolver(tuple(shape), tuple(dl), GAMMA, theta, riemann, rk, "outflow", arith=arith)
    s.upload(u)
    s.step(dt, nsteps)
    out = s.download()
    status = s.status_result()
    s.close()
    # the transverse cells must be identical copies
    line = np.moveaxis(out, axis, 0).reshape(n, -1, 5)
    uniform = bool(np.all(line == line[:, :1, :]))
    q = line[:, 0, :]
    rho = q[:, 0]
    vel = q[:, 1 + axis] / rho
    pre = (q[:, 4] - 0.5 * (q[:, 1] ** 2 + q[:, 2] ** 2 + q[:, 3] ** 2) / rho) * (GAMMA - 1)
    return rho, vel, pre, status, uniform


def front_position(x, q, lo, hi, x_exact, halfwidth):
    """where the profile q crosses the mean of its plateau values lo / hi, searched within halfwidth of the exact position"""
    mid = 0.5 * (lo + hi)
    win = np.where(np.abs(x - x_exact) <= halfwidth)[0]
    s = np.sign(q[win] - mid)
    k = np.where(s[:-1] * s[1:] <= 0)[0]
    if len(k) == 0:
        return None
    i = win[k[0]]
    return x[i] + (mid - q[i]) / (q[i + 1] - q[i]) * (x[i + 1] - x[i]) if q[i + 1] != q[i] else x[i]


def metrics(engine, test, n, **kw):
    rho, vel, pre, status, uniform = run_tube(engine, test, n, **kw)
    rho_e, u_e, p_e = cell_averages(test, n, GAMMA)
    x = (np.arange(n) + 0.5) / n
    w = wave_positions(test, GAMMA)
    rl, ul, pl, rr, ur, pr, x0, t_end = TORO[test]
    # star densities on both sides of the contact
    eps = 1e-7
    r_sl = sample(rl, ul, pl, rr, ur, pr, GAMMA, np.array([w["u_star"] - eps]))[0][0]
    r_sr = sample(rl, ul, pl, rr, ur, pr, GAMMA, np.array([w["u_star"] + eps]))[0][0]
    out = {"l1_rho": float(np.mean(np.abs(rho - rho_e))), "l1_u": float(np.mean(np.abs(vel - u_e))), "l1_p": float(np.mean(np.abs(pre - p_e))),
           "scale_rho": float(np.mean(np.abs(rho_e))), "scale_p": float(np.mean(np.abs(p_e))), "status": status, "uniform": uniform,
           "min_rho": float(rho.min()), "min_p": float(pre.min())}
    hw = 12.0 / n
    if abs(r_sl - r_sr) > 1e-3 * max(r_sl, r_sr):
        pos = front_position(x, rho, r_sl, r_sr, w["contact"], hw)
        out["contact_err_cells"] = None if pos is None else float((pos - w["contact"]) * n)
    if w["shock_r"] is not None:
        pos = front_position(x, pre, w["p_star"], pr, w["shock_r"], hw)
        out["shock_r_err_cells"] = None if pos is None else float((pos - w["shock_r"]) * n)
    if w["shock_l"] is not None:
        pos = front_position(x, pre, pl, w["p_star"], w["shock_l"], hw)
        out["shock_l_err_cells"] = None if pos is None else float((pos - w["shock_l"]) * n)
    return out
