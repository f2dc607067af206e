/*
 * mara_oracle_iso2d.c — TEST INFRASTRUCTURE (see mara_oracle.h).
 * Plain-C restatement of mara::iso2d (src/physics_iso2d.hpp): primitive <-> conserved
 * (linear- and angular-momentum forms), flux, wavespeeds, HLLE and HLLC.
 * Parity pinned by tests/golden/iso2d_functions.npz (reference headers via
 * oracle/ref_drivers/funcs_iso2d_ref.cpp), including the reference's own
 * known-answer test src/physics_test.cpp:143-153 (HLLC contact speed == 0).
 * Component order everywhere: (Sigma, px|vx, py|vy) - the LOGICAL order.
 */
#include "mara_oracle.h"
#include <math.h>

static inline double std_max(double a, double b) { return (a < b) ? b : a; }
static inline double std_min(double a, double b) { return (b < a) ? b : a; }
static const double NHAT[3][3] = {{1.0, 0.0, 0.0}, {0.0, 1.0, 0.0}, {0.0, 0.0, 1.0}};

/* physics_iso2d.hpp:249-258 */
void mo_iso2d_to_conserved(const double P[3], double U[3])
{
    U[0] = P[0];
    U[1] = P[0] * P[1];
    U[2] = P[0] * P[2];
}

/* physics_iso2d.hpp:351-362 ; returns 1 where the reference throws (sigma < 0) */
int mo_iso2d_recover_primitive(const double U[3], double P[3])
{
    P[0] = U[0];
    P[1] = U[1] / U[0];
    P[2] = U[2] / U[0];
    return U[0] < 0.0;
}

/* physics_iso2d.hpp:263-272 */
void mo_iso2d_to_conserved_angmom(const double P[3], const double x[2], double Q[3])
{
    Q[0] = P[0];
    Q[1] = P[0] * (x[0] * P[1] + x[1] * P[2]);
    Q[2] = P[0] * (x[0] * P[2] - x[1] * P[1]);
}

/* physics_iso2d.hpp:376-390 */
int mo_iso2d_recover_primitive_angmom(const double Q[3], const double x[2], double P[3])
{
    double sigma = Q[0];
    double sr = Q[1] / sigma;
    double lz = Q[2] / sigma;
    double r2 = x[0] * x[0] + x[1] * x[1];
    P[0] = sigma;
    P[1] = (sr * x[0] - lz * x[1]) / r2;
    P[2] = (sr * x[1] + lz * x[0]) / r2;
    return sigma < 0.0;
}

static inline double velocity_along(const double P[3], int axis)
{
    const double* n = NHAT[axis];
    return P[1] * n[0] + P[2] * n[1] + 0.0 * n[2];
}

/* physics_iso2d.hpp:299-307 */
void mo_iso2d_flux(const double P[3], int axis, double cs2, double F[3])
{
    const double* n = NHAT[axis];
    double v = velocity_along(P, axis);
    double p = P[0] * cs2;
    F[0] = v * P[0];
    F[1] = v * P[0] * P[1] + p * n[0];
    F[2] = v * P[0] * P[2] + p * n[1];
}

/* physics_iso2d.hpp:320-328, max_wavespeed :330-337 ; lam = (minus, plus, max over both axes) */
void mo_iso2d_wavespeeds(const double P[3], int axis, double cs2, double lam[3])
{
    double cs = sqrt(cs2);
    double vn = velocity_along(P, axis);
    lam[0] = vn - cs;
    lam[1] = vn + cs;
    double vx = velocity_along(P, 0), vy = velocity_along(P, 1);
    double ax = std_max(fabs(vx - cs), fabs(vx + cs));
    double ay = std_max(fabs(vy - cs), fabs(vy + cs));
    lam[2] = std_max(ax, ay);
}

/* physics_iso2d.hpp:488-506 */
void mo_iso2d_riemann_hlle(const double Pl[3], const double Pr[3], double cs2l, double cs2r, int axis, double F[3])
{
    double Ul[3], Ur[3], Al[3], Ar[3], Fl[3], Fr[3];
    mo_iso2d_to_conserved(Pl, Ul);
    mo_iso2d_to_conserved(Pr, Ur);
    mo_iso2d_wavespeeds(Pl, axis, cs2l, Al);
    mo_iso2d_wavespeeds(Pr, axis, cs2r, Ar);
    mo_iso2d_flux(Pl, axis, cs2l, Fl);
    mo_iso2d_flux(Pr, axis, cs2r, Fr);
    double ap = std_max(0.0, std_max(Al[1], Ar[1]));
    double am = std_min(0.0, std_min(Al[0], Ar[0]));
    for (int q = 0; q < 3; ++q)
        F[q] = (Fl[q] * ap - Fr[q] * am - (Ul[q] - Ur[q]) * ap * am) / (ap - am);
}

/* physics_iso2d.hpp:610-687 (variables), :556-583 (star states, flux selection), :704-712.
 * Returns 1 where the reference throws (NaN wave speeds); *contact = s_star. */
int mo_iso2d_riemann_hllc(const double Pl[3], const double Pr[3], double cs2l, double cs2r, int axis, double F[3], double* contact)
{
    const double* n = NHAT[axis];
    double ul = velocity_along(Pl, axis), ur = velocity_along(Pr, axis);
    double vperp_l[2] = {Pl[1] - n[0] * ul, Pl[2] - n[1] * ul};
    double vperp_r[2] = {Pr[1] - n[0] * ur, Pr[2] - n[1] * ur};
    double sigma_l = Pl[0], sigma_r = Pr[0];
    double sigma_bar = 0.5 * (sigma_l + sigma_r);
    double al = sqrt(cs2l), ar = sqrt(cs2r);
    double a_bar = 0.5 * (al + ar);
    double press_l = sigma_l * cs2l, press_r = sigma_r * cs2r;
    double ppvrs = 0.5 * (press_l + press_r) - 0.5 * (ur - ul) * sigma_bar * a_bar;
    double pstar = std_max(0.0, ppvrs);
    double ql = std_max(1.0, sqrt(pstar / press_l));
    double qr = std_max(1.0, sqrt(pstar / press_r));
    double sl = ul - al * ql;
    double sr = ur + ar * qr;
    double den = sigma_l * (sl - ul) - sigma_r * (sr - ur);
    double sstar = (press_r - press_l + ul * sigma_l * (sl - ul) - ur * sigma_r * (sr - ur)) / den;
    if (contact) *contact = sstar;

    double Ul[3], Ur[3], Fl[3], Fr[3];
    mo_iso2d_to_conserved(Pl, Ul);
    mo_iso2d_to_conserved(Pr, Ur);
    mo_iso2d_flux(Pl, axis, al * al, Fl);        /* Fl() = Pl.flux(nhat, al * al) (:554) */
    mo_iso2d_flux(Pr, axis, ar * ar, Fr);

    if (0.0 <= sl)
    {
        for (int q = 0; q < 3; ++q) F[q] = Fl[q];
    }
    else if (sl <= 0.0 && 0.0 <= sstar)
    {
        double Us[3];
        Us[0] = sigma_l * (sl - ul) / (sl - sstar);
        Us[1] = sigma_l * (sl - ul) / (sl - sstar) * (sstar * n[0] + vperp_l[0]);
        Us[2] = sigma_l * (sl - ul) / (sl - sstar) * (sstar * n[1] + vperp_l[1]);
        for (int q = 0; q < 3; ++q) F[q] = Fl[q] + (Us[q] - Ul[q]) * sl;
    }
    else if (sstar <= 0.0 && 0.0 <= sr)
    {
        double Us[3];
        Us[0] = sigma_r * (sr - ur) / (sr - sstar);
        Us[1] = sigma_r * (sr - ur) / (sr - sstar) * (sstar * n[0] + vperp_r[0]);
        Us[2] = sigma_r * (sr - ur) / (sr - sstar) * (sstar * n[1] + vperp_r[1]);
        for (int q = 0; q < 3; ++q) F[q] = Fr[q] + (Us[q] - Ur[q]) * sr;
    }
    else if (sr <= 0.0)
    {
        for (int q = 0; q < 3; ++q) F[q] = Fr[q];
    }
    else
    {
        for (int q = 0; q < 3; ++q) F[q] = NAN;
        return 1;
    }
    return 0;
}

void mo_iso2d_to_conserved_n(size_t n, const double* P, double* U)
{
    for (size_t i = 0; i < n; ++i) mo_iso2d_to_conserved(P + 3 * i, U + 3 * i);
}
void mo_iso2d_recover_primitive_n(size_t n, const double* U, double* P, int* threw)
{
    for (size_t i = 0; i < n; ++i) threw[i] = mo_iso2d_recover_primitive(U + 3 * i, P + 3 * i);
}
void mo_iso2d_to_conserved_angmom_n(size_t n, const double* P, const double* x, double* Q)
{
    for (size_t i = 0; i < n; ++i) mo_iso2d_to_conserved_angmom(P + 3 * i, x + 2 * i, Q + 3 * i);
}
void mo_iso2d_recover_primitive_angmom_n(size_t n, const double* Q, const double* x, double* P, int* threw)
{
    for (size_t i = 0; i < n; ++i) threw[i] = mo_iso2d_recover_primitive_angmom(Q + 3 * i, x + 2 * i, P + 3 * i);
}
void mo_iso2d_flux_n(size_t n, const double* P, const double* cs2, int axis, double* F)
{
    for (size_t i = 0; i < n; ++i) mo_iso2d_flux(P + 3 * i, axis, cs2[i], F + 3 * i);
}
void mo_iso2d_wavespeeds_n(size_t n, const double* P, const double* cs2, int axis, double* lam)
{
    for (size_t i = 0; i < n; ++i) mo_iso2d_wavespeeds(P + 3 * i, axis, cs2[i], lam + 3 * i);
}
void mo_iso2d_riemann_n(size_t n, const double* Pl, const double* Pr, const double* cs2l, const double* cs2r, int axis,
                        int solver, double* F, double* contact, int* threw)
{
    for (size_t i = 0; i < n; ++i)
    {
        if (solver == MO_RIEMANN_HLLC) threw[i] = mo_iso2d_riemann_hllc(Pl + 3 * i, Pr + 3 * i, cs2l[i], cs2r[i], axis, F + 3 * i, contact + i);
        else { mo_iso2d_riemann_hlle(Pl + 3 * i, Pr + 3 * i, cs2l[i], cs2r[i], axis, F + 3 * i); threw[i] = 0; contact[i] = 0.0; }
    }
}
