/*
 * mara_oracle_srhd.c — TEST INFRASTRUCTURE (see mara_oracle.h).
 * Plain-C restatement of mara::srhd (src/physics_srhd.hpp) and of the `cloud`
 * sub-program's stage (src/subprog_cloud.cpp:260-290 geometry, :466-509
 * boundary conditions, :511-584 advance). Parity pinned by
 * tests/golden/srhd_functions.npz and cloud_*.npz, which are produced by the
 * reference's own headers (oracle/ref_drivers/funcs_srhd_ref.cpp, cloud_ref.cpp).
 */
#include "mara_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

static inline double std_max(double a, double b) { return (a < b) ? b : a; }
static inline double std_min(double a, double b) { return (b < a) ? b : a; }
static const double NHAT[3][3] = {{1.0, 0.0, 0.0}, {0.0, 1.0, 0.0}, {0.0, 0.0, 1.0}};

/* physics_srhd.hpp:104-107 enthalpy_density, :165-168 lorentz_factor */
static inline double enthalpy_density(const double P[5], double gamma) { return P[0] + P[4] * (1.0 + 1.0 / (gamma - 1.0)); }
static inline double gamma_beta_squared(const double P[5]) { return P[1] * P[1] + P[2] * P[2] + P[3] * P[3]; }

/* physics_srhd.hpp:364-451. Returns 0 or a mask of MO_C2P_* failure bits (the reference throws). */
int mo_srhd_recover_primitive(const double U[5], double gm, double temperature_floor, double P[5])
{
    const double D = U[0], tau = U[4];
    const double SS = U[1] * U[1] + U[2] * U[2] + U[3] * U[3];
    int solution_found = 0, iteration = 0;
    double W0 = 1.0, p = 0.0;

    while (iteration < 50)
    {
        double v2 = std_min(SS / pow(tau + D + p, 2), 1.0 - 1e-10);
        double W2 = 1.0 / (1.0 - v2);
        double W = sqrt(W2);
        double e = (tau + D * (1.0 - W) + p * (1.0 - W2)) / (D * W);
        double d = D / W;
        double h = 1.0 + e + p / d;
        double cs2 = gm * p / (d * h);
        double f = d * e * (gm - 1.0) - p;
        double g = v2 * cs2 - 1.0;
        p -= f / g;
        if (fabs(f) < 1e-10)
        {
            W0 = W;
            solution_found = 1;
            break;
        }
        ++iteration;
    }
    if (temperature_floor > 0.0) p = std_max(p, temperature_floor * D / W0);
    P[0] = D / W0;
    P[1] = W0 * U[1] / (tau + D + p);
    P[2] = W0 * U[2] / (tau + D + p);
    P[3] = W0 * U[3] / (tau + D + p);
    P[4] = p;
    int status = 0;
    if (! solution_found) status |= MO_C2P_NOT_CONVERGED;
    if (P[0] <= 0.0) status |= MO_C2P_NEG_DENSITY;
    if (P[4] <= 0.0) status |= MO_C2P_NEG_PRESSURE;
    if (isnan(W0)) status |= MO_C2P_NAN;
    return status;
}

/* physics_srhd.hpp:213-227 */
void mo_srhd_to_conserved_density(const double P[5], double gamma, double U[5])
{
    double W = sqrt(1.0 + gamma_beta_squared(P));
    double h = enthalpy_density(P, gamma) / P[0];
    double D = P[0] * W;
    double p = P[4];
    U[0] = D;
    U[1] = D * P[1] * h;
    U[2] = D * P[2] * h;
    U[3] = D * P[3] * h;
    U[4] = D * h * W - p - D;
}

/* beta_along :181-185 */
static inline double beta_along(const double P[5], int axis)
{
    const double* n = NHAT[axis];
    return (P[1] * n[0] + P[2] * n[1] + P[3] * n[2]) / sqrt(1.0 + gamma_beta_squared(P));
}

/* physics_srhd.hpp:259-270 */
void mo_srhd_flux(const double P[5], const double U[5], int axis, double F[5])
{
    const double* n = NHAT[axis];
    double v = beta_along(P, axis);
    double p = P[4];
    F[0] = v * U[0];
    F[1] = v * U[1] + p * n[0];
    F[2] = v * U[2] + p * n[1];
    F[3] = v * U[3] + p * n[2];
    F[4] = v * U[4] + p * v;
}

/* physics_srhd.hpp:283-295 */
void mo_srhd_wavespeeds(const double P[5], int axis, double gamma, double lam[2])
{
    double c2 = gamma * P[4] / enthalpy_density(P, gamma);
    double vn = beta_along(P, axis);
    double uu = gamma_beta_squared(P);
    double vv = uu / (1 + uu);
    double v2 = vn * vn;
    double k0 = sqrt(c2 * (1 - vv) * (1 - vv * c2 - v2 * (1 - c2)));
    lam[0] = (vn * (1 - c2) - k0) / (1 - vv * c2);
    lam[1] = (vn * (1 - c2) + k0) / (1 - vv * c2);
}

/* physics_srhd.hpp:466-483 */
void mo_srhd_riemann_hlle(const double Pl[5], const double Pr[5], int axis, double gamma, double F[5])
{
    double Ul[5], Ur[5], Al[2], Ar[2], Fl[5], Fr[5];
    mo_srhd_to_conserved_density(Pl, gamma, Ul);
    mo_srhd_to_conserved_density(Pr, gamma, Ur);
    mo_srhd_wavespeeds(Pl, axis, gamma, Al);
    mo_srhd_wavespeeds(Pr, axis, gamma, Ar);
    mo_srhd_flux(Pl, Ul, axis, Fl);
    mo_srhd_flux(Pr, Ur, axis, Fr);
    double ap = std_max(0.0, std_max(Al[1], Ar[1]));
    double am = std_min(0.0, std_min(Al[0], Ar[0]));
    for (int q = 0; q < 5; ++q)
        F[q] = (Fl[q] * ap - Fr[q] * am - (Ul[q] - Ur[q]) * ap * am) / (ap - am);
}

/* physics_srhd.hpp:309-326 */
void mo_srhd_source_terms(const double P[5], double r, double theta, double gamma, double S[5])
{
    double cotq = tan(M_PI_2 - theta);
    double ur = P[1], uq = P[2], up = P[3], pg = P[4];
    double H = enthalpy_density(P, gamma);
    S[0] = 0.0;
    S[1] = (2.0  * pg + H * (uq * uq        + up * up)) / r;
    S[2] = (cotq * pg + H * (up * up * cotq - ur * uq)) / r;
    S[3] =        -up * H * (ur + uq * cotq) / r;
    S[4] = 0.0;
}

void mo_srhd_recover_primitive_n(size_t n, const double* U, double gamma, double tfloor, double* P, int* status)
{
    for (size_t i = 0; i < n; ++i) status[i] = mo_srhd_recover_primitive(U + 5 * i, gamma, tfloor, P + 5 * i);
}
void mo_srhd_to_conserved_density_n(size_t n, const double* P, double gamma, double* U)
{
    for (size_t i = 0; i < n; ++i) mo_srhd_to_conserved_density(P + 5 * i, gamma, U + 5 * i);
}
void mo_srhd_riemann_hlle_n(size_t n, const double* Pl, const double* Pr, int axis, double gamma, double* F)
{
    for (size_t i = 0; i < n; ++i) mo_srhd_riemann_hlle(Pl + 5 * i, Pr + 5 * i, axis, gamma, F + 5 * i);
}
void mo_srhd_source_terms_n(size_t n, const double* P, const double* r, const double* theta, double gamma, double* S)
{
    for (size_t i = 0; i < n; ++i) mo_srhd_source_terms(P + 5 * i, r[i], theta[i], gamma, S + 5 * i);
}

/* ------------------------------------------------------------------------ */
/* cloud geometry, subprog_cloud.cpp:260-290. Arrays: dAr[(nr+1)*nq], dAq[nr*(nq+1)], dv[nr*nq]. */
void mo_cloud_geometry(size_t nr, size_t nq, const double* rv, const double* qv, double* dAr, double* dAq, double* dv)
{
    for (size_t i = 0; i <= nr; ++i)
        for (size_t j = 0; j < nq; ++j)
        {
            double rc = (rv[i] + rv[i]) * 0.5;
            double dm = -cos(qv[j + 1]) - -cos(qv[j]);
            dAr[i * nq + j] = rc * rc * dm * 2 * M_PI;
        }
    for (size_t i = 0; i < nr; ++i)
        for (size_t j = 0; j <= nq; ++j)
        {
            double dr = rv[i + 1] - rv[i];
            double rc = (rv[i] + rv[i + 1]) * 0.5;
            double qc = (qv[j] + qv[j]) * 0.5;
            dAq[i * (nq + 1) + j] = rc * dr * sin(qc) * 2 * M_PI;
        }
    for (size_t i = 0; i < nr; ++i)
        for (size_t j = 0; j < nq; ++j)
        {
            double r0 = rv[i] * rv[i] * rv[i], r1 = rv[i + 1] * rv[i + 1] * rv[i + 1];
            double d3 = r1 - r0;
            double dvr = (d3 + d3) * 0.5;
            double dmj = -cos(qv[j + 1]) - -cos(qv[j]);
            double dm = (dmj + dmj) * 0.5;
            dv[i * nq + j] = dvr * dm * 2 * M_PI / 3.0;
        }
}

static void plm5(const double* a, const double* b, const double* c, double theta, double scale_zero, double G[5])
{
    for (int q = 0; q < 5; ++q)
    {
        double g = mo_plm_gradient(a[q], b[q], c[q], theta);   /* same arithmetic as subprog_cloud.cpp:450-464 */
        G[q] = scale_zero ? g * 0 : g;
    }
}

/* One stage. inflow: [nq][5] primitives of the inner ghost row. plm_theta < 0 => reconstruct_method 1.
 * Returns the OR of the c2p status bits over all cells. */
int mo_cloud_advance(size_t nr, size_t nq, const double* rv, const double* qv, const double* inflow,
                     double gamma, double plm_theta, double temperature_floor, double dt, const double* u0, double* u1)
{
    double* dAr = (double*) malloc(sizeof(double) * (nr + 1) * nq);
    double* dAq = (double*) malloc(sizeof(double) * nr * (nq + 1));
    double* dv  = (double*) malloc(sizeof(double) * nr * nq);
    double* pe  = (double*) malloc(sizeof(double) * 5 * (nr + 2) * nq);   /* primitives with radial ghost rows */
    double* Fr  = (double*) malloc(sizeof(double) * 5 * (nr + 1) * nq);
    double* Fq  = (double*) malloc(sizeof(double) * 5 * nr * (nq + 1));
    int status = 0;
    mo_cloud_geometry(nr, nq, rv, qv, dAr, dAq, dv);

#define PE(e, j) (pe + 5 * ((e) * nq + (j)))
    for (size_t i = 0; i < nr; ++i)
        for (size_t j = 0; j < nq; ++j)
        {
            double U[5];
            for (int q = 0; q < 5; ++q) U[q] = u0[5 * (i * nq + j) + q] / dv[i * nq + j];
            status |= mo_srhd_recover_primitive(U, gamma, temperature_floor, PE(i + 1, j));
        }
    for (size_t j = 0; j < nq; ++j)
    {
        memcpy(PE(0, j), inflow + 5 * j, 5 * sizeof(double));            /* extend_inflow_nozzle_inner :466-493 */
        memcpy(PE(nr + 1, j), PE(nr, j), 5 * sizeof(double));            /* extend_zero_gradient_outer :503-509 */
    }

    /* radial faces f = 0..nr between extended rows f and f+1 */
    for (size_t f = 0; f <= nr; ++f)
        for (size_t j = 0; j < nq; ++j)
        {
            double PL[5], PR[5];
            if (plm_theta < 0.0)
            {
                memcpy(PL, PE(f, j), sizeof PL);
                memcpy(PR, PE(f + 1, j), sizeof PR);
            }
            else
            {
                double GL[5], GR[5];
                /* gradient of extended row e: real rows e=1..nr; ghosts take the edge gradient times zero */
                size_t el = f, er = f + 1;
                size_t cl = el == 0 ? 1 : el, cr = er == nr + 1 ? nr : er;
                plm5(PE(cl - 1, j), PE(cl, j), PE(cl + 1, j), plm_theta, el == 0, GL);
                plm5(PE(cr - 1, j), PE(cr, j), PE(cr + 1, j), plm_theta, er == nr + 1, GR);
                for (int q = 0; q < 5; ++q)
                {
                    PL[q] = PE(el, j)[q] + GL[q] * 0.5;
                    PR[q] = PE(er, j)[q] - GR[q] * 0.5;
                }
            }
            mo_srhd_riemann_hlle(PL, PR, 0, gamma, Fr + 5 * (f * nq + j));
        }

    /* polar faces: interior faces g = 1..nq-1 between cells g-1 and g; pole faces are the neighbouring flux times zero */
    for (size_t i = 0; i < nr; ++i)
    {
        for (size_t g = 1; g < nq; ++g)
        {
            double PL[5], PR[5];
            const double* pl = PE(i + 1, g - 1);
            const double* pr = PE(i + 1, g);
            if (plm_theta < 0.0)
            {
                memcpy(PL, pl, sizeof PL);
                memcpy(PR, pr, sizeof PR);
            }
            else
            {
                double GL[5], GR[5];
                size_t jl = g - 1, jr = g;
                size_t cl = jl == 0 ? 1 : jl, cr = jr == nq - 1 ? nq - 2 : jr;
                plm5(PE(i + 1, cl - 1), PE(i + 1, cl), PE(i + 1, cl + 1), plm_theta, jl == 0, GL);
                plm5(PE(i + 1, cr - 1), PE(i + 1, cr), PE(i + 1, cr + 1), plm_theta, jr == nq - 1, GR);
                for (int q = 0; q < 5; ++q)
                {
                    PL[q] = pl[q] + GL[q] * 0.5;
                    PR[q] = pr[q] - GR[q] * 0.5;
                }
            }
            mo_srhd_riemann_hlle(PL, PR, 1, gamma, Fq + 5 * (i * (nq + 1) + g));
        }
        for (int q = 0; q < 5; ++q)
        {
            Fq[5 * (i * (nq + 1) + 0) + q]  = Fq[5 * (i * (nq + 1) + 1) + q] * 0;
            Fq[5 * (i * (nq + 1) + nq) + q] = Fq[5 * (i * (nq + 1) + nq - 1) + q] * 0;
        }
    }

    for (size_t i = 0; i < nr; ++i)
        for (size_t j = 0; j < nq; ++j)
        {
            double S[5];
            double rc = (rv[i] + rv[i + 1]) * 0.5, qc = (qv[j] + qv[j + 1]) * 0.5;
            mo_srhd_source_terms(PE(i + 1, j), rc, qc, gamma, S);
            for (int q = 0; q < 5; ++q)
            {
                double lr = Fr[5 * ((i + 1) * nq + j) + q] * -dAr[(i + 1) * nq + j] - Fr[5 * (i * nq + j) + q] * -dAr[i * nq + j];
                double lq = Fq[5 * (i * (nq + 1) + j + 1) + q] * -dAq[i * (nq + 1) + j + 1] - Fq[5 * (i * (nq + 1) + j) + q] * -dAq[i * (nq + 1) + j];
                double s0 = S[q] * dv[i * nq + j];
                u1[5 * (i * nq + j) + q] = u0[5 * (i * nq + j) + q] + (lr + lq + s0) * dt;
            }
        }
#undef PE
    free(dAr); free(dAq); free(dv); free(pe); free(Fr); free(Fq);
    return status;
}

/* nsteps steps; inflow: [nsteps][nq][5] (step-start time for both RK stages, subprog_cloud.cpp:468-473) */
int mo_cloud_run(size_t nr, size_t nq, const double* rv, const double* qv, const double* inflow, double gamma,
                 double plm_theta, double temperature_floor, int rk_order, double dt, int nsteps, double* u)
{
    size_t n = 5 * nr * nq;
    double* a = (double*) malloc(sizeof(double) * n);
    double* b = (double*) malloc(sizeof(double) * n);
    int status = 0;
    for (int s = 0; s < nsteps; ++s)
    {
        const double* in = inflow + (size_t) s * nq * 5;
        if (rk_order == 1)
        {
            status |= mo_cloud_advance(nr, nq, rv, qv, in, gamma, plm_theta, temperature_floor, dt, u, a);
            memcpy(u, a, sizeof(double) * n);
        }
        else
        {
            status |= mo_cloud_advance(nr, nq, rv, qv, in, gamma, plm_theta, temperature_floor, dt, u, a);
            status |= mo_cloud_advance(nr, nq, rv, qv, in, gamma, plm_theta, temperature_floor, dt, a, b);
            for (size_t m = 0; m < n; ++m) u[m] = u[m] * 0.5 + b[m] * 0.5;
        }
    }
    free(a);
    free(b);
    return status;
}


/* ---- CloudProblem::make_diagnostic_fields, subprog_cloud.cpp:334-433 -------------------------------------
 * u [nr][nq][5] cell-integrated conserved; units = {length, mass, time} of make_reference_units (:318-326, unit_system_t :177-195).
 * fields [5][nr][nq] = mass_density, gas_pressure, specific_entropy, radial_gamma_beta, radial_energy_flow;
 * columns [15][nq] in the order of diagnostic_fields_t (:147-161). The shock locator is post_shock_locator.hpp:73-170: the first
 * minimum of the entropy difference; scans that end in the reference's bounds_check exception return index 0. */
static double srhd_energy_flux_radial(const double P[5], double gamma)
{
    double U[5], F[5];
    mo_srhd_to_conserved_density(P, gamma, U);
    mo_srhd_flux(P, U, 0, F);
    return F[4];
}

int mo_cloud_diagnostics(size_t nr, size_t nq, const double* rv, const double* qv, const double* u, double gamma, double tfloor,
                         const double units[3], double* fields, double* columns)
{
    const double light_speed_cgs = 2.998e10;
    const double u_length = units[0], u_mass = units[1], u_time = units[2];
    const double u_energy = u_mass * pow(light_speed_cgs, 2);
    const double u_mass_density = u_mass / pow(u_length, 3);
    const double u_energy_density = u_energy / pow(u_length, 3);
    const double u_power = u_energy / u_time;
    double* dAr = (double*) malloc(sizeof(double) * (nr + 1) * nq);
    double* dAq = (double*) malloc(sizeof(double) * nr * (nq + 1));
    double* dv = (double*) malloc(sizeof(double) * nr * nq);
    double* P = (double*) malloc(sizeof(double) * nr * nq * 5);
    double* s0 = (double*) malloc(sizeof(double) * nr);
    double* L = (double*) malloc(sizeof(double) * nr);
    int status = 0;
    mo_cloud_geometry(nr, nq, rv, qv, dAr, dAq, dv);
    for (size_t c = 0; c < nr * nq; ++c)
    {
        double Ud[5];
        for (int q = 0; q < 5; ++q) Ud[q] = u[5 * c + q] / dv[c];
        status |= mo_srhd_recover_primitive(Ud, gamma, tfloor, P + 5 * c);
    }
    for (size_t i = 0; i < nr; ++i)
        for (size_t j = 0; j < nq; ++j)
        {
            const double* p = P + 5 * (i * nq + j);
            const size_t c = i * nq + j;
            fields[0 * nr * nq + c] = p[0] * u_mass_density;
            fields[1 * nr * nq + c] = p[4] * u_energy_density;
            fields[2 * nr * nq + c] = log(p[4] / pow(p[0], gamma));
            fields[3 * nr * nq + c] = p[1];
            fields[4 * nr * nq + c] = (srhd_energy_flux_radial(p, gamma) * dAr[i * nq + j]) * u_power;
        }
    for (size_t j = 0; j < nq; ++j)
    {
        for (size_t i = 0; i < nr; ++i)
        {
            const double* p = P + 5 * (i * nq + j);
            const double Aj = (dAr[i * nq + j] + dAr[(i + 1) * nq + j]) * 0.5;
            s0[i] = log(p[4] / pow(p[0], gamma));
            L[i] = (srhd_energy_flux_radial(p, gamma) * Aj) * u_power;
        }
#define PRESSURE(i) P[5 * ((i) * nq + j) + 4]
        /* find_shock_index: first index of the minimum entropy difference */
        size_t mid = 0;
        {
            double dsmin = s0[1] - s0[0];
            for (size_t i = 1; i + 1 < nr; ++i) { const double ds = s0[i + 1] - s0[i]; if (ds < dsmin) { dsmin = ds; mid = i; } }
        }
        /* find_index_of_pressure_plateau_ahead: dlogp has nr - 1 entries; an index outside them ends the scan with 0 */
        size_t up = mid;
        for (;;)
        {
            const size_t a = up - 1, b = up - 2;                      /* wrap like std::size_t */
            if (a >= nr - 1 || b >= nr - 1) { up = 0; break; }
            const double da = log(PRESSURE(a + 1)) - log(PRESSURE(a));
            const double db = log(PRESSURE(b + 1)) - log(PRESSURE(b));
            if (da < 0.5 * db) ++up; else break;
        }
        /* find_index_of_maximum_pressure_behind / find_index_of_maximum_behind */
        size_t pi = mid;
        for (;;)
        {
            const size_t a = pi - 1;
            if (a >= nr || pi >= nr) { pi = 0; break; }
            if (PRESSURE(a) > PRESSURE(pi)) --pi; else break;
        }
        size_t li = mid;
        for (;;)
        {
            const size_t a = li - 1;
            if (a >= nr || li >= nr) { li = 0; break; }
            if (L[a] > L[li]) --li; else break;
        }
#undef PRESSURE
        const size_t back[6] = {2, 4, 8, 16, 32, 64};
        double total = 0.0;
        for (size_t i = 0; i < nr; ++i) total = total + u[5 * (i * nq + j) + 4] * u_energy;
        const double* pp = P + 5 * (pi * nq + j);
        columns[0 * nq + j] = total;
        columns[1 * nq + j] = dAr[j] / rv[0] / rv[0];
        columns[2 * nq + j] = ((rv[mid] + rv[mid + 1]) * 0.5) * u_length;
        columns[3 * nq + j] = ((rv[up] + rv[up + 1]) * 0.5) * u_length;
        columns[4 * nq + j] = ((rv[pi] + rv[pi + 1]) * 0.5) * u_length;
        columns[5 * nq + j] = ((rv[li] + rv[li + 1]) * 0.5) * u_length;
        columns[6 * nq + j] = sqrt(1.0 + (pp[1] * pp[1] + pp[2] * pp[2] + pp[3] * pp[3]));
        columns[7 * nq + j] = L[pi];
        for (int k = 0; k < 6; ++k) columns[(8 + k) * nq + j] = L[mid > back[k] ? mid - back[k] : 0];
        columns[14 * nq + j] = L[li];
    }
    free(dAr); free(dAq); free(dv); free(P); free(s0); free(L);
    return status;
}
