/*
 * mara_oracle.c — TEST INFRASTRUCTURE (see mara_oracle.h). Plain C, scalar
 * loops, no FMA contraction (build with -ffp-contract=off). Every function
 * cites the reference lines (relative to /root/reference/src) it restates.
 */
#include "mara_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <pthread.h>

/* std::max(a,b) = (a<b)?b:a ; std::min(a,b) = (b<a)?b:a  (NaN behaviour kept) */
static inline double std_max(double a, double b) { return (a < b) ? b : a; }
static inline double std_min(double a, double b) { return (b < a) ? b : a; }

static const double NHAT[3][3] = {{1.0, 0.0, 0.0}, {0.0, 1.0, 0.0}, {0.0, 0.0, 1.0}};

/* ------------------------------------------------------------------------ */
/* math_interpolation.hpp:85-94 */
double mo_plm_gradient(double yl, double y0, double yr, double theta)
{
    double a = (y0 - yl) * theta;
    double b = (yr - yl) * 0.5;
    double c = (yr - y0) * theta;
    double sa = copysign(1.0, a), sb = copysign(1.0, b), sc = copysign(1.0, c);
    double m = std_min(std_min(fabs(a), fabs(b)), fabs(c));
    return 0.25 * fabs(sa + sb) * (sa + sc) * m;
}

/* physics_euler.hpp:555-575 */
void mo_euler_recover_primitive(const double U[5], double gamma, double tfloor, double P[5])
{
    double p_squared = U[1] * U[1] + U[2] * U[2] + U[3] * U[3];
    double d = U[0];
    P[0] = d;
    P[1] = U[1] / d;
    P[2] = U[2] / d;
    P[3] = U[3] / d;
    P[4] = (U[4] - 0.5 * p_squared / d) * (gamma - 1.0);
    if (P[4] < 0.0 && tfloor > 0.0)
        P[4] = tfloor * d;
}

/* physics_euler.hpp:209-220 (velocity_squared :160-164) */
void mo_euler_to_conserved_density(const double P[5], double gamma, double U[5])
{
    double d = P[0], p = P[4];
    double vsq = P[1] * P[1] + P[2] * P[2] + P[3] * P[3];
    U[0] = d;
    U[1] = d * P[1];
    U[2] = d * P[2];
    U[3] = d * P[3];
    U[4] = 0.5 * d * vsq + p / (gamma - 1);
}

/* velocity_along :177-181 -> unit_vector_t::project core_geometric.hpp:85-89 */
static inline double velocity_along(const double P[5], int axis)
{
    const double* n = NHAT[axis];
    return P[1] * n[0] + P[2] * n[1] + P[3] * n[2];
}

/* physics_euler.hpp:252-263 */
void mo_euler_flux(const double P[5], const double U[5], int axis, double F[5])
{
    const double* n = NHAT[axis];
    double v = velocity_along(P, axis);
    double p = P[4];
    F[0] = v * U[0];
    F[1] = v * U[1] + p * n[0];
    F[2] = v * U[2] + p * n[1];
    F[3] = v * U[3] + p * n[2];
    F[4] = v * U[4] + p * v;
}

/* physics_euler.hpp:276-284, sound_speed_squared :193-196 */
void mo_euler_wavespeeds(const double P[5], int axis, double gamma, double lam[2])
{
    double cs = sqrt(gamma * P[4] / P[0]);
    double vn = velocity_along(P, axis);
    lam[0] = vn - cs;
    lam[1] = vn + cs;
}

/* physics_euler.hpp:614-631 */
void mo_euler_riemann_hlle(const double Pl[5], const double Pr[5], int axis, double gamma, double F[5])
{
    double Ul[5], Ur[5], Al[2], Ar[2], Fl[5], Fr[5];
    mo_euler_to_conserved_density(Pl, gamma, Ul);
    mo_euler_to_conserved_density(Pr, gamma, Ur);
    mo_euler_wavespeeds(Pl, axis, gamma, Al);
    mo_euler_wavespeeds(Pr, axis, gamma, Ar);
    mo_euler_flux(Pl, Ul, axis, Fl);
    mo_euler_flux(Pr, Ur, axis, Fr);
    double ap = std_max(0.0, std_max(Al[1], Ar[1]));
    double am = std_min(0.0, std_min(Al[0], Ar[0]));
    for (int q = 0; q < 5; ++q)
        F[q] = (Fl[q] * ap - Fr[q] * am - (Ul[q] - Ur[q]) * ap * am) / (ap - am);
}

/* NO UPSTREAM COUNTERPART — parity unpinned. Structure follows
 * physics_iso2d.hpp:610-687 (variables) and :556-583 (star states, flux
 * selection); gamma-law closure from Toro eq. 10.69 and the energy row of
 * eq. 10.73. */
void mo_euler_riemann_hllc(const double Pl[5], const double Pr[5], int axis, double gamma, double F[5])
{
    const double* n = NHAT[axis];
    double Ul[5], Ur[5], Fl[5], Fr[5];
    mo_euler_to_conserved_density(Pl, gamma, Ul);
    mo_euler_to_conserved_density(Pr, gamma, Ur);
    mo_euler_flux(Pl, Ul, axis, Fl);
    mo_euler_flux(Pr, Ur, axis, Fr);

    double ul = velocity_along(Pl, axis);
    double ur = velocity_along(Pr, axis);
    double dl = Pl[0], dr = Pr[0], pl = Pl[4], pr = Pr[4];
    double dbar = 0.5 * (dl + dr);
    double al = sqrt(gamma * pl / dl);
    double ar = sqrt(gamma * pr / dr);
    double abar = 0.5 * (al + ar);
    double ppvrs = 0.5 * (pl + pr) - 0.5 * (ur - ul) * dbar * abar;
    double pstar = std_max(0.0, ppvrs);
    double gfac = (gamma + 1.0) / (2.0 * gamma);
    double ql = pstar <= pl ? 1.0 : sqrt(1.0 + gfac * (pstar / pl - 1.0));
    double qr = pstar <= pr ? 1.0 : sqrt(1.0 + gfac * (pstar / pr - 1.0));
    double sl = ul - al * ql;
    double sr = ur + ar * qr;
    double den = dl * (sl - ul) - dr * (sr - ur);
    double sstar = (pr - pl + ul * dl * (sl - ul) - ur * dr * (sr - ur)) / den;

    if (0.0 <= sl)
    {
        for (int q = 0; q < 5; ++q) F[q] = Fl[q];
    }
    else if (sl <= 0.0 && 0.0 <= sstar)
    {
        double fac = dl * (sl - ul) / (sl - sstar);
        double Us[5];
        Us[0] = fac;
        for (int k = 0; k < 3; ++k)
            Us[1 + k] = fac * (sstar * n[k] + (Pl[1 + k] - n[k] * ul));
        Us[4] = fac * (Ul[4] / dl + (sstar - ul) * (sstar + pl / (dl * (sl - ul))));
        for (int q = 0; q < 5; ++q) F[q] = Fl[q] + (Us[q] - Ul[q]) * sl;
    }
    else if (sstar <= 0.0 && 0.0 <= sr)
    {
        double fac = dr * (sr - ur) / (sr - sstar);
        double Us[5];
        Us[0] = fac;
        for (int k = 0; k < 3; ++k)
            Us[1 + k] = fac * (sstar * n[k] + (Pr[1 + k] - n[k] * ur));
        Us[4] = fac * (Ur[4] / dr + (sstar - ur) * (sstar + pr / (dr * (sr - ur))));
        for (int q = 0; q < 5; ++q) F[q] = Fr[q] + (Us[q] - Ur[q]) * sr;
    }
    else if (sr <= 0.0)
    {
        for (int q = 0; q < 5; ++q) F[q] = Fr[q];
    }
    else /* NaN wave speeds: the iso2d template throws here (:582) */
    {
        for (int q = 0; q < 5; ++q) F[q] = NAN;
    }
}

/* physics_euler.hpp:328-337 */
void mo_euler_source_terms_radial(const double P[5], double r, double S[5])
{
    double vq = P[2], pg = P[4], d = P[0];
    S[0] = 0.0;
    S[1] = (2.0 * pg + d * vq * vq) / r;
    S[2] = 0.0;
    S[3] = 0.0;
    S[4] = 0.0;
}

/* ---- array forms -------------------------------------------------------- */
void mo_plm_gradient_n(size_t n, const double* yl, const double* y0, const double* yr, double theta, double* g)
{
    for (size_t i = 0; i < n; ++i) g[i] = mo_plm_gradient(yl[i], y0[i], yr[i], theta);
}
void mo_euler_recover_primitive_n(size_t n, const double* U, double gamma, double tfloor, double* P)
{
    for (size_t i = 0; i < n; ++i) mo_euler_recover_primitive(U + 5 * i, gamma, tfloor, P + 5 * i);
}
void mo_euler_to_conserved_density_n(size_t n, const double* P, double gamma, double* U)
{
    for (size_t i = 0; i < n; ++i) mo_euler_to_conserved_density(P + 5 * i, gamma, U + 5 * i);
}
void mo_euler_riemann_n(size_t n, const double* Pl, const double* Pr, int axis, double gamma, int solver, double* F)
{
    for (size_t i = 0; i < n; ++i)
    {
        if (solver == MO_RIEMANN_HLLC) mo_euler_riemann_hllc(Pl + 5 * i, Pr + 5 * i, axis, gamma, F + 5 * i);
        else                           mo_euler_riemann_hlle(Pl + 5 * i, Pr + 5 * i, axis, gamma, F + 5 * i);
    }
}

/* ------------------------------------------------------------------------ */
/* Uniform cartesian Euler stage. Composition restated from
 * subprog_cloud.cpp:511-584 with cartesian geometry (see
 * oracle/ref_drivers/euler_cart_ref.cpp for the reference-header form):
 *   - ghost primitives: extend_zero_gradient (core_ndarray_ops.hpp:172-180)
 *     or extend_periodic_on_axis (:162-170)
 *   - ghost gradients (outflow): edge gradient * 0 (extend_zeros :182-190)
 *   - face states PL = P + G*0.5, PR = P - G*0.5 (subprog_cloud.cpp:566-568)
 */
typedef struct
{
    const mo_euler_cart_t* cfg;
    const double* u0;
    const double* p0;
    double* u1;
    double dt;
    size_t i0, i1;    /* axis-0 range of this slab */
    int phase;        /* 0: cons->prim, 1: update */
} cart_job_t;

static inline size_t cell_index(const mo_euler_cart_t* c, const long idx[3])
{
    return ((size_t) idx[0] * c->shape[1] + (size_t) idx[1]) * c->shape[2] + (size_t) idx[2];
}

/* Primitive of the (possibly ghost) cell at position k on `axis`; other coords from idx. */
static inline const double* prim_at(const mo_euler_cart_t* c, const double* p0, const long idx[3], int axis, long k)
{
    long n = (long) c->shape[axis];
    long id[3] = {idx[0], idx[1], idx[2]};
    if (c->bc == MO_BC_PERIODIC) k = ((k % n) + n) % n;
    else                         k = k < 0 ? 0 : (k >= n ? n - 1 : k);
    id[axis] = k;
    return p0 + 5 * cell_index(c, id);
}

/* PLM gradient of the (possibly ghost) cell k along axis. */
static inline void grad_at(const mo_euler_cart_t* c, const double* p0, const long idx[3], int axis, long k, double G[5])
{
    long n = (long) c->shape[axis];
    if (c->bc == MO_BC_PERIODIC || (k >= 0 && k < n))
    {
        const double* a = prim_at(c, p0, idx, axis, k - 1);
        const double* b = prim_at(c, p0, idx, axis, k);
        const double* d = prim_at(c, p0, idx, axis, k + 1);
        for (int q = 0; q < 5; ++q) G[q] = mo_plm_gradient(a[q], b[q], d[q], c->plm_theta);
    }
    else /* outflow ghost: nearest real cell's gradient times zero */
    {
        long ke = k < 0 ? 0 : n - 1;
        const double* a = prim_at(c, p0, idx, axis, ke - 1);
        const double* b = prim_at(c, p0, idx, axis, ke);
        const double* d = prim_at(c, p0, idx, axis, ke + 1);
        for (int q = 0; q < 5; ++q) G[q] = mo_plm_gradient(a[q], b[q], d[q], c->plm_theta) * 0;
    }
}

/* Godunov flux through the face between cells k-1 and k along axis. */
static inline void face_flux(const mo_euler_cart_t* c, const double* p0, const long idx[3], int axis, long k, double F[5])
{
    double PL[5], PR[5];
    const double* pl = prim_at(c, p0, idx, axis, k - 1);
    const double* pr = prim_at(c, p0, idx, axis, k);

    if (c->plm_theta < 0.0)
    {
        memcpy(PL, pl, sizeof PL);
        memcpy(PR, pr, sizeof PR);
    }
    else
    {
        double GL[5], GR[5];
        grad_at(c, p0, idx, axis, k - 1, GL);
        grad_at(c, p0, idx, axis, k, GR);
        for (int q = 0; q < 5; ++q)
        {
            PL[q] = pl[q] + GL[q] * 0.5;
            PR[q] = pr[q] - GR[q] * 0.5;
        }
    }
    if (c->riemann == MO_RIEMANN_HLLC) mo_euler_riemann_hllc(PL, PR, axis, c->gamma, F);
    else                               mo_euler_riemann_hlle(PL, PR, axis, c->gamma, F);
}

static void* cart_worker(void* arg)
{
    cart_job_t* job = (cart_job_t*) arg;
    const mo_euler_cart_t* c = job->cfg;
    size_t n1 = c->shape[1], n2 = c->shape[2];

    if (job->phase == 0)
    {
        size_t a = job->i0 * n1 * n2, b = job->i1 * n1 * n2;
        for (size_t m = a; m < b; ++m)
            mo_euler_recover_primitive(job->u0 + 5 * m, c->gamma, 0.0, (double*) job->p0 + 5 * m);
        return NULL;
    }
    for (size_t i = job->i0; i < job->i1; ++i)
    for (size_t j = 0; j < n1; ++j)
    for (size_t k = 0; k < n2; ++k)
    {
        long idx[3] = {(long) i, (long) j, (long) k};
        size_t m = cell_index(c, idx);
        double L[3][5];

        for (int axis = 0; axis < c->rank; ++axis)
        {
            double F0[5], F1[5];
            double dtdl = job->dt / c->dl[axis];
            face_flux(c, job->p0, idx, axis, idx[axis], F0);
            face_flux(c, job->p0, idx, axis, idx[axis] + 1, F1);
            for (int q = 0; q < 5; ++q) L[axis][q] = (F1[q] - F0[q]) * dtdl;
        }
        for (int q = 0; q < 5; ++q)
        {
            double s = L[0][q];
            if (c->rank > 1) s = s + L[1][q];
            if (c->rank > 2) s = s + L[2][q];
            job->u1[5 * m + q] = job->u0[5 * m + q] - s;
        }
    }
    return NULL;
}

static void cart_dispatch(cart_job_t proto, int phase)
{
    const mo_euler_cart_t* c = proto.cfg;
    int nt = c->nthreads < 1 ? 1 : c->nthreads;
    if ((size_t) nt > c->shape[0]) nt = (int) c->shape[0];
    pthread_t* th = (pthread_t*) malloc(sizeof(pthread_t) * nt);
    cart_job_t* jobs = (cart_job_t*) malloc(sizeof(cart_job_t) * nt);

    for (int t = 0; t < nt; ++t)
    {
        jobs[t] = proto;
        jobs[t].phase = phase;
        mo_partition_rows(c->shape[0], nt, t, &jobs[t].i0, &jobs[t].i1);
        if (nt == 1) cart_worker(&jobs[t]);
        else pthread_create(&th[t], NULL, cart_worker, &jobs[t]);
    }
    if (nt > 1) for (int t = 0; t < nt; ++t) pthread_join(th[t], NULL);
    free(th);
    free(jobs);
}

int mo_euler_cart_advance(const mo_euler_cart_t* cfg, const double* u0, double dt, double* u1)
{
    if (cfg->rank < 1 || cfg->rank > 3) return -1;
    size_t ncell = cfg->shape[0] * cfg->shape[1] * cfg->shape[2];
    double* p0 = (double*) malloc(sizeof(double) * 5 * ncell);
    if (! p0) return -2;
    cart_job_t proto = {cfg, u0, p0, u1, dt, 0, 0, 0};
    cart_dispatch(proto, 0);
    cart_dispatch(proto, 1);
    free(p0);
    return 0;
}

int mo_euler_cart_run(const mo_euler_cart_t* cfg, double* u, double dt, int nsteps)
{
    size_t n = 5 * cfg->shape[0] * cfg->shape[1] * cfg->shape[2];
    double* a = (double*) malloc(sizeof(double) * n);
    double* b = (double*) malloc(sizeof(double) * n);
    if (! a || ! b) { free(a); free(b); return -2; }

    for (int s = 0; s < nsteps; ++s)
    {
        if (cfg->rk_order == 1)
        {
            mo_euler_cart_advance(cfg, u, dt, a);
            memcpy(u, a, sizeof(double) * n);
        }
        else
        {
            mo_euler_cart_advance(cfg, u, dt, a);
            mo_euler_cart_advance(cfg, a, dt, b);
            for (size_t m = 0; m < n; ++m) u[m] = u[m] * 0.5 + b[m] * 0.5;
        }
    }
    free(a);
    free(b);
    return 0;
}

/* ------------------------------------------------------------------------ */
/* sedov */

/* subprog_sedov.cpp:366-371 ; nd::linspace core_ndarray.hpp:2544-2551 */
void mo_sedov_vertices(int nr, double outer_radius, size_t* nz_out, double* vertices)
{
    double radial_decades = log10(outer_radius);
    size_t count = (size_t) ((int) (radial_decades * nr) + 1);
    if (nz_out) *nz_out = count - 1;
    if (! vertices) return;
    double x0 = -0.5, x1 = radial_decades;
    for (size_t i = 0; i < count; ++i)
    {
        double y = x0 + (x1 - x0) * i / (count - 1);
        vertices[i] = pow(10.0, y);
    }
}

/* subprog_sedov.cpp:176-179 */
static inline double shell_volume(double r0, double r1)
{
    return (pow(r1, 3) - pow(r0, 3)) / 3;
}

/* subprog_sedov.cpp:353-363,373-380 */
void mo_sedov_initial_system(int srhd, size_t nz, const double* v, double gamma, double explosion_density,
                             double explosion_pressure, double density_index, double* u)
{
    for (size_t i = 0; i < nz; ++i)
    {
        double r = (v[i] + v[i + 1]) * 0.5;
        double P[5] = {0, 0, 0, 0, 0}, U[5];
        P[0] = r < 1.0 ? explosion_density  : pow(r, -density_index);
        P[4] = r < 1.0 ? explosion_pressure : pow(r, -density_index) * 1e-6;
        if (srhd) mo_srhd_to_conserved_density(P, gamma, U);
        else      mo_euler_to_conserved_density(P, gamma, U);
        double dv = shell_volume(v[i], v[i + 1]);
        for (int q = 0; q < 5; ++q) u[5 * i + q] = U[q] * dv;
    }
}

void mo_sedov_initial(size_t nz, const double* v, double gamma, double explosion_density,
                      double explosion_pressure, double density_index, double* u)
{
    for (size_t i = 0; i < nz; ++i)
    {
        double r = (v[i] + v[i + 1]) * 0.5;
        double P[5] = {0, 0, 0, 0, 0}, U[5];
        P[0] = r < 1.0 ? explosion_density  : pow(r, -density_index);
        P[4] = r < 1.0 ? explosion_pressure : pow(r, -density_index) * 1e-6;
        mo_euler_to_conserved_density(P, gamma, U);
        double dv = shell_volume(v[i], v[i + 1]);
        for (int q = 0; q < 5; ++q) u[5 * i + q] = U[q] * dv;
    }
}

/* subprog_sedov.cpp:404-405 */
double mo_sedov_timestep(const double* v, double cfl)
{
    return cfl * (v[1] - v[0]);
}

/* subprog_sedov.cpp:394-421 (+ :217-250 BCs and flux) */
/* the same with HydroSystem = mara::srhd (the sub-program's default, src/subprog_sedov.cpp:652-659);
 * radial source term src/physics_srhd.hpp:339-348. Returns the OR of the c2p failure bits. */
int mo_sedov_advance_srhd(size_t nz, const double* v, double gamma, double dt, const double* u0, double* u1)
{
    double* p = (double*) malloc(sizeof(double) * 5 * (nz + 2));
    double* F = (double*) malloc(sizeof(double) * 5 * (nz + 1));
    int status = 0;
    for (size_t i = 0; i < nz; ++i)
    {
        double dv = shell_volume(v[i], v[i + 1]);
        double U[5];
        for (int q = 0; q < 5; ++q) U[q] = u0[5 * i + q] / dv;
        status |= mo_srhd_recover_primitive(U, gamma, 0.0, p + 5 * (i + 1));
    }
    memcpy(p, p + 5, sizeof(double) * 5);
    p[1] = -p[1];
    memcpy(p + 5 * (nz + 1), p + 5 * nz, sizeof(double) * 5);
    for (size_t f = 0; f <= nz; ++f)
        mo_srhd_riemann_hlle(p + 5 * f, p + 5 * (f + 1), 0, gamma, F + 5 * f);
    for (size_t i = 0; i < nz; ++i)
    {
        const double* P = p + 5 * (i + 1);
        double dv = shell_volume(v[i], v[i + 1]);
        double rc = (v[i] + v[i + 1]) * 0.5;
        double H = P[0] + P[4] * (1.0 + 1.0 / (gamma - 1.0));
        double S[5] = {0.0, (2.0 * P[4] + H * P[2] * P[2]) / rc, 0.0, 0.0, 0.0};
        double da0 = v[i] * v[i], da1 = v[i + 1] * v[i + 1];
        for (int q = 0; q < 5; ++q)
        {
            double l0 = F[5 * (i + 1) + q] * (-da1) - F[5 * i + q] * (-da0);
            double s0 = S[q] * dv;
            u1[5 * i + q] = u0[5 * i + q] + (l0 + s0) * dt;
        }
    }
    free(p);
    free(F);
    return status;
}

void mo_sedov_advance(size_t nz, const double* v, double gamma, double dt, const double* u0, double* u1)
{
    double* p = (double*) malloc(sizeof(double) * 5 * (nz + 2)); /* extended primitives */
    double* F = (double*) malloc(sizeof(double) * 5 * (nz + 1));

    for (size_t i = 0; i < nz; ++i)
    {
        double dv = shell_volume(v[i], v[i + 1]);
        double U[5];
        for (int q = 0; q < 5; ++q) U[q] = u0[5 * i + q] / dv;
        mo_euler_recover_primitive(U, gamma, 0.0, p + 5 * (i + 1));
    }
    memcpy(p, p + 5, sizeof(double) * 5);                       /* reflecting inner: v_r -> -v_r (:231-241, :213-216) */
    p[1] = -p[1];
    memcpy(p + 5 * (nz + 1), p + 5 * nz, sizeof(double) * 5);   /* zero-gradient outer (:243-250) */

    for (size_t f = 0; f <= nz; ++f)
        mo_euler_riemann_hlle(p + 5 * f, p + 5 * (f + 1), 0, gamma, F + 5 * f);

    for (size_t i = 0; i < nz; ++i)
    {
        double dv = shell_volume(v[i], v[i + 1]);
        double rc = (v[i] + v[i + 1]) * 0.5;
        double S[5];
        mo_euler_source_terms_radial(p + 5 * (i + 1), rc, S);
        double da0 = v[i] * v[i], da1 = v[i + 1] * v[i + 1];
        for (int q = 0; q < 5; ++q)
        {
            double l0 = F[5 * (i + 1) + q] * (-da1) - F[5 * i + q] * (-da0);
            double s0 = S[q] * dv;
            u1[5 * i + q] = u0[5 * i + q] + (l0 + s0) * dt;
        }
    }
    free(p);
    free(F);
}

/* ------------------------------------------------------------------------ */
/* integer work */

/* core_ndarray.hpp:828-833 (also nd::divvy :2567-2583) */
void mo_partition_rows(size_t count, size_t nparts, size_t part, size_t* start, size_t* final_)
{
    *start  = (part + 0) * count / nparts;
    *final_ = (part + 1) * count / nparts;
}

/* app_parallel.hpp:185-197 */
static void factor_once(int num, int* d_out, int* rest)
{
    for (int d = 2; ; ++d)
    {
        if (num % d == 0) { *d_out = d; *rest = num / d; return; }
        if (d * d > num) break;
    }
    *d_out = num;
    *rest = 1;
}

/* app_parallel.hpp:199-212 */
static void prime_factors_impl(unsigned long* out, int* count, int num)
{
    int d, rest;
    factor_once(num, &d, &rest);
    if (rest == 1) out[(*count)++] = (unsigned long) d;
    else
    {
        prime_factors_impl(out, count, d);
        prime_factors_impl(out, count, rest);
    }
}

int mo_prime_factors(unsigned long n, unsigned long* factors)
{
    int count = 0;
    prime_factors_impl(factors, &count, (int) n);
    return count;
}

/* app_parallel.hpp:119-131 */
void mo_propose_block_decomposition(int rank, unsigned long nblocks, unsigned long* blocks_per_axis)
{
    unsigned long f[64];
    int nf = mo_prime_factors(nblocks, f);
    for (int g = 0; g < rank; ++g)
    {
        size_t a, b;
        mo_partition_rows((size_t) nf, (size_t) rank, (size_t) g, &a, &b);
        int prod = 1; /* std::accumulate(..., 1, multiplies) -> int */
        for (size_t k = a; k < b; ++k) prod *= (int) f[k];
        blocks_per_axis[g] = (unsigned long) prod;
    }
}

/* app_parallel.hpp:148-179 via nd::divvy */
void mo_block_extent(size_t n, size_t nblocks, size_t b, size_t* start, size_t* final_)
{
    mo_partition_rows(n, nblocks, b, start, final_);
}


/* ---- sedov diagnostics: SedovProblem::make_diagnostic_fields / compute_time_series_data, subprog_sedov.cpp:252-308 ---------------
 * fields [4][nz]: specific_entropy, gas_pressure, mass_density, radial velocity (Euler) or gamma-beta (SRHD); indices = shock,
 * downstream (maximum pressure behind), upstream (pressure plateau ahead) by post_shock_locator.hpp:73-170; series = time,
 * shock_radius, shock_radius_upstream, shock_radius_downstream, shock_radius_interpolated (math_polynomial.hpp:206-215),
 * shock_velocity (:96-114). The interpolated radius is NaN where the reference would index outside the array. */
int mo_sedov_diagnostics(int srhd, size_t nz, const double* v, const double* u, double gamma, double time, double* fields, int* indices, double* series)
{
    double* P = (double*) malloc(sizeof(double) * 5 * nz);
    int status = 0;
    for (size_t i = 0; i < nz; ++i)
    {
        const double dv = (pow(v[i + 1], 3) - pow(v[i], 3)) / 3;
        double U[5];
        for (int q = 0; q < 5; ++q) U[q] = u[5 * i + q] / dv;
        if (srhd) status |= mo_srhd_recover_primitive(U, gamma, 0.0, P + 5 * i);
        else      mo_euler_recover_primitive(U, gamma, 0.0, P + 5 * i);
        fields[0 * nz + i] = log(P[5 * i + 4] / pow(P[5 * i], gamma));
        fields[1 * nz + i] = P[5 * i + 4];
        fields[2 * nz + i] = P[5 * i];
        fields[3 * nz + i] = P[5 * i + 1];
    }
    const double* s0 = fields;
    const double* pr = fields + nz;
    size_t mid = 0;
    {
        double dsmin = s0[1] - s0[0];
        for (size_t i = 1; i + 1 < nz; ++i) { const double ds = s0[i + 1] - s0[i]; if (ds < dsmin) { dsmin = ds; mid = i; } }
    }
    size_t down = mid;
    for (;;)
    {
        const size_t a = down - 1;
        if (a >= nz || down >= nz) { down = 0; break; }
        if (pr[a] > pr[down]) --down; else break;
    }
    size_t up = mid;
    for (;;)
    {
        const size_t a = up - 1, b = up - 2;
        if (a >= nz - 1 || b >= nz - 1) { up = 0; break; }
        if (log(pr[a + 1]) - log(pr[a]) < 0.5 * (log(pr[b + 1]) - log(pr[b]))) ++up; else break;
    }
    indices[0] = (int) mid; indices[1] = (int) down; indices[2] = (int) up;
#define RC(i) ((v[i] + v[(i) + 1]) * 0.5)
#define VC(i) (P[5 * (i) + 1])
    series[0] = time;
    series[1] = v[mid];
    series[2] = RC(up);
    series[3] = RC(down);
    if (down >= 1 && down + 1 < nz)
    {
        const double x1 = RC(down - 1), x2 = RC(down), x3 = RC(down + 1), y1 = VC(down - 1), y2 = VC(down), y3 = VC(down + 1);
        const double d = (x1 - x2) * (x1 - x3) * (x2 - x3);
        const double A = (x3 * (y2 - y1) + x2 * (y1 - y3) + x1 * (y3 - y2)) / d;
        const double B = (x3 * x3 * (y1 - y2) + x2 * x2 * (y3 - y1) + x1 * x1 * (y2 - y3)) / d;
        series[4] = -B / (2 * A);
    }
    else series[4] = NAN;
    {
        const double* p1 = P + 5 * up;
        const double* p2 = P + 5 * down;
        if (srhd)
        {
            const double g1 = sqrt(1.0 + (p1[1] * p1[1] + p1[2] * p1[2] + p1[3] * p1[3])), g2 = sqrt(1.0 + (p2[1] * p2[1] + p2[2] * p2[2] + p2[3] * p2[3]));
            series[5] = (p2[0] * p2[1] - p1[0] * p1[1]) / (p2[0] * g2 - p1[0] * g1);
        }
        else series[5] = (p2[0] * p2[1] - p1[0] * p1[1]) / (p2[0] - p1[0]);
    }
#undef RC
#undef VC
    free(P);
    return status;
}
