/*
 * mara_oracle_binary.c — TEST INFRASTRUCTURE (see mara_oracle.h).
 *
 * Plain-C restatement of one stage of the circumbinary-disk scheme, binary::advance_u
 * (src/subprog_binary_scheme.cpp:790-904) and, with angmom_form set, binary::advance_q (:906-1020) on a uniform-depth block tree (every node
 * refined, so the grid is a periodic n x n tensor-product mesh of blocks of
 * block_size^2 cells), with the static solver data of
 * src/subprog_binary_solver_data.cpp:20-102, the initial model of
 * src/subprog_binary.cpp:105-153 and the time step of scheme.cpp:1107-1126.
 *
 * Parity status: PINNED AGAINST A REFERENCE-COMPOSED DRIVER, not against the
 * sub-program itself. oracle/ref_drivers/binary_ref.cpp calls the reference's header
 * functions (block tree of vertices, iso2d recover_primitive / plm_gradient / riemann_hlle /
 * angular_momentum / max_wavespeed, the Kepler two-body model) in the sub-program's
 * order; the arithmetic that the reference writes inline in its (here unbuildable)
 * translation unit - gravity, sinks, buffer, sound speed, viscosity, the update
 * expression - is a restatement in that driver too. tests/golden/binary_*.npz hold its output;
 * this file reproduces them bit for bit (same libm). The scheme-level composition is
 * therefore "partially pinned": leaf physics by the reference, glue by two independent
 * restatements that agree.
 */
#include "mara_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

static double phi_soft(const mo_binary_params* P, double x, double y, const double body[5])
{
    /* grav_phi_field scheme.cpp:101-111 ; body = (mass, x, y, vx, vy) */
    double d0 = x - body[1], d1 = y - body[2];
    double dr2 = d0 * d0 + d1 * d1;
    double rs2 = P->softening_radius * P->softening_radius;
    return -1.0 * body[0] / pow(dr2 + rs2, 0.5);
}

/* cs2_at_position scheme.cpp:160-175 */
double mo_binary_cs2(const mo_binary_params* P, double x, double y, const double bodies[10])
{
    double M = P->mach_number;
    if (P->axisymmetric_cs2)
        return 1.0 / sqrt(x * x + y * y) / M / M;
    return -(phi_soft(P, x, y, bodies) + phi_soft(P, x, y, bodies + 5)) / M / M;
}

/* nu_at_position scheme.cpp:177-193 */
double mo_binary_nu(const mo_binary_params* P, double x, double y, double cs2)
{
    double radius = sqrt(x * x + y * y);
    double rc = P->alpha_cutoff_radius;
    double profile = rc > 0.0 ? 0.5 * (1.0 + tanh(3.0 * (radius - rc))) : 1.0;
    if (P->nu > 0.0)
        return profile * P->nu;
    return profile * P->alpha * sqrt(cs2) * (radius / P->mach_number);
}

/* intercell_flux_u scheme.cpp:268-293 with viscous_flux :220-262 ; g = longitudinal, h = transverse gradients */
static void face_flux(const mo_binary_params* P, int axis, double spacing, double xf, double yf, const double bodies[10],
    const double* pl, const double* pr, const double* gl, const double* gr, const double* hl, const double* hr, double F[3])
{
    double pl_hat[3], pr_hat[3];
    for (int q = 0; q < 3; ++q)
    {
        pl_hat[q] = pl[q] + gl[q] * 0.5 * spacing;
        pr_hat[q] = pr[q] - gr[q] * 0.5 * spacing;
    }
    double cs2 = mo_binary_cs2(P, xf, yf, bodies);
    double nu = mo_binary_nu(P, xf, yf, cs2);
    double mu = 0.5 * nu * (pl_hat[0] + pr_hat[0]);
    mo_iso2d_riemann_hlle(pl_hat, pr_hat, cs2, cs2, axis, F);
    if (axis == 0)
    {
        double dx_ux = 0.5 * (gl[1] + gr[1]);
        double dx_uy = 0.5 * (gl[2] + gr[2]);
        double dy_ux = 0.5 * (hl[1] + hr[1]);
        double dy_uy = 0.5 * (hl[2] + hr[2]);
        double tauxx = mu * (dx_ux - dy_uy);
        double tauxy = mu * (dx_uy + dy_ux);
        F[0] = F[0] + 0.0;
        F[1] = F[1] + -tauxx;
        F[2] = F[2] + -tauxy;
    }
    else
    {
        double dx_ux = 0.5 * (hl[1] + hr[1]);
        double dx_uy = 0.5 * (hl[2] + hr[2]);
        double dy_ux = 0.5 * (gl[1] + gr[1]);
        double dy_uy = 0.5 * (gl[2] + gr[2]);
        double tauyx =  mu * (dx_uy + dy_ux);
        double tauyy = -mu * (dx_ux - dy_uy);
        F[0] = F[0] + 0.0;
        F[1] = F[1] + -tauyx;
        F[2] = F[2] + -tauyy;
    }
    if (P->angmom_form)
    {
        /* to_angmom_fluxes scheme.cpp:199-214 */
        double rd = P->domain_radius;
        double flux_sr = xf * F[1] + yf * F[2];
        double flux_lz = xf * F[2] - yf * F[1];
        if (axis == 0 && (xf == -rd || xf == rd)) flux_lz = 0.0;
        if (axis == 1 && (yf == -rd || yf == rd)) flux_lz = 0.0;
        F[1] = flux_sr;
        F[2] = flux_lz;
    }
}

/* arithmetic_binary_tree_t::sum core_tree.hpp:502 over sequence_t::sum core_sequence.hpp:216 */
static double fold_tree(const double* block_vals, int nb, int level, int depth, int bi, int bj)
{
    if (level == depth) return block_vals[bi * nb + bj];
    double r = 0.0;
    for (int c = 0; c < 4; ++c)
        r = r + fold_tree(block_vals, nb, level + 1, depth, bi * 2 + (c & 1), bj * 2 + ((c >> 1) & 1));
    return r;
}

static int tree_depth(int n, int bs)
{
    int depth = 0;
    while ((bs << depth) < n) ++depth;
    return depth;
}

static int binary_threads = 1;
void mo_binary_set_threads(int n) { binary_threads = n > 0 ? n : 1; }

int mo_binary_advance_u(const mo_binary_params* P, const double* xv, const double* yv, const double* u0, const double* u_init,
    const double* br, const double bodies[10], double dt, double* u1, double totals[MO_BINARY_NTOTALS])
{
    const int N = P->n, bs = P->block_size, nb = N / bs, depth = tree_depth(N, bs);
    const double th = P->plm_theta;
    const double spacing = 2.0 * P->domain_radius / bs / (1 << depth);
    const size_t ncell = (size_t) N * N;
#define AT(i, j) ((size_t) (((i) + N) % N) * N + (size_t) (((j) + N) % N))
    double* p  = malloc(ncell * 3 * sizeof(double));
    double* gx = malloc(ncell * 3 * sizeof(double));
    double* gy = malloc(ncell * 3 * sizeof(double));
    double* fx = malloc((size_t) (N + 1) * N * 3 * sizeof(double));
    double* fy = malloc((size_t) N * (N + 1) * 3 * sizeof(double));
    double* blk = calloc((size_t) MO_BINARY_NTOTALS * nb * nb, sizeof(double));

    /* threads (mo_binary_set_threads; 1 by default): every pass below writes disjoint cells / faces / blocks, and the totals are folded from the
     * per-block values in tree order afterwards, so the results do not depend on the thread count - the role of the reference's
     * tree.map(fn, pool) over blocks (core_tree.hpp:615-625, core_thread_pool.hpp:47-189) */
#pragma omp parallel for schedule(static) num_threads(binary_threads)
    for (int i = 0; i < N; ++i)
        for (int j = 0; j < N; ++j)
        {
            size_t n = AT(i, j);
            double x[2] = {(xv[i] + xv[i + 1]) * 0.5, (yv[j] + yv[j + 1]) * 0.5};
            if (P->angmom_form) mo_iso2d_recover_primitive_angmom(u0 + 3 * n, x, p + 3 * n);
            else                mo_iso2d_recover_primitive(u0 + 3 * n, p + 3 * n);
        }
#pragma omp parallel for schedule(static) num_threads(binary_threads)
    for (int i = 0; i < N; ++i)
        for (int j = 0; j < N; ++j)
            for (int q = 0; q < 3; ++q)
            {
                gx[3 * AT(i, j) + q] = mo_plm_gradient(p[3 * AT(i - 1, j) + q], p[3 * AT(i, j) + q], p[3 * AT(i + 1, j) + q], th) / spacing;
                gy[3 * AT(i, j) + q] = mo_plm_gradient(p[3 * AT(i, j - 1) + q], p[3 * AT(i, j) + q], p[3 * AT(i, j + 1) + q], th) / spacing;
            }
    /* block_fluxes_u :472-516 — a block's outer faces sit at its own vertices, so both sides of the periodic seam are evaluated */
#pragma omp parallel for schedule(static) num_threads(binary_threads)
    for (int i = 0; i <= N; ++i)
        for (int j = 0; j < N; ++j)
        {
            double xf = (xv[i] + xv[i]) * 0.5, yf = (yv[j] + yv[j + 1]) * 0.5, F[3];
            face_flux(P, 0, spacing, xf, yf, bodies, p + 3 * AT(i - 1, j), p + 3 * AT(i, j), gx + 3 * AT(i - 1, j), gx + 3 * AT(i, j), gy + 3 * AT(i - 1, j), gy + 3 * AT(i, j), F);
            double dy = yv[j + 1] - yv[j];
            for (int q = 0; q < 3; ++q) fx[((size_t) i * N + j) * 3 + q] = F[q] * dy;
        }
#pragma omp parallel for schedule(static) num_threads(binary_threads)
    for (int i = 0; i < N; ++i)
        for (int j = 0; j <= N; ++j)
        {
            double xf = (xv[i] + xv[i + 1]) * 0.5, yf = (yv[j] + yv[j]) * 0.5, F[3];
            face_flux(P, 1, spacing, xf, yf, bodies, p + 3 * AT(i, j - 1), p + 3 * AT(i, j), gy + 3 * AT(i, j - 1), gy + 3 * AT(i, j), gx + 3 * AT(i, j - 1), gx + 3 * AT(i, j), F);
            double dx = xv[i + 1] - xv[i];
            for (int q = 0; q < 3; ++q) fy[((size_t) i * (N + 1) + j) * 3 + q] = F[q] * dx;
        }

    const double rs2 = P->softening_radius * P->softening_radius;
    const double s2 = P->sink_radius * P->sink_radius;
    int negative = 0;
#pragma omp parallel for collapse(2) schedule(static) reduction(|:negative) num_threads(binary_threads)
    for (int bi = 0; bi < nb; ++bi)
        for (int bj = 0; bj < nb; ++bj)
        {
            double t[MO_BINARY_NTOTALS] = {0};
            double sink_sum[2][3] = {{0}};
            for (int i = bi * bs; i < (bi + 1) * bs; ++i)
                for (int j = bj * bs; j < (bj + 1) * bs; ++j)
                {
                    /* source_terms_u :345-411 */
                    const double xc = (xv[i] + xv[i + 1]) * 0.5, yc = (yv[j] + yv[j + 1]) * 0.5;
                    const double dA = (xv[i + 1] - xv[i]) * (yv[j + 1] - yv[j]);
                    const double* u = u0 + 3 * AT(i, j);
                    double s_grav[2][3], s_sink[2][3], fg[2][2], s_buffer[3], s_floor[3];
                    for (int b = 0; b < 2; ++b)
                    {
                        const double* body = bodies + 5 * b;
                        double d0 = xc - body[1], d1 = yc - body[2];
                        double den = pow(d0 * d0 + d1 * d1 + rs2, 1.5);
                        fg[b][0] = (-d0 / den * 1.0 * body[0]) * u[0];
                        fg[b][1] = (-d1 / den * 1.0 * body[0]) * u[0];
                        s_grav[b][0] = 0.0 * dt;
                        s_grav[b][1] = fg[b][0] * dt;
                        s_grav[b][2] = fg[b][1] * dt;
                        double a2 = (d0 * d0 + d1 * d1) / s2 / 2.0;
                        double rate = P->sink_rate * exp(-a2);
                        for (int q = 0; q < 3; ++q) s_sink[b][q] = -u[q] * rate * dt;
                    }
                    const double fl = (double) (u[0] < P->density_floor);
                    for (int q = 0; q < 3; ++q)
                    {
                        s_buffer[q] = (u_init[3 * AT(i, j) + q] - u[q]) * br[AT(i, j)] * dt;
                        s_floor[q] = u[q] * 1e-2 * fl;
                    }
                    double dps[2][3];
                    for (int b = 0; b < 2; ++b) for (int q = 0; q < 3; ++q) dps[b][q] = s_sink[b][q];
                    if (P->angmom_form)
                    {
                        /* source_terms_q :417-466 */
                        double sr2 = pow(P->gst_suppr_radius, 2.0);
                        double ramp = 1.0 - exp(-(xc * xc + yc * yc) / sr2);
                        const double* pc = p + 3 * AT(i, j);
                        double cs2 = mo_binary_cs2(P, xc, yc, bodies);
                        double Ek = 0.5 * pc[0] * (pc[1] * pc[1] + pc[2] * pc[2]);      /* source_terms_conserved_angmom physics_iso2d.hpp:277-285 */
                        double pg = pc[0] * cs2;
                        double sg[3] = {0.0, (Ek + pg) * 2.0, 0.0};
                        for (int b = 0; b < 2; ++b)
                        {
                            s_grav[b][1] = (xc * fg[b][0] + yc * fg[b][1]) * dt;
                            s_grav[b][2] = (xc * fg[b][1] - yc * fg[b][0]) * dt;
                            double r2 = 0.0 + xc * xc + yc * yc;                       /* to_conserved_per_area(Q, x) physics_iso2d.hpp:404-414 */
                            dps[b][1] = (s_sink[b][1] * xc - s_sink[b][2] * yc) / r2;
                            dps[b][2] = (s_sink[b][1] * yc + s_sink[b][2] * xc) / r2;
                        }
                        for (int q = 0; q < 3; ++q) s_floor[q] = sg[q] * ramp * dt;    /* s_geom takes the floor term's place in the sum */
                    }
#define LZ(s) (P->angmom_form ? (s)[2] : (xc * (s)[2] - yc * (s)[1]))   /* iso2d::angular_momentum physics_iso2d.hpp:444-447, or component 2 */
                    for (int b = 0; b < 2; ++b)
                    {
                        t[MO_T_MASS_ACC + b] = t[MO_T_MASS_ACC + b] + s_sink[b][0] * dA;
                        t[MO_T_L_ACC + b]    = t[MO_T_L_ACC + b] + LZ(s_sink[b]) * dA;
                        t[MO_T_TORQUE + b]   = t[MO_T_TORQUE + b] + LZ(s_grav[b]) * dA;
                        t[MO_T_FX + b]       = t[MO_T_FX + b] + fg[b][0] * dt * dA;
                        t[MO_T_FY + b]       = t[MO_T_FY + b] + fg[b][1] * dt * dA;
                        t[MO_T_PX_ACC + b]   = t[MO_T_PX_ACC + b] + dps[b][1] * dA;
                        t[MO_T_PY_ACC + b]   = t[MO_T_PY_ACC + b] + dps[b][2] * dA;
                        for (int q = 0; q < 3; ++q) sink_sum[b][q] = sink_sum[b][q] + s_sink[b][q] * dA;
                    }
                    t[MO_T_L_EJ]    = t[MO_T_L_EJ] + LZ(s_buffer) * dA;
                    t[MO_T_MASS_EJ] = t[MO_T_MASS_EJ] + s_buffer[0] * dA;
#undef LZ
                    /* block_update_u :568-587 */
                    for (int q = 0; q < 3; ++q)
                    {
                        double lx = fx[((size_t) (i + 1) * N + j) * 3 + q] - fx[((size_t) i * N + j) * 3 + q];
                        double ly = fy[((size_t) i * (N + 1) + j + 1) * 3 + q] - fy[((size_t) i * (N + 1) + j) * 3 + q];
                        double s = s_grav[0][q] + s_grav[1][q] + s_sink[0][q] + s_sink[1][q] + s_buffer[q] + s_floor[q];
                        u1[3 * AT(i, j) + q] = u[q] - (lx + ly) * dt / dA + s;
                    }
                    if (u1[3 * AT(i, j)] < 0.0) negative = 1;       /* validate_u :726-752 */
                }
            for (int k = 0; k < MO_BINARY_NTOTALS; ++k) t[k] = -t[k];
            for (int b = 0; b < 2; ++b)
            {
                /* work :356-365 with du = -(sum of s_sink dA) over THIS block */
                const double* body = bodies + 5 * b;
                double M0 = body[0], px0 = body[3] * M0, py0 = body[4] * M0;
                double M1 = M0 + -sink_sum[b][0], px1 = px0 + -sink_sum[b][1], py1 = py0 + -sink_sum[b][2];
                t[MO_T_WORK + b] = ((px1 * px1 + py1 * py1) / M1 - (px0 * px0 + py0 * py0) / M0) * 0.5;
                if (P->angmom_form) t[MO_T_WORK + b] = 0.0;    /* source_terms_q never sets work_done_on */
            }
            for (int k = 0; k < MO_BINARY_NTOTALS; ++k) blk[(size_t) k * nb * nb + bi * nb + bj] = t[k];
        }
    for (int k = 0; k < MO_BINARY_NTOTALS; ++k)
        totals[k] = fold_tree(blk + (size_t) k * nb * nb, nb, 0, depth, 0, 0);
#undef AT
    free(p); free(gx); free(gy); free(fx); free(fy); free(blk);
    return negative;
}

/* maximum_timestep scheme.cpp:1107-1126 : min over blocks of spacing / max over cells of max_wavespeed (physics_iso2d.hpp:330-337) */
double mo_binary_maximum_timestep(const mo_binary_params* P, const double* xv, const double* yv, const double* u, const double bodies[10])
{
    const int N = P->n, bs = P->block_size, nb = N / bs, depth = tree_depth(N, bs);
    const double spacing = 2.0 * P->domain_radius / bs / (1 << depth);
    double result = 0.0;
    for (int bi = 0; bi < nb; ++bi)
        for (int bj = 0; bj < nb; ++bj)
        {
            double a = 0.0;
            for (int i = bi * bs; i < (bi + 1) * bs; ++i)
                for (int j = bj * bs; j < (bj + 1) * bs; ++j)
                {
                    double xc = (xv[i] + xv[i + 1]) * 0.5, yc = (yv[j] + yv[j + 1]) * 0.5, prim[3], lx[3], ly[3];
                    double xx[2] = {xc, yc};
                    if (P->angmom_form) mo_iso2d_recover_primitive_angmom(u + 3 * ((size_t) i * N + j), xx, prim);
                    else                mo_iso2d_recover_primitive(u + 3 * ((size_t) i * N + j), prim);
                    double cs2 = mo_binary_cs2(P, xc, yc, bodies);
                    mo_iso2d_wavespeeds(prim, 0, cs2, lx);
                    mo_iso2d_wavespeeds(prim, 1, cs2, ly);
                    double vx = fmax(fabs(lx[0]), fabs(lx[1])), vy = fmax(fabs(ly[0]), fabs(ly[1]));
                    double w = vx < vy ? vy : vx;
                    a = (i == bi * bs && j == bj * bs) ? w : (a < w ? w : a);
                }
            double d = spacing / a;
            result = (bi == 0 && bj == 0) ? d : (d < result ? d : result);
        }
    return result;
}

/* create_vertex_quadtree mesh_tree_operators.hpp:158-190 for an always-true predicate: linspace(-1, 1, bs + 1)
 * (core_ndarray.hpp:2544-2551), `depth` rounds of prolong_verts (mesh_prolong_restrict.hpp:148-159: the mean of the two
 * nearest coarse vertices, which for an existing vertex is (v + v) * 0.5), then * domain_radius (subprog_binary.cpp:179-184) */
void mo_binary_vertices(int block_size, int depth, double domain_radius, double* v)
{
    int n = block_size;
    double* a = malloc(((size_t) (block_size << depth) + 1) * sizeof(double));
    double* b = malloc(((size_t) (block_size << depth) + 1) * sizeof(double));
    for (int i = 0; i <= n; ++i) a[i] = -1.0 + (1.0 - -1.0) * i / (double) ((n + 1) - 1);
    for (int l = 0; l < depth; ++l)
    {
        for (int i = 0; i <= 2 * n; ++i)
            b[i] = (a[i / 2] + a[(i + 1) / 2]) * 0.5;
        n *= 2;
        double* t = a; a = b; b = t;
    }
    for (int i = 0; i <= n; ++i) v[i] = a[i] * domain_radius;
    free(a); free(b);
}

/* create_disk_profile subprog_binary.cpp:105-153 */
void mo_binary_disk_profile(const mo_binary_model* m, double x, double y, double prim[3])
{
    double rs = m->softening_radius, rc = m->disk_radius, Ma = m->mach_number;
    double s0 = m->disk_mass / (17.0618 * rc * rc);
    double s1 = m->ambient_density * s0;
    double r2 = x * x + y * y;
    double r = sqrt(r2);
#define SIGMA(r_) (s0 * exp(-0.5 * ((r_) / rc - 1) * ((r_) / rc - 1)) + s1)
    double q = r / rc;
    double dp_dr = (1.0 / Ma / Ma / (r + rs)) * (q * (1 - q) * (1 - s1 / SIGMA(r)) - 1.0);
    double vp = sqrt(1.0 / (r + rs) + dp_dr) * (m->counter_rotate ? -1 : 1);
    double vr = -m->mdot / (SIGMA(r) * 2 * M_PI * r) * (r > 2.0);
    prim[0] = SIGMA(r);
    prim[1] = vr * (x / r) + vp * (-y / r);
    prim[2] = vr * (y / r) + vp * ( x / r);
#undef SIGMA
}

/* create_solver_data solver_data.cpp:20-102 : initial conserved field, buffer-rate field, recommended time step */
double mo_binary_solver_data(const mo_binary_model* m, int n, const double* xv, const double* yv, double* u_init, double* br)
{
    double min_dx = xv[1] - xv[0], min_dy = yv[1] - yv[0], max_v = 1.0;
    for (int i = 0; i < n; ++i)
    {
        if (xv[i + 1] - xv[i] < min_dx) min_dx = xv[i + 1] - xv[i];
        if (yv[i + 1] - yv[i] < min_dy) min_dy = yv[i + 1] - yv[i];
    }
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j)
        {
            double xc = (xv[i] + xv[i + 1]) * 0.5, yc = (yv[j] + yv[j + 1]) * 0.5, prim[3];
            mo_binary_disk_profile(m, xc, yc, prim);
            double xx[2] = {xc, yc};
            if (m->angmom_form) mo_iso2d_to_conserved_angmom(prim, xx, u_init + 3 * ((size_t) i * n + j));
            else                mo_iso2d_to_conserved(prim, u_init + 3 * ((size_t) i * n + j));
            double v = sqrt(prim[1] * prim[1] + prim[2] * prim[2]);
            if (max_v < v) max_v = v;
            double rc = pow(xc * xc + yc * yc, 0.5);
            br[(size_t) i * n + j] = m->buffer_damping_rate * (1.0 + tanh(3.0 * (rc - m->domain_radius)));
        }
    return (min_dx < min_dy ? min_dx : min_dy) / max_v * m->cfl_number;
}


/* ---- diagnostics: subprog_binary_diagnostics.cpp:21-82 on a block tree ------------------------------
 * blocks[nb][3] = (level, i, j) in tree order (depth first, children in orthant order), edges[nb][2][bs + 1] = x of the block's
 * vertex rows then y of its vertex columns, u[nb][bs][bs][3] conserved (or angular-momentum-form) per area.
 * totals = {disk_mass, disk_angular_momentum}: block sums are nd::sum() (sequential from 0, row-major, core_ndarray.hpp:1882-1898),
 * combined as arithmetic_binary_tree_t::sum() combines them: recursively, each node 0 + c0 + c1 + c2 + c3 (core_tree.hpp:502,
 * core_sequence.hpp:216-224). fields[nb][3][bs][bs] = sigma, radial_velocity, phi_velocity (:60-71; r = std::pow(r2, 0.5)). */
typedef struct { const int* blocks; const double* partial; int nb, next; } tree_sum_t;

static double tree_sum(tree_sum_t* T, int level, int i, int j)
{
    if (T->next < T->nb)
    {
        const int* b = T->blocks + 3 * T->next;
        if (b[0] == level && b[1] == i && b[2] == j) return T->partial[T->next++];
    }
    double result = 0.0;
    for (int c = 0; c < 4; ++c) result = result + tree_sum(T, level + 1, 2 * i + (c & 1), 2 * j + ((c >> 1) & 1));
    return result;
}

void mo_binary_diagnostics(int angmom_form, int bs, int nb, const int* blocks, const double* edges, const double* u, double totals[2], double* fields)
{
    double* mass = (double*) malloc(sizeof(double) * (size_t) nb);
    double* lz = (double*) malloc(sizeof(double) * (size_t) nb);
    for (int k = 0; k < nb; ++k)
    {
        const double* xe = edges + (size_t) k * 2 * (bs + 1);
        const double* ye = xe + bs + 1;
        double m = 0.0, l = 0.0;
        for (int i = 0; i < bs; ++i)
            for (int j = 0; j < bs; ++j)
            {
                const double* U = u + 3 * (((size_t) k * bs + i) * bs + j);
                const double xc = (xe[i] + xe[i + 1]) * 0.5, yc = (ye[j] + ye[j + 1]) * 0.5;      /* midpoint of midpoints of a tensor-product block: exact */
                const double dA = (xe[i + 1] - xe[i]) * (ye[j + 1] - ye[j]);
                const double x[2] = {xc, yc};
                double P[3];
                m = m + U[0] * dA;
                if (angmom_form)
                {
                    l = l + U[2] * dA;
                    mo_iso2d_recover_primitive_angmom(U, x, P);
                }
                else
                {
                    l = l + (xc * U[2] - yc * U[1]) * dA;                 /* iso2d::angular_momentum physics_iso2d.hpp:444-447 */
                    mo_iso2d_recover_primitive(U, P);
                }
                if (fields)
                {
                    const double rc = pow(xc * xc + yc * yc, 0.5);
                    const double rhat_x = xc / rc, rhat_y = yc / rc, phat_x = -yc / rc, phat_y = xc / rc;
                    double* F = fields + (size_t) k * 3 * bs * bs + (size_t) i * bs + j;
                    F[0] = P[0];
                    F[(size_t) bs * bs] = P[1] * rhat_x + P[2] * rhat_y;
                    F[(size_t) 2 * bs * bs] = P[1] * phat_x + P[2] * phat_y;
                }
            }
        mass[k] = m;
        lz[k] = l;
    }
    tree_sum_t T = {blocks, mass, nb, 0};
    totals[0] = tree_sum(&T, 0, 0, 0);
    T.partial = lz; T.next = 0;
    totals[1] = tree_sum(&T, 0, 0, 0);
    free(mass);
    free(lz);
}
