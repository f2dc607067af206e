/*
 * mara_oracle.h — TEST INFRASTRUCTURE. Not part of the product.
 *
 * Plain-C CPU restatement of the Mara3 per-step hot path, kept
 * deliberately simple: flat loops over row-major array-of-structs fields, the
 * same operation order as the reference expressions so that results are
 * bit-identical to the reference compiled without FMA contraction.
 *
 * Parity status: PINNED — every function here is checked against outputs of
 * the reference's own headers run in the build container (oracle/_ref drivers
 * -> tests/golden/ (npz files); generator: oracle/gen_golden.py). Exceptions, which
 * have no upstream counterpart and say so at their declaration: the Euler HLLC
 * solver ("parity unpinned").
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library.
 */
#ifndef MARA_ORACLE_H
#define MARA_ORACLE_H
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { MO_BC_OUTFLOW = 0, MO_BC_PERIODIC = 1 };
enum { MO_RIEMANN_HLLE = 0, MO_RIEMANN_HLLC = 1 };

/* ---- per-cell / per-face functions ------------------------------------- */

/* math_interpolation.hpp:85-94 */
double mo_plm_gradient(double yl, double y0, double yr, double theta);

/* physics_euler.hpp:555-575 */
void mo_euler_recover_primitive(const double U[5], double gamma, double temperature_floor, double P[5]);
/* physics_euler.hpp:209-220 */
void mo_euler_to_conserved_density(const double P[5], double gamma, double U[5]);
/* physics_euler.hpp:252-263 (axis selects unit_vector_t::on_axis, core_geometric.hpp:65-74) */
void mo_euler_flux(const double P[5], const double U[5], int axis, double F[5]);
/* physics_euler.hpp:276-284 ; lam[0] = minus, lam[1] = plus */
void mo_euler_wavespeeds(const double P[5], int axis, double gamma, double lam[2]);
/* physics_euler.hpp:614-631 */
void mo_euler_riemann_hlle(const double Pl[5], const double Pr[5], int axis, double gamma, double F[5]);
/* NO UPSTREAM COUNTERPART (parity unpinned): gamma-law generalisation of
 * physics_iso2d.hpp:556-583,610-687 per Toro 3rd ed. eq. 10.61-10.73 */
void mo_euler_riemann_hllc(const double Pl[5], const double Pr[5], int axis, double gamma, double F[5]);
/* physics_euler.hpp:328-337 */
void mo_euler_source_terms_radial(const double P[5], double r, double S[5]);

/* array forms for the golden-vector tests (n independent items) */
void mo_plm_gradient_n(size_t n, const double* yl, const double* y0, const double* yr, double theta, double* g);
void mo_euler_recover_primitive_n(size_t n, const double* U, double gamma, double tfloor, double* P);
void mo_euler_to_conserved_density_n(size_t n, const double* P, double gamma, double* U);
void mo_euler_riemann_n(size_t n, const double* Pl, const double* Pr, int axis, double gamma, int solver, double* F);

/* ---- uniform cartesian Euler step (BASELINE configs 2 and 5) ------------ */

typedef struct
{
    int    rank;            /* 1, 2 or 3 */
    size_t shape[3];        /* cells per axis (unused axes = 1) */
    double dl[3];           /* cell size per axis */
    double gamma;
    double plm_theta;       /* < 0 => piecewise constant */
    int    riemann;         /* MO_RIEMANN_* */
    int    bc;              /* MO_BC_* (same on every axis) */
    int    rk_order;        /* 1 or 2 */
    int    nthreads;        /* axis-0 slabs, app_parallel.hpp:75-103 */
} mo_euler_cart_t;

/* One stage: u1 = u0 - sum_axis diff(F)*(dt/dl). Fields: AoS [n0][n1][n2][5]. */
int mo_euler_cart_advance(const mo_euler_cart_t* cfg, const double* u0, double dt, double* u1);
/* nsteps full steps in place (RK1: advance; RK2: u*0.5 + advance(advance(u))*0.5,
 * subprog_cloud.cpp:682-695). */
int mo_euler_cart_run(const mo_euler_cart_t* cfg, double* u, double dt, int nsteps);

/* ---- sedov: 1-D spherical Euler, PCM + HLLE + forward Euler ------------- */
/* subprog_sedov.cpp:353-421. vertices[nz+1]; u = volume-integrated conserved AoS [nz][5]. */
void mo_sedov_vertices(int nr, double outer_radius, size_t* nz_out, double* vertices /* may be NULL to query nz */);
void mo_sedov_initial(size_t nz, const double* vertices, double gamma, double explosion_density,
                      double explosion_pressure, double density_index, double* u);
double mo_sedov_timestep(const double* vertices, double cfl);
void mo_sedov_advance(size_t nz, const double* vertices, double gamma, double dt, const double* u0, double* u1);
/* HydroSystem = mara::srhd (the sub-program's default system) */
void mo_sedov_initial_system(int srhd, size_t nz, const double* vertices, double gamma, double explosion_density,
                             double explosion_pressure, double density_index, double* u);
/* make_diagnostic_fields / compute_time_series_data (subprog_sedov.cpp:252-308): fields [4][nz], indices[3], series[6]; see the .c file */
int  mo_sedov_diagnostics(int srhd, size_t nz, const double* vertices, const double* u, double gamma, double time, double* fields, int* indices, double* series);
int  mo_sedov_advance_srhd(size_t nz, const double* vertices, double gamma, double dt, const double* u0, double* u1);

/* ---- mara::srhd and the `cloud` stage (mara_oracle_srhd.c) ------------------ */
enum { MO_C2P_NOT_CONVERGED = 4, MO_C2P_NEG_DENSITY = 1, MO_C2P_NEG_PRESSURE = 2, MO_C2P_NAN = 8 };
/* physics_srhd.hpp:364-451 ; returns failure bits where the reference throws */
int  mo_srhd_recover_primitive(const double U[5], double gamma, double temperature_floor, double P[5]);
/* physics_srhd.hpp:213-227 */
void mo_srhd_to_conserved_density(const double P[5], double gamma, double U[5]);
/* physics_srhd.hpp:259-270 */
void mo_srhd_flux(const double P[5], const double U[5], int axis, double F[5]);
/* physics_srhd.hpp:283-295 */
void mo_srhd_wavespeeds(const double P[5], int axis, double gamma, double lam[2]);
/* physics_srhd.hpp:466-483 */
void mo_srhd_riemann_hlle(const double Pl[5], const double Pr[5], int axis, double gamma, double F[5]);
/* physics_srhd.hpp:309-326 */
void mo_srhd_source_terms(const double P[5], double r, double theta, double gamma, double S[5]);
void mo_srhd_recover_primitive_n(size_t n, const double* U, double gamma, double tfloor, double* P, int* status);
void mo_srhd_to_conserved_density_n(size_t n, const double* P, double gamma, double* U);
void mo_srhd_riemann_hlle_n(size_t n, const double* Pl, const double* Pr, int axis, double gamma, double* F);
void mo_srhd_source_terms_n(size_t n, const double* P, const double* r, const double* theta, double gamma, double* S);
/* subprog_cloud.cpp:260-290 : dAr[(nr+1)*nq], dAq[nr*(nq+1)], dv[nr*nq] */
void mo_cloud_geometry(size_t nr, size_t nq, const double* rv, const double* qv, double* dAr, double* dAq, double* dv);
/* subprog_cloud.cpp:511-584 ; u = cell-integrated conserved AoS [nr][nq][5]; inflow [nq][5] primitives */
int  mo_cloud_advance(size_t nr, size_t nq, const double* rv, const double* qv, const double* inflow,
                      double gamma, double plm_theta, double temperature_floor, double dt, const double* u0, double* u1);
/* subprog_cloud.cpp:334-433 make_diagnostic_fields; units = {length, mass, time}; fields [5][nr][nq], columns [15][nq] (see the .c file) */
int  mo_cloud_diagnostics(size_t nr, size_t nq, const double* rv, const double* qv, const double* u, double gamma, double tfloor,
                          const double units[3], double* fields, double* columns);
/* subprog_cloud.cpp:676-697 ; inflow [nsteps][nq][5] */
int  mo_cloud_run(size_t nr, size_t nq, const double* rv, const double* qv, const double* inflow, double gamma,
                  double plm_theta, double temperature_floor, int rk_order, double dt, int nsteps, double* u);

/* ---- mara::iso2d (mara_oracle_iso2d.c); component order (Sigma, x, y) ------------ */
void mo_iso2d_to_conserved(const double P[3], double U[3]);                                   /* physics_iso2d.hpp:249-258 */
int  mo_iso2d_recover_primitive(const double U[3], double P[3]);                              /* :351-362 */
void mo_iso2d_to_conserved_angmom(const double P[3], const double x[2], double Q[3]);         /* :263-272 */
int  mo_iso2d_recover_primitive_angmom(const double Q[3], const double x[2], double P[3]);    /* :376-390 */
void mo_iso2d_flux(const double P[3], int axis, double cs2, double F[3]);                     /* :299-307 */
void mo_iso2d_wavespeeds(const double P[3], int axis, double cs2, double lam[3]);             /* :320-337 */
void mo_iso2d_riemann_hlle(const double Pl[3], const double Pr[3], double cs2l, double cs2r, int axis, double F[3]);   /* :488-506 */
int  mo_iso2d_riemann_hllc(const double Pl[3], const double Pr[3], double cs2l, double cs2r, int axis, double F[3], double* contact); /* :556-583,:610-712 */
void mo_iso2d_to_conserved_n(size_t n, const double* P, double* U);
void mo_iso2d_recover_primitive_n(size_t n, const double* U, double* P, int* threw);
void mo_iso2d_to_conserved_angmom_n(size_t n, const double* P, const double* x, double* Q);
void mo_iso2d_recover_primitive_angmom_n(size_t n, const double* Q, const double* x, double* P, int* threw);
void mo_iso2d_flux_n(size_t n, const double* P, const double* cs2, int axis, double* F);
void mo_iso2d_wavespeeds_n(size_t n, const double* P, const double* cs2, int axis, double* lam);
void mo_iso2d_riemann_n(size_t n, const double* Pl, const double* Pr, const double* cs2l, const double* cs2r, int axis,
                        int solver, double* F, double* contact, int* threw);

/* ---- integer / index work (bit-exact) ----------------------------------- */
/* core_ndarray.hpp:820-836 : slab n of N over `count` rows -> [start, final) */
void mo_partition_rows(size_t count, size_t nparts, size_t part, size_t* start, size_t* final_);
/* app_parallel.hpp:185-221 ; returns number of factors written (<= 64) */
int  mo_prime_factors(unsigned long n, unsigned long* factors);
/* app_parallel.hpp:119-131 */
void mo_propose_block_decomposition(int rank, unsigned long nblocks, unsigned long* blocks_per_axis);
/* app_parallel.hpp:148-179 : block b of B along an axis with n cells -> [start, final) */
void mo_block_extent(size_t n, size_t nblocks, size_t b, size_t* start, size_t* final_);

/* ---- circumbinary disk scheme (mara_oracle_binary.c; BASELINE config 3) ---- */
enum { MO_T_MASS_ACC = 0, MO_T_L_ACC = 2, MO_T_TORQUE = 4, MO_T_PX_ACC = 6, MO_T_PY_ACC = 8, MO_T_FX = 10, MO_T_FY = 12,
       MO_T_WORK = 14, MO_T_MASS_EJ = 16, MO_T_L_EJ = 17, MO_BINARY_NTOTALS = 18 };   /* source_term_total_t scheme.cpp:17-32; [2] = per body */
typedef struct
{
    int    n;                   /* cells per side = block_size << depth */
    int    block_size;
    double domain_radius, mach_number, alpha, nu, alpha_cutoff_radius;
    double sink_rate, sink_radius, softening_radius;
    double density_floor;       /* absolute surface density (config density_floor * disk_mass, solver_data.cpp:100) */
    double plm_theta;           /* 0 in safe mode (scheme.cpp:792) */
    int    axisymmetric_cs2;
    int    angmom_form;         /* conserve_linear_p == 0: fields are (Sigma, Sigma s_r, Sigma l_z) and the stage is advance_q */
    double gst_suppr_radius;    /* source_term_softening * min(dx, dy) (solver_data.cpp:91); used by advance_q only */
} mo_binary_params;
typedef struct
{
    double softening_radius, disk_radius, mach_number, disk_mass, ambient_density, mdot;
    int    counter_rotate;
    double buffer_damping_rate, domain_radius, cfl_number;
    int    angmom_form;
} mo_binary_model;
/* bodies = (mass, x, y, vx, vy) of body 1 then body 2; fields are [n][n][3] row-major in (Sigma, px, py); returns 1 where validate_u throws */
/* threads of mo_binary_advance_u (OpenMP over rows / blocks - the role of the reference's tree.map(fn, pool)); results do not depend on it */
void   mo_binary_set_threads(int n);
int    mo_binary_advance_u(const mo_binary_params* P, const double* xv, const double* yv, const double* u0, const double* u_init,
                           const double* br, const double bodies[10], double dt, double* u1, double totals[MO_BINARY_NTOTALS]);
double mo_binary_maximum_timestep(const mo_binary_params* P, const double* xv, const double* yv, const double* u, const double bodies[10]);
double mo_binary_cs2(const mo_binary_params* P, double x, double y, const double bodies[10]);
double mo_binary_nu(const mo_binary_params* P, double x, double y, double cs2);
void   mo_binary_vertices(int block_size, int depth, double domain_radius, double* v);   /* v[(block_size << depth) + 1] */
void   mo_binary_disk_profile(const mo_binary_model* m, double x, double y, double prim[3]);
double mo_binary_solver_data(const mo_binary_model* m, int n, const double* xv, const double* yv, double* u_init, double* br);  /* returns recommended_time_step */
/* binary::disk_mass, disk_angular_momentum, diagnostic_fields (subprog_binary_diagnostics.cpp:21-82) on a block tree; see the .c file */
void   mo_binary_diagnostics(int angmom_form, int bs, int nb, const int* blocks, const double* edges, const double* u, double totals[2], double* fields);

#ifdef __cplusplus
}
#endif
#endif
