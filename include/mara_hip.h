/*
 * mara_hip.h — C ABI of libmara_hip.so, the MI355X (gfx950) engine for the
 * per-time-step hot path of Mara3-style finite-volume Godunov hydro.
 *
 * What this boundary replaces. The reference (jzrake/Mara3) has no FFI; its
 * innermost seam is the pure step function + array evaluator
 *     advance(app_state, solution, dt) -> solution_t      src/subprog_cloud.cpp:511-584
 *     next_solution(state) -> state                        src/subprog_cloud.cpp:676-697, src/subprog_sedov.cpp:394-421
 *     `| mara::evaluate_on<N>()` / `| nd::to_shared()`     src/app_parallel.hpp:72-103
 * Each entry point below cites the reference interface it stands in for.
 * Everything is plain pointers and sizes; no C++ or torch types cross the ABI.
 *
 * Conventions
 *  - All floating point is fp64. Return value 0 = success, negative = MH_E_*;
 *    nothing throws across the ABI; mh_last_error() gives the message.
 *  - HOST fields at the boundary are row-major array-of-structs exactly like
 *    the reference's nd::shared_array<conserved_t, Rank>: cell (i,j[,k]) is
 *    nq consecutive doubles in the logical order (rho|D, m1, m2, m3, E|tau)
 *    (src/core_ndarray.hpp:777-792, src/physics_euler.hpp:46).
 *  - DEVICE fields are row-interleaved struct-of-arrays with two ghost rows on
 *    each side of axis 0 (the slab axis, src/core_ndarray.hpp:820-836):
 *        row i in [-2, n0+2), variable q, transverse index t:  ((i+2)*nq + q)*row_pitch + t
 *    where row_pitch = n1 (2-D) or n1*n2 (3-D). Every variable of a row is a
 *    contiguous, coalescable run of row_pitch doubles, and the two ghost rows
 *    of a side (all variables) are ONE contiguous block of 2*nq*row_pitch
 *    doubles - one message per neighbour in a slab decomposition.
 *    Ghost rows hold the boundary condition (outflow copies / periodic wrap)
 *    or the neighbour rank's rows; the stage kernels keep physical ghosts up
 *    to date themselves.
 *  - `stream` arguments are hipStream_t passed as void* (NULL = default stream).
 */
#ifndef MARA_HIP_H
#define MARA_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MH_VERSION 100

enum mh_error
{
    MH_OK = 0,
    MH_E_INVALID = -1,       /* bad argument / unsupported combination */
    MH_E_HIP = -2,           /* HIP runtime error (see mh_last_error) */
    MH_E_NOMEM = -3,
    MH_E_STATE = -4,         /* call order (e.g. step before upload) */
    MH_E_PHYSICS = -5        /* status word non-zero after a step (negative density/pressure, NaN) */
};

enum mh_system   { MH_SYSTEM_EULER = 0, MH_SYSTEM_ISO2D = 1, MH_SYSTEM_SRHD = 2 };
enum mh_riemann  { MH_RIEMANN_HLLE = 0, MH_RIEMANN_HLLC = 1 };
/* boundary kinds. On axis 0 each side is set separately because a slab cut is
 * MH_BC_EXTERNAL (ghost rows are written by the caller's halo exchange). */
enum mh_bc       { MH_BC_OUTFLOW = 0, MH_BC_PERIODIC = 1, MH_BC_EXTERNAL = 2, MH_BC_REFLECT = 3, MH_BC_INFLOW = 4 };
/* arithmetic contract: STRICT = no FMA contraction, IEEE division/sqrt: bit-identical
 * to the reference built for baseline x86-64. FAST = contraction allowed and shared
 * reciprocals; validated to the 1e-12 L1 bound of BASELINE.json, not bit-exact. */
enum mh_arith    { MH_ARITH_STRICT = 0, MH_ARITH_FAST = 1 };

/* status bits accumulated on the device by a stage (reference error contract:
 * src/physics_srhd.hpp:430-449, src/subprog_binary_scheme.cpp:726-752) */
enum mh_status
{
    MH_STATUS_NEG_DENSITY = 1, MH_STATUS_NEG_PRESSURE = 2, MH_STATUS_C2P_FAILED = 4, MH_STATUS_NAN = 8
};
/* What a step reports back: the OR of the mh_status bits raised by any cell, and the flat index (row-major, the
 * order of the host array handed to mh_upload / mh_slab_upload / mh_binary_set_solution) of the FIRST cell that raised one;
 * UINT64_MAX when status == 0. The Euler kernels raise NEG_DENSITY (updated density <= 0), NEG_PRESSURE (recovered pressure < 0)
 * and NAN (either is NaN); the SRHD kernels raise the bit of each `throw` of mara::srhd::recover_primitive
 * (src/physics_srhd.hpp:430-449); `binary` raises NEG_DENSITY where validate_u throws (src/subprog_binary_scheme.cpp:726-752).
 * On the device this is int32[2] = {bits, 0xFFFFFFFF - first local flat index (0: none)}, both order-independent atomics. */
typedef struct
{
    int32_t  status;
    int32_t  reserved;
    uint64_t first_bad_index;
} mh_step_result;

/* ------------------------------------------------------------------------ */
/* Uniform cartesian Euler (BASELINE configs 2 and 5): stateless launchers.   */
/* Replaces one `advance` evaluation of the composition in                    */
/* src/subprog_cloud.cpp:511-584 specialised to mara::euler on a cartesian    */
/* grid: recover_primitive (src/physics_euler.hpp:555-575), plm_gradient      */
/* (src/math_interpolation.hpp:85-94), riemann_hlle (src/physics_euler.hpp:   */
/* 614-631), flux difference, and the RK combine of                           */
/* src/subprog_cloud.cpp:682-695 fused into the second stage.                 */
/* ------------------------------------------------------------------------ */
typedef struct
{
    int    rank;            /* 2 or 3 */
    int    n[3];            /* LOCAL cells per axis on this device (axis 0 = slab axis) */
    double dl[3];           /* cell sizes */
    double gamma;           /* gamma-law index */
    double plm_theta;       /* < 0: piecewise constant */
    int    riemann;         /* enum mh_riemann */
    int    bc_lo0, bc_hi0;  /* enum mh_bc on the low / high side of axis 0 */
    int    bc_transverse;   /* MH_BC_OUTFLOW or MH_BC_PERIODIC on axes 1 (and 2) */
    int    arith;           /* enum mh_arith */
    int    chunk_rows;      /* rows marched per wave (0 = default) */
    int    tail_rows;       /* graded tail: the LAST tail_rows rows of a launch go to short waves of tail_chunk_rows rows */
    int    tail_chunk_rows; /* (0, 0 = the measured default on large grids; tail_rows < 0 = off). Read at configure time. */
    int    fuse_stages;     /* RK2 step of a whole 2-D field as ONE launch (mara3_amd/csrc/euler2d_fused.hip: the first-stage field stays in LDS,
                             * 80 instead of 200 B per zone-update): 0 = where available and worth it (MH_ARITH_FAST, PLM, rk_order 2: the context
                             * stepper and slabs without neighbours; slabs WITH neighbours from 384 rows per slab on - they then keep four
                             * rows of each neighbour and exchange once per step), < 0 = never, > 0 = required (configure fails
                             * where it is not available). The result is bit-identical to the two launches'. */
    int    planar;          /* rank-2 fields whose THIRD momentum is identically zero - every 2-D run of mara::euler's five-component state, which
                             * the reference carries as zeros through recover_primitive, the slopes, both Riemann problems and the update
                             * (src/physics_euler.hpp:209-220, :252-263, :555-575). The PLM kernels (the one-launch FAST step, and the stage kernels
                             * of both arithmetic modes) then neither read nor compute that component and write it as zero: the other four
                             * components keep their bits (x + 0 and fma(0, 0, x) are x), ~11 % fewer instructions. MH_ARITH_STRICT takes it only on
                             * the BIT PATTERN of +0.0 everywhere, for which the reference's own operations return +0.0: still bit-identical. 0 = where the stepper has VERIFIED it: the context and the slab steppers look at
                             * the uploaded field (one pass at upload, not per step; a field with a third momentum takes the general kernel),
                             * loopback groups at all members' rows; slabs that exchange with OTHER PROCESSES - and contexts with an MH_BC_EXTERNAL
                             * side, whose ghost rows the caller writes through mh_field_ptr - only under > 0 = the caller asserts
                             * it for the whole grid (each rank still verifies its own rows: upload fails otherwise); < 0 = never. */
} mh_euler_cart_desc;

/* number of doubles one device field of this description occupies: (n0+4) * 5 * row_pitch */
size_t mh_euler_cart_field_doubles(const mh_euler_cart_desc* d);

/*
 * One Runge-Kutta stage over rows [row_begin, row_end) of axis 0:
 *     stage_weight == 1.0 :  u_out = u_in - sum_axis diff(F)*(dt/dl)                      (RK1 / first RK2 stage)
 *     otherwise           :  u_out = u_base*(1-w) + (u_in - sum_axis diff(F)*(dt/dl))*w    (w = 0.5: RK2 combine,
 *                            src/subprog_cloud.cpp:693 `s0*0.5 + s2*0.5`; u_out may alias u_base)
 * All pointers are DEVICE SoA fields (layout above). Ghost rows of u_in must be valid.
 * Physical ghost rows of u_out (bc OUTFLOW / PERIODIC) are written by the same launch.
 * status (device, may be NULL): int32[2] = {OR of mh_status bits, unused}.
 */
int mh_euler_cart_stage(const mh_euler_cart_desc* d, const double* u_in, const double* u_base, double* u_out,
                        double dt, double stage_weight, int row_begin, int row_end, int32_t* status, void* stream);

/* Fill the physical ghost rows of a field (after an upload). External sides are left untouched. */
int mh_euler_cart_fill_ghosts(const mh_euler_cart_desc* d, double* u, void* stream);

/* Host-layout conversion on the device: AoS [n0][row_pitch][5] (no ghosts) <-> SoA field with ghosts.
 * Replaces nothing in the reference (its arrays are AoS throughout); this is the boundary's layout shim. */
int mh_aos_to_soa(const double* aos_dev, double* soa_dev, int nq, int n0, size_t row_pitch, void* stream);
int mh_soa_to_aos(const double* soa_dev, double* aos_dev, int nq, int n0, size_t row_pitch, void* stream);

/* Measurement aid: copies ndoubles doubles with the stage kernels' access shape (8 B per lane). Used to
 * calibrate rocprofv3's FETCH_SIZE / WRITE_SIZE on a known byte count (MI355X_MICROARCH.md, HBM section). */
int mh_calib_stream_copy(const double* src_dev, double* dst_dev, size_t ndoubles, void* stream);

/* ------------------------------------------------------------------------ */
/* Context API for compiled hosts (owns device memory and streams).           */
/* Mirrors the solution_t value that the reference's drivers thread through   */
/* `next` (src/subprog_cloud.cpp:676-697): upload once, step many, download   */
/* when a task is due.                                                        */
/* ------------------------------------------------------------------------ */
typedef struct mh_ctx mh_ctx;

int  mh_create(mh_ctx** ctx, int device_id);
void mh_destroy(mh_ctx* ctx);
const char* mh_last_error(const mh_ctx* ctx);      /* ctx may be NULL: last error of the calling thread */

int  mh_euler_cart_configure(mh_ctx* ctx, const mh_euler_cart_desc* d, int rk_order);

/* `sedov` sub-program (BASELINE config 1): 1-D spherical blast, piecewise constant + HLLE + forward Euler,
 * volume-integrated conserved variables, reflecting inner / zero-gradient outer boundary.
 * Replaces SedovProblem<HydroSystem>::next_solution (src/subprog_sedov.cpp:394-421). vertices_host[nz+1]
 * are the radial vertices (src/subprog_sedov.cpp:366-371); geometry factors are built host-side with the
 * reference's libm calls. After this call mh_upload / mh_step / mh_download act on the sedov state
 * (AoS [nz][5], volume-integrated); mh_step(ctx, dt, n) takes dt = cfl*(r1 - r0) from the caller (:404-405). */
typedef struct
{
    int    nz;              /* radial zones */
    double gamma;           /* 4/3 in the reference (#define at src/subprog_sedov.cpp:48) */
    int    system;          /* MH_SYSTEM_SRHD (the sub-program's default) or MH_SYSTEM_EULER (newtonian=1) */
    int    arith;           /* MH_ARITH_STRICT */
} mh_sedov_desc;
int  mh_sedov_configure(mh_ctx* ctx, const mh_sedov_desc* d, const double* vertices_host);
/* SedovProblem::make_diagnostic_fields and the shock locator of compute_time_series_data (src/subprog_sedov.cpp:252-308,
 * post_shock_locator.hpp:73-170) of the device-resident state. fields_host [4][nz]: specific_entropy,
 * gas_pressure, mass_density, radial velocity (Euler) or gamma-beta (SRHD); indices_host = {shock_index, downstream_index
 * (maximum pressure behind), upstream_index (pressure plateau ahead)}. The scalar time-series entries are a few host operations on
 * these (parabola_vertex, solve_for_shock_velocity) and stay in the driver. Either pointer may be NULL. */
int  mh_sedov_diagnostics(mh_ctx* ctx, double* fields_host, int32_t indices_host[3]);

/* `cloud` sub-program (BASELINE config 4): 2-D axisymmetric spherical-polar SRHD, PCM/PLM + HLLE, RK1/RK2,
 * cell-integrated conserved variables (D, S_r, S_theta, S_phi, tau). Replaces CloudProblem::advance and
 * next_solution (src/subprog_cloud.cpp:511-584, :676-697). Axis 0 = radius (the slab axis), axis 1 = polar angle.
 * Inner radial boundary: nozzle inflow given as PRIMITIVES per polar cell (src/subprog_cloud.cpp:466-493; the
 * host evaluates the jet model at the step-start time and passes the row); outer: zero-gradient; poles: zero
 * slope in the pole cells and zero flux through the pole faces (:563, :573).                                  */
typedef struct
{
    int    nr, nq;              /* LOCAL radial rows on this device, polar cells */
    int    nr_global;           /* radial cells of the whole grid (vertex array has nr_global + 1 entries) */
    int    row_offset;          /* global index of local row 0 (nd::partition_shape slab) */
    double gamma;               /* 4/3 in the reference (#define at src/subprog_cloud.cpp:52) */
    double plm_theta;           /* < 0: reconstruct_method 1 (piecewise constant) */
    double temperature_floor;   /* recover_primitive's floor (0 = none) */
    int    bc_lo0, bc_hi0;      /* MH_BC_INFLOW / MH_BC_OUTFLOW on physical sides, MH_BC_EXTERNAL on slab cuts */
    int    arith;               /* MH_ARITH_STRICT */
    int    chunk_rows;          /* rows marched per wave (0 = default) */
    int    tail_rows;           /* graded tail, as in the cartesian descriptor: 0, 0 = default; tail_rows < 0 = off */
    int    tail_chunk_rows;
    int    fuse_stages;         /* RK2 step of a whole field (both radial sides physical) as ONE launch - the first-stage field stays in LDS,
                                 * csrc/cloud_fused.hip; next_solution's s0 * 1/2 + advance(advance(s0)) * 1/2, src/subprog_cloud.cpp:676-697,
                                 * with the step-start nozzle row in both stages (:524): 0 = where available (MH_ARITH_FAST, PLM, rk_order 2;
                                 * results bit-identical to the two launches), < 0 = never, > 0 = required (configure fails otherwise) */
    int    planar;              /* field and nozzle row without AZIMUTHAL momentum - the `cloud` problem as upstream sets it up (radial envelope and
                                 * nozzle, src/subprog_cloud.cpp:626-660, :466-493), which the reference carries as zeros through every operator. The
                                 * PLM kernels (one-launch step and stage kernels, both arithmetic modes; MH_ARITH_STRICT on the bit pattern of
                                 * +0.0, bit-identical to the reference in all five components - csrc/srhd_device.hpp) then skip that component
                                 * (cf. mh_euler_cart_desc.planar): 0 = where the stepper has VERIFIED it (mh_upload / the slab uploads look at the
                                 * field, mh_cloud_set_inflow / mh_slab_set_inflow at the row; a step whose row has an azimuthal velocity takes the
                                 * general kernels from then on; slabs that exchange with other processes only under > 0), < 0 = never,
                                 * > 0 = asserted (upload / set_inflow fail otherwise) */
} mh_cloud_desc;

/* doubles of the packed device geometry block: rv[nr_global+1] | dmu[nq] | sinq[nq+1] | cotq[nq] | per-row factors [nr_global][8] |
 * per-column factors [nq][8] (the two factor tables are read by MH_ARITH_FAST only: the geometry products of
 * src/subprog_cloud.cpp:260-290 split into a radial and a polar part) */
size_t mh_cloud_geometry_doubles(const mh_cloud_desc* d);
/* Host side: fill that block from the vertex arrays with the reference's libm calls
 * (dmu_j = -cos q_{j+1} - -cos q_j, sin q_j, cot = tan(pi/2 - (q_j + q_{j+1})/2); src/subprog_cloud.cpp:260-290,
 * src/physics_srhd.hpp:314). */
int  mh_cloud_pack_geometry(const mh_cloud_desc* d, const double* r_vertices, const double* q_vertices, double* geom_host);
/* One stage over local rows [row_begin, row_end) (stateless launcher; all pointers DEVICE; inflow_dev = [5][nq]). */
int  mh_cloud_stage(const mh_cloud_desc* d, const double* geom_dev, const double* inflow_dev, const double* u_in,
                    const double* u_base, double* u_out, double dt, double stage_weight, int row_begin, int row_end,
                    int32_t* status, void* stream);
/* Context form (single device, both radial sides physical). After mh_cloud_configure, mh_upload / mh_step /
 * mh_download act on the cloud state (AoS [nr][nq][5]); mh_cloud_set_inflow takes host AoS [nq][5] primitives. */
int  mh_cloud_configure(mh_ctx* ctx, const mh_cloud_desc* d, const double* r_vertices_host, const double* q_vertices_host, int rk_order);
int  mh_cloud_set_inflow(mh_ctx* ctx, const double* inflow_prims_host);
/* CloudProblem::make_diagnostic_fields (src/subprog_cloud.cpp:334-433) of the device-resident solution, evaluated on the device
 * when the driver's write_diagnostics task is due. units = {length (cm), mass (g), time (s)} of
 * make_reference_units (:318-326). Host outputs, either may be NULL:
 *   fields  [5][nr][nq]: mass_density, gas_pressure, specific_entropy, radial_gamma_beta, radial_energy_flow   (cgs but the entropy)
 *   columns [15][nq]   : total_energy_at_theta, solid_angle_at_theta, shock_midpoint_radius, shock_upstream_radius, shock_pressure_radius,
 *                        shock_luminosity_radius, postshock_flow_gamma, postshock_flow_power, ..power02, 04, 08, 16, 32, 64, ..power_max
 * (the member order of diagnostic_fields_t, :147-161; the shock locator is post_shock_locator.hpp:73-170). Primitive recovery and
 * fluxes are the STRICT ones; log / pow are the device library's, so the entropy agrees with the reference to an ulp or two. */
int  mh_cloud_diagnostics(mh_ctx* ctx, const double units[3], double* fields_host, double* columns_host);

int  mh_upload(mh_ctx* ctx, const double* u_aos_host, size_t ncell);     /* host AoS -> device SoA (+ ghosts) */
int  mh_download(mh_ctx* ctx, double* u_aos_host, size_t ncell);         /* device SoA -> host AoS */
/* nsteps full time steps (all RK stages) with fixed dt, like the reference's cloud/sedov drivers. */
int  mh_step(mh_ctx* ctx, double dt, int nsteps);
int  mh_synchronize(mh_ctx* ctx);
/* OR of mh_status bits since the last call (reads back 8 bytes; synchronises) / the same with the first failing cell. Both clear. */
int  mh_status_word(mh_ctx* ctx, int32_t* status);
int  mh_status(mh_ctx* ctx, mh_step_result* result);
/* ONE full time step as a transaction (Euler and cloud contexts): the step's result goes to a third buffer and becomes the
 * solution only if no cell raised a status bit; otherwise the call returns MH_E_PHYSICS, `result` says what and where, and the
 * previous solution stays in place and downloadable, bit for bit - what the reference's callers rely on when they catch the
 * exception and retry from the OLD solution (src/subprog_binary.cpp:285-292). Synchronises (one 8-byte read-back per step);
 * mh_step stays the asynchronous in-place form for drivers that, like upstream's cloud / sedov, do not retry. */
int  mh_step_checked(mh_ctx* ctx, double dt, mh_step_result* result);
/* 1 if the steps of this context currently take their planar kernels (mh_euler_cart_desc.planar / mh_cloud_desc.planar: the uploaded 2-D
 * field - and the nozzle row - have no third momentum), else 0 */
int  mh_field_is_planar(const mh_ctx* ctx);
/* Raw device pointer of the current solution field (SoA with ghosts) and of the scratch field, for halo exchange. */
double* mh_field_ptr(mh_ctx* ctx, int which /* 0 = current solution, 1 = stage scratch */);
/* Average kernel time in ms of the stage launches since the last reset, measured with HIP events
 * on the context's stream (enable before stepping). */
int  mh_profile_enable(mh_ctx* ctx, int on);
int  mh_profile_read(mh_ctx* ctx, double* avg_stage_ms, int* nlaunches);

/* ------------------------------------------------------------------------ */
/* Native slab stepper (one process per GPU): the Euler step (2-D, 3-D) and    */
/* the `cloud` step (radial slabs) with the axis-0 ghost exchange as RCCL      */
/* send/recv over xGMI, overlapped with the interior update on a second        */
/* stream; without neighbours the Euler step is replayed from one HIP graph.   */
/* The cut is nd::partition_shape (src/core_ndarray.hpp:820-836), the          */
/* execution policy it replaces is mara::evaluate_on<N> (src/app_parallel.hpp  */
/* :75-103) - thread slabs that share one address space upstream.              */
/* ------------------------------------------------------------------------ */
typedef struct mh_slab mh_slab;

/* RCCL unique id (128 bytes) created on one rank; the host distributes it to the others (any transport). */
int  mh_comm_unique_id(void* id128);
/* One communicator per PROCESS, lent to the steppers the host builds one after the other (mh_slab_use_comm, mh_block_use_comm,
 * mh_binary_band_use_comm): ncclCommInitRank is collective and costs hundreds of milliseconds, so a host that sets up several steppers
 * in a row (bench.py's legs) enters it once. mh_comm_create is collective over `world` ranks; the communicator must outlive every
 * stepper that borrows it. The reference's analogue is the one thread pool a run creates (src/app_parallel.hpp:75-103 starts its
 * workers per evaluation; src/subprog_binary_scheme.cpp:12 keeps one). */
typedef struct mh_comm mh_comm;
int  mh_comm_create(mh_comm** comm, const void* comm_id128, int rank, int world, int device_id);
void mh_comm_destroy(mh_comm* comm);
/* `global`: the description of the WHOLE grid (n[0] = global rows; bc_lo0/bc_hi0 = the physical boundary
 * condition). comm_id128 may be NULL when the rank has no neighbours. self_exchange != 0 with world == 1 and a
 * periodic axis 0 makes the rank exchange with itself (exercises the whole path on one GPU). */
int  mh_slab_create(mh_slab** slab, const mh_euler_cart_desc* global, int rk_order, int rank, int world,
                    const void* comm_id128, int self_exchange, int device_id);
/* comm_id128 == NULL for a rank WITH neighbours defers the RCCL communicator: the host first lets every rank agree that creation
 * succeeded and only then calls mh_slab_connect on all of them (ncclCommInitRank is collective: a rank that failed earlier would
 * leave the others blocked inside it). */
int  mh_slab_connect(mh_slab* slab, const void* comm_id128);
/* ... or lend it the process's communicator (same rank / world / device); not collective */
int  mh_slab_use_comm(mh_slab* slab, mh_comm* comm);
/* `cloud` sub-program (BASELINE config 4: radial slabs + RCCL halo): CloudProblem::advance / next_solution
 * (src/subprog_cloud.cpp:511-584, :676-697) on rows partition_shape(nr, world)[rank] of the global grid. `global` describes the
 * WHOLE grid (nr == nr_global, row_offset 0); rank 0 keeps the nozzle-inflow boundary, the last rank the zero-gradient one, cut
 * sides are MH_BC_EXTERNAL. Geometry is evaluated from the GLOBAL vertex arrays, so every cell sees the same bits as in the
 * single-domain run. mh_slab_set_inflow: host AoS [nq][5] primitives of the inner ghost row at the step-start time (:466-493);
 * every rank may call it, only the rank owning row 0 reads it. Stepping: mh_slab_step(slab, dt, 1, 0) per time step. */
int  mh_slab_cloud_create(mh_slab** slab, const mh_cloud_desc* global, const double* r_vertices_host, const double* q_vertices_host,
                          int rk_order, int rank, int world, const void* comm_id128, int device_id);
int  mh_slab_set_inflow(mh_slab* slab, const double* inflow_prims_aos_host);
/* Planarity contract of the row (mh_cloud_desc.planar): a cloud slab starts out NOT knowing a nozzle row and therefore on the general kernels;
 * a row with an azimuthal velocity sends the slab it is handed to onto the general kernels until the next upload. A lone slab, or one whose
 * neighbours live in other processes (there only under planar > 0), re-resolves in mh_slab_set_inflow. The members of a loopback group are
 * resolved TOGETHER - hand the row to mh_slab_group_set_inflow, which updates every member in one call (the group is planar only if every
 * member's rows and the row are); through the per-slab call a member can lose its planar kernels but never gain them, so a host that hands
 * the row to the nozzle-side member only stays correct, on the general kernels. Likewise a rank of a multi-process run that is never
 * handed a row (only the rank owning row 0 READS it) stays on the general kernels: correct, slower - hand the row to every rank. */
int  mh_slab_group_set_inflow(mh_slab** slabs, int n, const double* inflow_prims_aos_host);
/* LOOPBACK groups: all `world` slabs of a decomposition as objects of ONE process on one GPU. A "receive" is a stream-ordered
 * device-to-device copy out of the neighbour object's field under the same event protocol; cut, ghost layout, edge / interior
 * split and staggering are the code the RCCL ranks run. (RCCL itself refuses two ranks on one device.) This is how the multi-rank
 * stepper is executed - ranks with lo != hi, mixed physical / external sides, uneven cuts - and compared bit for bit with the
 * single-domain run on a one-GPU box. slabs[world] receives the handles (destroy each with mh_slab_destroy); the group calls take
 * the global host array [N0][row_pitch][5] and drive the members in lockstep; mh_slab_download / _status / _rows work per member. */
int  mh_slab_group_create(mh_slab** slabs, const mh_euler_cart_desc* global, int rk_order, int world, int device_id);
int  mh_slab_cloud_group_create(mh_slab** slabs, const mh_cloud_desc* global, const double* r_vertices_host, const double* q_vertices_host,
                                int rk_order, int world, int device_id);
/* The same groups with member r on device_ids[r]: ONE process (one host thread) driving several GPUs - what the reference's thread slabs
 * (mara::evaluate_on<N>, src/app_parallel.hpp:75-103) become when a slab is a device; "receives" are peer copies. With all ids equal this is
 * the group above (the form that is tested on a one-GPU box); distinct devices have not been exercised there. */
int  mh_slab_group_create_on(mh_slab** slabs, const mh_euler_cart_desc* global, int rk_order, int world, const int* device_ids);
int  mh_slab_cloud_group_create_on(mh_slab** slabs, const mh_cloud_desc* global, const double* r_vertices_host, const double* q_vertices_host,
                                   int rk_order, int world, const int* device_ids);
int  mh_slab_group_upload(mh_slab** slabs, int world, const double* u_aos_global_host);
int  mh_slab_group_download(mh_slab** slabs, int world, double* u_aos_global_host);
int  mh_slab_group_step(mh_slab** slabs, int world, double dt, int nsteps);
void mh_slab_destroy(mh_slab* slab);
int  mh_slab_rows(const mh_slab* slab, int* row0, int* row1);                 /* this rank's rows [row0, row1) */
/* The decisions of a slab decomposition that touch no device, in ONE place (csrc/slab_plan.hpp) - pure host code, callable without a GPU: the
 * rows of a rank (nd::partition_shape, src/core_ndarray.hpp:820-836), its neighbours, the ghost rows that travel (2 per stage, or 4 once per
 * step under the one-launch RK2 step across the cuts), the edge rows stepped first, and the messages of one exchange IN ISSUE ORDER (rows are
 * local: ghost rows are negative / >= row1 - row0). The native stepper issues exactly these; mara3_amd/slab.py's torch.distributed stepper
 * (bench.py's fallback, the gloo tests on CPU) reads them here instead of restating them. */
typedef struct
{
    int row0, row1, lo, hi;               /* rows [row0, row1) of the global grid; neighbour ranks, -1 = physical side */
    int ghost_rows, edge_rows, exchanges_per_step;
    int nmsg;
    struct { int send, peer, first_row, rows; } msg[4];
} mh_slab_plan;
int  mh_slab_plan_make(int nrows_global, int world, int rank, int periodic, int self_exchange, int rk_order, int fused_cut, mh_slab_plan* plan);
int  mh_slab_plan_of(const mh_slab* slab, mh_slab_plan* plan);                 /* the plan this slab was created with */
int  mh_slab_is_planar(const mh_slab* slab);                                  /* 1: its launches skip the third momentum (the descriptors' planar) */
int  mh_slab_launches_per_step(const mh_slab* slab);                          /* 1: the RK2 step of its rows is ONE fused launch (fuse_stages), else rk_order */
int  mh_slab_upload(mh_slab* slab, const double* u_aos_slab_host);            /* host AoS [n0][n1][5] of this rank's rows */
int  mh_slab_download(mh_slab* slab, double* u_aos_slab_host);
int  mh_slab_step(mh_slab* slab, double dt, int nsteps, int use_graph);       /* use_graph: replay one captured step (RK2) */
int  mh_slab_synchronize(mh_slab* slab);
int  mh_slab_status_word(mh_slab* slab, int32_t* status);
int  mh_slab_status(mh_slab* slab, mh_step_result* result);                   /* first_bad_index in the GLOBAL host array */
double* mh_slab_field_ptr(mh_slab* slab, int which);
/* HIP-event timing of the bulk (interior) stage launches in eager mode: avg_ms[0|1] = first | second RK stage */
int  mh_slab_profile_enable(mh_slab* slab, int on);
int  mh_slab_profile_read(mh_slab* slab, double avg_ms[2], int nlaunches[2], int* bulk_rows);

/* ------------------------------------------------------------------------ */
/* Block stepper (one process per GPU): the 3-D Euler step under a 3-axis    */
/* block decomposition (BASELINE config 5: 1024^3 as (2,2,2) blocks on 8     */
/* GPUs) with the ghost exchange on all three axes as RCCL send/recv.        */
/* Blocks per axis: mara::propose_block_decomposition<3>(world)              */
/* (src/app_parallel.hpp:119-131); extents: nd::divvy as applied by          */
/* create_access_pattern_array (:148-179): block b of B on an axis of N      */
/* cells owns [b N / B, (b + 1) N / B); rank = (c0 B1 + c1) B2 + c2.         */
/* Per stage and cut side one message: axis 0 two contiguous planes straight */
/* out of the field (2 * 5 * plane doubles), axes 1 / 2 two rows / columns   */
/* of every plane packed to [n0][5][2][n2] / [n0][5][n1][2]; all in one RCCL */
/* group on a side stream while the interior updates on the main stream.     */
/* ------------------------------------------------------------------------ */
typedef struct mh_block mh_block;
/* `global`: the WHOLE grid (rank 3; bc_lo0 / bc_hi0 / bc_transverse = the physical boundary conditions). comm_id128 as for mh_slab_create
 * (NULL with neighbours: connect later with mh_block_connect, once every rank has agreed that creation succeeded). */
int  mh_block_create(mh_block** block, const mh_euler_cart_desc* global, int rk_order, int rank, int world, const void* comm_id128, int self_exchange,
                     int device_id);          /* self_exchange != 0 with world == 1: periodic axes wrap through RCCL send/recv to self (exercises the exchange on one GPU) */
int  mh_block_connect(mh_block* block, const void* comm_id128);
int  mh_block_use_comm(mh_block* block, mh_comm* comm);
void mh_block_destroy(mh_block* block);
/* blocks per axis, this block's coordinates, first global cell and cell count per axis (any pointer may be NULL) */
int  mh_block_extent(const mh_block* block, int blocks_per_axis[3], int coords[3], int start[3], int count[3]);
/* neighbour ranks {axis0 lo, hi, axis1 lo, hi, axis2 lo, hi} (-1: physical boundary) and the message size per axis in doubles */
int  mh_block_neighbours(const mh_block* block, int ranks[6], size_t message_doubles[3]);
int  mh_block_upload(mh_block* block, const double* u_aos_block_host);       /* host AoS [n0][n1][n2][5] of this block's cells */
int  mh_block_download(mh_block* block, double* u_aos_block_host);
int  mh_block_step(mh_block* block, double dt, int nsteps);
int  mh_block_synchronize(mh_block* block);
/* global_n = {N0, N1, N2}: first_bad_index in the GLOBAL host array; NULL: within the block */
int  mh_block_status(mh_block* block, mh_step_result* result, const int global_n[3]);
/* HIP-event timing of the INTERIOR launches: reads (and clears) the averages of the launches since the last call, then sets the switch */
int  mh_block_profile(mh_block* block, int enable, double avg_ms[2], int nlaunches[2], long* interior_cells);
/* LOOPBACK group (see mh_slab_group_create): all `world` blocks as objects of one process on one GPU, receives as device-to-device
 * copies under the same event protocol; the group calls take the GLOBAL host array [N0][N1][N2][5]. */
int  mh_block_group_create(mh_block** blocks, const mh_euler_cart_desc* global, int rk_order, int world, int device_id);
int  mh_block_group_upload(mh_block** blocks, int world, const double* u_aos_global_host);
int  mh_block_group_download(mh_block** blocks, int world, double* u_aos_global_host);
int  mh_block_group_step(mh_block** blocks, int world, double dt, int nsteps);

/* ------------------------------------------------------------------------ */
/* Per-function device entry points (parity tests call these through the ABI) */
/* inputs/outputs are DEVICE arrays of n items, AoS rows of 5 (or 3 / 1)      */
/* ------------------------------------------------------------------------ */
int mh_plm_gradient_n(size_t n, const double* yl, const double* y0, const double* yr, double theta, double* g, int arith, void* stream);
int mh_euler_recover_primitive_n(size_t n, const double* U, double gamma, double temperature_floor, double* P, int arith, void* stream);
int mh_euler_to_conserved_n(size_t n, const double* P, double gamma, double* U, int arith, void* stream);
int mh_euler_riemann_n(size_t n, const double* Pl, const double* Pr, int axis, double gamma, int riemann, double* F, int arith, void* stream);

/* mara::srhd (src/physics_srhd.hpp): recover_primitive :364-451 (status[i] = OR of mh_status bits where the
 * reference throws), to_conserved_density :213-227, riemann_hlle :466-483, spherical_geometry_source_terms :309-326
 * (cot_theta[i] = tan(pi/2 - theta_i) evaluated by the caller, as the host does for the cloud geometry). */
int mh_srhd_recover_primitive_n(size_t n, const double* U, double gamma, double temperature_floor, double* P, int32_t* status, void* stream);
int mh_srhd_to_conserved_n(size_t n, const double* P, double gamma, double* U, void* stream);
int mh_srhd_riemann_hlle_n(size_t n, const double* Pl, const double* Pr, int axis, double gamma, double* F, void* stream);
int mh_srhd_source_terms_n(size_t n, const double* P, const double* r, const double* cot_theta, double gamma, double* S, void* stream);

/* mara::iso2d (src/physics_iso2d.hpp), AoS rows of 3 in the logical order (Sigma, x, y); x = positions [n][2]:
 * to_conserved_per_area :249-258, recover_primitive(U) :351-362 (threw[i] = 1 where the reference throws on
 * negative density), to_conserved_angmom_per_area :263-272, recover_primitive(Q, x) :376-390, flux :299-307,
 * wavespeeds + max_wavespeed :320-337 (out rows = minus, plus, max), riemann_hlle :488-506 / riemann_hllc
 * :556-583,:610-712 (contact[i] = s_star, threw[i] = 1 where interface_flux throws). axis in {0, 1}. */
int mh_iso2d_to_conserved_n(size_t n, const double* P, double* U, void* stream);
int mh_iso2d_recover_primitive_n(size_t n, const double* U, double* P, int32_t* threw, void* stream);
int mh_iso2d_to_conserved_angmom_n(size_t n, const double* P, const double* x, double* Q, void* stream);
int mh_iso2d_recover_primitive_angmom_n(size_t n, const double* Q, const double* x, double* P, int32_t* threw, void* stream);
int mh_iso2d_flux_n(size_t n, const double* P, const double* cs2, int axis, double* F, void* stream);
int mh_iso2d_wavespeeds_n(size_t n, const double* P, const double* cs2, int axis, double* lam, void* stream);
int mh_iso2d_riemann_n(size_t n, const double* Pl, const double* Pr, const double* cs2l, const double* cs2r, int axis, int riemann,
                       double* F, double* contact, int32_t* threw, void* stream);

/* ------------------------------------------------------------------------ */
/* Integer / index work (host side, bit-exact with the reference)             */
/* ------------------------------------------------------------------------ */
/* nd::partition_shape, src/core_ndarray.hpp:820-836: rows [start, final) of slab `part` of `nparts` */
void mh_partition_rows(size_t count, size_t nparts, size_t part, size_t* start, size_t* final_);
/* mara::propose_block_decomposition<Rank>, src/app_parallel.hpp:119-131 */
int  mh_propose_block_decomposition(int rank, unsigned long nblocks, unsigned long* blocks_per_axis);
/* mara::create_access_pattern_array, src/app_parallel.hpp:148-179, for propose_block_decomposition<3>(world): blocks per axis, the
 * coordinates of block `rank` (row-major position in the array of access patterns), its first cell and cell count per axis
 * (nd::divvy: block b of B owns [b N / B, (b + 1) N / B)). Host only. MH_E_INVALID where the reference throws "too many blocks". */
int  mh_block_layout(const int global_n[3], int world, int rank, int blocks_per_axis[3], int coords[3], int start[3], int count[3]);

/* Kepler two-body model, host side (src/model_two_body.hpp). Bodies are (mass, x, y, vx, vy).
 * mh_two_body_state = mara::compute_two_body_state(full_orbital_elements_t, t) :209-268 (Newton solve for the
 * eccentric anomaly :168-208); mh_orbital_elements_from_state = mara::compute_orbital_elements :295-381, which
 * returns MH_E_PHYSICS where the reference throws (unbound orbit). */
typedef struct { double separation, total_mass, mass_ratio, eccentricity; } mh_orbital_elements;
typedef struct
{
    double pomega, tau, cm_position_x, cm_position_y, cm_velocity_x, cm_velocity_y;
    mh_orbital_elements elements;
} mh_full_orbital_elements;
typedef struct { double body1[5], body2[5]; } mh_two_body_t;
int  mh_two_body_state(const mh_full_orbital_elements* elements, double t, mh_two_body_t* out);
int  mh_orbital_elements_from_state(const mh_two_body_t* state, double t, mh_full_orbital_elements* out);

/* Difference of two sets of orbital elements, mara::diff(a, b) src/model_two_body.hpp:492-518 (b - a, with pomega wrapped
 * to the nearest image mod 2 pi and tau mod the orbital period of b). */
void mh_orbital_elements_diff(const mh_full_orbital_elements* a, const mh_full_orbital_elements* b, mh_full_orbital_elements* out);

/* ---- `binary` sub-program: isothermal circumbinary disk on a uniform-depth block tree (BASELINE config 3) --------------
 * Replaces binary::advance_u / next_solution / maximum_timestep (src/subprog_binary_scheme.cpp:790-904, :1107-1126;
 * src/subprog_binary.cpp:258-293) for a tree whose every node is refined (focus_factor large): the blocks tile a periodic
 * n x n tensor-product mesh, n = block_size << depth. Fields are iso2d conserved_per_area (Sigma, px, py) in the LOGICAL
 * order (never the std::tuple storage order): host AoS [n][n][3], axis 0 = x; device SoA as above with nq = 3.
 * Arithmetic: reference operation order, IEEE division / sqrt, no FMA contraction; the reference's pow(x, 1/2), pow(x, 3/2),
 * exp and tanh are evaluated by sqrt / x sqrt(x) / the device math library, so parity is to L1 <= 1e-12, not bit-exact. */
typedef struct mh_binary_desc
{
    int32_t n;                      /* cells per side */
    int32_t block_size;             /* cells per side of one tree block: grouping of the source-term totals (work_done_on) */
    double  domain_radius, mach_number, alpha, nu, alpha_cutoff_radius;
    double  sink_rate, sink_radius, softening_radius;
    double  density_floor;          /* absolute: run_config density_floor * disk_mass (solver_data.cpp:100) */
    double  plm_theta;
    int32_t axisymmetric_cs2;
    int32_t chunk_rows;             /* rows marched per wavefront; 0 = default */
    int32_t angmom_form;            /* 0: conserve_linear_p = 1, fields (Sigma, px, py), advance_u;
                                       1: conserve_linear_p = 0, fields (Sigma, Sigma s_r, Sigma l_z), advance_q (scheme.cpp:906-1020) */
    int32_t arith;                  /* MH_ARITH_STRICT (reference operation order, IEEE division / sqrt) or MH_ARITH_FAST */
    double  gst_suppr_radius;       /* source_term_softening * min(dx, dy) (solver_data.cpp:91); advance_q only */
} mh_binary_desc;

/* source_term_total_t (scheme.cpp:17-32), [2] = per body */
enum mh_binary_total { MH_T_MASS_ACC = 0, MH_T_L_ACC = 2, MH_T_TORQUE = 4, MH_T_PX_ACC = 6, MH_T_PY_ACC = 8, MH_T_FX = 10, MH_T_FY = 12,
                       MH_T_WORK = 14, MH_T_MASS_EJ = 16, MH_T_L_EJ = 17, MH_BINARY_NTOTALS = 18 };

size_t mh_binary_field_doubles(const mh_binary_desc* d);      /* 3 * (n + 4) * n */
size_t mh_binary_scratch_doubles(const mh_binary_desc* d);    /* reduction scratch of one stage */
/* One stage (stateless launcher; all pointers DEVICE except bodies = host (mass, x, y, vx, vy) x 2):
 *   u_out = advance_u(u_in)                      (stage_weight == 1)
 *   u_out = u_base (1 - w) + advance_u(u_in) w   (otherwise; w = 0.5 is the RK2 combine)
 * u_init is a field like u; buffer_rate is [n][n]; xv, yv are the n + 1 vertex coordinates; totals_dev receives the
 * MH_BINARY_NTOTALS source-term totals of this stage; status |= MH_STATUS_NEG_DENSITY where validate_u would throw. */
int  mh_binary_stage(const mh_binary_desc* d, const double* xv_dev, const double* yv_dev, const double* u_in, const double* u_base,
                     double* u_out, const double* u_init, const double* buffer_rate, const double* bodies_host, double dt,
                     double stage_weight, double* totals_dev, double* scratch_dev, int32_t* status, void* stream);
/* max over cells of primitive_t::max_wavespeed(cs2(x_c)) -> one double on the device (maximum_timestep = spacing / that) */
int  mh_binary_max_wavespeed(const mh_binary_desc* d, const double* xv_dev, const double* yv_dev, const double* u,
                             const double* bodies_host, double* result_dev, void* stream);

/* The scalar part of binary::solution_t (src/subprog_binary.hpp:130-150) */
typedef struct mh_binary_state
{
    double  time;
    int64_t iteration;
    double  mass_accreted_on[2], angular_momentum_accreted_on[2], integrated_torque_on[2], work_done_on[2];
    double  mass_ejected, angular_momentum_ejected;
    mh_full_orbital_elements orbital_elements_acc, orbital_elements_grav, orbital_elements;
} mh_binary_state;
typedef struct mh_binary_run
{
    int32_t rk_order;               /* 1 or 2 */
    int32_t fixed_dt;               /* use recommended_time_step instead of cfl * maximum_timestep */
    int32_t no_accretion_force;
    int32_t reserved;
    double  cfl_number, recommended_time_step, begin_live_binary;
} mh_binary_run;
/* Host-side set-up of the `binary` problem, evaluated with the host libm exactly as upstream (no device work):
 * mh_binary_vertices = the vertex coordinates of mara::create_vertex_quadtree for an always-true predicate
 * (mesh_tree_operators.hpp:158-190: linspace(-1, 1, block_size + 1), `depth` rounds of prolong_verts) times domain_radius
 * (subprog_binary.cpp:165-185); out[(block_size << depth) + 1], the same for both axes.
 * mh_binary_solver_data = create_disk_profile (subprog_binary.cpp:105-153) sampled at the cell centres as conserved_per_area,
 * the buffer-rate field and recommended_time_step of create_solver_data (subprog_binary_solver_data.cpp:20-102). */
typedef struct mh_binary_model
{
    double  softening_radius, disk_radius, mach_number, disk_mass, ambient_density, mdot;
    int32_t counter_rotate;
    int32_t angmom_form;            /* initial field as to_conserved_angmom_per_area(x) (subprog_binary.cpp:208-216) */
    double  buffer_damping_rate, domain_radius, cfl_number;
} mh_binary_model;
int  mh_binary_vertices(int block_size, int depth, double domain_radius, double* out_host);
int  mh_binary_solver_data(const mh_binary_model* m, int n, const double* xv_host, const double* yv_host, double* u_init_aos_host,
                           double* buffer_rate_host, double* recommended_time_step);
typedef struct mh_binary mh_binary;
/* Solver object: owns the device fields (solution, stage buffers, static solver data). Host arrays are copied. */
int  mh_binary_create(mh_binary** out, int device, const mh_binary_desc* d, const mh_binary_run* run, const double* xv_host,
                      const double* yv_host, const double* u_init_aos_host, const double* buffer_rate_host);
void mh_binary_destroy(mh_binary* b);
/* u_aos_host = NULL: start from the initial field */
int  mh_binary_set_solution(mh_binary* b, const double* u_aos_host, const mh_binary_state* state);
int  mh_binary_get_solution(mh_binary* b, double* u_aos_host /* may be NULL */, mh_binary_state* state /* may be NULL */);
/* nsteps x binary::next_solution: dt choice, RK stages, accumulators and orbital-element perturbations, and the safe-mode
 * retry (dt * 0.1, theta = 0) when a stage reports a negative density (the reference catches the exception and retries
 * from the old solution, subprog_binary.cpp:285-292). safe_mode_steps (may be NULL) counts the steps that needed it.
 * Returns MH_E_PHYSICS if the safe-mode retry fails too (the reference's exception then escapes). */
int  mh_binary_next(mh_binary* b, int nsteps, int* safe_mode_steps);
/* Multi-GPU for uniform-depth trees: the mesh is cut into BANDS of whole rows of tree blocks - rank r of N owns block rows
 * partition_shape(n / block_size, N)[r] (src/core_ndarray.hpp:820-836); the reference hands whole blocks to its thread pool
 * (tree.map(fn, pool), src/core_tree.hpp:615-625) and blocks must stay whole because work_done_on is nonlinear in each block's sink
 * sums (src/subprog_binary_scheme.cpp:356-365). Per stage one two-row ghost exchange with the periodic neighbours; per host
 * synchronisation one sum of the 2 x 18 totals (max of the wavespeed, of the status words) over the ranks, then every rank does the
 * same scalar bookkeeping. xv, yv, u_init_aos, buffer_rate describe the WHOLE mesh (a band keeps its rows); mh_binary_set_solution /
 * get_solution take the whole-mesh host array as well and touch the band's rows [row0, row1) only. mh_binary_next is collective over the
 * ranks. RCCL form (one process per GPU) and LOOPBACK group (all bands as objects of one process on one GPU, see mh_slab_group_create).
 * Fields do not depend on the partition bit for bit while the binary is not live; the totals agree to the order of summation. */
int  mh_binary_band_create(mh_binary** out, int device, const mh_binary_desc* d, const mh_binary_run* run, const double* xv_host,
                           const double* yv_host, const double* u_init_aos_host, const double* buffer_rate_host, int rank, int world,
                           const void* comm_id128, int self_exchange /* world == 1: the RCCL halo and all-reduces, to self */);
/* comm_id128 == NULL on a band with neighbours defers the communicator to mh_binary_band_use_comm. Both ways of connecting end with the
 * exchange of the initial solution's ghost rows (collective), so mh_binary_next may follow directly. */
int  mh_binary_band_use_comm(mh_binary* b, mh_comm* comm);
int  mh_binary_band_rows(const mh_binary* b, int* row0, int* row1);
/* A band with neighbours may step its EDGE rows first - the first and last `rows` rows, one small launch: RCCL bands run it on their
 * exchange stream with the ghost exchange of the stage behind it, both beside the launch of the interior rows on the main stream, which
 * waits for the exchange only behind the interior. The field does not depend on the cut bit for bit (either arithmetic mode); the totals
 * keep one fixed order of summation per cut (edge waves first). rows == 0 (the state after create): one launch per stage with the
 * exchange behind it; rows < 0: the recommended cut (2 rows: what a neighbour needs; 0 when the band is too thin); else 2 <= rows < n0 / 2.
 * OFF by default because it does not pay where it could be measured (one GPU, exchange to self through RCCL: 0.31 against 0.27 ms per
 * step at 2048^2 - an edge wave's 6 rows take 11 us, and two cross-stream events cost what the send / recv kernel costs).
 * The reference's analogue is the order in which the pool's workers take the blocks (src/core_tree.hpp:615-625): no result depends on it. */
int  mh_binary_band_set_edge_rows(mh_binary* b, int rows);
/* What made an attempt of the most recent mh_binary_next call fail (before its safe-mode retry, or before MH_E_PHYSICS): OR of the status
 * bits of every band and the first failing cell as a flat index into the WHOLE-mesh host array of mh_binary_set_solution; status 0 when
 * no attempt of that call failed. The same on every band. */
int  mh_binary_last_failure(const mh_binary* b, mh_step_result* result);
int  mh_binary_group_create(mh_binary** bands, int world, int device, const mh_binary_desc* d, const mh_binary_run* run, const double* xv_host,
                            const double* yv_host, const double* u_init_aos_host, const double* buffer_rate_host);
int  mh_binary_group_set_solution(mh_binary** bands, int world, const double* u_aos_host, const mh_binary_state* state);
int  mh_binary_group_get_solution(mh_binary** bands, int world, double* u_aos_host, mh_binary_state* state);
int  mh_binary_group_next(mh_binary** bands, int world, int nsteps, int* safe_mode_steps);
double mh_binary_last_dt(const mh_binary* b);
const double* mh_binary_field_ptr(mh_binary* b);
int  mh_binary_profile(mh_binary* b, int enable, double* avg_stage_ms, int* nlaunches);

/* Diagnostics of the device-resident solution, evaluated on the device when a task of the driver is due.
 * mh_binary_disk_totals: binary::disk_mass and binary::disk_angular_momentum (src/subprog_binary_diagnostics.cpp:21-46), the two
 * reductions of a time-series sample (subprog_binary.cpp:356-357). mh_binary_diagnostic_fields: binary::diagnostic_fields
 * (:52-82): sigma, v . rhat, v . phihat per cell into host arrays of n^2 (or nblocks * block_size^2) doubles in the order of
 * mh_binary_get_solution's cells; any of the three may be NULL. IEEE arithmetic whatever `arith` the stepping uses. Against the
 * reference: the fields agree to an ulp of the radius (sqrt where the reference calls pow(r2, 0.5)), the sums to reduction order. */
int  mh_binary_disk_totals(mh_binary* b, double* disk_mass, double* disk_angular_momentum);
int  mh_binary_diagnostic_fields(mh_binary* b, double* sigma, double* radial_velocity, double* phi_velocity);

/* ---- `binary` on a GRADED block tree (the sub-program's default: refinement towards the origin) ---------
 * Leaf blocks of block_size^2 cells at different levels; guard zones are prolonged (piecewise constant) from coarser and
 * restricted (averaged) from finer neighbours, and the face fluxes of a coarse block next to finer ones are replaced by the sum
 * of the fine fluxes (src/subprog_binary_scheme.cpp:132-142, :614-720; mesh_tree_operators.hpp:223-258). Both conserved-variable
 * forms. Block data are ordered as the reference traverses its tree (children in orthant order i + 2 j).
 * Host fields: [nblocks][bs][bs][3]; vertex edges: [nblocks][2][bs + 1] (x of the block's vertex columns, y of its rows). */
typedef struct mh_tree_block { int32_t level, i, j; } mh_tree_block;
/* mara::create_vertex_quadtree with the sub-program's refinement predicate (centroid_radius < focus_factor / level^focus_index,
 * subprog_binary.cpp:174-177) followed by ensure_valid_quadtree (2:1 balance, mesh_tree_operators.hpp:115-139). Returns the number of
 * leaf blocks (also when out is NULL or capacity too small: then nothing is written), or a negative mh_error. */
int  mh_binary_tree_build(int block_size, int depth, double focus_factor, double focus_index, mh_tree_block* out, int capacity);
int  mh_binary_tree_vertices(int block_size, double domain_radius, const mh_tree_block* blocks, int nblocks, double* edges_host);
int  mh_binary_tree_solver_data(const mh_binary_model* m, int block_size, const mh_tree_block* blocks, int nblocks, const double* edges_host,
                                double* u_init_host, double* buffer_rate_host, double* recommended_time_step);
/* Solver object on a graded tree; afterwards mh_binary_set_solution / get_solution / next / last_dt / destroy apply unchanged
 * (solution arrays in the block layout above). d->n is ignored; with d->angmom_form set, d->gst_suppr_radius is
 * source_term_softening times the smallest vertex spacing of any block. */
int  mh_binary_tree_create(mh_binary** out, int device, const mh_binary_desc* d, const mh_binary_run* run, const mh_tree_block* blocks,
                           int nblocks, const double* edges_host, const double* u_init_host, const double* buffer_rate_host);
/* Multi-GPU for GRADED trees. The leaves are ordered along the Hilbert curve of the finest level present (mh_binary_tree_curve_order) and
 * member r of N owns the r-th of N runs of equal length of that order (partition_shape's formula, src/core_ndarray.hpp:820-836). Every
 * member holds the whole tree; per stage it runs each of the three block kernels on its own blocks and the members' results are gathered
 * behind each kernel (primitives + slopes, fluxes, new field + per-tile sums). Totals and time-step bound are then formed by every member
 * over all blocks in the CALLER's block order: field, totals and scalars are the single-domain solver's bit for bit, with a live binary
 * too. The reference hands the leaves to its thread pool in traversal order (tree.map(fn, pool), src/core_tree.hpp:615-625); it declares
 * a hilbert_index (src/core_tree.hpp:1033-1069) that no sub-program uses. Arrays at this boundary (blocks, edges, u, buffer rate, the
 * failing-cell index) stay in the caller's block order. RCCL form: one process per GPU, comm_id128 or (NULL) mh_binary_band_use_comm;
 * LOOPBACK group: all members as objects of one process on one GPU, driven by the mh_binary_group_* calls. */
int  mh_binary_tree_curve_order(const mh_tree_block* blocks, int nblocks, int32_t* order /* [nblocks]: order[k] = the block standing k-th along the curve */);
int  mh_binary_tree_band_create(mh_binary** out, int device, const mh_binary_desc* d, const mh_binary_run* run, const mh_tree_block* blocks,
                                int nblocks, const double* edges_host, const double* u_init_aos_host, const double* buffer_rate_host,
                                int rank, int world, const void* comm_id128);
int  mh_binary_tree_group_create(mh_binary** members, int world, int device, const mh_binary_desc* d, const mh_binary_run* run,
                                 const mh_tree_block* blocks, int nblocks, const double* edges_host, const double* u_init_aos_host,
                                 const double* buffer_rate_host);
/* the blocks (caller's numbering, in curve order) a member runs the block kernels on; ids may be NULL */
int  mh_binary_tree_owned_blocks(const mh_binary* b, int32_t* ids, int* count);

/* device utilities used by bench / tests without torch */
int  mh_device_count(void);
int  mh_device_cu_count(void);         /* compute units of the current HIP device (256 on an MI355X): the launchers size their chunks for whole residency rounds of it */
int  mh_malloc(void** ptr, size_t bytes);
int  mh_free(void* ptr);
int  mh_memcpy_h2d(void* dst, const void* src, size_t bytes);
int  mh_memcpy_d2h(void* dst, const void* src, size_t bytes);
int  mh_device_synchronize(void);

/* Row-range guard of the row-marching kernels (csrc/row_check.hpp; a library built with -DMH_CHECK_ROWS, `make -C mara3_amd/csrc check`):
 * {smallest, largest} axis-0 row (plane) index the kernels of a family REQUESTED since the last reset - a check build holds every access
 * to the rows that exist and reports which kernel would have left them (the stored rows are -2 .. n0 + 1; -4 .. n0 + 3 on the cut sides of
 * the fused 2-D step). family (binary: the stage kernels of the uniform-depth mesh, whole or in bands): */
enum { MH_ROWS_EULER2D = 0, MH_ROWS_EULER2D_FUSED = 1, MH_ROWS_CLOUD = 2, MH_ROWS_CLOUD_FUSED = 3, MH_ROWS_EULER3D_STRICT = 4, MH_ROWS_EULER3D_FAST = 5,
       MH_ROWS_BINARY_STRICT = 6, MH_ROWS_BINARY_FAST = 7 };
/* MH_E_STATE from a product build (no guard compiled in); lo_hi = {INT_MAX, INT_MIN} when nothing was requested */
/* {chunk rows, chunk rows of the second segment, chunks of the first segment, chunks} of the LAST one-launch RK2 step this process issued
 * (family 1: 2-D Euler, 3: cloud): the tests of the tapered interior launch (a slab with neighbours: csrc/euler2d_fused.hip, TAPER) read it */
int  mh_debug_last_fused_cut(int family, int32_t out[4]);
int  mh_debug_row_range(int family, int32_t lo_hi[2], int reset);

#ifdef __cplusplus
}
#endif
#endif
