// `cloud` sub-program stage on gfx950: 2-D axisymmetric spherical-polar SRHD
// (BASELINE config 4). Replaces one evaluation of CloudProblem::advance
// (src/subprog_cloud.cpp:511-584) and, with stage_weight != 1, the RK combine
// of next_solution (:682-695):
//     p0 = recover_primitive(u0 / dv)                                   :533
//     s0 = spherical_geometry_source_terms(p0, r_c, theta_c) * dv       :534
//     lr = diff_r( hlle_r(PL, PR) * (-dAr) ),  lq = diff_q( hlle_q(PL, PR) * (-dAq) )   :572-573
//     u1 = u0 + (lr + lq + s0) * dt                                     :574
// Geometry (:260-290): dAr = ((r_i r_i) dmu_j) 2 pi, dAq = ((r_c dr) sin q_j) 2 pi,
// dv = (((r^3_{i+1} - r^3_i) dmu_j) 2 pi) / 3. The theta-dependent factors
// (dmu_j = -cos q_{j+1} - -cos q_j, sin q_j, cot theta_c) come from the host,
// which evaluates them with the same libm as the reference; products are formed
// here in the reference's order.
// Boundary conditions: inner radial ghost = nozzle-inflow PRIMITIVES (:466-493),
// outer = copy of the last row (:503-509), radial ghost slopes = edge slope * 0
// (extend_zeros on G, :563); polar: slopes of the two pole cells = neighbour's
// slope * 0, and the pole-face fluxes = neighbouring face flux * 0 (:573).
//
// Same wave-marching structure as euler2d.hip: a wavefront owns 60 polar columns
// and marches radially; radial fluxes are reused between iterations, polar
// neighbours move through DPP wave shifts. Conserved variables are
// cell-integrated; device layout as in include/mara_hip.h.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include "launch.hpp"
#include "status_device.hpp"
#include "euler_device.hpp"
#include "srhd_device.hpp"
#include "srhd_device_fast.hpp"
#include "cloud_rows.hpp"
#include "row_check.hpp"

namespace mh {

static constexpr int CWAVE = 64;
static constexpr int CHALO = 2;
static constexpr int CSTRIP = CWAVE - 2 * CHALO;
static constexpr int CWAVES_PER_BLOCK = 4;

template<class S, bool PLM, bool COMBINE>
__global__ __launch_bounds__(CWAVE * CWAVES_PER_BLOCK, S::min_waves_per_simd)
void cloud_stage_kernel(CloudParams p)
{
    int b = blockIdx.x;
    {
        // XCD-aware order; each XCD's share ends on the short waves of the graded tail (as euler2d.hip)
        const int per_xcd = gridDim.x >> 3, tail = p.tail_blocks_per_xcd, body = per_xcd - tail;
        if (b < per_xcd * 8)
        {
            const int x = b & 7, s = b >> 3;
            b = s < body ? x * body + s : 8 * body + x * tail + (s - body);
        }
    }
    const int w = __builtin_amdgcn_readfirstlane(b * CWAVES_PER_BLOCK + (int) (threadIdx.x >> 6));   // wave-uniform -> scalar registers
    if (w >= p.nstrips * p.nchunks) return;
    const int lane = threadIdx.x & 63;
    constexpr bool PL = S::planar;                      // no azimuthal component anywhere (the stepper has checked field and nozzle row)
    const int chunk = w / p.nstrips;
    const int strip = w - chunk * p.nstrips;
    int r0 = p.row_begin + chunk * p.chunk_rows;
    int r1 = min(r0 + p.chunk_rows, p.row_end);
    if (chunk >= p.nchunks_a)
    {
        r0 = p.row_begin2 + (chunk - p.nchunks_a) * p.chunk_rows2;
        r1 = min(r0 + p.chunk_rows2, p.row_end2);
    }

    const int col = strip * CSTRIP - CHALO + lane;
    const int jc = min(max(col, 0), p.n1 - 1);
    const bool writes = lane >= CHALO && lane < CWAVE - CHALO && col < p.n1;
    const bool pole_lo = col == 0, pole_hi = col == p.n1 - 1;

    const srhd::Gamma g = srhd::make_gamma(p.gamma);
    const Recip three = make_recip(3.0, 1.0);
    const double tfloor = p.tfloor;
    const typename S::Limiter lim = S::limiter(p.theta);
    const ColGeom cg = {p.dmu[jc], p.sinq[jc], p.sinq[jc + 1], p.cotq[jc]};
    // MH_ARITH_FAST: per-column factors of the host's table (dmu 2 pi, dmu 2 pi / 3, its inverse, sin q_j 2 pi, sin q_{j+1} 2 pi)
    double col_ar = 0.0, col_dv = 0.0, col_inv_dv = 0.0, col_aq_lo = 0.0, col_aq_hi = 0.0;
    if constexpr (S::table_geometry)
    {
        const double* cf = p.colf + 8L * jc;
        col_ar = cf[0]; col_dv = cf[1]; col_inv_dv = cf[2]; col_aq_lo = cf[3]; col_aq_hi = cf[4];
    }
    // MH_ARITH_FAST at the poles (extend_zeros on the polar slopes and fluxes, :563 / :570): the pole lanes' own constants are zero - the face
    // area of the pole face and the slope's weight in the face states - instead of selects on five slopes and ten fluxes per row
    typename S::Limiter lim_polar = lim;
    if constexpr (! S::exact_zero_products)
    {
        if (pole_lo) col_aq_lo = 0.0;
        if (pole_hi) col_aq_hi = 0.0;
        if (pole_lo || pole_hi) lim_polar.half_theta = 0.0;
    }
    // geometry of the cells of global row i: the reference's products in its order (strict), or per-row x per-column factors (fast)
    auto cell_geometry = [&] (int i) -> CellGeom
    {
        CellGeom c;
        if constexpr (S::table_geometry)
        {
            const double* rf = p.rowf + 8L * i;          // wave-uniform: scalar loads
            c.dv = rf[2] * col_dv;
            c.inv_dv = rf[3] * col_inv_dv;
            c.nAr_lo = -(rf[0] * col_ar);
            c.nAr_hi = -(rf[1] * col_ar);
            c.nAq_lo = -(rf[4] * col_aq_lo);
            c.nAq_hi = -(rf[4] * col_aq_hi);
            c.rc = rf[5];
            c.inv_rc = rf[6];
        }
        else
        {
            const RowGeom rg = row_geometry(p.rv, i);
            c.dv = cell_volume(rg, cg, three);
            c.inv_dv = 0.0;
            c.nAr_lo = -(rg.rr_lo * cg.dmu * 2 * M_PI);
            c.nAr_hi = -(rg.rr_hi * cg.dmu * 2 * M_PI);
            c.nAq_lo = -(rg.rcdr * cg.sin_lo * 2 * M_PI);
            c.nAq_hi = -(rg.rcdr * cg.sin_hi * 2 * M_PI);
            c.rc = rg.rc;
            c.inv_rc = 0.0;
        }
        return c;
    };

    const long row_stride = p.row_stride, plane = p.plane_stride;
    const double* in = p.u_in;
    const int rows_hi = p.n0 + 1;                     // the rows that exist: -2 .. n0 + 1 (row_check.hpp)
    auto row_off = [row_stride, rows_hi] (int r) { (void) rows_hi; return (long) (MH_ROW(r, -CHALO, rows_hi) + CHALO) * row_stride; };
    const unsigned jc8 = (unsigned) jc * 8u, col8 = (unsigned) (writes ? col : 0) * 8u;
    StatusAcc acc;        // error contract (status_device.hpp): recover_primitive's status bits + the flat index r * n1 + col of the cell

    // the five conserved variables of a stored row (clamped to the stored range: rows -2 .. n0+1)
    auto load_raw = [&] (int r) -> State5
    {
        const int rr = min(max(r, -CHALO), p.n0 + CHALO - 1);
        return cloud_load_row<PL>(in + row_off(rr), plane, jc8);
    };
    // MH_ARITH_FAST: the conserved values of rows r, r+1, r+2 wait for their update in a per-wave LDS ring (no barrier: private to the
    // wave) instead of being read a second time - that second read misses in L2 and made the launch's HBM traffic 1.6 x / 1.4 x algorithmic
    __shared__ double own_rows[S::lds_row_ring ? CWAVES_PER_BLOCK : 1][3][5][CWAVE];
    const int wave_in_block = (int) (threadIdx.x >> 6);
    auto ring_put = [&] (int slot, const State5& raw)
    {
        if constexpr (S::lds_row_ring)
        {
#pragma unroll
            for (int q = 0; q < 5; ++q) if (S::live(q)) own_rows[wave_in_block][slot][q][lane] = raw[q];
        }
    };
    auto ring_get = [&] (int slot) -> State5
    {
        State5 U;
#pragma unroll
        for (int q = 0; q < 5; ++q) U[q] = S::live(q) ? own_rows[S::lds_row_ring ? wave_in_block : 0][slot][q][lane] : 0.0;
        return U;
    };
    // primitive of a stored row (real row, or a ghost row received from the neighbouring slab) from its loaded variables
    auto prim_of_raw = [&] (int r, const State5& raw) -> State5
    {
        const CellGeom c = cell_geometry(p.row_offset + r);
        double x[5];
#pragma unroll
        for (int q = 0; q < 5; ++q) x[q] = raw[q];
        S::to_density(x, c.dv, c.inv_dv);
        State5 U, P;
#pragma unroll
        for (int q = 0; q < 5; ++q) U[q] = x[q];
        const int st = S::c2p(U, g, tfloor, P);
        if (__any(st != 0))       // where the reference throws (physics_srhd.hpp:430-449); a scalar branch never taken in a healthy run
        {
            if (writes && st != 0 && r >= 0 && r < p.n0) acc.note((uint32_t) st, (uint32_t) r * (uint32_t) p.n1 + (uint32_t) col);
        }
        return P;
    };
    auto prim_of_row = [&] (int r) -> State5 { return prim_of_raw(r, load_raw(r)); };
    // primitive of row r with the physical boundary conditions applied; `raw` = its variables if already loaded
    auto prim_bc_raw = [&] (int r, const State5& last, const State5& raw) -> State5
    {
        if (r < 0 && p.bc_lo0 != MH_BC_EXTERNAL)
        {
            State5 P;
#pragma unroll
            for (int q = 0; q < 5; ++q) P[q] = S::live(q) ? p.inflow[(long) q * p.n1 + jc] : 0.0;
            return P;
        }
        if (r >= p.n0 && p.bc_hi0 != MH_BC_EXTERNAL) return last;
        return prim_of_raw(r, raw);
    };
    auto prim_bc = [&] (int r, const State5& last) -> State5
    {
        if (r < 0 && p.bc_lo0 != MH_BC_EXTERNAL)
        {
            State5 P;
#pragma unroll
            for (int q = 0; q < 5; ++q) P[q] = S::live(q) ? p.inflow[(long) q * p.n1 + jc] : 0.0;
            return P;
        }
        if (r >= p.n0 && p.bc_hi0 != MH_BC_EXTERNAL) return last;     // zero-gradient outer: copy of the last real row
        return prim_of_row(r);
    };
    const bool phys_lo = p.bc_lo0 != MH_BC_EXTERNAL, phys_hi = p.bc_hi0 != MH_BC_EXTERNAL;

    // ---- prologue
    State5 P0, P1, G0, Fx_lo;
    {
        State5 dummy = {};
        const State5 Pb = prim_bc(r0 - 1, dummy);
        const State5 raw0 = load_raw(r0), raw1 = load_raw(r0 + 1);
        ring_put(0, raw0);
        ring_put(1, raw1);
        P0 = prim_of_raw(r0, raw0);
        P1 = prim_bc_raw(r0 + 1, P0, raw1);
        if constexpr (PLM)
        {
            G0 = S::plm(Pb, P0, P1, lim);
            State5 Gb;
            if (r0 == 0 && phys_lo) Gb = times_zero_p<PL>(G0);                       // extend_zeros on G
            else                    Gb = S::plm(prim_bc(r0 - 2, dummy), Pb, P0, lim);
            Fx_lo = S::template hlle<0>(S::plus(Pb, Gb, lim), S::minus(P0, G0, lim), g);
        }
        else
        {
            Fx_lo = S::template hlle<0>(Pb, P0, g);
        }
    }

    // software pipeline: row r+2 (needed now, for the primitives of the face after next) was requested one iteration ago;
    // row r+3 is requested at the top of the iteration
    State5 Uahead = load_raw(r0 + 2);
    int slot = 0;                                       // ring slot of row r; row r + 2 goes to slot + 2 (mod 3)
    for (int r = r0; r < r1; ++r)
    {
        const State5 Unext = load_raw(r + 3);

        // ---- radial face r+1/2
        const State5 P2 = prim_bc_raw(r + 2, P1, Uahead);
        ring_put(slot == 0 ? 2 : slot - 1, Uahead);
        Uahead = Unext;
        State5 G1, Fx_hi;
        if constexpr (PLM)
        {
            if (r + 1 == p.n0 && phys_hi) G1 = times_zero_p<PL>(G0);
            else                          G1 = S::plm(P0, P1, P2, lim);
            Fx_hi = S::template hlle<0>(S::plus(P0, G0, lim), S::minus(P1, G1, lim), g);
        }
        else
        {
            Fx_hi = S::template hlle<0>(P0, P1, g);
        }

        // ---- polar faces: this lane computes its LEFT face; pole faces carry the neighbouring flux times zero
        State5 Fy_lo, Fy_hi;
        if constexpr (PLM)
        {
            const State5 Graw = S::plm(dpp_left_p<PL>(P0), P0, dpp_right_p<PL>(P0), lim);
            State5 Gy;
            if constexpr (S::exact_zero_products)
            {
                // pole cells: the neighbour's slope times zero (extend_zeros, :563) - NaN and the sign of zero propagate as upstream
                const State5 Gl = dpp_left_p<PL>(Graw), Gr = dpp_right_p<PL>(Graw);
#pragma unroll
                for (int q = 0; q < 5; ++q) Gy[q] = ! S::live(q) ? 0.0 : (pole_lo ? Gr[q] * 0.0 : (pole_hi ? Gl[q] * 0.0 : Graw[q]));
            }
            else
            {
                Gy = Graw;          // (a pole lane's lim_polar gives it no weight)
            }
            const State5 SL = dpp_left_p<PL>(S::plus(P0, Gy, lim_polar));
            Fy_lo = S::template hlle<1>(SL, S::minus(P0, Gy, lim_polar), g);
        }
        else
        {
            Fy_lo = S::template hlle<1>(dpp_left_p<PL>(P0), P0, g);
        }
        Fy_hi = dpp_right_p<PL>(Fy_lo);
        if constexpr (S::exact_zero_products)
        {
            if (pole_lo) Fy_lo = times_zero_p<PL>(Fy_hi);
            if (pole_hi) Fy_hi = times_zero_p<PL>(Fy_lo);
        }
        // (MH_ARITH_FAST: the pole faces' area factors are zero, see lim_polar above)

        // ---- geometry, source terms, update
        // the row's own conserved values once more (no register ring: three waves per SIMD). FAST: requested as ONE group ahead of the
        // geometry and source arithmetic - left to itself the scheduler, at the 168-register limit, may split the group into load - wait - use
        // triples (measured: +12 % on the launch); any earlier - ahead of the polar Riemann problem - and 19 registers spill. STRICT (176-186
        // registers, two waves per SIMD either way): the compiler's own placement is 3 % faster than this one.
        State5 U0, Ubase;
        auto load_own_row = [&] () __attribute__((always_inline))
        {
            if constexpr (S::lds_row_ring) U0 = ring_get(slot);
            else                           U0 = cloud_load_row<PL>(in + row_off(r), plane, jc8);
            if constexpr (COMBINE) Ubase = cloud_load_row<PL>(p.u_base + row_off(r), plane, jc8);
        };
        if constexpr (S::group_own_row_loads)
        {
            load_own_row();
            __builtin_amdgcn_sched_barrier(0);
        }
        const CellGeom c = cell_geometry(p.row_offset + r);
        const State5 Src = S::source(P0, c.rc, c.inv_rc, cg.cot, g);
        if constexpr (! S::group_own_row_loads) load_own_row();

        State5 Un;
#pragma unroll
        for (int q = 0; q < 5; ++q)
        {
            if (! S::live(q)) { Un[q] = 0.0; continue; }      // planar: u0 + ((+0) + (+0) + (+-0)) dt and the combination of two +0 are +0
            const double u1 = S::update(U0[q], Fx_lo[q], Fx_hi[q], Fy_lo[q], Fy_hi[q], c.nAr_lo, c.nAr_hi, c.nAq_lo, c.nAq_hi, Src[q], c.dv, p.dt);
            if constexpr (COMBINE) Un[q] = S::combine(Ubase[q], u1, p.weight);
            else                   Un[q] = u1;
        }
        if (writes) cloud_store_row(p.u_out + row_off(r), plane, col8, Un);

        P0 = P1; P1 = P2;
        if constexpr (PLM) G0 = G1;
        Fx_lo = Fx_hi;
        slot = slot == 2 ? 0 : slot + 1;
    }

    acc.commit(p.status);
}

hipError_t cloud_stage_launch(const mh_cloud_desc* d, const double* geom_dev, const double* inflow_dev,
                              const double* u_in, const double* u_base, double* u_out, double dt, double weight,
                              int row_begin, int row_end, int32_t* status, hipStream_t stream)
{
    CloudParams p;
    p.u_in = u_in; p.u_base = u_base; p.u_out = u_out;
    p.n0 = d->nr; p.n1 = d->nq;
    // geom_dev: rv[nr_global+1] | dmu[nq] | sinq[nq+1] | cotq[nq]
    p.rv = geom_dev;
    p.dmu = p.rv + d->nr_global + 1;
    p.sinq = p.dmu + d->nq;
    p.cotq = p.sinq + d->nq + 1;
    p.rowf = p.cotq + d->nq;
    p.colf = p.rowf + 8L * d->nr_global;
    p.inflow = inflow_dev;
    p.status = status;
    p.plane_stride = d->nq;
    p.row_stride = 5L * d->nq;
    p.row_offset = d->row_offset;
    p.row_begin = row_begin; p.row_end = row_end;
    p.chunk_rows = d->chunk_rows > 0 ? d->chunk_rows : 32;
    p.nstrips = (p.n1 + CSTRIP - 1) / CSTRIP;
    // graded tail (see euler2d.hip): the last rows of a large launch go to short waves, so that the ragged end of the last residency
    // round lasts a short wave's duration. The descriptor's tail_rows / tail_chunk_rows override (tail_rows < 0 = off).
    p.row_begin2 = p.row_end2 = row_end;
    p.chunk_rows2 = p.chunk_rows;
    const bool tail_given = d->tail_rows != 0;          // explicit: applies to any launch with more rows than the tail (tests on small grids)
    if (tail_given || d->chunk_rows == 0)
    {
        int tail_rows = 512, tail_chunk = 8;
        if (tail_given) { tail_rows = d->tail_rows; tail_chunk = d->tail_chunk_rows > 0 ? d->tail_chunk_rows : 8; }
        if (tail_rows > 0 && tail_chunk >= 2 && row_end - row_begin >= (tail_given ? tail_rows + 1 : 4 * tail_rows))
        {
            p.row_begin2 = row_end - tail_rows;
            p.row_end2 = row_end;
            p.row_end = p.row_begin2;
            p.chunk_rows2 = tail_chunk;
        }
    }
    p.nchunks_a = (p.row_end - row_begin + p.chunk_rows - 1) / p.chunk_rows;
    p.nchunks = p.nchunks_a + (p.row_end2 - p.row_begin2 + p.chunk_rows2 - 1) / p.chunk_rows2;
    p.tail_blocks_per_xcd = p.chunk_rows2 != p.chunk_rows ? (int) (((long) p.nstrips * (p.nchunks - p.nchunks_a) / CWAVES_PER_BLOCK) >> 3) : 0;
    p.bc_lo0 = d->bc_lo0; p.bc_hi0 = d->bc_hi0;
    p.gamma = d->gamma; p.theta = d->plm_theta; p.tfloor = d->temperature_floor;
    p.dt = dt; p.weight = weight;
    if (p.nchunks <= 0) return hipSuccess;
    const int nwaves = p.nstrips * p.nchunks;
    const dim3 grid((nwaves + CWAVES_PER_BLOCK - 1) / CWAVES_PER_BLOCK), block(CWAVE * CWAVES_PER_BLOCK);
    const bool plm = d->plm_theta >= 0.0, combine = weight != 1.0;
    // planar (mh_cloud_desc.planar > 0: the stepper has checked the field and the nozzle row - api.hip, slab.hip): the PLM kernels without
    // the azimuthal component
    if (d->planar > 0 && plm && d->arith == MH_ARITH_FAST)
    {
        if (combine)              hipLaunchKernelGGL((cloud_stage_kernel<SrhdFastPlanar, true, true>), grid, block, 0, stream, p);
        else                      hipLaunchKernelGGL((cloud_stage_kernel<SrhdFastPlanar, true, false>), grid, block, 0, stream, p);
    }
    else if (d->planar > 0 && plm)
    {
        if (combine)              hipLaunchKernelGGL((cloud_stage_kernel<SrhdStrictPlanar, true, true>), grid, block, 0, stream, p);
        else                      hipLaunchKernelGGL((cloud_stage_kernel<SrhdStrictPlanar, true, false>), grid, block, 0, stream, p);
    }
    else if (d->arith == MH_ARITH_FAST)
    {
        if (plm && combine)       hipLaunchKernelGGL((cloud_stage_kernel<SrhdFast, true, true>), grid, block, 0, stream, p);
        else if (plm)             hipLaunchKernelGGL((cloud_stage_kernel<SrhdFast, true, false>), grid, block, 0, stream, p);
        else if (combine)         hipLaunchKernelGGL((cloud_stage_kernel<SrhdFast, false, true>), grid, block, 0, stream, p);
        else                      hipLaunchKernelGGL((cloud_stage_kernel<SrhdFast, false, false>), grid, block, 0, stream, p);
    }
    else
    {
        if (plm && combine)       hipLaunchKernelGGL((cloud_stage_kernel<SrhdStrict, true, true>), grid, block, 0, stream, p);
        else if (plm)             hipLaunchKernelGGL((cloud_stage_kernel<SrhdStrict, true, false>), grid, block, 0, stream, p);
        else if (combine)         hipLaunchKernelGGL((cloud_stage_kernel<SrhdStrict, false, true>), grid, block, 0, stream, p);
        else                      hipLaunchKernelGGL((cloud_stage_kernel<SrhdStrict, false, false>), grid, block, 0, stream, p);
    }
    return hipGetLastError();
}

// row-range guard (row_check.hpp): what this translation unit's kernels asked for; false in product builds
MH_ROW_RANGE_READER(rows_requested_cloud)

} // namespace mh
