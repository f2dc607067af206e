// 3-D uniform-cartesian Euler Runge-Kutta stage for gfx950 (BASELINE config 5).
//
// Same scheme and reference citations as euler2d.hip with a third axis:
//     u1 = u0 - (diff0(Fx)*(dt/dx) + diff1(Fy)*(dt/dy) + diff2(Fz)*(dt/dz))
// (oracle/ref_drivers/euler_cart_ref.cpp, rank 3; cloud::advance composition,
// src/subprog_cloud.cpp:511-584; RK combine :682-695).
//
// 2.5-D plane marching. A workgroup of 8 wavefronts owns a tile of 8 axis-1
// rows x 60 axis-2 columns and marches along axis 0 (STRICT since round 5: 4 wavefronts,
// 4 rows - two independent workgroups per CU, A::tile_rows):
//   * axis 0 (march): primitives of planes i..i+2, the slope of plane i and the
//     flux through face i-1/2 live in registers (three-slot rings, loop unrolled
//     x3); each flux is computed once;
//   * axis 2 (lanes): DPP wave shifts, exactly as in the 2-D kernel;
//   * axis 1 (across waves): ONE LDS exchange per plane - the primitives of the
//     tile's rows and of two rows on either side (fetched by four helper waves),
//     double-buffered by plane parity - and one workgroup barrier. Each wave then
//     computes both of its axis-1 face fluxes itself (a fourth Riemann problem per
//     cell; identical code on identical inputs on both sides of a face). An earlier
//     version exchanged face states and fluxes as well: three barriers per plane
//     and an edge wave with a ninth face left the VALU 57 % busy (rocprofv3 SQ
//     counters, profiles/r01); trading redundant arithmetic for two barriers and a
//     balanced workgroup is faster on this part.
// Conserved planes are read once (+ halo re-reads that hit in L2) and written
// once; nothing else touches HBM. Algorithmic bytes: 80 / 120 B per cell per
// stage as in 2-D. Device layout: include/mara_hip.h with row_pitch = n1*n2.
#include "euler3d_kernel.hpp"

namespace mh {

// axis-1 rows of a work item's tile: the arithmetic's choice (euler_device_fast.hpp: STRICT four - two workgroups per CU -, FAST eight)
int euler3d_tile_rows(const mh_euler_cart_desc* d) { return d->arith == MH_ARITH_FAST ? FastArith::tile_rows : StrictArith::tile_rows; }

void euler3d_tiling(const mh_euler_cart_desc* d, int* ntiles1, int* nstrips)
{
    const int rows = euler3d_tile_rows(d);
    *ntiles1 = (d->n[1] + rows - 1) / rows;
    *nstrips = (d->n[2] + STRIP3 - 1) / STRIP3;
}

// Planes per work item of a whole-field launch (descriptor's chunk_rows == 0). A work item occupies a CU (its LDS tile and 256-register waves
// leave room for one workgroup), so a launch runs in ceil(items / CUs) residency rounds, each as long as a chunk plus the pipeline fill
// (~1.3 plane-steps: four conversions, two slopes, one Riemann problem before the first update). The chunk with the lowest
// rounds x (chunk + 1.3) wins; measured at 512^3 FAST (profiles/r04/ab_3d_chunks.jsonl, ms per RK2 step - the model's ratio to the best in
// brackets): 16 planes 7.99 (1.07), 32: 7.69 (1.03, the fixed default of rounds 1 - 3), 43: 7.65 (1.03), 64: 7.51 (1.01), 86: 7.71 (1.05),
// 128: 7.46 (1.00), 256: 8.20 (1.10). Results do not depend on the cut (tests/test_gpu_euler3d.py).
static int euler3d_default_chunk(int planes, long items_per_layer, int tile_rows)
{
    const int cus = device_cu_count() * (tile_rows == 4 ? 2 : 1);          // resident work items: two four-row workgroups share a CU
    int best = planes < 32 ? planes : 32;
    double best_cost = 0.0;
    const int longest = planes < 128 ? planes : 128;          // (longer chunks were not measured; they balance worse over CUs whose speeds differ)
    for (int chunk = 8; chunk <= longest || chunk == 8; ++chunk)
    {
        const int c = chunk < longest ? chunk : longest;
        const long nchunks = (planes + c - 1) / c;
        if (nchunks > 1 && (planes + nchunks - 1) / nchunks != c) continue;          // only the even cuts
        const long rounds = (nchunks * items_per_layer + cus - 1) / cus;
        const double cost = (double) rounds * (c + 1.3);
        if (best_cost == 0.0 || cost < best_cost * 0.995) { best_cost = cost; best = c; }
        if (c == longest) break;
    }
    return best > 0 ? best : 1;
}

hipError_t euler3d_stage_launch_boxes(const mh_euler_cart_desc* d, const Euler3dLayout& lay, const Euler3dBox* boxes, int nboxes,
                                      const double* u_in, const double* u_base, double* u_out, double dt, double weight,
                                      int32_t* status, hipStream_t stream)
{
    if (nboxes < 0 || nboxes > MAX_BOXES) return hipErrorInvalidValue;
    Stage3dParams p;
    p.u_in = u_in; p.u_base = u_base; p.u_out = u_out; p.status = status;
    p.n0 = d->n[0]; p.n1 = d->n[1]; p.n2 = d->n[2];
    p.g1 = lay.g1; p.g2 = lay.g2;
    p.pitch2 = p.n2 + 2 * p.g2;
    p.plane_stride = (long) (p.n1 + 2 * p.g1) * p.pitch2;
    p.row_stride = 5L * p.plane_stride;
    int ntiles1, nstrips;
    euler3d_tiling(d, &ntiles1, &nstrips);
    p.chunk_rows = d->chunk_rows > 0 ? d->chunk_rows : 32;
    if (d->chunk_rows == 0 && nboxes == 1 && boxes[0].r1 > boxes[0].r0)          // (a block's shell of several boxes keeps the short chunks)
        p.chunk_rows = euler3d_default_chunk(boxes[0].r1 - boxes[0].r0, (long) (boxes[0].t1 - boxes[0].t0) * (boxes[0].s1 - boxes[0].s0), euler3d_tile_rows(d));
    int nblocks = 0;
    p.nboxes = 0;
    for (int k = 0; k < nboxes; ++k)
    {
        const Euler3dBox& bx = boxes[k];
        if (bx.r0 < 0 || bx.r1 > p.n0 || bx.t0 < 0 || bx.t1 > ntiles1 || bx.s0 < 0 || bx.s1 > nstrips) return hipErrorInvalidValue;
        if (bx.r1 <= bx.r0 || bx.t1 <= bx.t0 || bx.s1 <= bx.s0) continue;          // empty box
        const int nchunks = (bx.r1 - bx.r0 + p.chunk_rows - 1) / p.chunk_rows;
        p.box[p.nboxes] = bx;
        p.first_block[p.nboxes] = nblocks;
        nblocks += nchunks * (bx.t1 - bx.t0) * (bx.s1 - bx.s0);
        ++p.nboxes;
    }
    for (int k = p.nboxes; k <= MAX_BOXES; ++k) p.first_block[k] = nblocks;
    p.bc_lo0 = d->bc_lo0; p.bc_hi0 = d->bc_hi0;
    p.bc_lo1 = lay.bc_lo1; p.bc_hi1 = lay.bc_hi1; p.bc_lo2 = lay.bc_lo2; p.bc_hi2 = lay.bc_hi2;
    p.gamma = d->gamma; p.theta = d->plm_theta;
    p.cx = dt / d->dl[0]; p.cy = dt / d->dl[1]; p.cz = dt / d->dl[2];
    p.weight = weight;
    if (nblocks <= 0) return hipSuccess;

    const bool plm = d->plm_theta >= 0.0, combine = weight != 1.0;
    const int key = (d->arith == MH_ARITH_FAST ? 8 : 0) | (d->riemann == MH_RIEMANN_HLLC ? 4 : 0) | (plm ? 2 : 0) | (combine ? 1 : 0);
    switch (key)
    {
        case 0:  return launch3<StrictArith, 0, false, false>(p, nblocks, stream);
        case 1:  return launch3<StrictArith, 0, false, true >(p, nblocks, stream);
        case 2:  return launch3<StrictArith, 0, true,  false>(p, nblocks, stream);
        case 3:  return launch3<StrictArith, 0, true,  true >(p, nblocks, stream);
        case 4:  return launch3<StrictArith, 1, false, false>(p, nblocks, stream);
        case 5:  return launch3<StrictArith, 1, false, true >(p, nblocks, stream);
        case 6:  return launch3<StrictArith, 1, true,  false>(p, nblocks, stream);
        case 7:  return launch3<StrictArith, 1, true,  true >(p, nblocks, stream);
        default: return euler3d_launch_fast(key & 7, p, nblocks, stream);
    }
    return hipErrorInvalidValue;
}

// whole transverse extent, rows [row_begin, row_end): the single-device / axis-0-slab form (no stored transverse ghosts)
hipError_t euler3d_stage_launch(const mh_euler_cart_desc* d, const double* u_in, const double* u_base, double* u_out,
                                double dt, double weight, int row_begin, int row_end, int32_t* status, hipStream_t stream)
{
    Euler3dLayout lay;
    lay.bc_lo1 = lay.bc_hi1 = lay.bc_lo2 = lay.bc_hi2 = d->bc_transverse;
    int ntiles1, nstrips;
    euler3d_tiling(d, &ntiles1, &nstrips);
    const Euler3dBox all = {row_begin, row_end, 0, ntiles1, 0, nstrips};
    return euler3d_stage_launch_boxes(d, lay, &all, 1, u_in, u_base, u_out, dt, weight, status, stream);
}

// row-range guard (row_check.hpp): what this translation unit's kernels asked for; false in product builds
MH_ROW_RANGE_READER(rows_requested_euler3d)

} // namespace mh
