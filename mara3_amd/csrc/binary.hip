// `binary` sub-program stage on gfx950: 2-D locally isothermal circumbinary disk on a uniform-depth
// block tree (BASELINE config 3). Replaces one evaluation of
// binary::advance_u (src/subprog_binary_scheme.cpp:790-904) - or, with angmom_form set, binary::advance_q (:906-1020) - and, with stage_weight != 1, the conserved
// part of the RK combine s0 * 1/2 + s2 * 1/2 (:1033-1069, src/subprog_binary.cpp:264-277):
//     p0      = iso2d::recover_primitive(u0)                                          :802, physics_iso2d.hpp:351
//     gx, gy  = plm_gradient(p0 on axis 0 / 1, theta) / spacing                        :794-800
//     fhat_x  = (hlle(pl + gl h/2, pr - gr h/2, cs2(xf)) + viscous_flux) * dy          :472-516, :268-293, :220-262
//     fhat_y  = likewise * dx
//     u1      = u0 - (diff_x fhat_x + diff_y fhat_y) * dt / dA + s                     :568-587
//     s       = gravity(2) + sink(2) + buffer + floor                                   :345-411
// and the ten source-term totals of :390-408. On a tree whose every node is refined, the blocks tile a
// periodic n x n tensor-product mesh (block neighbours wrap, core_tree.hpp:203-204), no flux correction
// applies (:614-720 only acts at refinement jumps), and the only trace of the blocks in the arithmetic is
// (i) the position of a block's OUTER faces, which are its own vertices - so the two sides of the periodic
// seam are evaluated separately, at x = -R and x = +R - and (ii) the grouping of the totals: work_done_on
// is a nonlinear function of each block's sink sums (:356-365, :409-410).
//
// Kernels:
//  * binary_stage_kernel  - the wave-marching stencil of euler2d.hip (one wavefront = 60 columns + 2 halo
//    lanes per side, marching along axis 0; axis-0 face fluxes reused between iterations, axis-1 neighbours
//    through DPP wave shifts), extended by the transverse slopes the viscous flux needs, the position-
//    dependent sound speed / viscosity, the source terms, and per-wave partial sums of the 8 totals that
//    are linear in the cells (torque, force, ejected mass / angular momentum).
//  * binary_sink_kernel   - one workgroup per tree block: the sink sums of that block and, from them, the
//    block's work_done_on. Blocks further than the range of exp(-a2) from both bodies (a2 > 750: the rate
//    underflows to exactly 0 there, in glibc and here) contribute exact zeros and return at once.
//  * binary_reduce_kernel - fixed-order reduction of both partial sets into the 18 totals (deterministic).
//  * binary_maxw_kernel   - max over cells of primitive_t::max_wavespeed(cs2(x_c)) for maximum_timestep (:1107-1126).
//
// Arithmetic: reference operation order with IEEE division and sqrt and no FMA contraction (as STRICT in
// euler2d.hip), EXCEPT the three libm calls the reference makes per cell/face: pow(x, 1/2) is evaluated as
// sqrt(x), pow(x, 3/2) as x * sqrt(x), exp / tanh by the device math library. These differ from glibc in the
// last bit or two, so parity with the reference is to the north-star tolerance (L1 <= 1e-12), not bit-exact.
//
// Algorithmic HBM bytes per cell per stage: read u0 (24) + u_init (24) + buffer rate (8) + write u1 (24) = 80;
// the second RK stage also reads the step-start field (+24) = 104. RK2: 184 B per zone-update.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>
#include "binary_kernel.hpp"

namespace mh {

// ---- per-block sink sums and work --------------------------------------------------------------------------------
struct BinarySinkParams
{
    const double* u_in;
    const double* xv;
    const double* yv;
    double*       block_out;   // [nb * nb][NBLK]
    int    n, bs, nb, qform;   // nb: tree blocks per row of blocks (= n / bs); xv points at this band's first vertex
    double dt;
    BinaryConsts c;
};

__global__ __launch_bounds__(256)
void binary_sink_kernel(BinarySinkParams p)
{
    const int blk = blockIdx.x;
    const int bi = blk / p.nb, bj = blk - bi * p.nb;
    const int n = p.n, bs = p.bs;
    const BinaryConsts& c = p.c;
    double* out = p.block_out + (long) blk * NBLK;

    // out of range of both sinks? (distance from the body to the block's rectangle)
    const double x0 = p.xv[bi * bs], x1 = p.xv[(bi + 1) * bs], y0 = p.yv[bj * bs], y1 = p.yv[(bj + 1) * bs];
    bool near = false;
    for (int b = 0; b < 2; ++b)
    {
        const double bx = c.body[5 * b + 1], by = c.body[5 * b + 2];
        const double ex = fmax(fmax(x0 - bx, bx - x1), 0.0), ey = fmax(fmax(y0 - by, by - y1), 0.0);
        if ((ex * ex + ey * ey) / c.s2 / 2.0 < 760.0) near = true;
    }
    if (!near)
    {
        if (threadIdx.x < NBLK) out[threadIdx.x] = 0.0;
        return;
    }

    double acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};     // per body: mass, Lz, px, py of s_sink * dA
    for (int idx = threadIdx.x; idx < bs * bs; idx += 256)
    {
        const int i = bi * bs + idx / bs, j = bj * bs + idx % bs;
        const double xc = (p.xv[i] + p.xv[i + 1]) * 0.5, yc = (p.yv[j] + p.yv[j + 1]) * 0.5;
        const double dA = (p.xv[i + 1] - p.xv[i]) * (p.yv[j + 1] - p.yv[j]);
        const double* u = p.u_in + (long) (i + BHALO) * 3 * n + j;
        const double u0[3] = {u[0], u[n], u[2 * n]};
        for (int b = 0; b < 2; ++b)
        {
            const double d0 = xc - c.body[5 * b + 1], d1 = yc - c.body[5 * b + 2];
            const double a2 = (d0 * d0 + d1 * d1) / c.s2 / 2.0;
            const double rate = c.sink_rate * (a2 < 750.0 ? exp(-a2) : 0.0);
            double s[3];
            for (int q = 0; q < 3; ++q) s[q] = -u0[q] * rate * p.dt;
            acc[4 * b + 0] = acc[4 * b + 0] + s[0] * dA;
            if (p.qform)
            {
                // totals of source_terms_q :447-462: l_z is component 2; the accreted linear momentum goes through
                // iso2d::to_conserved_per_area(Q, x) physics_iso2d.hpp:404-414
                const double r2 = 0.0 + xc * xc + yc * yc;
                acc[4 * b + 1] = acc[4 * b + 1] + s[2] * dA;
                acc[4 * b + 2] = acc[4 * b + 2] + (s[1] * xc - s[2] * yc) / r2 * dA;
                acc[4 * b + 3] = acc[4 * b + 3] + (s[1] * yc + s[2] * xc) / r2 * dA;
            }
            else
            {
                acc[4 * b + 1] = acc[4 * b + 1] + (xc * s[2] - yc * s[1]) * dA;
                acc[4 * b + 2] = acc[4 * b + 2] + s[1] * dA;
                acc[4 * b + 3] = acc[4 * b + 3] + s[2] * dA;
            }
        }
    }
    __shared__ double red[8][256];
    for (int k = 0; k < 8; ++k) red[k][threadIdx.x] = acc[k];
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1)
    {
        if ((int) threadIdx.x < off)
            for (int k = 0; k < 8; ++k) red[k][threadIdx.x] = red[k][threadIdx.x] + red[k][threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0)
    {
        for (int b = 0; b < 2; ++b)
        {
            const double dm = -red[4 * b + 0][0], dl = -red[4 * b + 1][0], dpx = -red[4 * b + 2][0], dpy = -red[4 * b + 3][0];
            out[0 + b] = dm;
            out[2 + b] = dl;
            out[4 + b] = dpx;
            out[6 + b] = dpy;
            // work :356-365
            const double M0 = c.body[5 * b], px0 = c.body[5 * b + 3] * M0, py0 = c.body[5 * b + 4] * M0;
            const double M1 = M0 + dm, px1 = px0 + dpx, py1 = py0 + dpy;
            out[8 + b] = p.qform ? 0.0 : ((px1 * px1 + py1 * py1) / M1 - (px0 * px0 + py0 * py0) / M0) * 0.5;   // source_terms_q never sets it
        }
    }
}

// ---- totals: fixed-order reduction ----------------------------------------------------------------------------------
// totals[] in the order of mh_binary_total (include/mara_hip.h): mass_acc[2], L_acc[2], torque[2], px_acc[2], py_acc[2],
// fx[2], fy[2], work[2], mass_ejected, L_ejected
__global__ __launch_bounds__(1024)
void binary_reduce_kernel(const double* partials, int nwaves, const double* block_out, int nblocks, double* totals)
{
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int t = wave; t < MH_BINARY_NTOTALS; t += 16)
    {
        // which source does total t come from?
        //   blocks: 0,1 mass  2,3 L  6,7 px  8,9 py  14,15 work         waves (negated): 4,5 torque 10,11 fx 12,13 fy 16 mass_ej 17 L_ej
        int from_block = -1, from_wave = -1;
        if (t < 4) from_block = t;
        else if (t < 6) from_wave = t - 4;
        else if (t < 10) from_block = t - 2;
        else if (t < 14) from_wave = t - 8;
        else if (t < 16) from_block = t - 6;
        else from_wave = t - 10;
        // each lane adds its entries in index order; the loads of eight of them are issued together (a plain loop is a chain of
        // dependent load latencies: 12 us per stage at 2048^2)
        const double* src = from_block >= 0 ? block_out + from_block : partials + from_wave;
        const long pitch = from_block >= 0 ? NBLK : NPART;
        const int count = from_block >= 0 ? nblocks : nwaves;
        double s = 0.0;
        int k = lane;
        for (; k + 7 * 64 < count; k += 8 * 64)
        {
            double v[8];
#pragma unroll
            for (int m = 0; m < 8; ++m) v[m] = src[(long) (k + 64 * m) * pitch];
#pragma unroll
            for (int m = 0; m < 8; ++m) s = s + v[m];
        }
        for (; k < count; k += 64) s = s + src[(long) k * pitch];
        s = wave_sum(s);
        if (lane == 0) totals[t] = from_block >= 0 ? s : -s;
    }
}

// ---- maximum wavespeed ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256)
void binary_maxw_kernel(const double* u, const double* xv, const double* yv, int n0, int n, int qform, BinaryConsts c, unsigned long long* result)
{
    const Recip rmach = make_recip(c.mach, 1.0);
    double m = 0.0;
    const long total = (long) n0 * n;
    for (long idx = (long) blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long) gridDim.x * blockDim.x)
    {
        const int i = (int) (idx / n), j = (int) (idx - (long) i * n);
        const double* q = u + (long) (i + BHALO) * 3 * n + j;
        State3 U, P;
        U[0] = q[0]; U[1] = q[n]; U[2] = q[2 * n];
        const double xc = (xv[i] + xv[i + 1]) * 0.5, yc = (yv[j] + yv[j + 1]) * 0.5;
        if (qform) iso2d::recover_primitive_angmom(U, xc, yc, P);
        else       iso2d::recover_primitive(U, P);
        const double wsp = iso2d::max_wavespeed(P, binary_cs2(c, rmach, xc, yc));
        m = (m < wsp) ? wsp : m;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { const double o = __shfl_xor(m, off); m = (m < o) ? o : m; }
    __shared__ double wmax[4];
    if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0)
    {
        for (int k = 1; k < 4; ++k) m = (m < wmax[k]) ? wmax[k] : m;
        atomicMax(result, (unsigned long long) __double_as_longlong(m));   // non-negative doubles order as integers; one atomic per workgroup
    }
}

// ---- launchers ----------------------------------------------------------------------------------------------------------
BinaryConsts binary_make_consts(const mh_binary_desc* d, const double bodies[10])
{
    BinaryConsts c;
    int depth = 0;
    while ((d->block_size << depth) < d->n) ++depth;
    c.h0 = 2.0 * d->domain_radius / d->block_size;
    c.h = c.h0 / (1 << depth);                                       // scheme.cpp:274, :793
    c.mach = d->mach_number;
    c.alpha = d->alpha;
    c.nu = d->nu;
    c.rc_cut = d->alpha_cutoff_radius;
    c.sink_rate = d->sink_rate;
    c.s2 = d->sink_radius * d->sink_radius;
    c.rs2 = d->softening_radius * d->softening_radius;
    c.floor_sigma = d->density_floor;
    c.axisym = d->axisymmetric_cs2;
    c.rd = d->domain_radius;
    c.sr2 = d->gst_suppr_radius * d->gst_suppr_radius;
    binary_set_theta(c, d->plm_theta);
    for (int k = 0; k < 10; ++k) c.body[k] = bodies[k];
    return c;
}

// reserved: wave slots held by a launch that runs beside this one (a band's edge rows beside its interior)
static int binary_chunk_rows(const mh_binary_desc* d, int nstrips, int rows, int reserved = 0)
{
    if (d->chunk_rows > 0) return d->chunk_rows;
    const long chunks_max = (2048 - reserved) / nstrips > 0 ? (2048 - reserved) / nstrips : 1;     // one residency round at 2 waves / SIMD
    const long c = (rows + chunks_max - 1) / chunks_max;
    return (int) (c > 96 ? 32 : (c < 4 ? 4 : c));
}

// the recommended cut of a band whose edge rows are stepped first: the two rows a neighbour needs, no more - an edge wave marches its four
// pipeline-fill rows and then these, and how long that takes (11 us; 22 us with eight rows) is the head start the exchange loses
int binary_edge_rows(int n0)
{
    return n0 >= 5 ? 2 : 0;
}

// partial sums of the waves + per-block sink results of one stage. Sized for the most waves any cut of the rows can make (the shortest
// chunk the launcher picks is 4 rows, or the descriptor's; two edge chunks per strip when a band runs its edges first): a few MB at most.
size_t binary_scratch_doubles(const mh_binary_desc* d, const BinaryBand* band)
{
    const int rows = band ? band->n0 : d->n;
    const int nstrips = (d->n + BSTRIP - 1) / BSTRIP;
    const int shortest = d->chunk_rows > 0 && d->chunk_rows < 4 ? d->chunk_rows : 4;
    const long nwaves = (long) nstrips * (2 + (rows + shortest - 1) / shortest);
    const long nb = d->n / d->block_size;
    return (size_t) (nwaves * NPART + (long) (rows / d->block_size) * nb * NBLK);
}

hipError_t binary_stage_launch(const mh_binary_desc* d, const double* xv, const double* yv, const double* u_in, const double* u_base,
                               double* u_out, const double* u_init, const double* br, const double bodies[10], double dt, double weight,
                               double theta, double* totals, double* scratch, int32_t* status, hipStream_t stream, const BinaryBand* band,
                               const BinaryTotalsOverlap* overlap, int32_t* status_clear, const BinaryRows* rows)
{
    // xv: the x vertices of the WHOLE mesh; a band (rows [row0, row0 + n0), a multiple of the block size) sees its own slice of them
    const int n0 = band ? band->n0 : d->n, row0 = band ? band->row0 : 0;
    const int part = rows ? rows->part : BIN_ROWS_ALL, edge = rows ? rows->edge : 0;
    if (part != BIN_ROWS_ALL && (edge < 2 || n0 - 2 * edge < 1)) return hipErrorInvalidValue;
    BinaryStageParams p;
    p.u_in = u_in; p.u_base = u_base; p.u_out = u_out; p.u_init = u_init; p.br = br; p.xv = xv + row0; p.xvg = xv; p.yv = yv;
    p.status = status;
    p.status_clear = status_clear;
    p.n = d->n;
    p.n0 = n0; p.row0 = row0; p.ext0 = band ? band->ext0 : 0;
    p.nstrips = (d->n + BSTRIP - 1) / BSTRIP;
    p.seg1_begin = p.seg1_end = 0;
    p.wave_base = 0;
    if (part == BIN_ROWS_EDGES)
    {
        // one chunk per strip and side: rows [0, edge) and [n0 - edge, n0)
        p.chunk_rows = edge;
        p.seg0_begin = 0; p.seg0_end = edge; p.seg0_chunks = 1;
        p.seg1_begin = n0 - edge; p.seg1_end = n0;
        p.nchunks = 2;
    }
    else
    {
        const int lo = part == BIN_ROWS_INTERIOR ? edge : 0, hi = part == BIN_ROWS_INTERIOR ? n0 - edge : n0;
        // (interior: the edge launch's 2 x nstrips waves run beside it - together they must not spill into a second residency round, or the
        // last interior waves start when the first ones end: 137 instead of 102 us at 2048^2, kernel trace)
        p.chunk_rows = binary_chunk_rows(d, p.nstrips, hi - lo, part == BIN_ROWS_INTERIOR ? 2 * p.nstrips : 0);
        p.nchunks = (hi - lo + p.chunk_rows - 1) / p.chunk_rows;
        p.seg0_begin = lo; p.seg0_end = hi; p.seg0_chunks = p.nchunks;
        if (part == BIN_ROWS_INTERIOR) p.wave_base = 2 * p.nstrips;
    }
    p.theta = theta;
    p.dt = dt;
    p.weight = weight;
    p.c = binary_make_consts(d, bodies);
    binary_set_theta(p.c, theta);          // the STAGE's theta (safe mode steps with theta = 0, subprog_binary.cpp:285-292)
    const int launch_waves = p.nstrips * p.nchunks;
    const int nwaves = p.wave_base + launch_waves;          // of the whole stage, once its last part is issued
    p.partials = scratch;
    const int nblocks = (launch_waves + BWAVES_PER_BLOCK - 1) / BWAVES_PER_BLOCK;
    const dim3 grid(nblocks), block(BWAVE * BWAVES_PER_BLOCK);
    const bool combine = weight != 1.0;
    // the stage's completion event for the totals stream rides on the launch itself (no marker packet between two stage kernels)
    static const bool event_on_launch = ! (getenv("MH_BIN_EVENT_ON_LAUNCH") && atoi(getenv("MH_BIN_EVENT_ON_LAUNCH")) == 0);
    const hipEvent_t done = (overlap && part != BIN_ROWS_EDGES && event_on_launch) ? overlap->stage_done : nullptr;
    hipError_t e = d->arith == MH_ARITH_FAST ? binary_stage_dispatch_fast(p, grid, block, stream, combine, d->angmom_form != 0, done)
                                             : binary_stage_dispatch<BinStrict>(p, grid, block, stream, combine, d->angmom_form != 0, done);
    if (e != hipSuccess) return e;
    if (part == BIN_ROWS_EDGES) return hipSuccess;          // the totals follow the interior

    // The per-block sink sums read the stage's INPUT and the fixed-order reduction is two small latency-bound launches (16 + 10 us at
    // 2048^2 beside a ~120 us stage kernel): with `overlap` they run on a second stream - the sink sums beside the stage kernel, the
    // reduction behind both - and the next stage does not wait for them (the caller waits once, before it reads the totals).
    // overlap->input_ready is an event of `stream` behind which u_in is complete (the previous stage's stage_done, or one the caller recorded).
    hipStream_t tstream = stream;
    if (overlap)
    {
        tstream = overlap->stream;
        if (! done && (e = hipEventRecord(overlap->stage_done, stream)) != hipSuccess) return e;
        if ((e = hipStreamWaitEvent(tstream, overlap->input_ready, 0)) != hipSuccess) return e;
    }
    BinarySinkParams s;
    s.u_in = u_in; s.xv = xv + row0; s.yv = yv;
    s.block_out = scratch + (long) nwaves * NPART;
    s.n = d->n; s.bs = d->block_size; s.nb = d->n / d->block_size; s.qform = d->angmom_form;
    s.dt = dt;
    s.c = p.c;
    const int tree_blocks = (n0 / d->block_size) * s.nb;          // the band's rows of tree blocks x blocks per row
    hipLaunchKernelGGL(binary_sink_kernel, dim3(tree_blocks), dim3(256), 0, tstream, s);
    e = hipGetLastError();
    if (e != hipSuccess) return e;
    if (overlap && overlap->sink_done && (e = hipEventRecord(overlap->sink_done, tstream)) != hipSuccess) return e;
    if (overlap && (e = hipStreamWaitEvent(tstream, overlap->stage_done, 0)) != hipSuccess) return e;
    if (rows && rows->edges_done && (e = hipStreamWaitEvent(tstream, rows->edges_done, 0)) != hipSuccess) return e;      // the edge launch ran on another stream
    hipLaunchKernelGGL(binary_reduce_kernel, dim3(1), dim3(1024), 0, tstream, p.partials, nwaves, s.block_out, tree_blocks, totals);
    return hipGetLastError();
}

hipError_t binary_maxw_launch(const mh_binary_desc* d, const double* xv, const double* yv, const double* u, const double bodies[10],
                              double* result, hipStream_t stream, const BinaryBand* band)
{
    hipError_t e = hipMemsetAsync(result, 0, sizeof(double), stream);
    if (e != hipSuccess) return e;
    const BinaryConsts c = binary_make_consts(d, bodies);
    const int n0 = band ? band->n0 : d->n, row0 = band ? band->row0 : 0;
    const long total = (long) n0 * d->n;
    const int nblocks = (int) ((total + 256 * 4 - 1) / (256 * 4) < 2048 ? (total + 256 * 4 - 1) / (256 * 4) : 2048);
    hipLaunchKernelGGL(binary_maxw_kernel, dim3(nblocks), dim3(256), 0, stream, u, xv + row0, yv, n0, d->n, (int) d->angmom_form, c, reinterpret_cast<unsigned long long*>(result));
    return hipGetLastError();
}

// row-range guard (row_check.hpp): what this translation unit's kernels asked for; false in product builds
MH_ROW_RANGE_READER(rows_requested_binary)

} // namespace mh
