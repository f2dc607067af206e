// 3-D uniform-cartesian Euler Runge-Kutta stage for gfx950 (BASELINE config 5).
//
// Same scheme and reference citations as euler2d.hip with a third axis:
//     u1 = u0 - (diff0(Fx)*(dt/dx) + diff1(Fy)*(dt/dy) + diff2(Fz)*(dt/dz))
// (oracle/ref_drivers/euler_cart_ref.cpp, rank 3; cloud::advance composition,
// src/subprog_cloud.cpp:511-584; RK combine :682-695).
//
// 2.5-D plane marching. A workgroup of 8 wavefronts owns a tile of 8 axis-1
// rows x 60 axis-2 columns and marches along axis 0:
//   * axis 0 (march): primitives of planes i..i+2, the slope of plane i and the
//     flux through face i-1/2 live in registers; each flux is computed once;
//   * axis 2 (lanes): DPP wave shifts, exactly as in the 2-D kernel;
//   * axis 1 (across waves): three small LDS exchanges per plane - primitives,
//     right-going face states, face fluxes - separated by workgroup barriers.
//     All 8 waves own interior rows; the first wave also builds the face state
//     below the tile, the last wave the flux through the tile's top face, so no
//     wave slots are spent on halo rows.
// Conserved planes are read once (+ halo re-reads that hit in L2) and written
// once; nothing else touches HBM. Algorithmic bytes: 80 / 120 B per cell per
// stage as in 2-D. Device layout: include/mara_hip.h with row_pitch = n1*n2.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "euler_device.hpp"
#include "euler_device_fast.hpp"
#include "launch.hpp"

namespace mh {

static constexpr int W3 = 64;                  // lanes
static constexpr int H3 = 2;                   // halo
static constexpr int STRIP3 = W3 - 2 * H3;     // 60 output columns per wave
static constexpr int ROWS3 = 8;                // axis-1 rows per workgroup (= waves)

struct Stage3dParams
{
    const double* u_in;
    const double* u_base;
    double*       u_out;
    int32_t*      status;
    long   plane_stride;     // n1*n2: doubles between variables of one axis-0 plane
    long   row_stride;       // 5*n1*n2: doubles between consecutive axis-0 planes
    int    n0, n1, n2;
    int    row_begin, row_end, chunk_rows;
    int    ntiles1, nstrips, nchunks;
    int    bc_lo0, bc_hi0, bc_t;
    double gamma, theta, cx, cy, cz, weight;
};

__device__ inline double dpp3_left(double x)
{
    int lo = __double2loint(x), hi = __double2hiint(x);
    lo = __builtin_amdgcn_update_dpp(0, lo, 0x138, 0xf, 0xf, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, 0x138, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
__device__ inline double dpp3_right(double x)
{
    int lo = __double2loint(x), hi = __double2hiint(x);
    lo = __builtin_amdgcn_update_dpp(0, lo, 0x130, 0xf, 0xf, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, 0x130, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
__device__ inline State5 dpp3_left(const State5& s) { State5 r; for (int q = 0; q < 5; ++q) r[q] = dpp3_left(s[q]); return r; }
__device__ inline State5 dpp3_right(const State5& s) { State5 r; for (int q = 0; q < 5; ++q) r[q] = dpp3_right(s[q]); return r; }

__device__ inline int fold_index(int j, int n, int bc)
{
    if (bc == 1) j = j < 0 ? j + n : (j >= n ? j - n : j);
    return min(max(j, 0), n - 1);
}

// LDS exchange buffers: [slot][variable][lane]
struct Tile3d
{
    double P[ROWS3 + 2][5][W3];      // slot r+1 = primitives of tile row r (slots 0 and ROWS3+1: rows below / above)
    double S[ROWS3 + 1][5][W3];      // slot r+1 = P + G/2 of tile row r    (slot 0: row below the tile)
    double F[ROWS3 + 1][5][W3];      // slot f   = axis-1 flux through the face below tile row f (slot ROWS3: top face)
};

__device__ inline void lds_put(double (*dst)[W3], int lane, const State5& s)
{
#pragma unroll
    for (int q = 0; q < 5; ++q) dst[q][lane] = s[q];
}
__device__ inline State5 lds_get(double (*src)[W3], int lane)
{
    State5 s;
#pragma unroll
    for (int q = 0; q < 5; ++q) s[q] = src[q][lane];
    return s;
}

template<class A, int RIEMANN, bool PLM, bool COMBINE>
__global__ __launch_bounds__(W3 * ROWS3, 2)
void euler3d_stage_kernel(Stage3dParams p)
{
    extern __shared__ double lds_raw[];
    Tile3d& tile = *reinterpret_cast<Tile3d*>(lds_raw);

    // work item -> (chunk along axis 0, tile along axis 1, strip along axis 2); neighbouring items share an XCD
    int b = blockIdx.x;
    {
        const int per_xcd = gridDim.x >> 3;
        if (b < per_xcd * 8) b = (b & 7) * per_xcd + (b >> 3);
    }
    const int strip = b % p.nstrips;
    const int t1 = (b / p.nstrips) % p.ntiles1;
    const int chunk = b / (p.nstrips * p.ntiles1);
    const int lane = threadIdx.x & 63;
    const int row = __builtin_amdgcn_readfirstlane((int) (threadIdx.x >> 6));   // wave index = tile row (uniform -> scalar registers)
    const int r0 = p.row_begin + chunk * p.chunk_rows;
    const int r1 = min(r0 + p.chunk_rows, p.row_end);

    const int j = t1 * ROWS3 + row;                         // axis-1 index of this wave's row (may exceed n1 - 1 in the last tile)
    const int jc = fold_index(j, p.n1, p.bc_t);
    const int col = strip * STRIP3 - H3 + lane;
    const int kc = fold_index(col, p.n2, p.bc_t);
    const bool writes = lane >= H3 && lane < W3 - H3 && col < p.n2 && j < p.n1;
    const bool first = row == 0, last = row == ROWS3 - 1;

    const long row_stride = p.row_stride, plane = p.plane_stride;
    auto cell = [&] (int jj) { return (long) fold_index(jj, p.n1, p.bc_t) * p.n2 + kc; };
    auto row_off = [row_stride] (int r) { return (long) (r + H3) * row_stride; };
    auto load = [&] (const double* base, long c, int r) -> State5
    {
        State5 U;
#pragma unroll
        for (int q = 0; q < 5; ++q) U[q] = base[q * plane + c + row_off(r)];
        return U;
    };
    const long c0 = (long) jc * p.n2 + kc;
    // extra rows handled by the edge waves: two below the tile (first wave) / two above it (last wave)
    const int je1 = first ? t1 * ROWS3 - 1 : t1 * ROWS3 + ROWS3;
    const int je2 = first ? t1 * ROWS3 - 2 : t1 * ROWS3 + ROWS3 + 1;
    const long ce1 = cell(je1), ce2 = cell(je2);
    const bool edge = first || last;

    const double theta = p.theta;
    const typename A::Gamma gl = A::gamma_law(p.gamma);

    // ---- prologue along axis 0
    State5 U0 = load(p.u_in, c0, r0), U1 = load(p.u_in, c0, r0 + 1), U2 = load(p.u_in, c0, r0 + 2);
    State5 P0, P1, G0, Fx_lo;
    {
        const State5 Pa = A::c2p(load(p.u_in, c0, r0 - 2), gl);
        const State5 Pb = A::c2p(load(p.u_in, c0, r0 - 1), gl);
        P0 = A::c2p(U0, gl);
        P1 = A::c2p(U1, gl);
        if constexpr (PLM)
        {
            const State5 Gb = A::plm(Pa, Pb, P0, theta);
            G0 = A::plm(Pb, P0, P1, theta);
            Fx_lo = A::template flux<RIEMANN, 0>(A::plus(Pb, Gb), A::minus(P0, G0), gl);
        }
        else
        {
            Fx_lo = A::template flux<RIEMANN, 0>(Pb, P0, gl);
        }
    }
    int32_t bad = 0;

    for (int r = r0; r < r1; ++r)
    {
        const int rp = min(r + 3, p.n0 + 1);
        const State5 U3 = load(p.u_in, c0, rp);
        State5 Ubase;
        if constexpr (COMBINE) Ubase = load(p.u_base, c0, r);

        // ---- phase 1: publish this plane's primitives (tile rows, plus the row just outside for the edge waves)
        State5 Pe1, Pe2;
        lds_put(tile.P[row + 1], lane, P0);
        if (edge)
        {
            Pe1 = A::c2p(load(p.u_in, ce1, r), gl);
            if constexpr (PLM) Pe2 = A::c2p(load(p.u_in, ce2, r), gl);
            lds_put(tile.P[first ? 0 : ROWS3 + 1], lane, Pe1);
        }
        // with 8 rows the first and last wave are distinct, so each edge wave has exactly one outside row

        // ---- axis 0 while the others arrive
        const State5 P2 = A::c2p(U2, gl);
        State5 G1, Fx_hi;
        if constexpr (PLM)
        {
            G1 = A::plm(P0, P1, P2, theta);
            Fx_hi = A::template flux<RIEMANN, 0>(A::plus(P0, G0), A::minus(P1, G1), gl);
        }
        else
        {
            Fx_hi = A::template flux<RIEMANN, 0>(P0, P1, gl);
        }

        // ---- axis 2 (lanes)
        State5 Fz_lo, Fz_hi;
        if constexpr (PLM)
        {
            const State5 Gz = A::plm(dpp3_left(P0), P0, dpp3_right(P0), theta);
            const State5 SL = dpp3_left(A::plus(P0, Gz));
            Fz_lo = A::template flux<RIEMANN, 2>(SL, A::minus(P0, Gz), gl);
        }
        else
        {
            Fz_lo = A::template flux<RIEMANN, 2>(dpp3_left(P0), P0, gl);
        }
        Fz_hi = dpp3_right(Fz_lo);

        __syncthreads();

        // ---- phase 2 (axis 1): slopes and right-going face states
        const State5 Pdn = lds_get(tile.P[row], lane), Pup = lds_get(tile.P[row + 2], lane);
        State5 Gy, SRy;
        if constexpr (PLM)
        {
            Gy = A::plm(Pdn, P0, Pup, theta);
            lds_put(tile.S[row + 1], lane, A::plus(P0, Gy));
            SRy = A::minus(P0, Gy);
            if (first)
            {
                const State5 Ge = A::plm(Pe2, Pe1, P0, theta);          // slope of the row below the tile
                lds_put(tile.S[0], lane, A::plus(Pe1, Ge));
            }
        }
        else
        {
            lds_put(tile.S[row + 1], lane, P0);
            SRy = P0;
            if (first) lds_put(tile.S[0], lane, Pe1);
        }
        __syncthreads();

        // ---- phase 3: flux through the face below this row (and, for the last wave, the tile's top face)
        const State5 Fy_lo = A::template flux<RIEMANN, 1>(lds_get(tile.S[row], lane), SRy, gl);
        lds_put(tile.F[row], lane, Fy_lo);
        if (last)
        {
            State5 SRe;
            if constexpr (PLM) SRe = A::minus(Pe1, A::plm(P0, Pe1, Pe2, theta));
            else               SRe = Pe1;
            State5 SLown;
            if constexpr (PLM) SLown = A::plus(P0, Gy); else SLown = P0;
            lds_put(tile.F[ROWS3], lane, A::template flux<RIEMANN, 1>(SLown, SRe, gl));
        }
        __syncthreads();
        const State5 Fy_hi = lds_get(tile.F[row + 1], lane);

        // ---- update
        State5 Un;
#pragma unroll
        for (int q = 0; q < 5; ++q)
        {
            const double u1 = A::update3(U0[q], Fx_lo[q], Fx_hi[q], Fy_lo[q], Fy_hi[q], Fz_lo[q], Fz_hi[q], p.cx, p.cy, p.cz);
            if constexpr (COMBINE) Un[q] = A::combine(Ubase[q], u1, p.weight);
            else                   Un[q] = u1;
        }
        if (!(Un[0] > 0.0)) bad |= 1;

        if (writes)
        {
            double* out = p.u_out + (long) j * p.n2 + col;
#pragma unroll
            for (int q = 0; q < 5; ++q) out[q * plane + row_off(r)] = Un[q];
            if (r < H3)
            {
                if (p.bc_lo0 == 0 && r == 0)
                {
#pragma unroll
                    for (int q = 0; q < 5; ++q) { out[q * plane + row_off(-1)] = Un[q]; out[q * plane + row_off(-2)] = Un[q]; }
                }
                if (p.bc_hi0 == 1)
                {
#pragma unroll
                    for (int q = 0; q < 5; ++q) out[q * plane + row_off(p.n0 + r)] = Un[q];
                }
            }
            if (r >= p.n0 - H3)
            {
                if (p.bc_hi0 == 0 && r == p.n0 - 1)
                {
#pragma unroll
                    for (int q = 0; q < 5; ++q) { out[q * plane + row_off(p.n0)] = Un[q]; out[q * plane + row_off(p.n0 + 1)] = Un[q]; }
                }
                if (p.bc_lo0 == 1)
                {
#pragma unroll
                    for (int q = 0; q < 5; ++q) out[q * plane + row_off(r - p.n0)] = Un[q];
                }
            }
        }

        U0 = U1; U1 = U2; U2 = U3;
        P0 = P1; P1 = P2;
        if constexpr (PLM) G0 = G1;
        Fx_lo = Fx_hi;
    }

    if (p.status)
    {
        if (__any(writes && bad) && lane == 0) atomicOr(p.status, 1);
    }
}

template<class A, int RIEMANN, bool PLM, bool COMBINE>
static hipError_t launch3(const Stage3dParams& p, hipStream_t stream)
{
    const int nblocks = p.nstrips * p.ntiles1 * p.nchunks;
    auto kernel = euler3d_stage_kernel<A, RIEMANN, PLM, COMBINE>;
    static bool attr_set = false;           // 70 KB of LDS per workgroup: above the 64 KB static limit, so dynamic + opt-in
    if (! attr_set)
    {
        hipError_t e = hipFuncSetAttribute((const void*) kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int) sizeof(Tile3d));
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    hipLaunchKernelGGL(kernel, dim3(nblocks), dim3(W3 * ROWS3), sizeof(Tile3d), stream, p);
    return hipGetLastError();
}

hipError_t euler3d_stage_launch(const mh_euler_cart_desc* d, const double* u_in, const double* u_base, double* u_out,
                                double dt, double weight, int row_begin, int row_end, int32_t* status, hipStream_t stream)
{
    Stage3dParams p;
    p.u_in = u_in; p.u_base = u_base; p.u_out = u_out; p.status = status;
    p.n0 = d->n[0]; p.n1 = d->n[1]; p.n2 = d->n[2];
    p.plane_stride = (long) p.n1 * p.n2;
    p.row_stride = 5L * p.n1 * p.n2;
    p.row_begin = row_begin; p.row_end = row_end;
    p.chunk_rows = d->chunk_rows > 0 ? d->chunk_rows : 32;
    p.ntiles1 = (p.n1 + ROWS3 - 1) / ROWS3;
    p.nstrips = (p.n2 + STRIP3 - 1) / STRIP3;
    p.nchunks = (row_end - row_begin + p.chunk_rows - 1) / p.chunk_rows;
    p.bc_lo0 = d->bc_lo0; p.bc_hi0 = d->bc_hi0; p.bc_t = d->bc_transverse;
    p.gamma = d->gamma; p.theta = d->plm_theta;
    p.cx = dt / d->dl[0]; p.cy = dt / d->dl[1]; p.cz = dt / d->dl[2];
    p.weight = weight;
    if (p.nchunks <= 0) return hipSuccess;

    const bool plm = d->plm_theta >= 0.0, combine = weight != 1.0;
    const int key = (d->arith == MH_ARITH_FAST ? 8 : 0) | (d->riemann == MH_RIEMANN_HLLC ? 4 : 0) | (plm ? 2 : 0) | (combine ? 1 : 0);
    switch (key)
    {
        case 0:  return launch3<StrictArith, 0, false, false>(p, stream);
        case 1:  return launch3<StrictArith, 0, false, true >(p, stream);
        case 2:  return launch3<StrictArith, 0, true,  false>(p, stream);
        case 3:  return launch3<StrictArith, 0, true,  true >(p, stream);
        case 4:  return launch3<StrictArith, 1, false, false>(p, stream);
        case 5:  return launch3<StrictArith, 1, false, true >(p, stream);
        case 6:  return launch3<StrictArith, 1, true,  false>(p, stream);
        case 7:  return launch3<StrictArith, 1, true,  true >(p, stream);
        case 8:  return launch3<FastArith, 0, false, false>(p, stream);
        case 9:  return launch3<FastArith, 0, false, true >(p, stream);
        case 10: return launch3<FastArith, 0, true,  false>(p, stream);
        case 11: return launch3<FastArith, 0, true,  true >(p, stream);
        case 12: return launch3<FastArith, 1, false, false>(p, stream);
        case 13: return launch3<FastArith, 1, false, true >(p, stream);
        case 14: return launch3<FastArith, 1, true,  false>(p, stream);
        case 15: return launch3<FastArith, 1, true,  true >(p, stream);
    }
    return hipErrorInvalidValue;
}

} // namespace mh
