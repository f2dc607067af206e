// Diagnostics of the device-resident `binary` solution, evaluated where the data live when a task of the
// driver is due: binary::disk_mass / disk_angular_momentum for the time series and binary::diagnostic_fields (sigma, v_r, v_phi)
// for the diagnostics file (src/subprog_binary_diagnostics.cpp:21-82). Both field layouts of the solver object: one periodic
// grid [n + 4][3][n] (binary.hip) or blocks [nb][3][bs][bs] (binary_tree.hip).
//
// Arithmetic: IEEE (the STRICT primitive recovery, division, sqrt; no contraction) whatever `arith` the stepping uses - these
// numbers are written to files. What differs from the reference: the sums are tree reductions (wave -> workgroup -> one wave over
// the workgroup partials, a fixed order independent of timing) instead of the reference's sequential block sums, and the radius
// is sqrt(r2) where the reference calls std::pow(r2, 0.5). Tolerances: tests/test_gpu_binary_diagnostics.py.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "binary_device.hpp"

namespace mh {

struct DiagParams
{
    const double* u;
    const double* xv;         // grid: x of vertex rows [n + 1], y of vertex columns [n + 1]
    const double* yv;
    const double* edges;      // tree: [nb][2][bs + 1]
    int n, nb, bs, tree;
    long ncell;
    double* partial;          // [gridDim.x][2]
    double* fields;           // [3][ncell]: sigma, v_r, v_phi in solution order
};

static constexpr int DIAG_THREADS = 256;
static constexpr int DIAG_BLOCKS = 1024;

struct DiagCell { State3 U; double xc, yc, dA; };

__device__ inline DiagCell diag_cell(const DiagParams& p, long c)
{
    DiagCell k;
    if (p.tree)
    {
        const int per = p.bs * p.bs;
        const int b = (int) (c / per), r = (int) (c - (long) b * per);
        const int i = r / p.bs, j = r - i * p.bs;
        const double* xe = p.edges + (long) b * 2 * (p.bs + 1);
        const double* ye = xe + p.bs + 1;
        k.xc = (xe[i] + xe[i + 1]) * 0.5;
        k.yc = (ye[j] + ye[j + 1]) * 0.5;
        k.dA = (xe[i + 1] - xe[i]) * (ye[j + 1] - ye[j]);
        for (int q = 0; q < 3; ++q) k.U[q] = p.u[(((long) b * 3 + q) * p.bs + i) * p.bs + j];
    }
    else
    {
        const int i = (int) (c / p.n), j = (int) (c - (long) i * p.n);
        k.xc = (p.xv[i] + p.xv[i + 1]) * 0.5;
        k.yc = (p.yv[j] + p.yv[j + 1]) * 0.5;
        k.dA = (p.xv[i + 1] - p.xv[i]) * (p.yv[j + 1] - p.yv[j]);
        for (int q = 0; q < 3; ++q) k.U[q] = p.u[((long) (i + 2) * 3 + q) * p.n + j];
    }
    return k;
}

__device__ inline double diag_wave_sum(double x)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) x += __shfl_xor(x, off);
    return x;
}

// disk_mass = sum U0 dA; disk_angular_momentum = sum (x U2 - y U1) dA, or sum Q2 dA in the angular-momentum form (:21-46)
template<bool QFORM>
__global__ __launch_bounds__(DIAG_THREADS)
void binary_diag_sums_kernel(DiagParams p)
{
    double mass = 0.0, lz = 0.0;
    for (long c = (long) blockIdx.x * DIAG_THREADS + threadIdx.x; c < p.ncell; c += (long) gridDim.x * DIAG_THREADS)
    {
        const DiagCell k = diag_cell(p, c);
        mass += k.U[0] * k.dA;
        lz += QFORM ? k.U[2] * k.dA : (k.xc * k.U[2] - k.yc * k.U[1]) * k.dA;
    }
    __shared__ double red[2][DIAG_THREADS / 64];
    mass = diag_wave_sum(mass);
    lz = diag_wave_sum(lz);
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = mass; red[1][threadIdx.x >> 6] = lz; }
    __syncthreads();
    if (threadIdx.x == 0)
    {
        double m = 0.0, l = 0.0;
        for (int w = 0; w < DIAG_THREADS / 64; ++w) { m += red[0][w]; l += red[1][w]; }
        p.partial[2 * blockIdx.x] = m;
        p.partial[2 * blockIdx.x + 1] = l;
    }
}

__global__ __launch_bounds__(64)
void binary_diag_final_kernel(const double* partial, int nparts, double* out)
{
    double m = 0.0, l = 0.0;
    for (int k = threadIdx.x; k < nparts; k += 64) { m += partial[2 * k]; l += partial[2 * k + 1]; }
    m = diag_wave_sum(m);
    l = diag_wave_sum(l);
    if (threadIdx.x == 0) { out[0] = m; out[1] = l; }
}

// sigma, v_r = v . rhat, v_phi = v . phihat at the cell centres (:52-82)
template<bool QFORM>
__global__ __launch_bounds__(DIAG_THREADS)
void binary_diag_fields_kernel(DiagParams p)
{
    for (long c = (long) blockIdx.x * DIAG_THREADS + threadIdx.x; c < p.ncell; c += (long) gridDim.x * DIAG_THREADS)
    {
        const DiagCell k = diag_cell(p, c);
        State3 P;
        if (QFORM) iso2d::recover_primitive_angmom(k.U, k.xc, k.yc, P);
        else       iso2d::recover_primitive(k.U, P);
        const double rc = sqrt(k.xc * k.xc + k.yc * k.yc);
        const double rhat_x = k.xc / rc, rhat_y = k.yc / rc, phat_x = -k.yc / rc, phat_y = k.xc / rc;
        p.fields[c] = P[0];
        p.fields[p.ncell + c] = P[1] * rhat_x + P[2] * rhat_y;
        p.fields[2 * p.ncell + c] = P[1] * phat_x + P[2] * phat_y;
    }
}

static int diag_blocks(long ncell)
{
    const long b = (ncell + DIAG_THREADS - 1) / DIAG_THREADS;
    return (int) (b < 1 ? 1 : (b > DIAG_BLOCKS ? DIAG_BLOCKS : b));
}

size_t binary_diag_partial_doubles() { return 2 * (size_t) DIAG_BLOCKS; }

// out[2] (device) = {disk_mass, disk_angular_momentum}
hipError_t binary_diag_sums_launch(const double* u, const double* xv, const double* yv, const double* edges, int n, int nb, int bs, bool tree,
                                   bool qform, double* partial, double* out, hipStream_t stream)
{
    DiagParams p = {u, xv, yv, edges, n, nb, bs, tree ? 1 : 0, tree ? (long) nb * bs * bs : (long) n * n, partial, nullptr};
    const int g = diag_blocks(p.ncell);
    if (qform) hipLaunchKernelGGL(binary_diag_sums_kernel<true>, dim3(g), dim3(DIAG_THREADS), 0, stream, p);
    else       hipLaunchKernelGGL(binary_diag_sums_kernel<false>, dim3(g), dim3(DIAG_THREADS), 0, stream, p);
    hipLaunchKernelGGL(binary_diag_final_kernel, dim3(1), dim3(64), 0, stream, partial, g, out);
    return hipGetLastError();
}

// fields[3][ncell] (device)
hipError_t binary_diag_fields_launch(const double* u, const double* xv, const double* yv, const double* edges, int n, int nb, int bs, bool tree,
                                     bool qform, double* fields, hipStream_t stream)
{
    DiagParams p = {u, xv, yv, edges, n, nb, bs, tree ? 1 : 0, tree ? (long) nb * bs * bs : (long) n * n, nullptr, fields};
    const int g = diag_blocks(p.ncell);
    if (qform) hipLaunchKernelGGL(binary_diag_fields_kernel<true>, dim3(g), dim3(DIAG_THREADS), 0, stream, p);
    else       hipLaunchKernelGGL(binary_diag_fields_kernel<false>, dim3(g), dim3(DIAG_THREADS), 0, stream, p);
    return hipGetLastError();
}

} // namespace mh
