// Layout shims at the boundary: host-order array-of-structs <-> the device
// layout (row-interleaved struct-of-arrays: for each axis-0 row, nq contiguous
// plane-rows; two ghost rows per side of axis 0), and the physical ghost-row
// fill used after an upload.
//
// The reference keeps every field as row-major AoS (src/core_ndarray.hpp:777-792)
// and builds ghost zones on the fly with extend_zero_gradient /
// extend_periodic_on_axis (src/core_ndarray_ops.hpp:162-180); here the same
// cells are materialised once as stored ghost rows.
#include "launch.hpp"

namespace mh {

static constexpr int HALO = 2;

// One thread per cell; AoS side is read/written with 8-B accesses at stride nq*8
// (L2 merges them: each 64-lane wave touches nq*512 contiguous bytes), SoA side is fully coalesced.
template<bool TO_SOA>
__global__ __launch_bounds__(256)
void transpose_kernel(const double* __restrict__ src, double* __restrict__ dst, int nq, size_t ncell, size_t row_pitch)
{
    for (size_t c = (size_t) blockIdx.x * blockDim.x + threadIdx.x; c < ncell; c += (size_t) gridDim.x * blockDim.x)
    {
        const size_t row = c / row_pitch, t = c - row * row_pitch;
        const size_t base = (row + HALO) * nq * row_pitch + t;
        for (int q = 0; q < nq; ++q)
        {
            if (TO_SOA) dst[base + q * row_pitch] = src[c * nq + q];
            else        dst[c * nq + q] = src[base + q * row_pitch];
        }
    }
}

__global__ __launch_bounds__(256)
void fill_ghost_rows_kernel(double* u, int nq, int n0, size_t row_pitch, int bc_lo0, int bc_hi0)
{
    const size_t total = (size_t) nq * HALO * row_pitch;

    for (size_t m = (size_t) blockIdx.x * blockDim.x + threadIdx.x; m < total; m += (size_t) gridDim.x * blockDim.x)
    {
        const size_t t = m % row_pitch;
        const int g = (int) ((m / row_pitch) % HALO);     // ghost layer 0,1
        const int q = (int) (m / (row_pitch * HALO));
        auto at = [=] (int row) -> double& { return u[((size_t) (row + HALO) * nq + q) * row_pitch + t]; };

        if (bc_lo0 == MH_BC_OUTFLOW)  at(-1 - g) = at(0);
        if (bc_lo0 == MH_BC_PERIODIC) at(-1 - g) = at(n0 - 1 - g);
        if (bc_hi0 == MH_BC_OUTFLOW)  at(n0 + g) = at(n0 - 1);
        if (bc_hi0 == MH_BC_PERIODIC) at(n0 + g) = at(g);
    }
}

// Known-byte-count stream with the stage kernels' access shape (8 B per lane, 512 B per wave instruction):
// calibrates rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 for this pattern.
__global__ __launch_bounds__(256)
void stream_copy_kernel(const double* __restrict__ src, double* __restrict__ dst, size_t n)
{
    for (size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t) gridDim.x * blockDim.x)
        dst[i] = src[i];
}

// ---- fields with stored transverse ghost layers (blocks of a 3-axis decomposition, launch.hpp: Euler3dLayout) --------------------
// host AoS [n0][n1][n2][5] <-> padded device field; one thread per cell
template<bool TO_SOA>
__global__ __launch_bounds__(256)
void block_transpose_kernel(const double* __restrict__ src, double* __restrict__ dst, int n0, int n1, int n2, int g1, int g2)
{
    const size_t ncell = (size_t) n0 * n1 * n2, pitch2 = (size_t) n2 + 2 * g2, plane = (size_t) (n1 + 2 * g1) * pitch2;
    for (size_t c = (size_t) blockIdx.x * blockDim.x + threadIdx.x; c < ncell; c += (size_t) gridDim.x * blockDim.x)
    {
        const size_t k = c % n2, j = (c / n2) % n1, i = c / ((size_t) n1 * n2);
        const size_t base = (i + HALO) * 5 * plane + (j + g1) * pitch2 + (k + g2);
        for (int q = 0; q < 5; ++q)
        {
            if (TO_SOA) dst[base + q * plane] = src[c * 5 + q];
            else        dst[c * 5 + q] = src[base + q * plane];
        }
    }
}

// One transverse face of a padded field <-> a contiguous message buffer [n0][5][len1][len2]: cells (i, j0 + jj, k0 + kk) of the
// interior planes i in [0, n0). PACK gathers the two edge layers that a neighbour needs, UNPACK scatters what it sent into the
// ghost layers (j0 or k0 = -2 / n). An axis-2 face is 2 doubles per row (16-byte pieces): latency, not bandwidth, and small.
template<bool PACK>
__global__ __launch_bounds__(256)
void block_face_kernel(double* __restrict__ field, double* __restrict__ buf, int n0, int j0, int len1, int k0, int len2,
                       long pitch2, long plane, int g1, int g2)
{
    const size_t total = (size_t) n0 * 5 * len1 * len2;
    for (size_t m = (size_t) blockIdx.x * blockDim.x + threadIdx.x; m < total; m += (size_t) gridDim.x * blockDim.x)
    {
        const size_t kk = m % len2, jj = (m / len2) % len1, iq = m / ((size_t) len1 * len2);      // iq = i * 5 + q
        const size_t f = (iq + (size_t) HALO * 5) * plane + (size_t) (j0 + (long) jj + g1) * pitch2 + (size_t) (k0 + (long) kk + g2);
        if (PACK) buf[m] = field[f];
        else      field[f] = buf[m];
    }
}

static int grid_for(size_t n)
{
    size_t b = (n + 255) / 256;
    return (int) (b > 2048 ? 2048 : (b ? b : 1));
}

hipError_t aos_to_soa_launch(const double* aos, double* soa, int nq, int n0, size_t row_pitch, hipStream_t stream)
{
    const size_t ncell = (size_t) n0 * row_pitch;
    hipLaunchKernelGGL(transpose_kernel<true>, dim3(grid_for(ncell)), dim3(256), 0, stream,
                       aos, soa, nq, ncell, row_pitch);
    return hipGetLastError();
}

hipError_t soa_to_aos_launch(const double* soa, double* aos, int nq, int n0, size_t row_pitch, hipStream_t stream)
{
    const size_t ncell = (size_t) n0 * row_pitch;
    hipLaunchKernelGGL(transpose_kernel<false>, dim3(grid_for(ncell)), dim3(256), 0, stream,
                       soa, aos, nq, ncell, row_pitch);
    return hipGetLastError();
}

hipError_t block_transpose_launch(bool to_soa, const double* src, double* dst, int n0, int n1, int n2, int g1, int g2, hipStream_t stream)
{
    const size_t ncell = (size_t) n0 * n1 * n2;
    if (to_soa) hipLaunchKernelGGL(block_transpose_kernel<true>, dim3(grid_for(ncell)), dim3(256), 0, stream, src, dst, n0, n1, n2, g1, g2);
    else        hipLaunchKernelGGL(block_transpose_kernel<false>, dim3(grid_for(ncell)), dim3(256), 0, stream, src, dst, n0, n1, n2, g1, g2);
    return hipGetLastError();
}

hipError_t block_face_launch(bool pack, double* field, double* buf, int n0, int j0, int len1, int k0, int len2,
                             int n1, int n2, int g1, int g2, hipStream_t stream)
{
    const size_t total = (size_t) n0 * 5 * len1 * len2;
    if (total == 0) return hipSuccess;
    const long pitch2 = n2 + 2 * g2, plane = (long) (n1 + 2 * g1) * pitch2;
    if (pack) hipLaunchKernelGGL(block_face_kernel<true>, dim3(grid_for(total)), dim3(256), 0, stream, field, buf, n0, j0, len1, k0, len2, pitch2, plane, g1, g2);
    else      hipLaunchKernelGGL(block_face_kernel<false>, dim3(grid_for(total)), dim3(256), 0, stream, field, buf, n0, j0, len1, k0, len2, pitch2, plane, g1, g2);
    return hipGetLastError();
}

hipError_t stream_copy_launch(const double* src, double* dst, size_t n, hipStream_t stream)
{
    hipLaunchKernelGGL(stream_copy_kernel, dim3(grid_for(n)), dim3(256), 0, stream, src, dst, n);
    return hipGetLastError();
}

// *flag |= 1 if variable q of any cell of rows [0, n0) is not zero (NaN counts as not zero; -0 as zero): the steppers' check, at upload, that a 2-D
// field carries no third momentum (mh_euler_cart_desc.planar) / no azimuthal momentum (mh_cloud_desc.planar)
__global__ void plane_nonzero_kernel(const double* u, int nq, int q, int n0, size_t row_pitch, int32_t* flag, int exact_bits)
{
    const size_t n = (size_t) n0 * row_pitch;
    bool any = false;
    for (size_t t = (size_t) blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (size_t) gridDim.x * blockDim.x)
    {
        const size_t r = t / row_pitch, j = t - r * row_pitch;
        const double x = u[((r + HALO) * nq + q) * row_pitch + j];
        any |= exact_bits ? __double_as_longlong(x) != 0 : ! (x == 0.0);          // (STRICT: +0.0 and nothing else - the reference keeps a -0.0 a -0.0)
    }
    if (__any(any) && (threadIdx.x & 63) == 0) atomicOr(flag, 1);
}

hipError_t plane_nonzero_launch(const double* u, int nq, int q, int n0, size_t row_pitch, int32_t* flag, hipStream_t stream, bool exact_bits)
{
    const size_t n = (size_t) n0 * row_pitch;
    const unsigned blocks = (unsigned) (n / 256 + 1 < 2048 ? n / 256 + 1 : 2048);
    hipLaunchKernelGGL(plane_nonzero_kernel, dim3(blocks), dim3(256), 0, stream, u, nq, q, n0, row_pitch, flag, exact_bits ? 1 : 0);
    return hipGetLastError();
}

hipError_t fill_ghost_rows_launch(double* u, int nq, int n0, size_t row_pitch, int bc_lo0, int bc_hi0, hipStream_t stream)
{
    hipLaunchKernelGGL(fill_ghost_rows_kernel, dim3(grid_for((size_t) nq * HALO * row_pitch)), dim3(256), 0, stream,
                       u, nq, n0, row_pitch, bc_lo0, bc_hi0);
    return hipGetLastError();
}

} // namespace mh
