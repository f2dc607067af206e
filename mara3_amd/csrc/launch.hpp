// Internal launcher declarations shared by the translation units of libmara_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/mara_hip.h"

namespace mh {

// Events carried by a stage launch itself (hipExtLaunchKernel): `stop` fires on the dispatch packet's own completion signal, `start`
// (optional) on its start - no marker packets on the stream. With nothing to launch they are recorded on the stream instead.
struct LaunchEvents { hipEvent_t start = nullptr, stop = nullptr; };

// compute units of the CURRENT device (256 on an MI355X; a partitioned device or another part has fewer): the launchers size their chunks
// for whole residency rounds of the chip they run on. Queried once per device.
inline int device_cu_count()
{
    static int cached[64] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
    if (cached[dev] == 0)
    {
        int n = 0;
        cached[dev] = (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) ? n : 256;
    }
    return cached[dev];
}

hipError_t euler2d_stage_launch(const mh_euler_cart_desc* d, const double* u_in, const double* u_base, double* u_out,
                                double dt, double weight, int row_begin, int row_end, int32_t* status, hipStream_t stream,
                                LaunchEvents ev = LaunchEvents());

// two disjoint row ranges [a0,a1) and [b0,b1) in one launch (the two edge strips of a slab)
hipError_t euler2d_stage_launch2(const mh_euler_cart_desc* d, const double* u_in, const double* u_base, double* u_out,
                                 double dt, double weight, int a0, int a1, int b0, int b1, int32_t* status, hipStream_t stream,
                                 LaunchEvents ev = LaunchEvents());

// both stages of an RK2 step of a whole 2-D field in one launch (euler2d_fused.hip): u_out = u_in * 0.5 + advance(advance(u_in)) * 0.5
// with_cuts: MH_BC_EXTERNAL sides on axis 0 too - the field then holds FOUR rows of the neighbour beyond such a side (rows -4 .. -1 /
// n0 .. n0 + 3 at the usual row pitch: two more than the stored ghost rows), which the caller allocates and exchanges once per step
bool euler2d_fused_rk2_available(const mh_euler_cart_desc* d, bool with_cuts = false);
hipError_t euler2d_fused_rk2_launch(const mh_euler_cart_desc* d, const double* u_in, double* u_out, double dt, int32_t* status, hipStream_t stream,
                                    LaunchEvents ev = LaunchEvents(), bool with_cuts = false);
// ... over rows [a, b) and, in the same launch, [a2, b2) (none if b2 <= a2): a slab with neighbours runs its two edge strips, then the rest
// late_blocks: workgroups of another launch that hold slots when this one starts (the slab's edge launch): the launch then ends in shorter
// chunks, launched last, so that the workgroups that start late do not end late (euler2d_fused.hip: TAPER)
hipError_t euler2d_fused_rk2_launch_rows(const mh_euler_cart_desc* d, const double* u_in, double* u_out, double dt, int a, int b, int a2, int b2,
                                         int32_t* status, hipStream_t stream, LaunchEvents ev = LaunchEvents(), bool with_cuts = false, int late_blocks = 0);
int euler2d_fused_rk2_blocks_per_chunk(const mh_euler_cart_desc* d);          // workgroups per chunk of rows (strips of 116 columns)

hipError_t euler3d_stage_launch(const mh_euler_cart_desc* d, const double* u_in, const double* u_base, double* u_out,
                                double dt, double weight, int row_begin, int row_end, int32_t* status, hipStream_t stream);

// 3-D stage on a field WITH stored transverse ghost layers (a block of a 3-axis decomposition), over a list of boxes.
// Layout: plane_stride = (n1 + 2 g1) * (n2 + 2 g2); cell (i, j, k), variable q at ((i + 2) * 5 + q) * plane_stride + (j + g1) * (n2 + 2 g2) + (k + g2).
// Each transverse side has its own boundary kind; MH_BC_EXTERNAL sides read the stored ghost cells.
struct Euler3dLayout { int g1 = 0, g2 = 0; int bc_lo1 = 0, bc_hi1 = 0, bc_lo2 = 0, bc_hi2 = 0; };
// rows [r0, r1) of axis 0 x tiles [t0, t1) of euler3d_tile_rows() axis-1 rows x strips [s0, s1) of 60 axis-2 columns (the kernel's work items)
struct Euler3dBox { int r0, r1, t0, t1, s0, s1; };
void euler3d_tiling(const mh_euler_cart_desc* d, int* ntiles1, int* nstrips);
int euler3d_tile_rows(const mh_euler_cart_desc* d);          // axis-1 rows of a tile: 4 (STRICT) or 8 (FAST)
hipError_t euler3d_stage_launch_boxes(const mh_euler_cart_desc* d, const Euler3dLayout& lay, const Euler3dBox* boxes, int nboxes,
                                      const double* u_in, const double* u_base, double* u_out, double dt, double weight,
                                      int32_t* status, hipStream_t stream);

// *flag |= 1 if variable q of any cell of rows [0, n0) of a field is not zero (the steppers' planarity check at upload)
// (exact_bits: anything but the bit pattern of +0.0 counts - the STRICT kernels' condition; otherwise -0.0 is a zero too)
hipError_t plane_nonzero_launch(const double* u, int nq, int q, int n0, size_t row_pitch, int32_t* flag, hipStream_t stream, bool exact_bits = false);
hipError_t fill_ghost_rows_launch(double* u, int nq, int n0, size_t row_pitch, int bc_lo0, int bc_hi0, hipStream_t stream);
hipError_t aos_to_soa_launch(const double* aos, double* soa, int nq, int n0, size_t row_pitch, hipStream_t stream);
hipError_t stream_copy_launch(const double* src, double* dst, size_t n, hipStream_t stream);
hipError_t soa_to_aos_launch(const double* soa, double* aos, int nq, int n0, size_t row_pitch, hipStream_t stream);

// padded block fields (Euler3dLayout): host AoS [n0][n1][n2][5] <-> device, and one transverse face <-> a message buffer [n0][5][len1][len2]
hipError_t block_transpose_launch(bool to_soa, const double* src, double* dst, int n0, int n1, int n2, int g1, int g2, hipStream_t stream);
hipError_t block_face_launch(bool pack, double* field, double* buf, int n0, int j0, int len1, int k0, int len2,
                             int n1, int n2, int g1, int g2, hipStream_t stream);

hipError_t sedov_stage_launch(int system, const double* u0, double* u1, const double* dv, const double* da, const double* rc,
                              int n, double gamma, double dt, int32_t* status, hipStream_t stream);

hipError_t sedov_diagnostics_launch(int system, const double* u, const double* dv, int n, double gamma, double* fields, int32_t* indices,
                                    int32_t* status, hipStream_t stream);

hipError_t cloud_stage_launch(const mh_cloud_desc* d, const double* geom_dev, const double* inflow_dev,
                              const double* u_in, const double* u_base, double* u_out, double dt, double weight,
                              int row_begin, int row_end, int32_t* status, hipStream_t stream);

// both stages of an RK2 step of a whole `cloud` field in one launch (cloud_fused.hip; MH_ARITH_FAST, PLM, both radial sides physical):
// u_out = u_in * 0.5 + advance(advance(u_in)) * 0.5 with the nozzle row of the step-start time in both stages
// with_cuts: MH_BC_EXTERNAL radial sides too - the field then holds FOUR rows of the neighbour beyond such a side (rows -4 .. -1 / nr .. nr + 3),
// which the caller allocates and exchanges once per step; the nozzle rows apply on the slab that owns row 0 only
bool cloud_fused_rk2_available(const mh_cloud_desc* d, bool with_cuts = false);
hipError_t cloud_fused_rk2_launch(const mh_cloud_desc* d, const double* geom_dev, const double* inflow_dev, const double* u_in, double* u_out,
                                  double dt, int32_t* status, hipStream_t stream);
// ... over rows [a, b) and, in the same launch, [a2, b2) (none if b2 <= a2): a radial slab with neighbours runs its two edge strips, then the rest
hipError_t cloud_fused_rk2_launch_rows(const mh_cloud_desc* d, const double* geom_dev, const double* inflow_dev, const double* u_in, double* u_out,
                                       double dt, int a, int b, int a2, int b2, int32_t* status, hipStream_t stream, bool with_cuts = false, int late_blocks = 0);
int cloud_fused_rk2_blocks_per_chunk(const mh_cloud_desc* d);
void euler2d_fused_last_cut(int out[4]);          // {chunk rows, chunk rows of the second segment, chunks of the first segment, chunks} of the last launch (tests)
void cloud_fused_last_cut(int out[4]);

// cloud_diag.hip: make_diagnostic_fields of a device-resident cloud state; fields [5][nr][nq], work [4][nr][nq], columns [15][nq] (device)
hipError_t cloud_diagnostics_launch(const mh_cloud_desc* d, const double* geom_dev, const double* u, const double units[3],
                                    double* fields, double* work, double* columns, int32_t* status, hipStream_t stream);

// row-range guard (row_check.hpp, -DMH_CHECK_ROWS): {smallest, largest} row index the kernels of a translation unit requested since the last
// reset; false where the library was built without the guard
bool rows_requested_euler2d(int32_t out[2], int reset);
bool rows_requested_euler2d_fused(int32_t out[2], int reset);
bool rows_requested_cloud(int32_t out[2], int reset);
bool rows_requested_cloud_fused(int32_t out[2], int reset);
bool rows_requested_euler3d(int32_t out[2], int reset);
bool rows_requested_euler3d_fast(int32_t out[2], int reset);
bool rows_requested_binary(int32_t out[2], int reset);
bool rows_requested_binary_fast(int32_t out[2], int reset);

// thread-local error text for the C ABI
void set_error(const char* fmt, ...);
int  hip_fail(hipError_t e, const char* what);

} // namespace mh

#define MH_HIP_TRY(call) do { hipError_t _e = (call); if (_e != hipSuccess) return mh::hip_fail(_e, #call); } while (0)
