// Internal launcher declarations shared by the translation units of libmara_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/mara_hip.h"

namespace mh {

// Events carried by a stage launch itself (hipExtLaunchKernel): `stop` fires on the dispatch packet's own completion signal, `start`
// (optional) on its start - no marker packets on the stream. With nothing to launch they are recorded on the stream instead.
struct LaunchEvents { hipEvent_t start = nullptr, stop = nullptr; };

hipError_t euler2d_stage_launch(const mh_euler_cart_desc* d, const double* u_in, const double* u_base, double* u_out,
                                double dt, double weight, int row_begin, int row_end, int32_t* status, hipStream_t stream,
                                LaunchEvents ev = LaunchEvents());

// two disjoint row ranges [a0,a1) and [b0,b1) in one launch (the two edge strips of a slab)
hipError_t euler2d_stage_launch2(const mh_euler_cart_desc* d, const double* u_in, const double* u_base, double* u_out,
                                 double dt, double weight, int a0, int a1, int b0, int b1, int32_t* status, hipStream_t stream,
                                 LaunchEvents ev = LaunchEvents());

hipError_t euler3d_stage_launch(const mh_euler_cart_desc* d, const double* u_in, const double* u_base, double* u_out,
                                double dt, double weight, int row_begin, int row_end, int32_t* status, hipStream_t stream);

hipError_t fill_ghost_rows_launch(double* u, int nq, int n0, size_t row_pitch, int bc_lo0, int bc_hi0, hipStream_t stream);
hipError_t aos_to_soa_launch(const double* aos, double* soa, int nq, int n0, size_t row_pitch, hipStream_t stream);
hipError_t stream_copy_launch(const double* src, double* dst, size_t n, hipStream_t stream);
hipError_t soa_to_aos_launch(const double* soa, double* aos, int nq, int n0, size_t row_pitch, hipStream_t stream);

hipError_t sedov_stage_launch(int system, const double* u0, double* u1, const double* dv, const double* da, const double* rc,
                              int n, double gamma, double dt, int32_t* status, hipStream_t stream);

hipError_t sedov_diagnostics_launch(int system, const double* u, const double* dv, int n, double gamma, double* fields, int32_t* indices,
                                    int32_t* status, hipStream_t stream);

hipError_t cloud_stage_launch(const mh_cloud_desc* d, const double* geom_dev, const double* inflow_dev,
                              const double* u_in, const double* u_base, double* u_out, double dt, double weight,
                              int row_begin, int row_end, int32_t* status, hipStream_t stream);

// cloud_diag.hip: make_diagnostic_fields of a device-resident cloud state; fields [5][nr][nq], work [4][nr][nq], columns [15][nq] (device)
hipError_t cloud_diagnostics_launch(const mh_cloud_desc* d, const double* geom_dev, const double* u, const double units[3],
                                    double* fields, double* work, double* columns, int32_t* status, hipStream_t stream);

// thread-local error text for the C ABI
void set_error(const char* fmt, ...);
int  hip_fail(hipError_t e, const char* what);

} // namespace mh

#define MH_HIP_TRY(call) do { hipError_t _e = (call); if (_e != hipSuccess) return mh::hip_fail(_e, #call); } while (0)
