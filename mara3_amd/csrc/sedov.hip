// 1-D spherical stage for the `sedov` sub-program (BASELINE config 1), for both of its hydro systems
// (SedovProblem<HydroSystem>, src/subprog_sedov.cpp:652-659: mara::srhd by default, mara::euler with newtonian=1):
// piecewise-constant reconstruction, HLLE, forward Euler, volume-integrated
// conserved variables, reflecting inner / zero-gradient outer boundary.
//
// Replaces one evaluation of SedovProblem<mara::euler>::next_solution
// (src/subprog_sedov.cpp:394-421; boundary conditions :231-250; flux :217-229;
// radial source term src/physics_euler.hpp:328-337). Per-cell expression order
// as in SURVEY.md Appendix B ("sedov"). The geometry arrays dv, da, rc are
// built on the host with the same libm calls as the reference (std::pow is not
// bit-identical between glibc and the device library) and live in HBM.
//
// Config 1 is plumbing (512 zones): one thread per zone, neighbours'
// primitives recomputed instead of exchanged. Not a performance kernel.
#include <hip/hip_runtime.h>
#include "launch.hpp"
#include "euler_device.hpp"
#include "srhd_device.hpp"

namespace mh {

// physics traits of the two systems the sub-program is instantiated for
struct SedovEuler
{
    using Gamma = GammaLaw;
    static __device__ Gamma gamma(double g) { return make_gamma_law(g); }
    static __device__ int c2p(const State5& U, const Gamma& g, State5& P) { P = recover_primitive(U, g.gamma, 0.0); return 0; }
    static __device__ State5 hlle(const State5& l, const State5& r, const Gamma& g) { return riemann_hlle<0>(l, r, g); }
    // physics_euler.hpp:328-337
    static __device__ double radial_source(const State5& P, double r, const Gamma&) { return (2.0 * P[4] + P[0] * P[2] * P[2]) / r; }
};
struct SedovSrhd
{
    using Gamma = srhd::Gamma;
    static __device__ Gamma gamma(double g) { return srhd::make_gamma(g); }
    static __device__ int c2p(const State5& U, const Gamma& g, State5& P) { return srhd::recover_primitive(U, g, 0.0, P); }
    static __device__ State5 hlle(const State5& l, const State5& r, const Gamma& g) { return srhd::riemann_hlle<0>(l, r, g); }
    // physics_srhd.hpp:339-348
    static __device__ double radial_source(const State5& P, double r, const Gamma& g) { return (2.0 * P[4] + srhd::enthalpy_density(P, g) * P[2] * P[2]) / r; }
};

template<class S>
__device__ inline State5 sedov_primitive(const double* u, const double* dv, int n, int i, const typename S::Gamma& g, int& status)
{
    // u0 / dv | map(recover_primitive)  (:411)
    double x[5];
#pragma unroll
    for (int q = 0; q < 5; ++q) x[q] = u[(size_t) q * n + i];
    divide_group<5>(x, make_recip(dv[i], 1.0));
    State5 U, P;
#pragma unroll
    for (int q = 0; q < 5; ++q) U[q] = x[q];
    status |= S::c2p(U, g, P);
    return P;
}

template<class S>
__global__ __launch_bounds__(64)
void sedov_stage_kernel(const double* __restrict__ u0, double* __restrict__ u1, const double* __restrict__ dv,
                        const double* __restrict__ da, const double* __restrict__ rc, int n, double gamma, double dt, int32_t* status)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const typename S::Gamma g = S::gamma(gamma);
    int bad = 0, ignore = 0;

    const State5 P0 = sedov_primitive<S>(u0, dv, n, i, g, bad);
    State5 Pl, Pr;
    if (i > 0) Pl = sedov_primitive<S>(u0, dv, n, i - 1, g, ignore);
    else { Pl = P0; Pl[1] = -P0[1]; }                       // reflecting inner: v_r -> -v_r
    if (i < n - 1) Pr = sedov_primitive<S>(u0, dv, n, i + 1, g, ignore);
    else Pr = P0;                                           // zero-gradient outer

    const State5 Flo = S::hlle(Pl, P0, g);
    const State5 Fhi = S::hlle(P0, Pr, g);

    // spherical_geometry_source_terms_radial: only the radial-momentum row is non-zero
    State5 Src;
    Src[0] = 0.0;
    Src[1] = S::radial_source(P0, rc[i], g);
    Src[2] = 0.0;
    Src[3] = 0.0;
    Src[4] = 0.0;

    const double na0 = -da[i], na1 = -da[i + 1];
#pragma unroll
    for (int q = 0; q < 5; ++q)
    {
        const double l0 = Fhi[q] * na1 - Flo[q] * na0;
        const double s0 = Src[q] * dv[i];
        u1[(size_t) q * n + i] = u0[(size_t) q * n + i] + (l0 + s0) * dt;
    }
    if (bad && status)          // {bits, first failing zone}: include/mara_hip.h, mh_step_result
    {
        atomicOr(status, bad);
        atomicMax(reinterpret_cast<unsigned int*>(status) + 1, 0xFFFFFFFFu - (unsigned int) i);
    }
}

// SedovProblem::make_diagnostic_fields and the indices of compute_time_series_data (src/subprog_sedov.cpp:252-308; the shock locator is
// post_shock_locator.hpp:73-170) of the device-resident state. fields [4][n]: specific_entropy, gas_pressure, mass_density, radial
// velocity (Euler) or gamma-beta (SRHD). STRICT primitive recovery; log / pow from the device library.
template<class S>
__global__ __launch_bounds__(64)
void sedov_diag_fields_kernel(const double* __restrict__ u, const double* __restrict__ dv, int n, double gamma, double* fields, int32_t* status)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const typename S::Gamma g = S::gamma(gamma);
    int bad = 0;
    const State5 P = sedov_primitive<S>(u, dv, n, i, g, bad);
    fields[i] = log(P[4] / pow(P[0], gamma));
    fields[(size_t) n + i] = P[4];
    fields[(size_t) 2 * n + i] = P[0];
    fields[(size_t) 3 * n + i] = P[1];
    if (bad) atomicOr(status, bad);
}

// indices[3] = shock, downstream (maximum pressure behind), upstream (pressure plateau ahead); one thread: the scans are sequential
__global__ void sedov_diag_indices_kernel(const double* __restrict__ fields, int n, int32_t* indices)
{
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    const unsigned nn = (unsigned) n;
    const double* s0 = fields;
    const double* pr = fields + n;
    unsigned mid = 0;
    double dsmin = 0.0;
    for (unsigned i = 0; i + 1 < nn; ++i)
    {
        const double ds = s0[i + 1] - s0[i];
        if (i == 0 || ds < dsmin) { dsmin = ds; mid = i; }
    }
    unsigned down = mid;
    for (;;)
    {
        const unsigned a = down - 1;
        if (a >= nn || down >= nn) { down = 0; break; }
        if (pr[a] > pr[down]) --down; else break;
    }
    unsigned up = mid;
    for (;;)
    {
        const unsigned a = up - 1, b = up - 2;
        if (a >= nn - 1 || b >= nn - 1) { up = 0; break; }
        const double da = log(pr[a + 1]) - log(pr[a]);
        const double db = log(pr[b + 1]) - log(pr[b]);
        if (da < 0.5 * db) ++up; else break;
    }
    indices[0] = (int32_t) mid;
    indices[1] = (int32_t) down;
    indices[2] = (int32_t) up;
}

hipError_t sedov_diagnostics_launch(int system, const double* u, const double* dv, int n, double gamma, double* fields, int32_t* indices,
                                    int32_t* status, hipStream_t stream)
{
    if (system == MH_SYSTEM_SRHD) hipLaunchKernelGGL(sedov_diag_fields_kernel<SedovSrhd>, dim3((n + 63) / 64), dim3(64), 0, stream, u, dv, n, gamma, fields, status);
    else                          hipLaunchKernelGGL(sedov_diag_fields_kernel<SedovEuler>, dim3((n + 63) / 64), dim3(64), 0, stream, u, dv, n, gamma, fields, status);
    hipLaunchKernelGGL(sedov_diag_indices_kernel, dim3(1), dim3(64), 0, stream, fields, n, indices);
    return hipGetLastError();
}

hipError_t sedov_stage_launch(int system, const double* u0, double* u1, const double* dv, const double* da, const double* rc,
                              int n, double gamma, double dt, int32_t* status, hipStream_t stream)
{
    if (system == MH_SYSTEM_SRHD) hipLaunchKernelGGL(sedov_stage_kernel<SedovSrhd>, dim3((n + 63) / 64), dim3(64), 0, stream, u0, u1, dv, da, rc, n, gamma, dt, status);
    else                          hipLaunchKernelGGL(sedov_stage_kernel<SedovEuler>, dim3((n + 63) / 64), dim3(64), 0, stream, u0, u1, dv, da, rc, n, gamma, dt, status);
    return hipGetLastError();
}

} // namespace mh
