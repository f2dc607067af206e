// 1-D spherical Euler stage for the `sedov` sub-program (BASELINE config 1):
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
#include "euler_device.hpp"
#include "launch.hpp"

namespace mh {

__device__ inline State5 sedov_primitive(const double* u, const double* dv, int n, int i, const GammaLaw& g)
{
    // u0 / dv | map(recover_primitive)  (:411)
    double x[5];
#pragma unroll
    for (int q = 0; q < 5; ++q) x[q] = u[(size_t) q * n + i];
    divide_group<5>(x, make_recip(dv[i], 1.0));
    State5 U;
#pragma unroll
    for (int q = 0; q < 5; ++q) U[q] = x[q];
    return recover_primitive(U, g.gamma, 0.0);
}

__global__ __launch_bounds__(64)
void sedov_stage_kernel(const double* __restrict__ u0, double* __restrict__ u1, const double* __restrict__ dv,
                        const double* __restrict__ da, const double* __restrict__ rc, int n, double gamma, double dt)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const GammaLaw g = make_gamma_law(gamma);

    const State5 P0 = sedov_primitive(u0, dv, n, i, g);
    State5 Pl, Pr;
    if (i > 0) Pl = sedov_primitive(u0, dv, n, i - 1, g);
    else { Pl = P0; Pl[1] = -P0[1]; }                       // reflecting inner: v_r -> -v_r
    if (i < n - 1) Pr = sedov_primitive(u0, dv, n, i + 1, g);
    else Pr = P0;                                           // zero-gradient outer

    const State5 Flo = riemann_hlle<0>(Pl, P0, g);
    const State5 Fhi = riemann_hlle<0>(P0, Pr, g);

    // spherical_geometry_source_terms_radial: only the radial-momentum row is non-zero
    const double vq = P0[2], pg = P0[4], d = P0[0];
    State5 S;
    S[0] = 0.0;
    S[1] = (2.0 * pg + d * vq * vq) / rc[i];
    S[2] = 0.0;
    S[3] = 0.0;
    S[4] = 0.0;

    const double na0 = -da[i], na1 = -da[i + 1];
#pragma unroll
    for (int q = 0; q < 5; ++q)
    {
        const double l0 = Fhi[q] * na1 - Flo[q] * na0;
        const double s0 = S[q] * dv[i];
        u1[(size_t) q * n + i] = u0[(size_t) q * n + i] + (l0 + s0) * dt;
    }
}

hipError_t sedov_stage_launch(const double* u0, double* u1, const double* dv, const double* da, const double* rc,
                              int n, double gamma, double dt, hipStream_t stream)
{
    hipLaunchKernelGGL(sedov_stage_kernel, dim3((n + 63) / 64), dim3(64), 0, stream, u0, u1, dv, da, rc, n, gamma, dt);
    return hipGetLastError();
}

} // namespace mh
