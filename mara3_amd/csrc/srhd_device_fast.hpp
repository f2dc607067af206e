// MH_ARITH_FAST device physics for mara::srhd on gfx950: the formulas of srhd_device.hpp (same reference lines) with the
// arithmetic freedoms of euler_device_fast.hpp - x / d as x * (v_rcp_f64 + one third-order step), sqrt via v_rsq_f64 +
// one third-order step and a correction, explicit FMAs, no literal 0/1 normal-vector products. NOT bit-exact: bound by the north star's tolerance
// (conserved-variable L1 <= 1e-12 relative to the field scale, asserted in tests/test_gpu_srhd_cloud.py::test_fast_*).
// The Newton iteration of recover_primitive keeps the reference's start value, update and stopping rule, so it walks the
// same sequence of iterates up to rounding; where |f| lands within rounding of the 1e-10 threshold one more or one fewer
// step is taken, which moves p by O(f / f') ~ 1e-10 p in that cell - rare, and far inside the L1 bound.
#pragma once
#include <hip/hip_runtime.h>
#include "euler_device_fast.hpp"
#include "srhd_device.hpp"

namespace mh {
namespace srhd_fast {

using srhd::Gamma;

// The iteration starts from p = 0 like the reference's (src/physics_srhd.hpp:378) and that is part of the RESULT, not only of the cost:
// in cold gas |f| < 1e-10 holds at the first test, so the accepted pressure is ONE Newton step from the start value. Starting from the
// neighbouring cell's pressure instead (measured: it would save a step or two per cell) lands 1e-9 away in relative terms - closer to
// the root, but outside the 1e-12 agreement with the reference that this arithmetic mode promises.
// PLANAR (here and below): the azimuthal four-velocity / momentum is identically zero - the `cloud` problem as upstream sets it up (radial
// envelope, radial nozzle, 2-D axisymmetric: src/subprog_cloud.cpp:626-660, :466-493), which the reference carries as zeros through every
// operator. Every term with that component is left out and the component is returned as 0: `x + 0` and `fma(0, 0, x)` are x exactly, so
// the other four components keep their bits (cloud_fused.hip takes this form where the stepper has verified field and nozzle row).
template<bool PLANAR = false>
__device__ inline int recover_primitive(const State5& U, const Gamma& g, double temperature_floor, State5& P)
{
    const double gm = g.gamma;
    const double D = U[0], tau = U[4];
    const double SS2 = __builtin_fma(U[2], U[2], U[1] * U[1]);
    const double SS = PLANAR ? SS2 : __builtin_fma(U[3], U[3], SS2);
    const double rD = fast::rcp_nr(D);
    bool solution_found = false;
    int iteration = 0;
    double W0 = 1.0, rW0 = 1.0;
    double p = 0.0;

    while (iteration < 50)
    {
        const double x = tau + D + p;
        const double rx = fast::rcp_nr(x);
        const double v2 = __builtin_fmin(SS * rx * rx, 1.0 - 1e-10);
        // W = 1 / sqrt(1 - v2) straight from the inverse root (the reference forms W2 = 1 / (1 - v2) and takes its root), 1 / W = (1 - v2) W
        const double omv = 1.0 - v2;
        const double W = fast::rsqrt_fast(omv);
        const double rW = omv * W;
        const double W2 = W * W;
        const double e = __builtin_fma(p, 1.0 - W2, __builtin_fma(D, 1.0 - W, tau)) * rD * rW;
        const double d = D * rW;
        const double h = 1.0 + e + p * W * rD;
        // f / g with g = v2 cs2 - 1, cs2 = gamma p / (d h):  f d h / (v2 gamma p - d h) - one reciprocal for the sound speed and the Newton step
        const double f = __builtin_fma(d * e, gm - 1.0, -p);
        const double dh = d * h;
        p = __builtin_fma(-(f * dh), fast::rcp_nr(__builtin_fma(v2, gm * p, -dh)), p);
        if (fabs(f) < 1e-10)
        {
            W0 = W;
            rW0 = rW;
            solution_found = true;
            break;
        }
        ++iteration;
    }
    if (temperature_floor > 0.0) p = __builtin_fmax(p, temperature_floor * D * rW0);
    const double s = W0 * fast::rcp_nr(tau + D + p);
    P[0] = D * rW0;
    P[1] = s * U[1];
    P[2] = s * U[2];
    P[3] = PLANAR ? 0.0 : s * U[3];
    P[4] = p;
    int status = 0;
    if (! solution_found) status |= MH_STATUS_C2P_FAILED;
    if (P[0] <= 0.0) status |= MH_STATUS_NEG_DENSITY;
    if (P[4] <= 0.0) status |= MH_STATUS_NEG_PRESSURE;
    if (W0 != W0) status |= MH_STATUS_NAN;
    return status;
}

template<int AXIS, bool PLANAR = false>
__device__ inline void side_fast(const State5& P, const Gamma& g, State5& U, State5& F, double& lam_m, double& lam_p)
{
    static_assert(! PLANAR || AXIS != 2, "a planar state has no azimuthal axis");
    const double uu2 = __builtin_fma(P[2], P[2], P[1] * P[1]);
    const double uu = PLANAR ? uu2 : __builtin_fma(P[3], P[3], uu2);
    const double x1 = 1.0 + uu;
    const double rW = fast::rsqrt_fast(x1);
    const double W = x1 * rW;                      // sqrt(x) = x rsqrt(x), to an ulp or two (x >= 1 here)
    const double H = __builtin_fma(P[4], g.hfac, P[0]);
    const double D = P[0] * W;
    const double p = P[4];
    // 1 / H and 1 / (1 - vv c2) = H / (H - vv gamma p) from ONE reciprocal, that of H (H - vv gamma p)
    const double vv = uu * rW * rW;                // uu / (1 + uu)
    const double B = __builtin_fma(-vv, g.gamma * p, H);
    const double rHB = fast::rcp_nr(H * B);
    const double rH = B * rHB;
    const double Dh = H * W;                       // D h = rho W (H / rho)
    U[0] = D;
    U[1] = Dh * P[1];
    U[2] = Dh * P[2];
    U[3] = PLANAR ? 0.0 : Dh * P[3];
    U[4] = __builtin_fma(Dh, W, -p) - D;
    const double v = P[1 + AXIS] * rW;
    F[0] = v * U[0];
    F[1] = AXIS == 0 ? __builtin_fma(v, U[1], p) : v * U[1];
    F[2] = AXIS == 1 ? __builtin_fma(v, U[2], p) : v * U[2];
    F[3] = PLANAR ? 0.0 : (AXIS == 2 ? __builtin_fma(v, U[3], p) : v * U[3]);
    F[4] = v * (U[4] + p);
    const double c2 = g.gamma * p * rH;
    const double v2 = v * v;
    // sqrt(x) = x rsqrt(x) without the correction step (an ulp or two: it only enters the signal speeds). x == 0 would give 0 * inf and is
    // answered with 0; a NEGATIVE or NaN radicand (an unphysical face state: c2 outside [0, 1]) stays NaN, as the reference's sqrt makes it
    // (src/physics_srhd.hpp:283-295) - riemann_hlle below carries it into the flux, so the next recover_primitive raises MH_STATUS_NAN
    const double k2 = c2 * (1 - vv) * (1 - vv * c2 - v2 * (1 - c2));
    const double k0 = k2 == 0.0 ? 0.0 : k2 * fast::rsqrt_fast(k2);
    const double rden = H * H * rHB;
    const double a = v * (1 - c2);
    lam_m = (a - k0) * rden;
    lam_p = (a + k0) * rden;
}

template<int AXIS, bool PLANAR = false> __device__ inline State5 riemann_hlle(const State5& Pl, const State5& Pr, const Gamma& g)
{
    State5 Ul, Ur, Fl, Fr;
    double alm, alp, arm, arp;
    side_fast<AXIS, PLANAR>(Pl, g, Ul, Fl, alm, alp);
    side_fast<AXIS, PLANAR>(Pr, g, Ur, Fr, arm, arp);
    const double ap = __builtin_fmax(0.0, __builtin_fmax(alp, arp));
    const double am = __builtin_fmin(0.0, __builtin_fmin(alm, arm));
    const double rden = fast::rcp_nr(ap - am);
    // fmax / fmin drop a NaN signal speed (v_max_f64 returns the other operand); it re-enters through the weight of the jump term,
    // so an invalid face state poisons the whole flux instead of being stepped over
    const double poison = (alm + arm) * 0.0;
    const double wl = ap * rden, wr = am * rden, wu = __builtin_fma(wl, am, poison);       // the three weights divided once (euler_device_fast.hpp)
    State5 F;
#pragma unroll
    for (int q = 0; q < 5; ++q)
    {
        if (PLANAR && q == 3) { F[q] = 0.0; continue; }
        F[q] = __builtin_fma(Ur[q] - Ul[q], wu, __builtin_fma(-Fr[q], wr, Fl[q] * wl));
    }
    return F;
}

// rr = 1 / r (the cloud kernel takes it from the host's per-row table)
template<bool PLANAR = false>
__device__ inline State5 source_terms_rinv(const State5& P, double rr, double cotq, const Gamma& g)
{
    const double ur = P[1], uq = P[2], up = P[3], pg = P[4];
    const double H = __builtin_fma(P[4], g.hfac, P[0]);
    State5 S;
    S[0] = 0.0;
    if constexpr (PLANAR)
    {
        // (up = 0: fma(uq, uq, 0) = uq uq and fma(0, cotq, -ur uq) = -ur uq to the bit)
        S[1] = __builtin_fma(H, uq * uq, 2.0 * pg) * rr;
        S[2] = __builtin_fma(H, -ur * uq, cotq * pg) * rr;
        S[3] = 0.0;
    }
    else
    {
        S[1] = __builtin_fma(H, __builtin_fma(uq, uq, up * up), 2.0 * pg) * rr;
        S[2] = __builtin_fma(H, __builtin_fma(up * up, cotq, -ur * uq), cotq * pg) * rr;
        S[3] = -up * H * __builtin_fma(uq, cotq, ur) * rr;
    }
    S[4] = 0.0;
    return S;
}
__device__ inline State5 source_terms(const State5& P, double r, double cotq, const Gamma& g) { return source_terms_rinv(P, fast::rcp_nr(r), cotq, g); }

} // namespace srhd_fast

// ---- arithmetic policies of the cloud kernel ---------------------------------------------------------------------
// PLANAR: the field and the nozzle row carry no azimuthal component (STRICT: the bit pattern of +0.0; see srhd_device.hpp) - the stage kernel
// neither loads nor computes it and stores +0.0, which is what the reference's own operations return for it
template<bool PLANAR>
struct SrhdStrictT
{
    static constexpr bool planar = PLANAR;
    static constexpr bool live(int q) { return ! (PLANAR && q == 3); }
#ifndef MH_CLOUD_STRICT_PLANAR_WAVES
#define MH_CLOUD_STRICT_PLANAR_WAVES 2
#endif
    static constexpr int min_waves_per_simd = PLANAR ? MH_CLOUD_STRICT_PLANAR_WAVES : 2;      // general: 163-167 VGPRs, three waves fit anyway
    static constexpr bool table_geometry = false;     // geometry factors formed per cell in the reference's order
    static constexpr bool exact_zero_products = true; // pole slopes / fluxes as (neighbour's value) * 0, like extend_zeros: NaN and -0 propagate
    static constexpr bool group_own_row_loads = false;// (cloud.hip: where the update's loads of the row's conserved values are requested)
    static constexpr bool lds_row_ring = true;        // the row's conserved values wait for the update in a per-wave LDS ring
    static __device__ inline void to_density(double (&x)[5], double dv, double)
    {
        if constexpr (PLANAR)
        {
            double y[4] = {x[0], x[1], x[2], x[4]};
            divide_group<4>(y, make_recip(dv, 1.0));
            x[0] = y[0]; x[1] = y[1]; x[2] = y[2]; x[3] = 0.0; x[4] = y[3];
        }
        else divide_group<5>(x, make_recip(dv, 1.0));
    }
    static __device__ inline int c2p(const State5& U, const srhd::Gamma& g, double tf, State5& P) { return srhd::recover_primitive<PLANAR>(U, g, tf, P); }
    static __device__ inline State5 source(const State5& P, double r, double, double cot, const srhd::Gamma& g) { return srhd::source_terms<PLANAR>(P, r, cot, g); }
    using Limiter = typename StrictArithT<PLANAR>::Limiter;
    static __device__ inline Limiter limiter(double theta) { return StrictArithT<PLANAR>::limiter(theta); }
    static __device__ inline State5 plm(const State5& l, const State5& c, const State5& r, const Limiter& lim) { return StrictArithT<PLANAR>::plm(l, c, r, lim); }
    static __device__ inline State5 plus(const State5& P, const State5& G, const Limiter& lim) { return StrictArithT<PLANAR>::plus(P, G, lim); }
    static __device__ inline State5 minus(const State5& P, const State5& G, const Limiter& lim) { return StrictArithT<PLANAR>::minus(P, G, lim); }
    template<int AXIS> static __device__ inline State5 hlle(const State5& Pl, const State5& Pr, const srhd::Gamma& g) { return srhd::riemann_hlle<AXIS, PLANAR>(Pl, Pr, g); }
    // u0 + ((Fr_hi (-dAr_hi) - Fr_lo (-dAr_lo)) + (Fq_hi (-dAq_hi) - Fq_lo (-dAq_lo)) + S dv) dt     src/subprog_cloud.cpp:572-574
    static __device__ inline double update(double u0, double fxl, double fxh, double fyl, double fyh, double nArl, double nArh, double nAql, double nAqh, double s, double dv, double dt)
    {
        const double lr = fxh * nArh - fxl * nArl;
        const double lq = fyh * nAqh - fyl * nAql;
        const double s0 = s * dv;
        return u0 + (lr + lq + s0) * dt;
    }
    static __device__ inline double combine(double base, double u1, double w) { return base * (1.0 - w) + u1 * w; }
};
using SrhdStrict = SrhdStrictT<false>;
using SrhdStrictPlanar = SrhdStrictT<true>;

template<bool PLANAR>
struct SrhdFastT
{
    static constexpr bool planar = PLANAR;
    static constexpr bool live(int q) { return ! (PLANAR && q == 3); }
    static constexpr int min_waves_per_simd = 3;      // hold the allocation at 168 VGPRs
    static constexpr bool table_geometry = true;      // per-row x per-column factors from the host's tables (mh_cloud_pack_geometry)
    static constexpr bool exact_zero_products = false;// pole slopes / fluxes are plain zeros (zero lane constants, cloud.hip: lim_polar)
    static constexpr bool group_own_row_loads = true;
    static constexpr bool lds_row_ring = true;        // ... or wait for it in a per-wave LDS ring
    static __device__ inline void to_density(double (&x)[5], double, double inv_dv) { for (int q = 0; q < 5; ++q) if (live(q)) x[q] *= inv_dv; }
    static __device__ inline int c2p(const State5& U, const srhd::Gamma& g, double tf, State5& P) { return srhd_fast::recover_primitive<PLANAR>(U, g, tf, P); }
    static __device__ inline State5 source(const State5& P, double, double inv_r, double cot, const srhd::Gamma& g) { return srhd_fast::source_terms_rinv<PLANAR>(P, inv_r, cot, g); }
    // the limiter on unscaled differences, theta / 2 in the face-state FMA (euler_device_fast.hpp: FastArith)
    using Limiter = typename FastArithT<PLANAR>::Limiter;
    static __device__ inline Limiter limiter(double theta) { return FastArithT<PLANAR>::limiter(theta); }
    static __device__ inline State5 plm(const State5& l, const State5& c, const State5& r, const Limiter& lim) { return FastArithT<PLANAR>::plm(l, c, r, lim); }
    static __device__ inline State5 plus(const State5& P, const State5& G, const Limiter& lim) { return FastArithT<PLANAR>::plus(P, G, lim); }
    static __device__ inline State5 minus(const State5& P, const State5& G, const Limiter& lim) { return FastArithT<PLANAR>::minus(P, G, lim); }
    template<int AXIS> static __device__ inline State5 hlle(const State5& Pl, const State5& Pr, const srhd::Gamma& g) { return srhd_fast::riemann_hlle<AXIS, PLANAR>(Pl, Pr, g); }
    static __device__ inline double update(double u0, double fxl, double fxh, double fyl, double fyh, double nArl, double nArh, double nAql, double nAqh, double s, double dv, double dt)
    {
        const double lr = __builtin_fma(fxh, nArh, -fxl * nArl);
        const double lq = __builtin_fma(fyh, nAqh, -fyl * nAql);
        return __builtin_fma(__builtin_fma(s, dv, lr + lq), dt, u0);
    }
    static __device__ inline double combine(double base, double u1, double w) { return __builtin_fma(u1, w, base * (1.0 - w)); }
};
using SrhdFast = SrhdFastT<false>;
using SrhdFastPlanar = SrhdFastT<true>;

} // namespace mh
