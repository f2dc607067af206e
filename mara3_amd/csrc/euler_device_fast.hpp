// MH_ARITH_FAST device physics for gfx950: the same scheme as euler_device.hpp
// (same formulas, citing the same reference lines) with the arithmetic freedoms
// BASELINE.json's tolerance allows (conserved-variable L1 <= 1e-12 vs the
// reference; NOT bit-exact):
//   * explicit fused multiply-adds,
//   * x / d  ->  x * r with r = v_rcp_f64(d) + one third-order step (0.5 ulp measured), one r per denominator,
//   * inverse roots via v_rsq_f64 + one third-order step (< 1 ulp); sqrt(x) = x * rsqrt(x) + one correction,
//   * the literal 0.0 / 1.0 normal-vector products removed (exact for finite data),
//   * the limiter from min / max alone (minmod_between: identical to the sign-product form for finite arguments except for the sign of
//     an exact zero), on UNSCALED one-sided differences formed once per face where the kernel can share them (theta rides in the
//     face-state FMA),
//   * HLLC with wave speeds from the primitives, the conserved state and flux of the sampled side only, the star flux as a blend
//     F_K + c (S* U_K - F_K + p* D) with c = 0 outside the star region instead of selects on the results.
// Less than half the executed instructions of the strict path; see DESIGN.md §5.2 / §6 for the measured difference.
#pragma once
#include <hip/hip_runtime.h>
#include "euler_device.hpp"

namespace mh {
namespace fast {

// v_rcp_f64 / v_rsq_f64 are good to ~2^-25 (measured: 2.5e8 ulp, scripts/probes/rcp_rsq_accuracy.hip); ONE third-order step brings
// either below an ulp (measured: 0.5 ulp for the reciprocal, < 1 ulp for the inverse root), one instruction less than two Newton steps
// for the reciprocal and about half the Goldschmidt sequence for the root.
__device__ inline double rcp_nr(double x)
{
    const double r = __builtin_amdgcn_rcp(x);
    const double e = __builtin_fma(-x, r, 1.0);
    return __builtin_fma(r, __builtin_fma(e, e, e), r);          // r (1 + e + e^2)
}

// 1 / sqrt(x), x > 0 (x == 0 gives NaN: callers guard with fmax)
__device__ inline double rsqrt_fast(double x)
{
    const double y = __builtin_amdgcn_rsq(x);
    const double e = __builtin_fma(-(x * y), y, 1.0);
    return __builtin_fma(y, __builtin_fma(0.375, e, 0.5) * e, y); // y (1 + e / 2 + 3 e^2 / 8)
}

// returns g ~ sqrt(x) and h2 ~ 1/sqrt(x) (x > 0; x == 0 gives g = 0, h2 = NaN; x < 0 gives NaN for both, as sqrt does)
__device__ inline void sqrt_rsqrt(double x, double& g, double& h2)
{
    h2 = rsqrt_fast(x);
    const double gg = x * h2;
    const double gc = __builtin_fma(__builtin_fma(-gg, gg, x), 0.5 * h2, gg);             // one correction step
    g = x == 0.0 ? 0.0 : gc;                                                               // a negative x keeps its NaN
}

__device__ inline double sqrt_fast(double x)
{
    double g, h2;
    sqrt_rsqrt(x, g, h2);
    return g;
}

// minmod(a, b, c) with b BETWEEN a and c in sign (b is a positive multiple of a + c): the common-sign smallest magnitude, else 0, as
//     max(min(b, max(min(a, c), 0)), min(max(a, c), 0))
// - both positive: min(b, min(a, c)); both negative: max(b, max(a, c)); mixed: 0. Same value as the reference's
// 0.25*|sgn a + sgn b|*(sgn a + sgn c)*min(|a|,|b|,|c|) for finite arguments (the sign of an exact zero may differ); 6 fp64
// instructions, no selects and no integer sign logic.
// The six operations are written as the instructions themselves (MH_FAST_ASM_MINMOD, default): v_min_f64 / v_max_f64 are what __builtin_fmin /
// __builtin_fmax become, but in front of them the compiler canonicalises (v_max_f64 x, x) every operand it cannot prove free of signalling NaNs -
// each value that reached the lane through a DPP move, 10 of the row loop's ~465 VALU instructions per wave and row in the 2-D kernels. The
// instructions quiet such operands themselves (IEEE mode), so the result is the same bit for bit.
#ifndef MH_FAST_ASM_MINMOD
#define MH_FAST_ASM_MINMOD 1
#endif
__device__ inline double minmod_between(double a, double b, double c)
{
#if MH_FAST_ASM_MINMOD && defined(__HIP_DEVICE_COMPILE__)
    double lo, hi, t;
    asm("v_min_f64 %0, %1, %2" : "=v"(lo) : "v"(a), "v"(c));
    asm("v_max_f64 %0, %1, %2" : "=v"(hi) : "v"(a), "v"(c));
    asm("v_max_f64 %0, %1, 0" : "=v"(lo) : "v"(lo));
    asm("v_min_f64 %0, %1, 0" : "=v"(hi) : "v"(hi));
    asm("v_min_f64 %0, %1, %2" : "=v"(t) : "v"(b), "v"(lo));
    asm("v_max_f64 %0, %1, %2" : "=v"(t) : "v"(t), "v"(hi));
    return t;
#else
    const double lo = __builtin_fmin(a, c), hi = __builtin_fmax(a, c);
    return __builtin_fmax(__builtin_fmin(b, __builtin_fmax(lo, 0.0)), __builtin_fmin(hi, 0.0));
#endif
}

__device__ inline double plm_gradient(double yl, double y0, double yr, double theta)
{
    const double a = (y0 - yl) * theta;
    const double b = (yr - yl) * 0.5;
    const double c = (yr - y0) * theta;
    return minmod_between(a, b, c);
}

__device__ inline State5 plm_gradient(const State5& l, const State5& c, const State5& r, double theta)
{
    State5 g;
#pragma unroll
    for (int q = 0; q < 5; ++q) g[q] = plm_gradient(l[q], c[q], r[q], theta);
    return g;
}

struct GammaLawFast
{
    double gamma;
    double inv_gm1;     // 1 / (gamma - 1), correctly rounded
    double gm1;
    double gfac;        // (gamma + 1) / (2 gamma)
};

__device__ inline GammaLawFast make_gamma_law(double gamma)
{
    GammaLawFast g;
    g.gamma = gamma;
    g.gm1 = gamma - 1.0;
    g.inv_gm1 = 1.0 / (gamma - 1.0);
    g.gfac = (gamma + 1.0) / (2.0 * gamma);
    return g;
}

// PLANAR (here and below): the third velocity / momentum is identically zero - a 2-D run of the five-component state, whose out-of-plane
// momentum the reference carries as zeros (euler2d_fused.hip: chosen where the uploaded field has none). Every term with that component is
// left out and the component is returned as 0: `x + 0` and `fma(0, 0, x)` are x exactly, so the other four components keep their bits.
template<bool PLANAR = false>
__device__ inline State5 recover_primitive(const State5& U, const GammaLawFast& g, double temperature_floor)
{
    const double rd = rcp_nr(U[0]);
    const double psq2 = __builtin_fma(U[2], U[2], U[1] * U[1]);
    const double psq = PLANAR ? psq2 : __builtin_fma(U[3], U[3], psq2);
    State5 P;
    P[0] = U[0];
    P[1] = U[1] * rd;
    P[2] = U[2] * rd;
    P[3] = PLANAR ? 0.0 : U[3] * rd;
    P[4] = __builtin_fma(-0.5 * psq, rd, U[4]) * g.gm1;
    if (P[4] < 0.0 && temperature_floor > 0.0) P[4] = temperature_floor * U[0];
    return P;
}

// conserved state, normal velocity, flux and sound speed of one face state
template<int AXIS, bool PLANAR = false>
__device__ inline void face_quantities(const State5& P, const GammaLawFast& g, State5& U, State5& F, double& vn, double& cs)
{
    static_assert(! PLANAR || AXIS != 2, "a planar state has no third axis");
    const double d = P[0], p = P[4];
    const double vsq2 = __builtin_fma(P[2], P[2], P[1] * P[1]);
    const double vsq = PLANAR ? vsq2 : __builtin_fma(P[3], P[3], vsq2);
    U[0] = d;
    U[1] = d * P[1];
    U[2] = d * P[2];
    U[3] = PLANAR ? 0.0 : d * P[3];
    U[4] = __builtin_fma(0.5 * d, vsq, p * g.inv_gm1);
    vn = P[1 + AXIS];
    F[0] = vn * U[0];
    F[1] = AXIS == 0 ? __builtin_fma(vn, U[1], p) : vn * U[1];
    F[2] = AXIS == 1 ? __builtin_fma(vn, U[2], p) : vn * U[2];
    F[3] = PLANAR ? 0.0 : (AXIS == 2 ? __builtin_fma(vn, U[3], p) : vn * U[3]);
    F[4] = vn * (U[4] + p);
    // cs = sqrt(gamma p / d) = gamma p / sqrt(gamma p d); p == 0 gives 0 * NaN, which fmax turns into 0
    const double gp = g.gamma * p;
    cs = __builtin_fmax(gp * rsqrt_fast(gp * d), 0.0);
}

template<int AXIS, bool PLANAR = false> __device__ inline State5 riemann_hlle(const State5& Pl, const State5& Pr, const GammaLawFast& g)
{
    State5 Ul, Ur, Fl, Fr;
    double vl, vr, csl, csr;
    face_quantities<AXIS, PLANAR>(Pl, g, Ul, Fl, vl, csl);
    face_quantities<AXIS, PLANAR>(Pr, g, Ur, Fr, vr, csr);
    const double ap = __builtin_fmax(0.0, __builtin_fmax(vl + csl, vr + csr));
    const double am = __builtin_fmin(0.0, __builtin_fmin(vl - csl, vr - csr));
    // (Fl ap - Fr am - (Ul - Ur) ap am) / (ap - am) with the three weights divided once: four instructions per component
    const double rden = rcp_nr(ap - am);
    const double wl = ap * rden, wr = am * rden, wu = wl * am;
    State5 F;
#pragma unroll
    for (int q = 0; q < 5; ++q)
    {
        if (PLANAR && q == 3) { F[q] = 0.0; continue; }
        F[q] = __builtin_fma(Ur[q] - Ul[q], wu, __builtin_fma(-Fr[q], wr, Fl[q] * wl));
    }
    return F;
}

// sound speed alone: sqrt(gamma p / d) = gamma p / sqrt(gamma p d)
__device__ inline double sound_speed(const State5& P, const GammaLawFast& g)
{
    const double gp = g.gamma * P[4];
    return __builtin_fmax(gp * rsqrt_fast(gp * P[0]), 0.0);      // p == 0: 0 * NaN -> 0
}

// conserved state and flux of one face state (what face_quantities computes besides the sound speed)
template<int AXIS, bool PLANAR = false>
__device__ inline void conserved_and_flux(const State5& P, const GammaLawFast& g, State5& U, State5& F)
{
    const double d = P[0], p = P[4], vn = P[1 + AXIS];
    const double vsq2 = __builtin_fma(P[2], P[2], P[1] * P[1]);
    const double vsq = PLANAR ? vsq2 : __builtin_fma(P[3], P[3], vsq2);
    U[0] = d;
    U[1] = d * P[1];
    U[2] = d * P[2];
    U[3] = PLANAR ? 0.0 : d * P[3];
    U[4] = __builtin_fma(0.5 * d, vsq, p * g.inv_gm1);
    F[0] = vn * U[0];
    F[1] = AXIS == 0 ? __builtin_fma(vn, U[1], p) : vn * U[1];
    F[2] = AXIS == 1 ? __builtin_fma(vn, U[2], p) : vn * U[2];
    F[3] = PLANAR ? 0.0 : (AXIS == 2 ? __builtin_fma(vn, U[3], p) : vn * U[3]);
    F[4] = vn * (U[4] + p);
}

// primitive_t::to_conserved_density (physics_euler.hpp:209-220)
__device__ inline State5 to_conserved(const State5& P, const GammaLawFast& g)
{
    const double vsq = __builtin_fma(P[3], P[3], __builtin_fma(P[2], P[2], P[1] * P[1]));
    State5 U;
    U[0] = P[0];
    U[1] = P[0] * P[1];
    U[2] = P[0] * P[2];
    U[3] = P[0] * P[3];
    U[4] = __builtin_fma(0.5 * P[0], vsq, P[4] * g.inv_gm1);
    return U;
}

// HLLC (Toro 3rd ed. section 10.4-10.6, pressure-based wave speeds as physics_iso2d.hpp:610-687 generalised to a gamma law). The wave
// speeds need only (d, u_n, p, a) of the two sides; the conserved state and flux are then formed for the ONE side the
// sampled region belongs to.
template<int AXIS, bool PLANAR = false> __device__ inline State5 riemann_hllc(const State5& Pl, const State5& Pr, const GammaLawFast& g)
{
    const double ul = Pl[1 + AXIS], ur = Pr[1 + AXIS];
    const double al = sound_speed(Pl, g), ar = sound_speed(Pr, g);
    const double dl = Pl[0], dr = Pr[0], pl = Pl[4], pr = Pr[4];
    const double dbar = 0.5 * (dl + dr);
    const double abar = 0.5 * (al + ar);
    const double ppvrs = __builtin_fma(-0.5 * (ur - ul), dbar * abar, 0.5 * (pl + pr));
    const double pstar = __builtin_fmax(0.0, ppvrs);
    // q_K = sqrt(1 + gfac (p* / p_K - 1)) = sqrt(x / p_K) = x rsqrt(x p_K), x = p_K + gfac (p* - p_K): one inverse root and no reciprocal
    // (the reciprocal + square root form cost 2.5 % of the smooth-wave step, where one side of nearly every face takes this branch)
    const double xl = __builtin_fma(g.gfac, pstar - pl, pl), xr = __builtin_fma(g.gfac, pstar - pr, pr);
    // x <= p_K exactly when p* <= p_K, where the root is <= 1: fmax picks the 1 (and turns the NaN of p_K == 0 into it)
    // (measured and not taken, profiles/r03/ab_fused_hllc_qskip.jsonl: a wave-wide vote that skips both inverse roots where no face of the
    // wave has p* > p_K - most of the blast's quiescent gas - costs more than it saves: 0.636 against 0.630 ms per fused 4096^2 step on the
    // blast, 0.667 against 0.651 on the smooth wave)
    const double ql = __builtin_fmax(xl * rsqrt_fast(xl * pl), 1.0);
    const double qr = __builtin_fmax(xr * rsqrt_fast(xr * pr), 1.0);
    const double sl = __builtin_fma(-al, ql, ul);
    const double sr = __builtin_fma(ar, qr, ur);
    const double ml = dl * (sl - ul);       // mass flux relative to the left wave
    const double mr = dr * (sr - ur);
    const double sstar = (pr - pl + ul * ml - ur * mr) * rcp_nr(ml - mr);

    // Region selection without control flow (four-way branches cost more in instruction-fetch stalls than the selects do in issue
    // slots): K = the side of the contact the face lies on; its own flux if the K wave moves away from the face, else the star flux
    //   F*_K = (S* (S_K U_K - F_K) + S_K p* D) / (S_K - S*), D = (0, n, S*), p* = p_K + rho_K (S_K - u_K)(S* - u_K)   (Toro eq. 10.41-10.43),
    // algebraically F_K + S_K (U*_K - U_K) of eq. 10.38-10.39 with one division instead of three. Conditions in the reference's
    // order (physics_iso2d.hpp:576-583): 0 <= S_L, S_L <= 0 <= S*, S* <= 0 <= S_R, S_R <= 0.
    // (0 <= S_L comes first upstream: where the pressure-based estimates cross, S_R < 0 < S_L - strongly colliding flows at gamma near 1 -
    // the face takes the left flux whatever the sign of S*; tests/test_gpu_toro.py, the isothermal-limit test, found this case)
    const bool left = (0.0 <= sstar) | (0.0 <= sl);
    State5 Pk;
#pragma unroll
    for (int q = 0; q < 5; ++q) Pk[q] = (PLANAR && q == 3) ? 0.0 : (left ? Pl[q] : Pr[q]);
    const double sk = left ? sl : sr, mk = left ? ml : mr;
    const bool star = left ? ! (0.0 <= sl) : (0.0 <= sr);
    State5 U, Fk, F;
    conserved_and_flux<AXIS, PLANAR>(Pk, g, U, Fk);
    const double rinv = rcp_nr(sk - sstar);
    // F = F_K + c (S* U_K - F_K + p* D), c = S_K / (S_K - S*) in the star region and 0 outside it: the same F*_K (subtract F_K from
    // eq. 10.41 over the common denominator) as a blend, two instructions per component and no selects on the five results
    const double pk_star = __builtin_fma(mk, sstar - Pk[1 + AXIS], Pk[4]);
    const double c = star ? sk * rinv : 0.0;
#pragma unroll
    for (int q = 0; q < 5; ++q) F[q] = (PLANAR && q == 3) ? 0.0 : __builtin_fma(sstar, U[q], -Fk[q]);
    F[1 + AXIS] += pk_star;
    F[4] = __builtin_fma(pk_star, sstar, F[4]);
#pragma unroll
    for (int q = 0; q < 5; ++q) F[q] = (PLANAR && q == 3) ? 0.0 : __builtin_fma(c, F[q], Fk[q]);
    return F;
}

} // namespace fast

// ---- arithmetic policies used by the stage kernels -------------------------
// PLANAR: see euler_device.hpp (recover_primitive) - the third momentum is +0.0 in every cell, bit for bit
template<bool PLANAR>
struct StrictArithT
{
    static constexpr bool planar = PLANAR;
    // planar: 170 / 180 registers - a third wave per SIMD is within reach, at 8 - 39 spilled registers. Measured at 4096^2, ms per RK2 step
    // (profiles/r04/ab_strict_planar.jsonl): HLLE general 0.945, planar at two waves 0.827, at three 0.806; HLLC 0.968 / 0.886 / 0.838
#ifndef MH_STRICT_PLANAR_WAVES
#define MH_STRICT_PLANAR_WAVES 3
#endif
    static constexpr int min_waves_per_simd = PLANAR ? MH_STRICT_PLANAR_WAVES : 2;
    static constexpr int min_waves_first_stage = PLANAR ? MH_STRICT_PLANAR_WAVES : 2;
    static constexpr bool deferred_axis1 = false;           // (euler3d_kernel.hpp) the update keeps the reference's order of terms
    // (euler3d_kernel.hpp) rows of a 3-D tile = waves of a workgroup. Four: the tile's LDS (70 KB) lets TWO workgroups share a CU, each with its own
    // barrier per plane, so one runs while the other waits for its slowest wave, an LDS read or a load - measured at 512^3 (profiles/r05/
    // ab_3d_four_row_tiles.json): STRICT 8.82 -> 9.35 Gzones/s. (FAST loses 6 % on four rows - every wave then fetches an outside row, five
    // axis-1 faces per four rows instead of nine per eight, and the kernel is within 6 % of the board's power limit already - and keeps eight.)
#ifndef MH_E3D_STRICT_ROWS
#define MH_E3D_STRICT_ROWS 4
#endif
    static constexpr int tile_rows = MH_E3D_STRICT_ROWS;
    static constexpr bool shared_differences = false;       // the reference's plm_gradient takes the three values, bit for bit
    static constexpr bool recompute_conserved = false;      // the update starts from the stored conserved state, bit for bit
    static constexpr bool lds_conserved_ring = false;
    using Gamma = GammaLaw;
    static __device__ inline Gamma gamma_law(double gamma) { return make_gamma_law(gamma); }
    static __device__ inline State5 c2p(const State5& U, const Gamma& g) { return recover_primitive<PLANAR>(U, g.gamma, 0.0); }
    struct Limiter { double theta; };
    static __device__ inline Limiter limiter(double theta) { return Limiter{theta}; }
    static __device__ inline State5 plm(const State5& l, const State5& c, const State5& r, const Limiter& lim) { return plm_gradient<PLANAR>(l, c, r, lim.theta); }
    static __device__ inline State5 plus(const State5& P, const State5& G, const Limiter&) { return face_plus<PLANAR>(P, G); }
    static __device__ inline State5 minus(const State5& P, const State5& G, const Limiter&) { return face_minus<PLANAR>(P, G); }
    template<int RIEMANN, int AXIS>
    static __device__ inline State5 flux(const State5& Pl, const State5& Pr, const Gamma& g) { return riemann<RIEMANN, AXIS, PLANAR>(Pl, Pr, g); }
    // u - ((Fxhi - Fxlo)*cx + (Fyhi - Fylo)*cy)
    static __device__ inline double update2(double u, double fxl, double fxh, double fyl, double fyh, double cx, double cy)
    {
        const double lx = (fxh - fxl) * cx;
        const double ly = (fyh - fyl) * cy;
        return u - (lx + ly);
    }
    static __device__ inline double update3(double u, double fxl, double fxh, double fyl, double fyh, double fzl, double fzh, double cx, double cy, double cz)
    {
        const double lx = (fxh - fxl) * cx;
        const double ly = (fyh - fyl) * cy;
        const double lz = (fzh - fzl) * cz;
        return u - (lx + ly + lz);
    }
    // src/subprog_cloud.cpp:693: s0*0.5 + s2*0.5
    static __device__ inline double combine(double base, double u1, double w) { return base * (1.0 - w) + u1 * w; }
};
using StrictArith = StrictArithT<false>;
using StrictArithPlanar = StrictArithT<true>;

// PLANAR: see fast::recover_primitive - the policy of the kernels that advance a field whose third momentum is identically zero
template<bool PLANAR>
struct FastArithT
{
    static constexpr bool planar = PLANAR;
    static constexpr bool live(int q) { return ! (PLANAR && q == 3); }
#ifndef MH_FAST_MIN_WAVES
#define MH_FAST_MIN_WAVES 2
#endif
    static constexpr int min_waves_per_simd = MH_FAST_MIN_WAVES;
#ifndef MH_FAST_MIN_WAVES_STAGE1
#define MH_FAST_MIN_WAVES_STAGE1 2          // 3 (168 VGPRs, ~40 spills) was measured: no faster, DESIGN.md §6
#endif
    static constexpr int min_waves_first_stage = MH_FAST_MIN_WAVES_STAGE1;
    // The limiter's one-sided differences y_{i+1} - y_i belong to a FACE: each is formed once and used by the two cells it separates
    // (carried in the register ring along the marching axis, passed by DPP across lanes). The limiter works on them UNSCALED:
    //     minmod(theta dl, (dl + dr) / 2, theta dr) = theta minmod(dl, (dl + dr) / (2 theta), dr),
    // and the factor theta / 2 of the half-cell extrapolation rides in the FMA that forms the face state. 1 + 2 + 6 fp64 instructions
    // per variable and axis (the difference, the central term, the limiter) where the three-value form takes 13.
    static constexpr bool shared_differences = true;
    // euler3d_kernel.hpp: one Riemann problem per axis-1 face, handed to the neighbouring wave through LDS, the update of a plane completed one
    // plane step later (MH_E3D_DEFER=0 builds the redundant-flux form: two solves per interior face, update in one piece)
#ifndef MH_E3D_DEFER
#define MH_E3D_DEFER 1
#endif
    static constexpr bool deferred_axis1 = MH_E3D_DEFER != 0;
    static constexpr int tile_rows = 8;                      // (the deferred axis-1 faces are written for eight rows: one workgroup per CU)
    // The 2-D stage kernel converts a loaded row to primitives once and keeps only those in its register window (no register-to-register
    // moves in the row loop, 20 VGPRs fewer); the conserved values the update starts from wait in a per-wave LDS ring meanwhile
    // (lds_conserved_ring; euler2d.hip). The first version re-formed them from the primitives instead (p2c, 8 instructions, equal to the
    // stored ones to rounding): same speed, but an ulp of energy per cell and step - 2e-13 of the total after 8000 steps where the stored
    // state conserves to 3e-16.
    static constexpr bool recompute_conserved = true;
    static constexpr bool lds_conserved_ring = true;
    static __device__ inline State5 p2c(const State5& P, const fast::GammaLawFast& g) { return fast::to_conserved(P, g); }
    struct Limiter
    {
        double central;      // 1 / (2 theta): weight of dl + dr in the limiter's central argument
        double half_theta;   // theta / 2: (slope in limiter units) -> (half-cell extrapolation)
    };
    static __device__ inline Limiter limiter(double theta) { return Limiter{0.5 / theta, 0.5 * theta}; }
    static __device__ inline State5 difference(const State5& P, const State5& Pnext)
    {
        State5 D;
#pragma unroll
        for (int q = 0; q < 5; ++q) D[q] = live(q) ? Pnext[q] - P[q] : 0.0;
        return D;
    }
    // the limited slope in units of theta (see shared_differences): plus / minus below scale it
    static __device__ inline State5 plm_from_differences(const State5& Dl, const State5& Dr, const Limiter& lim)
    {
        State5 G;
#pragma unroll
        for (int q = 0; q < 5; ++q) G[q] = live(q) ? fast::minmod_between(Dl[q], (Dl[q] + Dr[q]) * lim.central, Dr[q]) : 0.0;
        return G;
    }
    using Gamma = fast::GammaLawFast;
    static __device__ inline Gamma gamma_law(double gamma) { return fast::make_gamma_law(gamma); }
    static __device__ inline State5 c2p(const State5& U, const Gamma& g) { return fast::recover_primitive<PLANAR>(U, g, 0.0); }
    static __device__ inline State5 plm(const State5& l, const State5& c, const State5& r, const Limiter& lim)
    {
        return plm_from_differences(difference(l, c), difference(c, r), lim);
    }
    static __device__ inline State5 plus(const State5& P, const State5& G, const Limiter& lim)
    {
        State5 S;
#pragma unroll
        for (int q = 0; q < 5; ++q) S[q] = live(q) ? __builtin_fma(G[q], lim.half_theta, P[q]) : 0.0;
        return S;
    }
    static __device__ inline State5 minus(const State5& P, const State5& G, const Limiter& lim)
    {
        State5 S;
#pragma unroll
        for (int q = 0; q < 5; ++q) S[q] = live(q) ? __builtin_fma(G[q], -lim.half_theta, P[q]) : 0.0;
        return S;
    }
    template<int RIEMANN, int AXIS>
    static __device__ inline State5 flux(const State5& Pl, const State5& Pr, const Gamma& g)
    {
        if constexpr (RIEMANN == 1) return fast::riemann_hllc<AXIS, PLANAR>(Pl, Pr, g);
        else                        return fast::riemann_hlle<AXIS, PLANAR>(Pl, Pr, g);
    }
    static __device__ inline double update2(double u, double fxl, double fxh, double fyl, double fyh, double cx, double cy)
    {
        return __builtin_fma(-(fyh - fyl), cy, __builtin_fma(-(fxh - fxl), cx, u));
    }
    static __device__ inline double update3(double u, double fxl, double fxh, double fyl, double fyh, double fzl, double fzh, double cx, double cy, double cz)
    {
        return __builtin_fma(-(fzh - fzl), cz, __builtin_fma(-(fyh - fyl), cy, __builtin_fma(-(fxh - fxl), cx, u)));
    }
    static __device__ inline double combine(double base, double u1, double w) { return __builtin_fma(u1, w, base * (1.0 - w)); }
};
using FastArith = FastArithT<false>;
using FastArithPlanar = FastArithT<true>;

} // namespace mh
