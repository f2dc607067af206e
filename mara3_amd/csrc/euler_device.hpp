// Device-side gamma-law Euler physics and PLM reconstruction for gfx950.
//
// Operation order follows the reference expressions exactly (SURVEY.md §8a,
// Appendix B) so that, compiled with -ffp-contract=off, every result is
// bit-identical to the reference built for baseline x86-64:
//   plm_gradient                      src/math_interpolation.hpp:85-94
//   euler::recover_primitive          src/physics_euler.hpp:555-575
//   primitive_t::to_conserved_density src/physics_euler.hpp:209-220
//   primitive_t::flux                 src/physics_euler.hpp:252-263
//   primitive_t::wavespeeds           src/physics_euler.hpp:276-284
//   euler::riemann_hlle               src/physics_euler.hpp:614-631
//   unit_vector_t::project            src/core_geometric.hpp:85-89
// The HLLC solver has no upstream Euler counterpart; it is the gamma-law form
// of src/physics_iso2d.hpp:556-583,610-687 (Toro 3rd ed. §10.6).
//
// Two instruction-count optimisations keep every finite result bit-identical:
//  * shared-denominator division (struct Recip): x/d for several x and one d runs
//    the exact v_div_scale / v_rcp / Newton / v_div_fmas / v_div_fixup sequence
//    hipcc emits for `x / d`, but the reciprocal refinement (rcp + 4 fma) is done
//    once per denominator. If a numerator would make v_div_scale rescale the
//    denominator differently (extreme exponents), that lane recomputes it.
//  * plm_gradient takes the minimum with v_min_f64 on |.| operands. For finite
//    arguments this equals std::min; only the NaN pattern differs (std::min is
//    not commutative for NaN), and a NaN state is reported through the status
//    word in any case.
#pragma once
#include <hip/hip_runtime.h>

namespace mh {

// ---- IEEE fp64 division with a shared denominator -------------------------
struct Recip
{
    double den;   // the denominator as given
    double ds;    // v_div_scale'd denominator the reciprocal was refined for
    double r;     // refined reciprocal of ds
};

__device__ inline double refine_rcp(double ds)
{
    double r = __builtin_amdgcn_rcp(ds);
    double e = __builtin_fma(-ds, r, 1.0);
    r = __builtin_fma(r, e, r);
    e = __builtin_fma(-ds, r, 1.0);
    r = __builtin_fma(r, e, r);
    return r;
}

// num0: a representative numerator (only its exponent class matters; 1.0 covers every normal-range numerator)
__device__ inline Recip make_recip(double den, double num0)
{
    bool flag;
    Recip R;
    R.den = den;
    R.ds = __builtin_amdgcn_div_scale(num0, den, false, &flag);
    R.r = refine_rcp(R.ds);
    return R;
}

// N quotients x[k] / R.den in place. One cold fallback per group instead of one per quotient.
template<int N>
__device__ inline void divide_group(double (&x)[N], const Recip& R)
{
    double ds[N], ns[N];
    bool flag[N];
    bool same = true;
#pragma unroll
    for (int k = 0; k < N; ++k)
    {
        bool unused;
        ds[k] = __builtin_amdgcn_div_scale(x[k], R.den, false, &unused);
        ns[k] = __builtin_amdgcn_div_scale(x[k], R.den, true, &flag[k]);
        // A ZERO numerator makes v_div_scale return NaN for both of its forms (there is nothing to scale), which is not the refined
        // denominator - but it needs none: v_div_fixup answers 0 / den from the operands' signs alone, whatever quotient it is handed. Without
        // this case every group with an identically zero component took the fallback: the third momentum of a 2-D run, the momenta of
        // gas at rest (round 3: -14 % of the STRICT 2-D kernels' executed instructions, profiles/r03/strict_zero_numerators.md).
        same = same && ((__double_as_longlong(ds[k]) == __double_as_longlong(R.ds)) | (x[k] == 0.0));
    }
#ifdef MH_PROBE_NO_DIV_FALLBACK      // instruction-count probes only (scripts/strict_isa_table.py): the hot path without the branch around it
    same = true;
#endif
    if (__builtin_expect(same, 1))
    {
#pragma unroll
        for (int k = 0; k < N; ++k)
        {
            const double q = ns[k] * R.r;
            const double rem = __builtin_fma(-R.ds, q, ns[k]);
            const double res = __builtin_amdgcn_div_fmas(rem, R.r, q, flag[k]);
            x[k] = __builtin_amdgcn_div_fixup(res, R.den, x[k]);
        }
    }
    else
    {
#pragma unroll
        for (int k = 0; k < N; ++k) x[k] = x[k] / R.den;
    }
}

__device__ inline double divide(double num, const Recip& R)
{
    double x[1] = {num};
    divide_group<1>(x, R);
    return x[0];
}

struct State5
{
    double v[5];
    __device__ double& operator[](int i) { return v[i]; }
    __device__ const double& operator[](int i) const { return v[i]; }
};

// std::max / std::min semantics (the reference uses those, not fmax/fmin)
__device__ inline double std_max(double a, double b) { return (a < b) ? b : a; }
__device__ inline double std_min(double a, double b) { return (b < a) ? b : a; }

__device__ inline double plm_gradient(double yl, double y0, double yr, double theta)
{
    const double a = (y0 - yl) * theta;
    const double b = (yr - yl) * 0.5;
    const double c = (yr - y0) * theta;
    const double sa = copysign(1.0, a);
    const double sb = copysign(1.0, b);
    const double sc = copysign(1.0, c);
    const double m = __builtin_fmin(__builtin_fmin(__builtin_fabs(a), __builtin_fabs(b)), __builtin_fabs(c));
    return 0.25 * fabs(sa + sb) * (sa + sc) * m;
}

// PLANAR (here and below): the third momentum of every cell is +0.0, BIT FOR BIT (a 2-D run of the five-component state; the steppers check
// the bit pattern at upload, mh_euler_cart_desc.planar). With that input the reference's own operations return +0.0 for every quantity of the
// third direction - 0 / d, plm_gradient(0, 0, 0), P + G * 0.5, vn * (+0) + p * 0.0, the HLLE and HLLC combinations, u - (0 * cx + 0 * cy),
// 0 * 0.5 + 0 * 0.5: a sum of zeros of either sign with a +0 is +0 - so those operations are left out and +0.0 is written, and the other
// four components see exactly the operands they see upstream: a sum of squares is never -0, so `x + (+0)` may go; the ONE place where the
// zero's sign could act on another component, velocity_along's `... + P[3] * 0.0`, keeps its addition of +0.0. Bit-identical to the
// reference, component 3 included (the hashes of tests/test_gpu_long_runs_vs_reference.py are taken on this kernel).
template<bool PLANAR = false>
__device__ inline State5 plm_gradient(const State5& l, const State5& c, const State5& r, double theta)
{
    State5 g;
#pragma unroll
    for (int q = 0; q < 5; ++q) g[q] = (PLANAR && q == 3) ? 0.0 : plm_gradient(l[q], c[q], r[q], theta);
    return g;
}

template<bool PLANAR = false>
__device__ inline State5 recover_primitive(const State5& U, double gamma, double temperature_floor)
{
    const double d = U[0];
    const Recip Rd = make_recip(d, 1.0);
    State5 P;
    P[0] = d;
    if constexpr (PLANAR)
    {
        double x[3] = {U[1], U[2], 0.5 * (U[1] * U[1] + U[2] * U[2])};
        divide_group<3>(x, Rd);
        P[1] = x[0];
        P[2] = x[1];
        P[3] = 0.0;
        P[4] = (U[4] - x[2]) * (gamma - 1.0);
    }
    else
    {
        const double p_squared = U[1] * U[1] + U[2] * U[2] + U[3] * U[3];
        double x[4] = {U[1], U[2], U[3], 0.5 * p_squared};
        divide_group<4>(x, Rd);
        P[1] = x[0];
        P[2] = x[1];
        P[3] = x[2];
        P[4] = (U[4] - x[3]) * (gamma - 1.0);
    }
    if (P[4] < 0.0 && temperature_floor > 0.0) P[4] = temperature_floor * d;
    return P;
}

// gamma-law constants prepared once per kernel: gm1 = make_recip(gamma - 1, 1.0)
struct GammaLaw
{
    double gamma;
    Recip  gm1;
};
__device__ inline GammaLaw make_gamma_law(double gamma)
{
    GammaLaw g;
    g.gamma = gamma;
    g.gm1 = make_recip(gamma - 1, 1.0);
    return g;
}

template<bool PLANAR = false>
__device__ inline State5 to_conserved_with(const State5& P, double p_over_gm1)
{
    const double d = P[0];
    const double vsq = PLANAR ? P[1] * P[1] + P[2] * P[2] : P[1] * P[1] + P[2] * P[2] + P[3] * P[3];
    State5 U;
    U[0] = d;
    U[1] = d * P[1];
    U[2] = d * P[2];
    U[3] = PLANAR ? 0.0 : d * P[3];
    U[4] = 0.5 * d * vsq + p_over_gm1;
    return U;
}

template<bool PLANAR = false>
__device__ inline State5 to_conserved_density(const State5& P, const GammaLaw& g)
{
    return to_conserved_with<PLANAR>(P, divide(P[4], g.gm1));
}

// both sides of a face share one division group for p / (gamma - 1)
template<bool PLANAR = false>
__device__ inline void to_conserved_pair(const State5& Pl, const State5& Pr, const GammaLaw& g, State5& Ul, State5& Ur)
{
    double x[2] = {Pl[4], Pr[4]};
    divide_group<2>(x, g.gm1);
    Ul = to_conserved_with<PLANAR>(Pl, x[0]);
    Ur = to_conserved_with<PLANAR>(Pr, x[1]);
}

// nhat = on_axis(AXIS): components are the literal 1.0 / 0.0 of the reference;
// the products with 0.0 are kept (they are not no-ops for -0.0, inf and NaN).
template<int AXIS> struct Normal
{
    static constexpr double n1 = AXIS == 0 ? 1.0 : 0.0;
    static constexpr double n2 = AXIS == 1 ? 1.0 : 0.0;
    static constexpr double n3 = AXIS == 2 ? 1.0 : 0.0;
};

template<int AXIS, bool PLANAR = false> __device__ inline double velocity_along(const State5& P)
{
    using N = Normal<AXIS>;
    static_assert(! PLANAR || AXIS != 2, "a planar state has no third axis");
    if constexpr (PLANAR) return P[1] * N::n1 + P[2] * N::n2 + 0.0;          // (+0) * 0.0 = +0.0: the addition stays, it can turn a -0 sum into +0
    else                  return P[1] * N::n1 + P[2] * N::n2 + P[3] * N::n3;
}

template<int AXIS, bool PLANAR = false> __device__ inline State5 flux(const State5& P, const State5& U, double vn)
{
    using N = Normal<AXIS>;
    const double p = P[4];
    State5 F;
    F[0] = vn * U[0];
    F[1] = vn * U[1] + p * N::n1;
    F[2] = vn * U[2] + p * N::n2;
    F[3] = PLANAR ? 0.0 : vn * U[3] + p * N::n3;           // vn * (+0) + p * 0.0: (+-0) + (+0) = +0
    F[4] = vn * U[4] + p * vn;
    return F;
}

template<int AXIS, bool PLANAR = false> __device__ inline State5 riemann_hlle(const State5& Pl, const State5& Pr, const GammaLaw& g)
{
    const double gamma = g.gamma;
    State5 Ul, Ur;
    to_conserved_pair<PLANAR>(Pl, Pr, g, Ul, Ur);
    const double csl = sqrt(gamma * Pl[4] / Pl[0]);
    const double vl = velocity_along<AXIS, PLANAR>(Pl);
    const double csr = sqrt(gamma * Pr[4] / Pr[0]);
    const double vr = velocity_along<AXIS, PLANAR>(Pr);
    const State5 Fl = flux<AXIS, PLANAR>(Pl, Ul, vl);
    const State5 Fr = flux<AXIS, PLANAR>(Pr, Ur, vr);
    const double ap = std_max(0.0, std_max(vl + csl, vr + csr));
    const double am = std_min(0.0, std_min(vl - csl, vr - csr));
    const Recip Rden = make_recip(ap - am, 1.0);
    if constexpr (PLANAR)
    {
        // (+0) ap - (+0) am - ((+0) - (+0)) ap am with ap >= 0 >= am is +0, and +0 / (ap - am) is +0: four quotients instead of five
        double x[4];
        constexpr int at[4] = {0, 1, 2, 4};
#pragma unroll
        for (int k = 0; k < 4; ++k) x[k] = Fl[at[k]] * ap - Fr[at[k]] * am - (Ul[at[k]] - Ur[at[k]]) * ap * am;
        divide_group<4>(x, Rden);
        State5 N;
        N[0] = x[0]; N[1] = x[1]; N[2] = x[2]; N[3] = 0.0; N[4] = x[3];
        return N;
    }
    else
    {
        State5 N;
#pragma unroll
        for (int q = 0; q < 5; ++q) N[q] = Fl[q] * ap - Fr[q] * am - (Ul[q] - Ur[q]) * ap * am;
        divide_group<5>(N.v, Rden);
        return N;
    }
}

template<int AXIS, bool PLANAR = false> __device__ inline State5 riemann_hllc(const State5& Pl, const State5& Pr, const GammaLaw& g)
{
    using N = Normal<AXIS>;
    const double nh[3] = {N::n1, N::n2, N::n3};
    const double gamma = g.gamma;
    // The wave speeds need the primitives only; the conserved state and the flux are formed for the side the sampled region belongs
    // to, inside its branch (same expressions, same bits as forming both sides up front - a division gives the same result alone
    // or in a group - and about 25 instructions fewer per face where a wave takes one branch).
    const double ul = velocity_along<AXIS, PLANAR>(Pl);
    const double ur = velocity_along<AXIS, PLANAR>(Pr);
    const double dl = Pl[0], dr = Pr[0], pl = Pl[4], pr = Pr[4];
    const double dbar = 0.5 * (dl + dr);
    const double al = sqrt(gamma * pl / dl);
    const double ar = sqrt(gamma * pr / dr);
    const double abar = 0.5 * (al + ar);
    const double ppvrs = 0.5 * (pl + pr) - 0.5 * (ur - ul) * dbar * abar;
    const double pstar = std_max(0.0, ppvrs);
    const double gfac = (gamma + 1.0) / (2.0 * gamma);
    const double ql = pstar <= pl ? 1.0 : sqrt(1.0 + gfac * (pstar / pl - 1.0));
    const double qr = pstar <= pr ? 1.0 : sqrt(1.0 + gfac * (pstar / pr - 1.0));
    const double sl = ul - al * ql;
    const double sr = ur + ar * qr;
    const double den = dl * (sl - ul) - dr * (sr - ur);
    const double sstar = (pr - pl + ul * dl * (sl - ul) - ur * dr * (sr - ur)) / den;

    State5 F;
    if (0.0 <= sl)
    {
        F = flux<AXIS, PLANAR>(Pl, to_conserved_density<PLANAR>(Pl, g), ul);
    }
    else if (sl <= 0.0 && 0.0 <= sstar)
    {
        const State5 Ul = to_conserved_density<PLANAR>(Pl, g);
        const State5 Fl = flux<AXIS, PLANAR>(Pl, Ul, ul);
        const double fac = dl * (sl - ul) / (sl - sstar);
        State5 Us;
        Us[0] = fac;
#pragma unroll
        for (int k = 0; k < 3; ++k) Us[1 + k] = (PLANAR && k == 2) ? 0.0 : fac * (sstar * nh[k] + (Pl[1 + k] - nh[k] * ul));
        Us[4] = fac * (Ul[4] / dl + (sstar - ul) * (sstar + pl / (dl * (sl - ul))));
#pragma unroll
        for (int q = 0; q < 5; ++q) F[q] = (PLANAR && q == 3) ? 0.0 : Fl[q] + (Us[q] - Ul[q]) * sl;
    }
    else if (sstar <= 0.0 && 0.0 <= sr)
    {
        const State5 Ur = to_conserved_density<PLANAR>(Pr, g);
        const State5 Fr = flux<AXIS, PLANAR>(Pr, Ur, ur);
        const double fac = dr * (sr - ur) / (sr - sstar);
        State5 Us;
        Us[0] = fac;
#pragma unroll
        for (int k = 0; k < 3; ++k) Us[1 + k] = (PLANAR && k == 2) ? 0.0 : fac * (sstar * nh[k] + (Pr[1 + k] - nh[k] * ur));
        Us[4] = fac * (Ur[4] / dr + (sstar - ur) * (sstar + pr / (dr * (sr - ur))));
#pragma unroll
        for (int q = 0; q < 5; ++q) F[q] = (PLANAR && q == 3) ? 0.0 : Fr[q] + (Us[q] - Ur[q]) * sr;
    }
    else if (sr <= 0.0)
    {
        F = flux<AXIS, PLANAR>(Pr, to_conserved_density<PLANAR>(Pr, g), ur);
    }
    else
    {
#pragma unroll
        for (int q = 0; q < 5; ++q) F[q] = __builtin_nan("");
    }
    return F;
}

template<int RIEMANN, int AXIS, bool PLANAR = false> __device__ inline State5 riemann(const State5& Pl, const State5& Pr, const GammaLaw& g)
{
    if constexpr (RIEMANN == 1) return riemann_hllc<AXIS, PLANAR>(Pl, Pr, g);
    else                        return riemann_hlle<AXIS, PLANAR>(Pl, Pr, g);
}

// face states: PL = P + G*0.5, PR = P - G*0.5 (src/subprog_cloud.cpp:566-568)
template<bool PLANAR = false>
__device__ inline State5 face_plus(const State5& P, const State5& G)
{
    State5 S;
#pragma unroll
    for (int q = 0; q < 5; ++q) S[q] = (PLANAR && q == 3) ? 0.0 : P[q] + G[q] * 0.5;
    return S;
}
template<bool PLANAR = false>
__device__ inline State5 face_minus(const State5& P, const State5& G)
{
    State5 S;
#pragma unroll
    for (int q = 0; q < 5; ++q) S[q] = (PLANAR && q == 3) ? 0.0 : P[q] - G[q] * 0.5;
    return S;
}

} // namespace mh
