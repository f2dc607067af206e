// Device code shared by the two kernel families of the `binary` path (binary.hip: uniform-depth trees as one periodic grid,
// binary_tree.hip: graded trees block by block): the scheme's constants, the STRICT / FAST arithmetic policies and the
// inter-cell flux. Reference lines are cited at each function.
#pragma once
#include <hip/hip_runtime.h>
#include "euler_device.hpp"
#include "euler_device_fast.hpp"
#include "iso2d_device.hpp"
#include "../../include/mara_hip.h"

namespace mh {

using iso2d::State3;

// A band of the uniform-depth mesh held by one device field: rows [row0, row0 + n0) of the n x n cells (whole rows of tree blocks);
// ext0: the ghost rows of the field belong to other bands (filled by the exchange) instead of being the periodic image of its own rows
struct BinaryBand { int n0, row0, ext0; };
// The rows of a band one call of binary_stage_launch covers. A band with neighbours runs its EDGE rows first (the first and the last
// `edge` rows, one small launch), so that their exchange travels beside the INTERIOR launch; the stage's sink sums and reduction
// follow the interior. The partial sums keep one fixed order: edge waves, then interior waves.
enum { BIN_ROWS_ALL = 0, BIN_ROWS_EDGES = 1, BIN_ROWS_INTERIOR = 2 };
struct BinaryRows { int part, edge; hipEvent_t edges_done; };          // edges_done (interior, or null): the reduction also waits for it

// binary_stage_launch with the totals (per-block sink sums + reduction) on a second stream: `input_ready` was recorded by the caller on the
// stage's stream before the launch (the stage's input field is complete), `stage_done` is recorded by the launcher behind the stage kernel
// sink_done (optional): recorded by the launcher behind the sink-sums kernel - the last reader of the stage's input on that stream
struct BinaryTotalsOverlap { hipStream_t stream; hipEvent_t input_ready, stage_done, sink_done; };

struct BinaryConsts
{
    double h;                 // grid spacing 2 R / block_size / 2^level of the grid or block being processed
    double h0;                // spacing at the root of the tree, 2 R / block_size (scheme.cpp:793)
    double mach, alpha, nu, rc_cut;
    double sink_rate, s2, rs2, floor_sigma;
    int    axisym;
    double rd;                // domain radius (the angular-momentum flux through x, y = +-rd is set to zero, scheme.cpp:208-209)
    double sr2;               // gst_suppr_radius^2: range of the ramp on advance_q's geometrical source term (:421, :440)
    double body[10];          // (mass, x, y, vx, vy) x 2
    double plm_theta, plm_central;   // theta and 1 / (2 theta) (host division): MH_ARITH_FAST limits the unscaled one-sided differences and scales once
};
// theta = 0 (the safe mode's piecewise-constant step): minmod(dl, 0, dr) * 0 - zero slopes without a division by zero
inline void binary_set_theta(BinaryConsts& c, double theta) { c.plm_theta = theta; c.plm_central = theta > 0.0 ? 0.5 / theta : 0.0; }


// ---- arithmetic policies ----------------------------------------------------------------------------------------------
// BinStrict: reference operation order, IEEE division / sqrt (shared-denominator form), no contraction.
// BinFast:   MH_ARITH_FAST as in euler_device_fast.hpp: reciprocal and inverse root = hardware estimate + one third-order step, one per denominator; square
//            roots (1 / sqrt directly where the reference divides by a root), FMAs, min/max limiter. Tolerance as STRICT's
//            (which already differs from the reference through libm): 1e-12 of the field scale, tests/test_gpu_binary.py.
struct BinStrict
{
    static __device__ inline double carried(double x) { return x; }          // (no contraction in this mode: nothing to settle)
    static constexpr int arith = MH_ARITH_STRICT;
    struct Ctx { Recip rmach, rh; };
    static __device__ inline Ctx make(const BinaryConsts& c) { return {make_recip(c.mach, 1.0), make_recip(c.h, 1.0)}; }

    // cs2_at_position :160-175 with grav_phi_field :101-111
    static __device__ inline double cs2(const BinaryConsts& c, const Ctx& k, double x, double y)
    {
        if (c.axisym)
        {
            const double a = 1.0 / sqrt(x * x + y * y);
            return divide(divide(a, k.rmach), k.rmach);
        }
        const double d0 = x - c.body[1], d1 = y - c.body[2];
        const double e0 = x - c.body[6], e1 = y - c.body[7];
        const double phi1 = (-1.0 * c.body[0]) / sqrt(d0 * d0 + d1 * d1 + c.rs2);
        const double phi2 = (-1.0 * c.body[5]) / sqrt(e0 * e0 + e1 * e1 + c.rs2);
        return divide(divide(-(phi1 + phi2), k.rmach), k.rmach);
    }
    // nu_at_position :177-193
    static __device__ inline double nu(const BinaryConsts& c, const Ctx& k, double x, double y, double cs2v)
    {
        const double radius = sqrt(x * x + y * y);
        const double profile = c.rc_cut > 0.0 ? 0.5 * (1.0 + tanh(3.0 * (radius - c.rc_cut))) : 1.0;
        if (c.nu > 0.0)
            return profile * c.nu;
        return profile * c.alpha * sqrt(cs2v) * divide(radius, k.rmach);
    }
    template<int AXIS> static __device__ inline State3 hlle(const State3& pl, const State3& pr, double cs2v) { return iso2d::riemann_hlle<AXIS>(pl, pr, cs2v, cs2v); }
    static __device__ inline State3 plm_per_length(const State3& l, const State3& m, const State3& r, double theta, const Ctx& k)
    {
        State3 g;
#pragma unroll
        for (int q = 0; q < 3; ++q) g[q] = plm_gradient(l[q], m[q], r[q], theta);
        divide_group<3>(g.v, k.rh);
        return g;
    }
    // iso2d::recover_primitive(U) physics_iso2d.hpp:351-362, or (Q, x) :376-390 at the centre of the cell the data belongs to
    template<bool QFORM> static __device__ inline State3 c2p(const State3& U, double xc, double yc)
    {
        State3 P;
        if constexpr (QFORM) iso2d::recover_primitive_angmom(U, xc, yc, P);
        else                 iso2d::recover_primitive(U, P);
        return P;
    }
    // grav_vdot_field :85-95 times sigma (:369-370)
    static __device__ inline void gravity(const BinaryConsts& c, int b, double d0, double d1, double sigma, double (&fg)[2])
    {
        const double r2s = d0 * d0 + d1 * d1 + c.rs2;
        const double den = r2s * sqrt(r2s);                     // pow<3, 2>
        double a[2] = {-d0, -d1};
        divide_group<2>(a, make_recip(den, 1.0));
        fg[0] = (a[0] * 1.0 * c.body[5 * b]) * sigma;
        fg[1] = (a[1] * 1.0 * c.body[5 * b]) * sigma;
    }
    static __device__ inline double sink_a2(const BinaryConsts& c, double d0, double d1) { return (d0 * d0 + d1 * d1) / c.s2 / 2.0; }
    static __device__ inline void over_area(double (&l)[3], double dA) { divide_group<3>(l, make_recip(dA, 1.0)); }
};

// Round 3 compiled binary_fast.hip with -ffp-contract=fast, where the compiler fuses a product into every sum that uses it - also across the
// inlined function that formed it - so that a value used in the iteration that forms it AND carried to the next would differ in the last bit
// between a chunk's prologue and the row loop. Values that are carried were therefore SETTLED where they are formed: an empty asm the
// optimiser cannot see through (no instruction). No file is compiled with contraction any more (Makefile); the marks stay as what they are -
// the places where a carried value is formed.
__device__ inline double settled(double x) { asm("" : "+v"(x)); return x; }

// DISK (round 5): the sub-program's default disk - two-body sound speed (axisymmetric_cs2 = 0), alpha viscosity (nu = 0) without the tanh cut-off
// (alpha_cutoff_radius = 0) - with those three run-time-uniform switches pinned at COMPILE time. The generic kernel keeps them as scalar
// branches inside the row loop: six per cell-row around ~380 instructions of device-libm tanh and the axisymmetric forms that a default run
// never executes (profiles/r05/binary_instruction_classes.md) - the loop body is then 845 VALU instructions per row where 447 run. Same
// formulas, same bits (the launcher picks the instantiation from the constants: binary_fast.hip).
template<bool DISK>
struct BinFastT
{
    static constexpr int arith = MH_ARITH_FAST;
    static __device__ inline double carried(double x) { return settled(x); }
    struct Ctx { double inv_mach, inv_mach2, inv_h, inv_2s2, half_h, inv_h2, slope_scale, central; };
    static __device__ inline Ctx make(const BinaryConsts& c)
    {
        const double im = fast::rcp_nr(c.mach), ih = fast::rcp_nr(c.h);
        return {im, im * im, ih, fast::rcp_nr(2.0 * c.s2), 0.5 * c.h, ih * ih, c.plm_theta * ih, c.plm_central};
    }
    static __device__ inline double rsqrt(double x) { return fast::rsqrt_fast(x); }
    static __device__ inline double cs2(const BinaryConsts& c, const Ctx& k, double x, double y)
    {
        if constexpr (! DISK) { if (c.axisym) return rsqrt(__builtin_fma(x, x, y * y)) * k.inv_mach2; }
        const double d0 = x - c.body[1], d1 = y - c.body[2];
        const double e0 = x - c.body[6], e1 = y - c.body[7];
        const double r1 = rsqrt(__builtin_fma(d0, d0, __builtin_fma(d1, d1, c.rs2)));
        const double r2 = rsqrt(__builtin_fma(e0, e0, __builtin_fma(e1, e1, c.rs2)));
        return __builtin_fma(c.body[0], r1, c.body[5] * r2) * k.inv_mach2;
    }
    // a root that only scales the viscosity: x rsqrt(x) without the correction step (an ulp or two); x == 0 would give 0 * inf, answered with 0
    static __device__ inline double root(double x) { return __builtin_fmax(x * fast::rsqrt_fast(x), 0.0); }
    static __device__ inline double nu(const BinaryConsts& c, const Ctx& k, double x, double y, double cs2v)
    {
        const double radius = root(__builtin_fma(x, x, y * y));
        if constexpr (DISK) return 1.0 * c.alpha * fast::sqrt_fast(cs2v) * (radius * k.inv_mach);      // (profile = 1.0: the product keeps its bits)
        const double profile = c.rc_cut > 0.0 ? 0.5 * (1.0 + tanh(3.0 * (radius - c.rc_cut))) : 1.0;
        if (c.nu > 0.0)
            return profile * c.nu;
        return profile * c.alpha * fast::sqrt_fast(cs2v) * (radius * k.inv_mach);      // the same root as hlle's sound speed: formed once
    }
    // the constants this instantiation is valid for
    static inline bool serves(const BinaryConsts& c) { return ! DISK || (c.axisym == 0 && !(c.rc_cut > 0.0) && !(c.nu > 0.0)); }
    // iso2d::riemann_hlle physics_iso2d.hpp:488-506 with one sound speed for both sides (the scheme passes cs2 twice, :288)
    template<int AXIS> static __device__ inline State3 hlle(const State3& pl, const State3& pr, double cs2v)
    {
        const double cs = fast::sqrt_fast(cs2v);
        const double vl = pl[1 + AXIS], vr = pr[1 + AXIS];
        const double ap = __builtin_fmax(0.0, __builtin_fmax(vl + cs, vr + cs));
        const double am = __builtin_fmin(0.0, __builtin_fmin(vl - cs, vr - cs));
        const double rden = fast::rcp_nr(ap - am), apam = ap * am;
        State3 F;
        const State3* side[2] = {&pl, &pr};
        double Ul[3], Ur[3], Fl[3], Fr[3];
#pragma unroll
        for (int s = 0; s < 2; ++s)
        {
            const State3& P = *side[s];
            double* U = s == 0 ? Ul : Ur;
            double* Fx = s == 0 ? Fl : Fr;
            const double v = P[1 + AXIS], p = P[0] * cs2v;
            U[0] = P[0]; U[1] = P[0] * P[1]; U[2] = P[0] * P[2];
            Fx[0] = v * U[0];
            Fx[1] = AXIS == 0 ? __builtin_fma(v, U[1], p) : v * U[1];
            Fx[2] = AXIS == 1 ? __builtin_fma(v, U[2], p) : v * U[2];
        }
#pragma unroll
        for (int q = 0; q < 3; ++q)
            F[q] = __builtin_fma(-(Ul[q] - Ur[q]), apam, __builtin_fma(-Fr[q], am, Fl[q] * ap)) * rden;
        return F;
    }
    static __device__ inline State3 plm_per_length(const State3& l, const State3& m, const State3& r, double theta, const Ctx& k)
    {
        State3 g;
#pragma unroll
        for (int q = 0; q < 3; ++q)
        {
            // minmod(theta dl, (dl + dr) / 2, theta dr) / h = (theta / h) minmod(dl, (dl + dr) / (2 theta), dr): the limiter on the unscaled
            // one-sided differences, one scaling (round 5: 11 instead of 13 instructions per variable and axis)
            const double dl = m[q] - l[q], dr = r[q] - m[q];
            g[q] = settled(fast::minmod_between(dl, (dl + dr) * k.central, dr) * k.slope_scale);
        }
        (void) theta;
        return g;
    }
    template<bool QFORM> static __device__ inline State3 c2p(const State3& U, double xc, double yc)
    {
        State3 P;
        const double rs = fast::rcp_nr(U[0]);
        P[0] = U[0];
        if constexpr (QFORM)
        {
            const double sr = U[1] * rs, lz = U[2] * rs;
            const double rr2 = fast::rcp_nr(__builtin_fma(xc, xc, yc * yc));
            P[1] = settled(__builtin_fma(sr, xc, -lz * yc) * rr2);
            P[2] = settled(__builtin_fma(sr, yc, lz * xc) * rr2);
        }
        else
        {
            P[1] = settled(U[1] * rs);
            P[2] = settled(U[2] * rs);
        }
        return P;
    }
    static __device__ inline void gravity(const BinaryConsts& c, int b, double d0, double d1, double sigma, double (&fg)[2])
    {
        const double rs = rsqrt(__builtin_fma(d0, d0, __builtin_fma(d1, d1, c.rs2)));
        const double w = -(rs * rs * rs) * c.body[5 * b] * sigma;
        fg[0] = d0 * w;
        fg[1] = d1 * w;
    }
    static __device__ inline double sink_a2(const BinaryConsts& c, double d0, double d1) { return __builtin_fma(d0, d0, d1 * d1) * fast::rcp_nr(2.0 * c.s2); }
    static __device__ inline void over_area(double (&l)[3], double dA) { const double r = fast::rcp_nr(dA); for (int q = 0; q < 3; ++q) l[q] *= r; }
};
using BinFast = BinFastT<false>;
using BinFastDisk = BinFastT<true>;

// strict helpers used by the small kernels (sink sums, maximum wavespeed)
__device__ inline double binary_cs2(const BinaryConsts& c, const Recip& rmach, double x, double y)
{
    BinStrict::Ctx k = {rmach, rmach};
    return BinStrict::cs2(c, k, x, y);
}

// intercell_flux_u :268-293 + viscous_flux :220-262 ; g = slopes along AXIS, t = transverse slopes (both per length)
template<class A, int AXIS, bool QFORM>
__device__ inline State3 binary_face_flux(const BinaryConsts& c, const typename A::Ctx& k, double xf, double yf,
    const State3& pl, const State3& pr, const State3& gl, const State3& gr, const State3& tl, const State3& tr)
{
    State3 pl_hat, pr_hat;
    if constexpr (A::arith == MH_ARITH_FAST)
    {
#pragma unroll
        for (int q = 0; q < 3; ++q)
        {
            pl_hat[q] = __builtin_fma(gl[q], k.half_h, pl[q]);
            pr_hat[q] = __builtin_fma(gr[q], -k.half_h, pr[q]);
        }
    }
    else
    {
#pragma unroll
        for (int q = 0; q < 3; ++q)
        {
            pl_hat[q] = pl[q] + gl[q] * 0.5 * c.h;
            pr_hat[q] = pr[q] - gr[q] * 0.5 * c.h;
        }
    }
    const double cs2 = A::cs2(c, k, xf, yf);
    const double nu = A::nu(c, k, xf, yf, cs2);
    State3 F = A::template hlle<AXIS>(pl_hat, pr_hat, cs2);
    if constexpr (A::arith == MH_ARITH_FAST)
    {
        // the same stress with the three factors 1/2 (of mu and of the two face averages) gathered: mu/2 = nu (sigma_l + sigma_r) / 4
        const double muh = 0.25 * nu * (pl_hat[0] + pr_hat[0]);
        const double g_ux = gl[1] + gr[1], g_uy = gl[2] + gr[2], t_ux = tl[1] + tr[1], t_uy = tl[2] + tr[2];      // along the face normal / transverse
        if constexpr (AXIS == 0)
        {
            F[1] = __builtin_fma(-muh, g_ux - t_uy, F[1]);      // tauxx = mu (dx_ux - dy_uy)
            F[2] = __builtin_fma(-muh, g_uy + t_ux, F[2]);      // tauxy = mu (dx_uy + dy_ux)
        }
        else
        {
            F[1] = __builtin_fma(-muh, t_uy + g_ux, F[1]);      // tauyx = mu (dx_uy + dy_ux)
            F[2] = __builtin_fma(muh, t_ux - g_uy, F[2]);       // tauyy = -mu (dx_ux - dy_uy)
        }
    }
    else if constexpr (AXIS == 0)
    {
        const double mu = 0.5 * nu * (pl_hat[0] + pr_hat[0]);
        const double dx_ux = 0.5 * (gl[1] + gr[1]);
        const double dx_uy = 0.5 * (gl[2] + gr[2]);
        const double dy_ux = 0.5 * (tl[1] + tr[1]);
        const double dy_uy = 0.5 * (tl[2] + tr[2]);
        const double tauxx = mu * (dx_ux - dy_uy);
        const double tauxy = mu * (dx_uy + dy_ux);
        F[0] = F[0] + 0.0;
        F[1] = F[1] + -tauxx;
        F[2] = F[2] + -tauxy;
    }
    else
    {
        const double mu = 0.5 * nu * (pl_hat[0] + pr_hat[0]);
        const double dx_ux = 0.5 * (tl[1] + tr[1]);
        const double dx_uy = 0.5 * (tl[2] + tr[2]);
        const double dy_ux = 0.5 * (gl[1] + gr[1]);
        const double dy_uy = 0.5 * (gl[2] + gr[2]);
        const double tauyx =  mu * (dx_uy + dy_ux);
        const double tauyy = -mu * (dx_ux - dy_uy);
        F[0] = F[0] + 0.0;
        F[1] = F[1] + -tauyx;
        F[2] = F[2] + -tauyy;
    }
    if constexpr (QFORM)
    {
        // to_angmom_fluxes scheme.cpp:199-214
        const double flux_sr = xf * F[1] + yf * F[2];
        double flux_lz = xf * F[2] - yf * F[1];
        if (AXIS == 0 && (xf == -c.rd || xf == c.rd)) flux_lz = 0.0;
        if (AXIS == 1 && (yf == -c.rd || yf == c.rd)) flux_lz = 0.0;
        F[1] = flux_sr;
        F[2] = flux_lz;
    }
    return F;
}

// the sink rate of one body at a cell, sink_rate_field :117-126. exp(-a2) == 0 exactly for a2 > 750 (glibc and ocml
// both underflow to zero below exp(-745.2)), so the call is skipped when no lane of the wave is in range.
template<class A>
__device__ inline double binary_sink_rate(const BinaryConsts& c, double d0, double d1)
{
    const double a2 = A::sink_a2(c, d0, d1);
    double e = 0.0;
    if (__any(a2 < 750.0)) e = exp(-a2);
    return c.sink_rate * e;
}


} // namespace mh
