// Device-side special-relativistic hydrodynamics (mara::srhd) for gfx950, strict
// arithmetic: operation order of the reference expressions, -ffp-contract=off,
// IEEE division and sqrt, so results are bit-identical to the reference built
// for baseline x86-64:
//   srhd::recover_primitive          src/physics_srhd.hpp:364-451  (Newton on p from p = 0, <= 50 iterations,
//                                                                   |f| < 1e-10 tested AFTER the update)
//   primitive_t::to_conserved_density src/physics_srhd.hpp:213-227
//   primitive_t::flux                 src/physics_srhd.hpp:259-270
//   primitive_t::wavespeeds           src/physics_srhd.hpp:283-295
//   srhd::riemann_hlle                src/physics_srhd.hpp:466-483
//   spherical_geometry_source_terms   src/physics_srhd.hpp:309-326  (cot theta is passed in: tan() is evaluated on the host)
// Where the reference throws (c2p) the device returns status bits (include/mara_hip.h, enum mh_status).
#pragma once
#include <hip/hip_runtime.h>
#include "euler_device.hpp"
#include "../../include/mara_hip.h"

namespace mh {
namespace srhd {

struct Gamma
{
    double gamma;
    double gm1;        // gamma - 1.0
    double hfac;       // 1.0 + 1.0 / (gamma - 1.0)   (enthalpy_density, :104-107)
};

__device__ inline Gamma make_gamma(double gamma)
{
    Gamma g;
    g.gamma = gamma;
    g.gm1 = gamma - 1.0;
    g.hfac = 1.0 + 1.0 / (gamma - 1.0);
    return g;
}

__device__ inline double enthalpy_density(const State5& P, const Gamma& g) { return P[0] + P[4] * g.hfac; }
// PLANAR (here and below): the azimuthal momentum / four-velocity of every cell and of the nozzle row is +0.0, BIT FOR BIT - upstream's `cloud`
// problem (radial envelope and nozzle, 2-D axisymmetric); the steppers check the bit pattern (mh_cloud_desc.planar). With that input the
// reference's own operations return +0.0 for every quantity of the azimuthal direction (W0 * (+0) / x, the slopes of zeros, v * (+0) + p * 0.0,
// the HLLE combination and its quotient, u + ((+0) + (+0) + (+-0) dv) dt, 0 * 1/2 + 0 * 1/2: a zero of either sign plus +0 is +0), so those
// operations are left out and +0.0 is written. The other four components see the operands they see upstream: a sum of squares is never -0, so
// `x + (+0)` may go; where the zero's SIGN could act on another component the operation stays with a literal zero in the place of the product -
// beta_along's `... + P[3] * 0.0` and the polar source term's `up * up * cot - ur * uq` (with ur * uq = +0, as in the initial state, the sign of
// cot decides the sign of the difference). Bit-identical to the reference, azimuthal component included (the golden `cloud` steps and the
// 388-step run of tests/test_gpu_long_runs_vs_reference.py are taken on this kernel).
template<bool PLANAR = false>
__device__ inline double gamma_beta_squared(const State5& P) { return PLANAR ? P[1] * P[1] + P[2] * P[2] : P[1] * P[1] + P[2] * P[2] + P[3] * P[3]; }

// returns status bits; P is always written (as in the reference before it throws)
template<bool PLANAR = false>
__device__ inline int recover_primitive(const State5& U, const Gamma& g, double temperature_floor, State5& P)
{
    const double gm = g.gamma;
    const double D = U[0], tau = U[4];
    const double SS = PLANAR ? U[1] * U[1] + U[2] * U[2] : U[1] * U[1] + U[2] * U[2] + U[3] * U[3];
    bool solution_found = false;
    int iteration = 0;
    double W0 = 1.0;
    double p = 0.0;

    while (iteration < 50)
    {
        const double x = tau + D + p;
        const double v2 = std_min(SS / (x * x), 1.0 - 1e-10);      // std::pow(x, 2): g++ -O2 emits x * x
        const double W2 = 1.0 / (1.0 - v2);
        const double W = sqrt(W2);
        const double e = (tau + D * (1.0 - W) + p * (1.0 - W2)) / (D * W);
        const double d = D / W;
        const double h = 1.0 + e + p / d;
        const double cs2 = gm * p / (d * h);
        const double f = d * e * (gm - 1.0) - p;
        const double gg = v2 * cs2 - 1.0;
        p -= f / gg;
        if (fabs(f) < 1e-10)
        {
            W0 = W;
            solution_found = true;
            break;
        }
        ++iteration;
    }
    if (temperature_floor > 0.0) p = std_max(p, temperature_floor * D / W0);
    const Recip Rx = make_recip(tau + D + p, 1.0);
    P[0] = D / W0;
    if constexpr (PLANAR)
    {
        double m[2] = {W0 * U[1], W0 * U[2]};
        divide_group<2>(m, Rx);
        P[1] = m[0];
        P[2] = m[1];
        P[3] = 0.0;
    }
    else
    {
        double m[3] = {W0 * U[1], W0 * U[2], W0 * U[3]};
        divide_group<3>(m, Rx);
        P[1] = m[0];
        P[2] = m[1];
        P[3] = m[2];
    }
    P[4] = p;
    int status = 0;
    if (! solution_found) status |= MH_STATUS_C2P_FAILED;
    if (P[0] <= 0.0) status |= MH_STATUS_NEG_DENSITY;
    if (P[4] <= 0.0) status |= MH_STATUS_NEG_PRESSURE;
    if (W0 != W0) status |= MH_STATUS_NAN;
    return status;
}

// everything riemann_hlle needs from one side: U, F, wavespeeds
template<int AXIS, bool PLANAR = false>
__device__ inline void side(const State5& P, const Gamma& g, State5& U, State5& F, double& lam_m, double& lam_p)
{
    using N = Normal<AXIS>;
    static_assert(! PLANAR || AXIS != 2, "a planar state has no azimuthal axis");
    const double uu = gamma_beta_squared<PLANAR>(P);
    const double W = sqrt(1.0 + uu);
    const double H = enthalpy_density(P, g);
    const double h = H / P[0];
    const double D = P[0] * W;
    const double p = P[4];
    U[0] = D;
    U[1] = D * P[1] * h;
    U[2] = D * P[2] * h;
    U[3] = PLANAR ? 0.0 : D * P[3] * h;
    U[4] = D * h * W - p - D;
    // beta_along (planar: (+0) * 0.0 = +0.0 - the addition stays, it can turn a -0 sum into +0)
    const double v = (PLANAR ? P[1] * N::n1 + P[2] * N::n2 + 0.0 : P[1] * N::n1 + P[2] * N::n2 + P[3] * N::n3) / W;
    F[0] = v * U[0];
    F[1] = v * U[1] + p * N::n1;
    F[2] = v * U[2] + p * N::n2;
    F[3] = PLANAR ? 0.0 : v * U[3] + p * N::n3;           // v * (+0) + p * 0.0: (+-0) + (+0) = +0
    F[4] = v * U[4] + p * v;
    const double c2 = g.gamma * p / H;
    const double vv = uu / (1 + uu);
    const double v2 = v * v;
    const double k0 = sqrt(c2 * (1 - vv) * (1 - vv * c2 - v2 * (1 - c2)));
    double lam[2] = {v * (1 - c2) - k0, v * (1 - c2) + k0};
    divide_group<2>(lam, make_recip(1 - vv * c2, 1.0));
    lam_m = lam[0];
    lam_p = lam[1];
}

template<int AXIS, bool PLANAR = false> __device__ inline State5 riemann_hlle(const State5& Pl, const State5& Pr, const Gamma& g)
{
    State5 Ul, Ur, Fl, Fr;
    double alm, alp, arm, arp;
    side<AXIS, PLANAR>(Pl, g, Ul, Fl, alm, alp);
    side<AXIS, PLANAR>(Pr, g, Ur, Fr, arm, arp);
    const double ap = std_max(0.0, std_max(alp, arp));
    const double am = std_min(0.0, std_min(alm, arm));
    if constexpr (PLANAR)
    {
        // (+0) ap - (+0) am - ((+0) - (+0)) ap am with ap >= 0 >= am is +0, and +0 / (ap - am) is +0: four quotients instead of five
        double x[4];
        constexpr int at[4] = {0, 1, 2, 4};
#pragma unroll
        for (int k = 0; k < 4; ++k) x[k] = Fl[at[k]] * ap - Fr[at[k]] * am - (Ul[at[k]] - Ur[at[k]]) * ap * am;
        divide_group<4>(x, make_recip(ap - am, 1.0));
        State5 N;
        N[0] = x[0]; N[1] = x[1]; N[2] = x[2]; N[3] = 0.0; N[4] = x[3];
        return N;
    }
    else
    {
        State5 N;
#pragma unroll
        for (int q = 0; q < 5; ++q) N[q] = Fl[q] * ap - Fr[q] * am - (Ul[q] - Ur[q]) * ap * am;
        divide_group<5>(N.v, make_recip(ap - am, 1.0));
        return N;
    }
}

__device__ inline State5 to_conserved_density(const State5& P, const Gamma& g)
{
    const double W = sqrt(1.0 + gamma_beta_squared(P));
    const double h = enthalpy_density(P, g) / P[0];
    const double D = P[0] * W;
    State5 U;
    U[0] = D;
    U[1] = D * P[1] * h;
    U[2] = D * P[2] * h;
    U[3] = D * P[3] * h;
    U[4] = D * h * W - P[4] - D;
    return U;
}

template<bool PLANAR = false>
__device__ inline State5 source_terms(const State5& P, double r, double cotq, const Gamma& g)
{
    const double ur = P[1], uq = P[2], up = P[3], pg = P[4];
    const double H = enthalpy_density(P, g);
    State5 S;
    S[0] = 0.0;
    S[4] = 0.0;
    if constexpr (PLANAR)
    {
        // up * up = +0; `(+0) * cot - ur * uq` keeps its subtraction (see the note on PLANAR above). The azimuthal source -up H (...) / r is a zero
        // of either sign that meets +0 terms in the update: not formed
        double s[2] = {2.0 * pg + H * (uq * uq), cotq * pg + H * (0.0 * cotq - ur * uq)};
        divide_group<2>(s, make_recip(r, 1.0));
        S[1] = s[0];
        S[2] = s[1];
        S[3] = 0.0;
    }
    else
    {
        double s[3] = {2.0 * pg + H * (uq * uq + up * up), cotq * pg + H * (up * up * cotq - ur * uq), -up * H * (ur + uq * cotq)};
        divide_group<3>(s, make_recip(r, 1.0));
        S[1] = s[0];
        S[2] = s[1];
        S[3] = s[2];
    }
    return S;
}

} // namespace srhd
} // namespace mh
