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
__device__ inline double gamma_beta_squared(const State5& P) { return P[1] * P[1] + P[2] * P[2] + P[3] * P[3]; }

// returns status bits; P is always written (as in the reference before it throws)
__device__ inline int recover_primitive(const State5& U, const Gamma& g, double temperature_floor, State5& P)
{
    const double gm = g.gamma;
    const double D = U[0], tau = U[4];
    const double SS = U[1] * U[1] + U[2] * U[2] + U[3] * U[3];
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
    double m[3] = {W0 * U[1], W0 * U[2], W0 * U[3]};
    divide_group<3>(m, Rx);
    P[0] = D / W0;
    P[1] = m[0];
    P[2] = m[1];
    P[3] = m[2];
    P[4] = p;
    int status = 0;
    if (! solution_found) status |= MH_STATUS_C2P_FAILED;
    if (P[0] <= 0.0) status |= MH_STATUS_NEG_DENSITY;
    if (P[4] <= 0.0) status |= MH_STATUS_NEG_PRESSURE;
    if (W0 != W0) status |= MH_STATUS_NAN;
    return status;
}

// everything riemann_hlle needs from one side: U, F, wavespeeds
template<int AXIS>
__device__ inline void side(const State5& P, const Gamma& g, State5& U, State5& F, double& lam_m, double& lam_p)
{
    using N = Normal<AXIS>;
    const double uu = gamma_beta_squared(P);
    const double W = sqrt(1.0 + uu);
    const double H = enthalpy_density(P, g);
    const double h = H / P[0];
    const double D = P[0] * W;
    const double p = P[4];
    U[0] = D;
    U[1] = D * P[1] * h;
    U[2] = D * P[2] * h;
    U[3] = D * P[3] * h;
    U[4] = D * h * W - p - D;
    const double v = (P[1] * N::n1 + P[2] * N::n2 + P[3] * N::n3) / W;      // beta_along
    F[0] = v * U[0];
    F[1] = v * U[1] + p * N::n1;
    F[2] = v * U[2] + p * N::n2;
    F[3] = v * U[3] + p * N::n3;
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

template<int AXIS> __device__ inline State5 riemann_hlle(const State5& Pl, const State5& Pr, const Gamma& g)
{
    State5 Ul, Ur, Fl, Fr;
    double alm, alp, arm, arp;
    side<AXIS>(Pl, g, Ul, Fl, alm, alp);
    side<AXIS>(Pr, g, Ur, Fr, arm, arp);
    const double ap = std_max(0.0, std_max(alp, arp));
    const double am = std_min(0.0, std_min(alm, arm));
    State5 N;
#pragma unroll
    for (int q = 0; q < 5; ++q) N[q] = Fl[q] * ap - Fr[q] * am - (Ul[q] - Ur[q]) * ap * am;
    divide_group<5>(N.v, make_recip(ap - am, 1.0));
    return N;
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

__device__ inline State5 source_terms(const State5& P, double r, double cotq, const Gamma& g)
{
    const double ur = P[1], uq = P[2], up = P[3], pg = P[4];
    const double H = enthalpy_density(P, g);
    double s[3] = {2.0 * pg + H * (uq * uq + up * up), cotq * pg + H * (up * up * cotq - ur * uq), -up * H * (ur + uq * cotq)};
    divide_group<3>(s, make_recip(r, 1.0));
    State5 S;
    S[0] = 0.0;
    S[1] = s[0];
    S[2] = s[1];
    S[3] = s[2];
    S[4] = 0.0;
    return S;
}

} // namespace srhd
} // namespace mh
