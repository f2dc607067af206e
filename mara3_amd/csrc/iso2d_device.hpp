// Device-side 2-D (locally) isothermal hydrodynamics, mara::iso2d, for gfx950; strict arithmetic
// (reference operation order, no FMA contraction, IEEE division and sqrt):
//   to_conserved_per_area         src/physics_iso2d.hpp:249-258      recover_primitive(U)      :351-362
//   to_conserved_angmom_per_area  :263-272                           recover_primitive(Q, x)   :376-390
//   flux :299-307   wavespeeds :320-328   max_wavespeed :330-337     riemann_hlle :488-506
//   compute_hllc_variables :610-687, Ul_star/Ur_star/interface_flux :556-583, riemann_hllc :704-712
// Component order is the LOGICAL one (Sigma, x, y); the reference's std::tuple storage order is an ABI
// detail that never reaches the device. Where the reference throws, a flag is returned.
#pragma once
#include <hip/hip_runtime.h>
#include "euler_device.hpp"

namespace mh {
namespace iso2d {

struct State3
{
    double v[3];
    __device__ double& operator[](int i) { return v[i]; }
    __device__ const double& operator[](int i) const { return v[i]; }
};

__device__ inline State3 to_conserved(const State3& P)
{
    State3 U;
    U[0] = P[0];
    U[1] = P[0] * P[1];
    U[2] = P[0] * P[2];
    return U;
}

__device__ inline bool recover_primitive(const State3& U, State3& P)
{
    double m[2] = {U[1], U[2]};
    divide_group<2>(m, make_recip(U[0], 1.0));
    P[0] = U[0];
    P[1] = m[0];
    P[2] = m[1];
    return U[0] < 0.0;
}

__device__ inline State3 to_conserved_angmom(const State3& P, double x0, double x1)
{
    State3 Q;
    Q[0] = P[0];
    Q[1] = P[0] * (x0 * P[1] + x1 * P[2]);
    Q[2] = P[0] * (x0 * P[2] - x1 * P[1]);
    return Q;
}

__device__ inline bool recover_primitive_angmom(const State3& Q, double x0, double x1, State3& P)
{
    const double sigma = Q[0];
    double a[2] = {Q[1], Q[2]};
    divide_group<2>(a, make_recip(sigma, 1.0));
    const double sr = a[0], lz = a[1];
    const double r2 = x0 * x0 + x1 * x1;
    double b[2] = {sr * x0 - lz * x1, sr * x1 + lz * x0};
    divide_group<2>(b, make_recip(r2, 1.0));
    P[0] = sigma;
    P[1] = b[0];
    P[2] = b[1];
    return sigma < 0.0;
}

template<int AXIS> __device__ inline double velocity_along(const State3& P)
{
    using N = Normal<AXIS>;
    return P[1] * N::n1 + P[2] * N::n2 + 0.0 * N::n3;
}

template<int AXIS> __device__ inline State3 flux(const State3& P, double cs2)
{
    using N = Normal<AXIS>;
    const double v = velocity_along<AXIS>(P);
    const double p = P[0] * cs2;
    State3 F;
    F[0] = v * P[0];
    F[1] = v * P[0] * P[1] + p * N::n1;
    F[2] = v * P[0] * P[2] + p * N::n2;
    return F;
}

__device__ inline double max_wavespeed(const State3& P, double cs2)
{
    const double cs = sqrt(cs2);
    const double vx = velocity_along<0>(P), vy = velocity_along<1>(P);
    const double ax = std_max(fabs(vx - cs), fabs(vx + cs));
    const double ay = std_max(fabs(vy - cs), fabs(vy + cs));
    return std_max(ax, ay);
}

template<int AXIS> __device__ inline State3 riemann_hlle(const State3& Pl, const State3& Pr, double cs2l, double cs2r)
{
    const State3 Ul = to_conserved(Pl), Ur = to_conserved(Pr);
    const double csl = sqrt(cs2l), csr = sqrt(cs2r);
    const double vl = velocity_along<AXIS>(Pl), vr = velocity_along<AXIS>(Pr);
    const State3 Fl = flux<AXIS>(Pl, cs2l), Fr = flux<AXIS>(Pr, cs2r);
    const double ap = std_max(0.0, std_max(vl + csl, vr + csr));
    const double am = std_min(0.0, std_min(vl - csl, vr - csr));
    State3 N;
#pragma unroll
    for (int q = 0; q < 3; ++q) N[q] = Fl[q] * ap - Fr[q] * am - (Ul[q] - Ur[q]) * ap * am;
    divide_group<3>(N.v, make_recip(ap - am, 1.0));
    return N;
}

// returns true where the reference's interface_flux throws; contact = s_star
template<int AXIS> __device__ inline bool riemann_hllc(const State3& Pl, const State3& Pr, double cs2l, double cs2r, State3& F, double& contact)
{
    using N = Normal<AXIS>;
    const double nh[2] = {N::n1, N::n2};
    const double ul = velocity_along<AXIS>(Pl), ur = velocity_along<AXIS>(Pr);
    const double vperp_l[2] = {Pl[1] - nh[0] * ul, Pl[2] - nh[1] * ul};
    const double vperp_r[2] = {Pr[1] - nh[0] * ur, Pr[2] - nh[1] * ur};
    const double sigma_l = Pl[0], sigma_r = Pr[0];
    const double sigma_bar = 0.5 * (sigma_l + sigma_r);
    const double al = sqrt(cs2l), ar = sqrt(cs2r);
    const double a_bar = 0.5 * (al + ar);
    const double press_l = sigma_l * cs2l, press_r = sigma_r * cs2r;
    const double ppvrs = 0.5 * (press_l + press_r) - 0.5 * (ur - ul) * sigma_bar * a_bar;
    const double pstar = std_max(0.0, ppvrs);
    const double ql = std_max(1.0, sqrt(pstar / press_l));
    const double qr = std_max(1.0, sqrt(pstar / press_r));
    const double sl = ul - al * ql;
    const double sr = ur + ar * qr;
    const double den = sigma_l * (sl - ul) - sigma_r * (sr - ur);
    const double sstar = (press_r - press_l + ul * sigma_l * (sl - ul) - ur * sigma_r * (sr - ur)) / den;
    contact = sstar;
    const State3 Ul = to_conserved(Pl), Ur = to_conserved(Pr);
    const State3 Fl = flux<AXIS>(Pl, al * al), Fr = flux<AXIS>(Pr, ar * ar);

    if (0.0 <= sl)
    {
        F = Fl;
    }
    else if (sl <= 0.0 && 0.0 <= sstar)
    {
        const double f = sigma_l * (sl - ul) / (sl - sstar);
        const State3 Us = {{f, f * (sstar * nh[0] + vperp_l[0]), f * (sstar * nh[1] + vperp_l[1])}};
#pragma unroll
        for (int q = 0; q < 3; ++q) F[q] = Fl[q] + (Us[q] - Ul[q]) * sl;
    }
    else if (sstar <= 0.0 && 0.0 <= sr)
    {
        const double f = sigma_r * (sr - ur) / (sr - sstar);
        const State3 Us = {{f, f * (sstar * nh[0] + vperp_r[0]), f * (sstar * nh[1] + vperp_r[1])}};
#pragma unroll
        for (int q = 0; q < 3; ++q) F[q] = Fr[q] + (Us[q] - Ur[q]) * sr;
    }
    else if (sr <= 0.0)
    {
        F = Fr;
    }
    else
    {
#pragma unroll
        for (int q = 0; q < 3; ++q) F[q] = __builtin_nan("");
        return true;
    }
    return false;
}

} // namespace iso2d
} // namespace mh
