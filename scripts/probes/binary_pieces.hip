// Dev probe (never linked into the library): the pieces of one cell-row of binary_stage_kernel<BinFast, COMBINE, false> (binary_kernel.hpp: BASELINE
// config 3, advance_u of src/subprog_binary_scheme.cpp:790-905) as kernels of their own, so that scripts/binary_isa_table.py can count their
// instructions by class. hipcc -S --cuda-device-only -I mara3_amd/csrc
#include <hip/hip_runtime.h>
#include "binary_kernel.hpp"
using namespace mh;
using A = BinFast;
#define LOAD(i) State3 s##i; for (int q = 0; q < 3; ++q) s##i[q] = in[(i * 3 + q) * n + t];
#define STORE(x) for (int q = 0; q < 3; ++q) out[q * n + t] = x[q];
// C3's run-time-uniform switches are pinned at compile time (alpha viscosity without tanh cut-off, two-body sound speed), so that the static count
// of a piece is what a wave executes; the per-launch constants A::Ctx arrive as an argument, as the kernel forms them once per wave, not per row
#define HEAD int t = threadIdx.x; LOAD(0) LOAD(1) LOAD(2) LOAD(3) LOAD(4) LOAD(5) BinaryConsts c = c_in; c.axisym = 0; c.rc_cut = 0.0; c.nu = 0.0; \
             const double xf = in[18 * n + t], yf = in[19 * n + t];
#define TAIL(r) STORE(r) out[3 * n + t] = s0[0] + s1[0] + s2[0] + s3[0] + s4[0] + s5[0] + xf + yf;
// every probe loads six states and two positions and adds them up at the end (so that no load is dropped): the baseline is that alone
extern "C" __global__ void piece_baseline(const double* in, double* out, int n, BinaryConsts c_in, A::Ctx k) { HEAD State3 r = s0; TAIL(r) }
extern "C" __global__ void piece_c2p(const double* in, double* out, int n, BinaryConsts c_in, A::Ctx k) { HEAD State3 r = A::c2p<false>(s0, xf, yf); TAIL(r) }
extern "C" __global__ void piece_plm(const double* in, double* out, int n, BinaryConsts c_in, A::Ctx k, double theta) { HEAD State3 r = A::plm_per_length(s0, s1, s2, theta, k); TAIL(r) }
extern "C" __global__ void piece_dpp3(const double* in, double* out, int n, BinaryConsts c_in, A::Ctx k) { HEAD State3 r = bdpp_left(s0); TAIL(r) }
extern "C" __global__ void piece_cs2(const double* in, double* out, int n, BinaryConsts c_in, A::Ctx k) { HEAD State3 r = s0; r[0] = A::cs2(c, k, xf, yf); TAIL(r) }
extern "C" __global__ void piece_cs2_nu(const double* in, double* out, int n, BinaryConsts c_in, A::Ctx k) { HEAD State3 r = s0; const double cs2 = A::cs2(c, k, xf, yf); r[0] = cs2; r[1] = A::nu(c, k, xf, yf, cs2); TAIL(r) }
extern "C" __global__ void piece_hlle0(const double* in, double* out, int n, BinaryConsts c_in, A::Ctx k) { HEAD State3 r = A::hlle<0>(s0, s1, xf); TAIL(r) }
extern "C" __global__ void piece_hlle1(const double* in, double* out, int n, BinaryConsts c_in, A::Ctx k) { HEAD State3 r = A::hlle<1>(s0, s1, xf); TAIL(r) }
extern "C" __global__ void piece_face0(const double* in, double* out, int n, BinaryConsts c_in, A::Ctx k) { HEAD State3 r = binary_face_flux<A, 0, false>(c, k, xf, yf, s0, s1, s2, s3, s4, s5); TAIL(r) }
extern "C" __global__ void piece_face1(const double* in, double* out, int n, BinaryConsts c_in, A::Ctx k) { HEAD State3 r = binary_face_flux<A, 1, false>(c, k, xf, yf, s0, s1, s2, s3, s4, s5); TAIL(r) }
extern "C" __global__ void piece_gravity2(const double* in, double* out, int n, BinaryConsts c_in, A::Ctx k)
{
    HEAD State3 r = s0; double fg[2][2];
    for (int b = 0; b < 2; ++b) A::gravity(c, b, xf - c.body[5 * b + 1], yf - c.body[5 * b + 2], s0[0], fg[b]);
    r[1] = fg[0][0] + fg[1][0]; r[2] = fg[0][1] + fg[1][1]; TAIL(r)
}
extern "C" __global__ void piece_sink2_near(const double* in, double* out, int n, BinaryConsts c_in, A::Ctx k)
{
    HEAD State3 r = s0; double rate = 0.0;
    for (int b = 0; b < 2; ++b) rate += binary_sink_rate<A>(c, xf - c.body[5 * b + 1], yf - c.body[5 * b + 2]);
    r[0] = rate; TAIL(r)
}
extern "C" __global__ void piece_sink2_far(const double* in, double* out, int n, BinaryConsts c_in, A::Ctx k)
{
    HEAD State3 r = s0; double rate = 0.0;
    for (int b = 0; b < 2; ++b) rate += __any(A::sink_a2(c, xf - c.body[5 * b + 1], yf - c.body[5 * b + 2]) < 750.0) ? 1.0 : 0.0;
    r[0] = rate; TAIL(r)
}
// the FAST source terms and update of the kernel as they stand there (gravity, sinks, buffer, floor, totals, update; COMBINE): fluxes are s1 .. s4
template<bool COMBINE, bool NEAR> __device__ inline void sources_update(const double* in, double* out, int n, const BinaryConsts& c_in, const A::Ctx& k, double dt, double weight, double brate, bool writes)
{
    HEAD
    const State3 u0 = s0, Uinit = s5;
    const double xc = xf, yc = yf, dx = c.h, dy = c.h;
    const double dA = dx * dy, dtA = dt * dA;
    double part[NPART];
    for (int i = 0; i < NPART; ++i) part[i] = in[(20 + i) * n + t];
    double fg[2][2], rate = 0.0;
    for (int bdy = 0; bdy < 2; ++bdy)
    {
        const double d0 = xc - c.body[5 * bdy + 1], d1 = yc - c.body[5 * bdy + 2];
        A::gravity(c, bdy, d0, d1, u0[0], fg[bdy]);
        if constexpr (NEAR) rate += binary_sink_rate<A>(c, d0, d1);
        else                rate += __any(A::sink_a2(c, d0, d1) < 750.0) ? 1.0 : 0.0;          // a wave out of range of the sink: the test alone
    }
    const double w0 = __builtin_fma(-rate, dt, u0[0] < c.floor_sigma ? 1e-2 : 0.0);
    const double bw = brate * dt;
    double sb[3], s[3];
    for (int q = 0; q < 3; ++q) sb[q] = (Uinit[q] - u0[q]) * bw;
    s[0] = __builtin_fma(u0[0], w0, sb[0]);
    s[1] = __builtin_fma(u0[1], w0, __builtin_fma(fg[0][0] + fg[1][0], dt, sb[1]));
    s[2] = __builtin_fma(u0[2], w0, __builtin_fma(fg[0][1] + fg[1][1], dt, sb[2]));
    if (writes)
    {
        for (int bdy = 0; bdy < 2; ++bdy)
        {
            part[0 + bdy] = __builtin_fma(__builtin_fma(xc, fg[bdy][1], -yc * fg[bdy][0]), dtA, part[0 + bdy]);
            part[2 + bdy] = __builtin_fma(fg[bdy][0], dtA, part[2 + bdy]);
            part[4 + bdy] = __builtin_fma(fg[bdy][1], dtA, part[4 + bdy]);
        }
        part[6] = __builtin_fma(sb[0], dA, part[6]);
        part[7] = __builtin_fma(__builtin_fma(xc, sb[2], -yc * sb[1]), dA, part[7]);
    }
    const double rA = dt * k.inv_h2;
    State3 Un;
    for (int q = 0; q < 3; ++q)
    {
        const double l = (s2[q] - s1[q]) + (s4[q] - s3[q]);
        const double u1 = __builtin_fma(-l, rA, u0[q] + s[q]);
        if constexpr (COMBINE) Un[q] = __builtin_fma(u1, weight, in[(30 + q) * n + t] * (1.0 - weight));
        else                   Un[q] = u1;
    }
    for (int i = 0; i < NPART; ++i) out[(4 + i) * n + t] = part[i];
    TAIL(Un)
}
extern "C" __global__ void piece_sources_update(const double* in, double* out, int n, BinaryConsts c, A::Ctx k, double dt, double w, double brate, int writes) { sources_update<false, false>(in, out, n, c, k, dt, w, brate, writes != 0); }
extern "C" __global__ void piece_sources_update_combine(const double* in, double* out, int n, BinaryConsts c, A::Ctx k, double dt, double w, double brate, int writes) { sources_update<true, false>(in, out, n, c, k, dt, w, brate, writes != 0); }
extern "C" __global__ void piece_sources_update_near(const double* in, double* out, int n, BinaryConsts c, A::Ctx k, double dt, double w, double brate, int writes) { sources_update<false, true>(in, out, n, c, k, dt, w, brate, writes != 0); }
