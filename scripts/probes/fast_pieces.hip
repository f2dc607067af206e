// Dev probe (never linked into the library): each piece of the MH_ARITH_FAST 2-D Euler PLM row step - the arithmetic both waves of a pair of
// euler2d_fused_rk2_kernel run per row - as a kernel of its own, so that scripts/fast_isa_table.py can count its instructions by class.
// hipcc -S --cuda-device-only -I mara3_amd/csrc
#include <hip/hip_runtime.h>
#include "euler_device.hpp"
#include "euler_device_fast.hpp"
#include "euler2d_rows.hpp"
using namespace mh;
using A = FastArith;
#define LOAD(i) State5 s##i; for (int q = 0; q < 5; ++q) s##i[q] = in[(i * 5 + q) * n + t];
#define STORE(x) for (int q = 0; q < 5; ++q) out[q * n + t] = x[q];
extern "C" __global__ void piece_baseline(const double* in, double* out, int n) { int t = threadIdx.x; LOAD(0) LOAD(1) LOAD(2) State5 r; for (int q = 0; q < 5; ++q) r[q] = s0[q]; STORE(r) out[5 * n + t] = s1[0] + s2[0]; }
extern "C" __global__ void piece_c2p(const double* in, double* out, int n, double gamma) { int t = threadIdx.x; LOAD(0) LOAD(1) LOAD(2) const A::Gamma gl = A::gamma_law(gamma); State5 r = A::c2p(s0, gl); STORE(r) out[5 * n + t] = s1[0] + s2[0]; }
extern "C" __global__ void piece_difference(const double* in, double* out, int n) { int t = threadIdx.x; LOAD(0) LOAD(1) LOAD(2) State5 r = A::difference(s0, s1); STORE(r) out[5 * n + t] = s2[0]; }
extern "C" __global__ void piece_plm(const double* in, double* out, int n, double theta) { int t = threadIdx.x; LOAD(0) LOAD(1) LOAD(2) const A::Limiter lim = A::limiter(theta); State5 r = A::plm_from_differences(s0, s1, lim); STORE(r) out[5 * n + t] = s2[0]; }
extern "C" __global__ void piece_faces(const double* in, double* out, int n, double theta) { int t = threadIdx.x; LOAD(0) LOAD(1) LOAD(2) const A::Limiter lim = A::limiter(theta); State5 a = A::plus(s0, s1, lim), b = A::minus(s0, s2, lim); STORE(a) for (int q = 0; q < 5; ++q) out[(5 + q) * n + t] = b[q]; }
#define FLUX(name, R, AX) extern "C" __global__ void name(const double* in, double* out, int n, double gamma) { int t = threadIdx.x; LOAD(0) LOAD(1) LOAD(2) const A::Gamma gl = A::gamma_law(gamma); State5 r = A::template flux<R, AX>(s0, s1, gl); STORE(r) out[5 * n + t] = s2[0]; }
FLUX(piece_hllc0, 1, 0)
FLUX(piece_hllc1, 1, 1)
FLUX(piece_hlle0, 0, 0)
FLUX(piece_hlle1, 0, 1)
extern "C" __global__ void piece_update(const double* in, double* out, int n, double cx, double cy)
{
    int t = threadIdx.x; LOAD(0) LOAD(1) LOAD(2) State5 r;
    for (int q = 0; q < 5; ++q) r[q] = A::update2(in[(15 + q) * n + t], s0[q], s1[q], s1[q], s2[q], cx, cy);
    STORE(r)
}
extern "C" __global__ void piece_update_combine(const double* in, double* out, int n, double cx, double cy)
{
    int t = threadIdx.x; LOAD(0) LOAD(1) LOAD(2) State5 r;
    for (int q = 0; q < 5; ++q) r[q] = A::combine(in[(20 + q) * n + t], A::update2(in[(15 + q) * n + t], s0[q], s1[q], s1[q], s2[q], cx, cy), 0.5);
    STORE(r)
}
// the four lane-to-lane exchanges of a row: primitives from the right, differences from the left, face states from the left, fluxes from the right
extern "C" __global__ void piece_lane_moves(const double* in, double* out, int n)
{
    int t = threadIdx.x; LOAD(0) LOAD(1) LOAD(2)
    State5 a = from_right(s0), b = from_left(s1), c = from_left(s2), d = from_right(s1), r;
    for (int q = 0; q < 5; ++q) r[q] = a[q] + b[q] + c[q] + d[q];          // (15 additions of the probe itself: subtracted by the script)
    STORE(r)
}
