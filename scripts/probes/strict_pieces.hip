// Dev probe (never linked into the library): each piece of the STRICT 2-D Euler PLM + HLLE row step as a kernel of its own, so that
// scripts/strict_isa_table.py can count its instructions by class. hipcc -S --cuda-device-only -I mara3_amd/csrc
#include <hip/hip_runtime.h>
#include "euler_device.hpp"
using namespace mh;
#define LOAD(i) State5 s##i; for (int q = 0; q < 5; ++q) s##i[q] = in[(i * 5 + q) * n + t];
#define STORE(x) for (int q = 0; q < 5; ++q) out[q * n + t] = x[q];
extern "C" __global__ void piece_baseline(const double* in, double* out, int n) { int t = threadIdx.x; LOAD(0) LOAD(1) LOAD(2) State5 r; for (int q = 0; q < 5; ++q) r[q] = s0[q]; STORE(r) out[5 * n + t] = s1[0] + s2[0]; }
extern "C" __global__ void piece_c2p(const double* in, double* out, int n, double gamma) { int t = threadIdx.x; LOAD(0) LOAD(1) LOAD(2) State5 r = recover_primitive(s0, gamma, 0.0); STORE(r) out[5 * n + t] = s1[0] + s2[0]; }
extern "C" __global__ void piece_plm(const double* in, double* out, int n, double theta) { int t = threadIdx.x; LOAD(0) LOAD(1) LOAD(2) State5 r = plm_gradient(s0, s1, s2, theta); STORE(r) }
extern "C" __global__ void piece_faces(const double* in, double* out, int n) { int t = threadIdx.x; LOAD(0) LOAD(1) LOAD(2) State5 a = face_plus(s0, s1), b = face_minus(s0, s2); State5 r; for (int q = 0; q < 5; ++q) r[q] = a[q] + b[q]; STORE(r) }
extern "C" __global__ void piece_hlle0(const double* in, double* out, int n, double gamma) { int t = threadIdx.x; LOAD(0) LOAD(1) LOAD(2) const GammaLaw g = make_gamma_law(gamma); State5 r = riemann_hlle<0>(s0, s1, g); STORE(r) out[5 * n + t] = s2[0]; }
extern "C" __global__ void piece_hlle1(const double* in, double* out, int n, double gamma) { int t = threadIdx.x; LOAD(0) LOAD(1) LOAD(2) const GammaLaw g = make_gamma_law(gamma); State5 r = riemann_hlle<1>(s0, s1, g); STORE(r) out[5 * n + t] = s2[0]; }
extern "C" __global__ void piece_update(const double* in, double* out, int n, double cx, double cy, double w)
{
    int t = threadIdx.x; LOAD(0) LOAD(1) LOAD(2) State5 r;
    for (int q = 0; q < 5; ++q)
    {
        const double lx = (s1[q] - s0[q]) * cx, ly = (s2[q] - s1[q]) * cy;
        const double u1 = in[(15 + q) * n + t] - (lx + ly);
        r[q] = in[(20 + q) * n + t] * (1.0 - w) + u1 * w;
    }
    STORE(r)
}
