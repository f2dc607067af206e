// Dev probe: accuracy of v_rcp_f64 / v_rsq_f64 and of the refinement variants used by MH_ARITH_FAST (max relative error in ulps
// of the exact result over log-uniform random inputs). hipcc --offload-arch=gfx950 -O2 -o rcp_rsq_accuracy rcp_rsq_accuracy.hip
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <random>
#include <vector>

__global__ void probe(const double* x, double* out, int n)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double v = x[i];
    // raw
    double r0 = __builtin_amdgcn_rcp(v);
    double y0 = __builtin_amdgcn_rsq(v);
    // rcp: two Newton steps (current)
    double r = r0, e = __builtin_fma(-v, r, 1.0); r = __builtin_fma(r, e, r); e = __builtin_fma(-v, r, 1.0); r = __builtin_fma(r, e, r);
    // rcp: one cubic step
    double e1 = __builtin_fma(-v, r0, 1.0);
    double rc = __builtin_fma(r0, __builtin_fma(e1, e1, e1), r0);
    // rsq: one cubic step  y (1 + e/2 + 3 e^2 / 8), e = 1 - x y^2
    double t = v * y0;
    double es = __builtin_fma(-t, y0, 1.0);
    double p = __builtin_fma(0.375, es, 0.5);
    double yc = __builtin_fma(y0, p * es, y0);
    // rsq: two Newton steps
    double yn = y0;
    for (int k = 0; k < 2; ++k) { double tt = v * yn; double ee = __builtin_fma(-tt, yn, 1.0); yn = __builtin_fma(0.5 * yn, ee, yn); }
    out[6 * i + 0] = r0; out[6 * i + 1] = y0; out[6 * i + 2] = r; out[6 * i + 3] = rc; out[6 * i + 4] = yc; out[6 * i + 5] = yn;
}

int main()
{
    const int n = 1 << 22;
    std::vector<double> x(n), out(6 * (size_t) n);
    std::mt19937_64 rng(12345);
    std::uniform_real_distribution<double> ex(-30.0, 30.0);
    for (auto& v : x) v = std::exp2(ex(rng));
    double *dx, *dout;
    hipMalloc(&dx, n * 8); hipMalloc(&dout, 6 * (size_t) n * 8);
    hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice);
    probe<<<n / 256, 256>>>(dx, dout, n);
    hipMemcpy(out.data(), dout, 6 * (size_t) n * 8, hipMemcpyDeviceToHost);
    double worst[6] = {0};
    for (int i = 0; i < n; ++i)
    {
        const long double v = x[i];
        const long double exact[6] = {1.0L / v, 1.0L / sqrtl(v), 1.0L / v, 1.0L / v, 1.0L / sqrtl(v), 1.0L / sqrtl(v)};
        for (int k = 0; k < 6; ++k)
        {
            const double ulp = std::fabs(std::nextafter((double) exact[k], INFINITY) - (double) exact[k]);
            const double err = (double) fabsl((long double) out[6 * (size_t) i + k] - exact[k]) / ulp;
            if (err > worst[k]) worst[k] = err;
        }
    }
    const char* names[6] = {"v_rcp_f64", "v_rsq_f64", "rcp + 2 Newton", "rcp + 1 cubic", "rsq + 1 cubic", "rsq + 2 Newton"};
    for (int k = 0; k < 6; ++k) std::printf("%-16s max error %.3g ulp\n", names[k], worst[k]);
    return 0;
}
