// Dev probe: issue cost (cycles per wave64 instruction, one wave on its SIMD, independent instructions) of the fp64 operations the row loops are
// made of - in particular the "quarter-rate" v_rcp_f64 / v_rsq_f64 against an f32 estimate + conversions.
//   hipcc --offload-arch=gfx950 -O2 scripts/probes/valu_issue_cost.hip -o build/probes/valu_issue_cost && build/probes/valu_issue_cost
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#define BODY(NAME, ASM, TYPE, INIT)                                                                                  \
    __global__ void NAME(long* cycles, TYPE* sink, int iters)                                                        \
    {                                                                                                                \
        TYPE a0 = INIT + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7; \
        TYPE b0 = a0, b1 = a1, b2 = a2, b3 = a3, b4 = a4, b5 = a5, b6 = a6, b7 = a7;                                  \
        const long t0 = clock64();                                                                                   \
        for (int i = 0; i < iters; ++i)                                                                              \
        {                                                                                                            \
            asm volatile(ASM(0) ASM(1) ASM(2) ASM(3) ASM(4) ASM(5) ASM(6) ASM(7)                                      \
                         ASM(0) ASM(1) ASM(2) ASM(3) ASM(4) ASM(5) ASM(6) ASM(7)                                      \
                         ASM(0) ASM(1) ASM(2) ASM(3) ASM(4) ASM(5) ASM(6) ASM(7)                                      \
                         ASM(0) ASM(1) ASM(2) ASM(3) ASM(4) ASM(5) ASM(6) ASM(7)                                      \
                         : "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3), "+v"(b4), "+v"(b5), "+v"(b6), "+v"(b7)            \
                         : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(a4), "v"(a5), "v"(a6), "v"(a7));                  \
        }                                                                                                            \
        const long t1 = clock64();                                                                                   \
        if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;                                                          \
        sink[threadIdx.x] = b0 + b1 + b2 + b3 + b4 + b5 + b6 + b7;                                                   \
    }
// operand numbering: %0..%7 = b (in/out), %8..%15 = a
#define A_FMA(i)   "v_fma_f64 %" #i ", %" #i ", %8, %9\n"
#define A_MUL(i)   "v_mul_f64 %" #i ", %" #i ", %8\n"
#define A_ADD(i)   "v_add_f64 %" #i ", %" #i ", %8\n"
#define A_MIN(i)   "v_min_f64 %" #i ", %" #i ", %8\n"
#define A_RCP(i)   "v_rcp_f64 %" #i ", %" #i "\n"
#define A_RSQ(i)   "v_rsq_f64 %" #i ", %" #i "\n"
#define A_SQRT(i)  "v_sqrt_f64 %" #i ", %" #i "\n"
#define A_RSQ32(i) "v_rsq_f32 %" #i ", %" #i "\n"
#define A_RCP32(i) "v_rcp_f32 %" #i ", %" #i "\n"
#define A_FMA32(i) "v_fma_f32 %" #i ", %" #i ", %8, %9\n"
#define A_MOV(i)   "v_mov_b32 %" #i ", %8\n"
#define A_DPP(i)   "v_mov_b32_dpp %" #i ", %8 wave_shr:1 row_mask:0xf bank_mask:0xf\n"
#define A_CNDMASK(i) "v_cndmask_b32 %" #i ", %8, %9, vcc\n"
BODY(k_fma_f64, A_FMA, double, 1.0)
BODY(k_mul_f64, A_MUL, double, 1.0)
BODY(k_add_f64, A_ADD, double, 1.0)
BODY(k_min_f64, A_MIN, double, 1.0)
BODY(k_rcp_f64, A_RCP, double, 1.5)
BODY(k_rsq_f64, A_RSQ, double, 1.5)
BODY(k_sqrt_f64, A_SQRT, double, 1.5)
BODY(k_rsq_f32, A_RSQ32, float, 1.5f)
BODY(k_rcp_f32, A_RCP32, float, 1.5f)
BODY(k_fma_f32, A_FMA32, float, 1.0f)
BODY(k_mov_b32, A_MOV, float, 1.0f)
BODY(k_dpp_b32, A_DPP, float, 1.0f)

// conversions need mixed register widths: written out
__global__ void k_cvt_f32_f64(long* cycles, float* sink, int iters)
{
    double a = 1.5 + threadIdx.x; float b0 = 0, b1 = 0, b2 = 0, b3 = 0;
    const long t0 = clock64();
    for (int i = 0; i < iters; ++i)
        asm volatile("v_cvt_f32_f64 %0, %4\nv_cvt_f32_f64 %1, %4\nv_cvt_f32_f64 %2, %4\nv_cvt_f32_f64 %3, %4\nv_cvt_f32_f64 %0, %4\nv_cvt_f32_f64 %1, %4\nv_cvt_f32_f64 %2, %4\nv_cvt_f32_f64 %3, %4\n"
                     "v_cvt_f32_f64 %0, %4\nv_cvt_f32_f64 %1, %4\nv_cvt_f32_f64 %2, %4\nv_cvt_f32_f64 %3, %4\nv_cvt_f32_f64 %0, %4\nv_cvt_f32_f64 %1, %4\nv_cvt_f32_f64 %2, %4\nv_cvt_f32_f64 %3, %4\n"
                     "v_cvt_f32_f64 %0, %4\nv_cvt_f32_f64 %1, %4\nv_cvt_f32_f64 %2, %4\nv_cvt_f32_f64 %3, %4\nv_cvt_f32_f64 %0, %4\nv_cvt_f32_f64 %1, %4\nv_cvt_f32_f64 %2, %4\nv_cvt_f32_f64 %3, %4\n"
                     "v_cvt_f32_f64 %0, %4\nv_cvt_f32_f64 %1, %4\nv_cvt_f32_f64 %2, %4\nv_cvt_f32_f64 %3, %4\nv_cvt_f32_f64 %0, %4\nv_cvt_f32_f64 %1, %4\nv_cvt_f32_f64 %2, %4\nv_cvt_f32_f64 %3, %4\n"
                     : "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3) : "v"(a));
    const long t1 = clock64();
    if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
    sink[threadIdx.x] = b0 + b1 + b2 + b3;
}
__global__ void k_cvt_f64_f32(long* cycles, double* sink, int iters)
{
    float a = 1.5f + threadIdx.x; double b0 = 0, b1 = 0, b2 = 0, b3 = 0;
    const long t0 = clock64();
    for (int i = 0; i < iters; ++i)
        asm volatile("v_cvt_f64_f32 %0, %4\nv_cvt_f64_f32 %1, %4\nv_cvt_f64_f32 %2, %4\nv_cvt_f64_f32 %3, %4\nv_cvt_f64_f32 %0, %4\nv_cvt_f64_f32 %1, %4\nv_cvt_f64_f32 %2, %4\nv_cvt_f64_f32 %3, %4\n"
                     "v_cvt_f64_f32 %0, %4\nv_cvt_f64_f32 %1, %4\nv_cvt_f64_f32 %2, %4\nv_cvt_f64_f32 %3, %4\nv_cvt_f64_f32 %0, %4\nv_cvt_f64_f32 %1, %4\nv_cvt_f64_f32 %2, %4\nv_cvt_f64_f32 %3, %4\n"
                     "v_cvt_f64_f32 %0, %4\nv_cvt_f64_f32 %1, %4\nv_cvt_f64_f32 %2, %4\nv_cvt_f64_f32 %3, %4\nv_cvt_f64_f32 %0, %4\nv_cvt_f64_f32 %1, %4\nv_cvt_f64_f32 %2, %4\nv_cvt_f64_f32 %3, %4\n"
                     "v_cvt_f64_f32 %0, %4\nv_cvt_f64_f32 %1, %4\nv_cvt_f64_f32 %2, %4\nv_cvt_f64_f32 %3, %4\nv_cvt_f64_f32 %0, %4\nv_cvt_f64_f32 %1, %4\nv_cvt_f64_f32 %2, %4\nv_cvt_f64_f32 %3, %4\n"
                     : "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3) : "v"(a));
    const long t1 = clock64();
    if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
    sink[threadIdx.x] = b0 + b1 + b2 + b3;
}

template<class K, class T> void run(const char* name, K kernel, T*, int waves_per_simd)
{
    long* d_c; T* d_s;
    const int iters = 2000, blocks = 1;
    hipMalloc(&d_c, sizeof(long) * 64); hipMalloc(&d_s, sizeof(T) * 1024);
    // waves_per_simd waves on ONE SIMD cannot be forced from here; a workgroup of 64 * 4 * w threads puts w waves on each SIMD of one CU
    const int threads = 64 * (waves_per_simd == 1 ? 1 : 4 * waves_per_simd);
    for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(kernel, dim3(blocks), dim3(threads > 1024 ? 1024 : threads), 0, 0, d_c, d_s, iters); hipDeviceSynchronize(); }
    long c = 0; hipMemcpy(&c, d_c, sizeof(long), hipMemcpyDeviceToHost);
    printf("%-16s %d wave(s)/SIMD: %7.2f clock64 ticks per instruction and wave\n", name, waves_per_simd, (double) c / (iters * 32.0));
    hipFree(d_c); hipFree(d_s);
}

int main()
{
    for (int w : {1, 2})
    {
        run("v_fma_f64", k_fma_f64, (double*) 0, w); run("v_mul_f64", k_mul_f64, (double*) 0, w); run("v_add_f64", k_add_f64, (double*) 0, w);
        run("v_min_f64", k_min_f64, (double*) 0, w); run("v_rcp_f64", k_rcp_f64, (double*) 0, w); run("v_rsq_f64", k_rsq_f64, (double*) 0, w);
        run("v_sqrt_f64", k_sqrt_f64, (double*) 0, w); run("v_rsq_f32", k_rsq_f32, (float*) 0, w); run("v_rcp_f32", k_rcp_f32, (float*) 0, w);
        run("v_fma_f32", k_fma_f32, (float*) 0, w); run("v_mov_b32", k_mov_b32, (float*) 0, w); run("v_mov_b32_dpp", k_dpp_b32, (float*) 0, w);
        run("v_cvt_f32_f64", k_cvt_f32_f64, (float*) 0, w); run("v_cvt_f64_f32", k_cvt_f64_f32, (double*) 0, w);
    }
    printf("(clock64 = s_memtime: ticks of the shader clock domain's counter; compare the rows with each other - v_fma_f64 is 4 cycles of its SIMD's 16 fp64 lanes)\n");
    return 0;
}
