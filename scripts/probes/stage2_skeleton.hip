// Probe: the memory skeleton of the second RK2 stage (read U1 with its row stencil, read U0 point-wise, write U; five planes per row,
// wave-marching over 32-row chunks) with 8-byte accesses in the row layout against 16-byte accesses in a row-PAIR-interleaved layout
// (two successive rows of a column adjacent in memory). No arithmetic beyond one add per value. What would the pair layout buy?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
constexpr int N = 4096, CHUNK = 32, STRIPS = (N + 59) / 60;

__global__ __launch_bounds__(256, 2) void rows8(const double* __restrict__ a, const double* __restrict__ b, double* __restrict__ c)
{
    const int w = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (w >= STRIPS * (N / CHUNK)) return;
    const int chunk = w / STRIPS, strip = w % STRIPS;
    const int col = min(max(strip * 60 - 2 + lane, 0), N - 1);
    const bool writes = lane >= 2 && lane < 62 && strip * 60 - 2 + lane < N;
    for (int r = chunk * CHUNK; r < (chunk + 1) * CHUNK; ++r)
        for (int q = 0; q < 5; ++q)
        {
            const size_t i = ((size_t) (r + 2) * 5 + q) * N + col;
            const double v = a[i] + b[i];
            if (writes) c[i] = v;
        }
}

__global__ __launch_bounds__(256, 2) void pairs16(const double2* __restrict__ a, const double2* __restrict__ b, double2* __restrict__ c)
{
    const int w = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (w >= STRIPS * (N / CHUNK)) return;
    const int chunk = w / STRIPS, strip = w % STRIPS;
    const int col = min(max(strip * 60 - 2 + lane, 0), N - 1);
    const bool writes = lane >= 2 && lane < 62 && strip * 60 - 2 + lane < N;
    for (int k = chunk * CHUNK / 2; k < (chunk + 1) * CHUNK / 2; ++k)          // row pairs
        for (int q = 0; q < 5; ++q)
        {
            const size_t i = ((size_t) (k + 1) * 5 + q) * N + col;
            const double2 x = a[i], y = b[i];
            if (writes) c[i] = make_double2(x.x + y.x, x.y + y.y);
        }
}

int main()
{
    const size_t doubles = (size_t) 5 * (N + 4) * N;
    double *a, *b, *c;
    CHECK(hipMalloc(&a, doubles * 8)); CHECK(hipMalloc(&b, doubles * 8)); CHECK(hipMalloc(&c, doubles * 8));
    CHECK(hipMemset(a, 0, doubles * 8)); CHECK(hipMemset(b, 0, doubles * 8));
    const int waves = STRIPS * (N / CHUNK), blocks = (waves + 3) / 4;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int variant = 0; variant < 2; ++variant)
        for (int rep = 0; rep < 3; ++rep)
        {
            for (int i = 0; i < 5; ++i) { if (variant) hipLaunchKernelGGL(pairs16, dim3(blocks), dim3(256), 0, 0, (double2*) a, (double2*) b, (double2*) c); else hipLaunchKernelGGL(rows8, dim3(blocks), dim3(256), 0, 0, a, b, c); }
            hipEventRecord(e0);
            for (int i = 0; i < 20; ++i) { if (variant) hipLaunchKernelGGL(pairs16, dim3(blocks), dim3(256), 0, 0, (double2*) a, (double2*) b, (double2*) c); else hipLaunchKernelGGL(rows8, dim3(blocks), dim3(256), 0, 0, a, b, c); }
            hipEventRecord(e1); CHECK(hipEventSynchronize(e1));
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double gb = 3.0 * 5 * N * (double) N * 8 / 1e9;
            printf("%s: %.4f ms per launch, %.2f TB/s of %.3f GB\n", variant ? "pairs16" : "rows8  ", ms / 20, gb / (ms / 20), gb);
        }
    return 0;
}
