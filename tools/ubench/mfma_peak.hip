// MFMA issue-rate microbenchmark for gfx950: V_MFMA_I32_32X32X32_I8 throughput per SIMD as a function of resident
// waves per SIMD and of independent accumulators per wave, plus the shader clock actually sustained.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_peak.hip -o /tmp/mfma_peak && /tmp/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ void mfma_loop(int iters, int *out, long long *clk)
{
    v16i acc[NACC];
    for (int i = 0; i < NACC; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = 0;
    v4i a = {(int)threadIdx.x, 1, 2, 3}, b = {4, 5, (int)blockIdx.x, 7};
    const long long t0 = clock64();
    const long long w0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, acc[i], 0, 0, 0);
    }
    const long long t1 = clock64();
    const long long w1 = wall_clock64();
    int s = 0;
    for (int i = 0; i < NACC; ++i)
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    if (s == 0x7fffffff) out[0] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = t1 - t0; clk[1] = w1 - w0; }
}

template <int NACC>
static void run(int waves_per_simd, int iters)
{
    int *out; long long *clk;
    hipMalloc(&out, 4); hipMalloc(&clk, 16);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int threads = 64 * 4 * waves_per_simd;  // one workgroup per CU
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(mfma_loop<NACC>, dim3(256), dim3(threads), 0, 0, iters, out, clk);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
    }
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long h[2]; hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
    const double ops = 2.0 * 32 * 32 * 32 * (double)NACC * iters * (threads / 64) * 256;
    printf("acc/wave %d waves/SIMD %d: %.1f us  %.0f TOP/s | clock64 %lld ticks, wall %lld ticks (100 MHz) -> clock64 rate %.0f MHz, "
           "%.1f clock64-ticks per MFMA per SIMD\n",
           NACC, waves_per_simd, ms * 1e3, ops / (ms * 1e-3) / 1e12, h[0], h[1], (double)h[0] / ((double)h[1] / 100.0),
           (double)h[0] / ((double)NACC * iters * waves_per_simd));
}

int main()
{
    for (int w = 1; w <= 4; w *= 2) {
        run<1>(w, 20000);
        run<2>(w, 10000);
        run<4>(w, 5000);
        run<6>(w, 4000);
        run<12>(w, 2000);
    }
    return 0;
}
