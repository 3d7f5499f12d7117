// How long does a wave take to ISSUE global_load_lds (LDS-DMA) instructions on gfx950, and what does the CU sustain?
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/dma_issue.hip -o /tmp/dma_issue && /tmp/dma_issue
// Each wave issues `n` DMA instructions back to back from an L2-resident 64 KiB window and measures (a) the shader
// clocks until all are issued and (b) until all have landed (vmcnt(0)).
#include <hip/hip_runtime.h>
#include <cstdio>
#define DMA(gsrc, ldst, BYTES)                                                                    \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(gsrc),      \
                                     (__attribute__((address_space(3))) void *)(ldst), BYTES, 0, 0)

template <int BYTES>
__global__ void dma_loop(const char *src, int iters, long long *out)
{
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const char *p = src + wave * 8192 + lane * BYTES;
    char *dst = lds + wave * 8 * 1024;
    __syncthreads();
    long long issue = 0, land = 0;
    for (int it = 0; it < iters; ++it) {
        const long long t0 = __builtin_readcyclecounter();
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if constexpr (BYTES == 16) DMA(p + u * 64 * BYTES, dst + u * 64 * BYTES, 16);
            else DMA(p + u * 64 * BYTES, dst + u * 64 * BYTES, 4);
        }
        const long long t1 = __builtin_readcyclecounter();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const long long t2 = __builtin_readcyclecounter();
        issue += t1 - t0;
        land += t2 - t0;
    }
    if (lane == 0) { out[(blockIdx.x * 16 + wave) * 2] = issue; out[(blockIdx.x * 16 + wave) * 2 + 1] = land; }
}

template <int BYTES>
static void run(int waves, int blocks)
{
    char *src; long long *out;
    hipMalloc(&src, 1 << 20); hipMemset(src, 1, 1 << 20);
    hipMalloc(&out, blocks * 16 * 16); hipMemset(out, 0, blocks * 16 * 16);
    const int iters = 200;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(dma_loop<BYTES>, dim3(blocks), dim3(64 * waves), 128 * 1024, 0, src, iters, out);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(dma_loop<BYTES>, dim3(blocks), dim3(64 * waves), 128 * 1024, 0, src, iters, out);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long h[32]; hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
    const double n = (double)iters * 8;
    printf("%2d B/lane, %2d waves/CU, %3d CUs: issue %6.1f ticks/instr, issue+land %6.1f ticks/instr (wave 0) | %6.1f ns per instr per CU, %6.1f GB/s per CU\n",
           BYTES, waves, blocks, h[0] / n, h[1] / n, ms * 1e6 / (n * waves), 64.0 * BYTES * n * waves / (ms * 1e6));
    hipFree(src); hipFree(out);
}

int main()
{
    hipFuncSetAttribute((const void *)dma_loop<16>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    hipFuncSetAttribute((const void *)dma_loop<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    for (int blocks : {1, 256})
        for (int waves : {1, 2, 4, 8}) {
            run<16>(waves, blocks);
            run<4>(waves, blocks);
        }
    return 0;
}
