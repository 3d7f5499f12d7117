// Can the matrix pipe and the VALU of one SIMD be busy at the same time on gfx950?  The conv + pool kernels and
// conv_ws3 spend about as many VALU-issue clocks requantising as MFMA clocks multiplying; their run time is close to the
// SUM of the two.  One workgroup of 8 waves per CU (2 waves per SIMD), 256 workgroups.
//   mode 0: every wave: MFMA only (2 independent accumulators)                     -> t_m
//   mode 1: every wave: VALU only (the requantise mix: cvt, FP64 mul, cvt, int ops) -> t_v
//   mode 2: waves 0-3 MFMA (twice the count), waves 4-7 VALU (twice the count): the two kinds of wave share each SIMD
//   mode 3: as 2, the MFMA waves at s_setprio 3
//   mode 4: every wave alternates one MFMA and VPM VALU instructions in one instruction stream (same totals as 0 + 1)
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_valu_overlap.hip -o /tmp/ov && /tmp/ov
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

constexpr int VPM = 8;  // VALU instructions per MFMA in the per-wave totals (32 clk of VALU issue per 32-clk MFMA)

__device__ __forceinline__ void valu_block(int &x, double m)  // 8 VALU instructions, one dependency chain
{
    const double d = (double)x * m;        // cvt, mul
    int q = (int)d;                        // cvt
    const int p = max(q, 0);               // max
    const int nq = p - q;                  // sub
    const int t = __mul24(nq, -0xCCCD) + 0x12345;  // mad
    x = (int)(((unsigned)p << 19) + (unsigned)t) >> 19;  // lshl_add / add3, ashr
    x ^= q;                                // keep q alive
}

template <int MODE>
__global__ __launch_bounds__(512, 2) void k(int iters, int *out, double m)
{
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    v16i acc0, acc1;
    for (int r = 0; r < 16; ++r) { acc0[r] = 0; acc1[r] = 0; }
    v4i a = {(int)threadIdx.x, 1, 2, 3}, b = {4, 5, (int)blockIdx.x, 7};
    int x0 = threadIdx.x, x1 = threadIdx.x * 3, x2 = threadIdx.x * 5, x3 = threadIdx.x * 7;
    const bool mf = MODE == 0 || ((MODE == 2 || MODE == 3) && wave < 4);
    const bool va = MODE == 1 || ((MODE == 2 || MODE == 3) && wave >= 4);
    const int n = (MODE == 2 || MODE == 3) ? 2 * iters : iters;
    if (MODE == 4) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                acc0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, acc0, 0, 0, 0);
                valu_block(x0, m);
                acc1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, acc1, 0, 0, 0);
                valu_block(x1, m);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    } else if (mf) {
        if (MODE == 3) __builtin_amdgcn_s_setprio(3);
        for (int it = 0; it < n; ++it) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                acc0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, acc1, 0, 0, 0);
            }
        }
    } else if (va) {
        for (int it = 0; it < n; ++it) {
#pragma unroll
            for (int u = 0; u < 2; ++u) {  // 8 MFMA-equivalents: 8 x VPM VALU instructions, four independent chains
                valu_block(x0, m); valu_block(x1, m); valu_block(x2, m); valu_block(x3, m);
            }
        }
    }
    int s = x0 + x1 + x2 + x3;
    for (int r = 0; r < 16; ++r) s += acc0[r] + acc1[r];
    if (s == 0x7fffffff) out[0] = s;
}

template <int MODE>
static float run(const char *name, int iters)
{
    int *out; hipMalloc(&out, 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(512), 0, 0, iters, out, 0.37);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(512), 0, 0, iters, out, 0.37);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-78s %8.1f us\n", name, ms * 1e3);
    hipFree(out);
    return ms;
}

int main()
{
    const int iters = 4000;  // 8 MFMAs (+ 64 VALU) per iteration and wave
    const float tm = run<0>("mode 0: all 8 waves MFMA only (16 MFMAs per SIMD and iteration)", iters);
    const float tv = run<1>("mode 1: all 8 waves VALU only (128 VALU instructions per SIMD and iteration)", iters);
    const float t2 = run<2>("mode 2: per SIMD one MFMA wave + one VALU wave (same totals)", iters);
    const float t3 = run<3>("mode 3: as 2, MFMA waves at s_setprio 3", iters);
    const float t4 = run<4>("mode 4: every wave interleaves 1 MFMA : 8 VALU (same totals)", iters);
    printf("sum of the separate runs %.1f us, max %.1f us; overlap achieved: two kinds of wave %.0f %% (prio %.0f %%), one stream %.0f %%\n",
           (tm + tv) * 1e3, (tm > tv ? tm : tv) * 1e3, 100 * (tm + tv - t2) / (tm < tv ? tm : tv), 100 * (tm + tv - t3) / (tm < tv ? tm : tv),
           100 * (tm + tv - t4) / (tm < tv ? tm : tv));
    return 0;
}
