// What keeps V_MFMA_I32_32X32X32_I8 from issuing back to back?  Variants of a 12-accumulator loop (the shape of the
// conv_rows.hip K-step: 6 MFMAs per k-half), 2 waves per SIMD, one workgroup of 512 threads per CU.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_issue.hip -o /tmp/mfma_issue && /tmp/mfma_issue
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

// MODE 0: same A/B registers, straight line.  1: distinct A/B registers per MFMA.  2: + wave-uniform branch after every
// 2 MFMAs (never taken).  3: + s_barrier once per 12 MFMAs.  4: branches taken for the last third on one wave in four.
template <int MODE>
__global__ __launch_bounds__(512, 2) void mfma_loop(int iters, int *out, int nact_in)
{
    v16i acc[2][3];
    for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 3; ++j)
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0;
    v4i a[2][2], b[2][3];
    for (int h = 0; h < 2; ++h) {
        for (int i = 0; i < 2; ++i) a[h][i] = v4i{(int)threadIdx.x + i, h, 2, 3};
        for (int j = 0; j < 3; ++j) b[h][j] = v4i{4, 5 + j, (int)blockIdx.x, 7 + h};
    }
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int nact = nact_in;  // 3 = all sub-tiles active
    if (MODE == 4 && (wave & 3) == 3) nact = 2;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                if (MODE >= 2 && j >= nact) continue;
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    if (MODE == 0) acc[i][j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[0][0], b[0][0], acc[i][j], 0, 0, 0);
                    else acc[i][j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[h][i], b[h][j], acc[i][j], 0, 0, 0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (MODE == 3) __builtin_amdgcn_s_barrier();
    }
    int s = 0;
    for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 3; ++j)
            for (int r = 0; r < 16; ++r) s += acc[i][j][r];
    if (s == 0x7fffffff) out[0] = s;
}

template <int MODE>
static void run(const char *name)
{
    int *out; hipMalloc(&out, 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 2000;
    hipLaunchKernelGGL(mfma_loop<MODE>, dim3(256), dim3(512), 0, 0, iters, out, 3);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(mfma_loop<MODE>, dim3(256), dim3(512), 0, 0, iters, out, 3);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double mfma_per_simd = (double)iters * 12 * 2;
    printf("%-64s %7.1f us  %5.2f ns per MFMA per SIMD (32 clk = %.2f ns at 2.4 GHz)\n", name, ms * 1e3, ms * 1e6 / mfma_per_simd, 32 / 2.4);
    hipFree(out);
}

int main()
{
    run<0>("same operand registers, straight line");
    run<1>("distinct A/B registers");
    run<2>("distinct registers + uniform branch every 2 MFMAs");
    run<3>("... + s_barrier every 12 MFMAs");
    run<4>("... one wave in four skips a third (SIMD 3 lighter)");
    return 0;
}
