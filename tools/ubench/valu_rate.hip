// VALU issue-rate microbenchmark for gfx950: cycles per wave64 instruction for the ops the requantise epilogue uses.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/valu_rate.hip -o /tmp/valu_rate && /tmp/valu_rate
// Each kernel runs 8 independent dependency chains of one instruction per lane, 4 waves per SIMD, so the figure is
// throughput (issue cycles per instruction per SIMD), not latency.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define CHAINS 8
#define UNROLL 8

#define BENCH_KERNEL(NAME, TYPE, INIT, ASM, CONSTRAINT)                                            \
    __global__ void k_##NAME(int iters, TYPE *out, long long *clk)                                 \
    {                                                                                              \
        TYPE v[CHAINS];                                                                            \
        for (int i = 0; i < CHAINS; ++i) v[i] = (TYPE)(INIT + threadIdx.x + i);                    \
        TYPE c = (TYPE)(out[1]);                                                                   \
        const long long t0 = clock64();                                                            \
        for (int it = 0; it < iters; ++it) {                                                       \
            _Pragma("unroll") for (int u = 0; u < UNROLL; ++u) {                                   \
                _Pragma("unroll") for (int i = 0; i < CHAINS; ++i)                                 \
                    asm volatile(ASM : "+" CONSTRAINT(v[i]) : CONSTRAINT(c));                      \
            }                                                                                      \
        }                                                                                          \
        const long long t1 = clock64();                                                            \
        TYPE s = 0;                                                                                \
        for (int i = 0; i < CHAINS; ++i) s += v[i];                                                \
        if (s == (TYPE)12345) out[0] = s;                                                          \
        if (threadIdx.x == 0 && blockIdx.x == 0) clk[0] = t1 - t0;                                 \
    }

BENCH_KERNEL(add_u32, unsigned, 1, "v_add_u32 %0, %0, %1", "v")
BENCH_KERNEL(mul_lo_u32, unsigned, 1, "v_mul_lo_u32 %0, %0, %1", "v")
BENCH_KERNEL(mul_hi_u32, unsigned, 1, "v_mul_hi_u32 %0, %0, %1", "v")
BENCH_KERNEL(mul_u32_u24, unsigned, 1, "v_mul_u32_u24 %0, %0, %1", "v")
BENCH_KERNEL(mad_u32_u24, unsigned, 1, "v_mad_u32_u24 %0, %0, %1, %0", "v")
BENCH_KERNEL(perm_b32, unsigned, 1, "v_perm_b32 %0, %0, %1, %0", "v")
BENCH_KERNEL(med3_i32, int, 1, "v_med3_i32 %0, %0, %1, %0", "v")
BENCH_KERNEL(cvt_f32_i32, unsigned, 1, "v_cvt_f32_i32 %0, %0", "v")
BENCH_KERNEL(cvt_i32_f32, unsigned, 1, "v_cvt_i32_f32 %0, %0", "v")
BENCH_KERNEL(mul_f32, float, 1, "v_mul_f32 %0, %0, %1", "v")
BENCH_KERNEL(fma_f32, float, 1, "v_fma_f32 %0, %0, %1, %0", "v")
BENCH_KERNEL(mul_f64, double, 1, "v_mul_f64 %0, %0, %1", "v")
BENCH_KERNEL(add_f64, double, 1, "v_add_f64 %0, %0, %1", "v")
BENCH_KERNEL(fma_f64, double, 1, "v_fma_f64 %0, %0, %1, %0", "v")
BENCH_KERNEL(trunc_f64, double, 1, "v_trunc_f64 %0, %0", "v")
BENCH_KERNEL(lshlrev_b64, unsigned long long, 1, "v_lshlrev_b64 %0, 1, %0", "v")
BENCH_KERNEL(dot4_i32_i8, int, 1, "v_dot4_i32_i8 %0, %0, %1, %0", "v")

BENCH_KERNEL(max_u32, unsigned, 1, "v_max_u32 %0, %0, %1", "v")
BENCH_KERNEL(max3_u32, unsigned, 1, "v_max3_u32 %0, %0, %1, %0", "v")
BENCH_KERNEL(ashrrev_i32, int, 1, "v_ashrrev_i32 %0, 3, %0", "v")
BENCH_KERNEL(lshlrev_b32, unsigned, 1, "v_lshlrev_b32 %0, 1, %0", "v")
BENCH_KERNEL(mul_hi_i32, int, 1, "v_mul_hi_i32 %0, %0, %1", "v")
__global__ void k_mad_u64_u32(int iters, unsigned long long *out, long long *clk)
{
    unsigned long long v[CHAINS];
    for (int i = 0; i < CHAINS; ++i) v[i] = threadIdx.x + i;
    const unsigned c = (unsigned)out[1] + 3u;
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < UNROLL; ++u)
#pragma unroll
            for (int i = 0; i < CHAINS; ++i) asm volatile("v_mad_u64_u32 %0, vcc, %1, %1, %0" : "+v"(v[i]) : "v"(c) : "vcc");
    }
    const long long t1 = clock64();
    unsigned long long s = 0;
    for (int i = 0; i < CHAINS; ++i) s += v[i];
    if (s == 12345ull) out[0] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) clk[0] = t1 - t0;
}
BENCH_KERNEL(sub_u32_clamp, unsigned, 1, "v_sub_u32_e64 %0, %0, %1 clamp", "v")
BENCH_KERNEL(or3_b32, unsigned, 1, "v_or3_b32 %0, %0, %1, %0", "v")
BENCH_KERNEL(xor_b32, unsigned, 1, "v_xor_b32 %0, %0, %1", "v")
BENCH_KERNEL(and_b32, unsigned, 1, "v_and_b32 %0, %0, %1", "v")
BENCH_KERNEL(add3_u32, unsigned, 1, "v_add3_u32 %0, %0, %1, %0", "v")
BENCH_KERNEL(lshl_add_u32, unsigned, 1, "v_lshl_add_u32 %0, %0, 2, %1", "v")
BENCH_KERNEL(mad_i32_i24, int, 1, "v_mad_i32_i24 %0, %0, %1, %0", "v")
BENCH_KERNEL(min_i32, int, 1, "v_min_i32 %0, %0, %1", "v")
BENCH_KERNEL(mov_b32, unsigned, 1, "v_mov_b32 %0, %1", "v")
BENCH_KERNEL(cndmask_b32, unsigned, 1, "v_cndmask_b32 %0, %0, %1, vcc", "v")
BENCH_KERNEL(cmp_gt_u32, unsigned, 1, "v_cmp_gt_u32 vcc, %0, %1", "v")
BENCH_KERNEL(bfe_u32, unsigned, 1, "v_bfe_u32 %0, %0, 3, 8", "v")
BENCH_KERNEL(ashrrev_i64, long long, 1, "v_ashrrev_i64 %0, 3, %0", "v")
BENCH_KERNEL(pk_max_i16, unsigned, 1, "v_pk_max_i16 %0, %0, %1", "v")

// conversions between 32- and 64-bit register operands need separate source/destination registers
__global__ void k_cvt_f64_i32(int iters, double *out, long long *clk)
{
    int v[CHAINS]; double d[CHAINS];
    for (int i = 0; i < CHAINS; ++i) v[i] = threadIdx.x + i;
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < UNROLL; ++u)
#pragma unroll
            for (int i = 0; i < CHAINS; ++i) asm volatile("v_cvt_f64_i32 %0, %1" : "=v"(d[i]) : "v"(v[i]));
    }
    const long long t1 = clock64();
    double s = 0;
    for (int i = 0; i < CHAINS; ++i) s += d[i];
    if (s == 12345.0) out[0] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) clk[0] = t1 - t0;
}
__global__ void k_cvt_i32_f64(int iters, double *out, long long *clk)
{
    int v[CHAINS]; double d[CHAINS];
    for (int i = 0; i < CHAINS; ++i) d[i] = threadIdx.x + i + out[1];
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < UNROLL; ++u)
#pragma unroll
            for (int i = 0; i < CHAINS; ++i) asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(v[i]) : "v"(d[i]));
    }
    const long long t1 = clock64();
    int s = 0;
    for (int i = 0; i < CHAINS; ++i) s += v[i];
    if (s == 12345) out[0] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) clk[0] = t1 - t0;
}

template <typename T, typename K>
static void run(const char *name, K kern)
{
    T *out; long long *clk;
    hipMalloc(&out, 64); hipMemset(out, 0, 64); hipMalloc(&clk, 16);
    const int iters = 2000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(256), dim3(1024), 0, 0, iters, out, clk);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(kern, dim3(256), dim3(1024), 0, 0, iters, out, clk);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long h; hipMemcpy(&h, clk, 8, hipMemcpyDeviceToHost);
    // 4 waves per SIMD (1024 threads / CU = 16 waves / 4 SIMDs)
    const double n = (double)iters * UNROLL * CHAINS * 4;
    printf("%-14s %6.2f clock64 ticks | %6.2f ns (event time) per wave64 instruction per SIMD\n", name, (double)h / n, ms * 1e6 / n);
    hipFree(out); hipFree(clk);
}

int main()
{
#define RUN(NAME, TYPE) run<TYPE>(#NAME, k_##NAME)
    RUN(add_u32, unsigned); RUN(mul_lo_u32, unsigned); RUN(mul_hi_u32, unsigned); RUN(mul_u32_u24, unsigned);
    RUN(mad_u32_u24, unsigned); RUN(perm_b32, unsigned); RUN(med3_i32, int); RUN(cvt_f32_i32, unsigned);
    RUN(cvt_i32_f32, unsigned); RUN(mul_f32, float); RUN(fma_f32, float); RUN(mul_f64, double); RUN(add_f64, double);
    RUN(fma_f64, double); RUN(trunc_f64, double); RUN(lshlrev_b64, unsigned long long); RUN(dot4_i32_i8, int);
    RUN(max_u32, unsigned); RUN(max3_u32, unsigned); RUN(ashrrev_i32, int); RUN(lshlrev_b32, unsigned); RUN(mul_hi_i32, int);
    RUN(mad_u64_u32, unsigned long long); RUN(sub_u32_clamp, unsigned); RUN(or3_b32, unsigned); RUN(xor_b32, unsigned); RUN(and_b32, unsigned);
    RUN(add3_u32, unsigned); RUN(lshl_add_u32, unsigned); RUN(mad_i32_i24, int); RUN(min_i32, int); RUN(mov_b32, unsigned);
    RUN(cndmask_b32, unsigned); RUN(cmp_gt_u32, unsigned); RUN(bfe_u32, unsigned); RUN(ashrrev_i64, long long); RUN(pk_max_i16, unsigned);
    run<double>("cvt_f64_i32", k_cvt_f64_i32);
    run<double>("cvt_i32_f64", k_cvt_i32_f64);
    return 0;
}
