// Does the MFMA rate the chip sustains depend on the operand DATA?  V_MFMA_I32_32X32X32_I8 in a loop that only issues MFMAs
// (2 waves per SIMD, 6 independent accumulators per wave = the conv_rows K loop's shape), operands streamed from 8 register
// sets holding (a) zeros, (b) small values, (c) uniform random bytes.  Prints TOP/s and the shader clock (clock64 / wall clock).
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_data_power.hip -o /tmp/mfma_data_power && /tmp/mfma_data_power
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

__device__ __forceinline__ uint32_t mix(uint32_t x)
{
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}

// ORDER: which operand registers consecutive MFMAs name.  0: both operands change with every MFMA; 1: the 2 x 3 register tile of
// the conv_rows K loop in its order (a0b0 a1b0 a0b1 a1b1 a0b2 a1b2); 2: the same tile as a snake (a0b0 a0b1 a0b2 a1b2 a1b1 a1b0: one
// operand changes per MFMA)
template <int ORDER>
__global__ __launch_bounds__(512) void mfma_stream(int iters, int mode, int *out, long long *clk)
{
    v16i acc[6];
    for (int i = 0; i < 6; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = 0;
    v4i a[8], b[8];
    for (int s = 0; s < 8; ++s)
        for (int k = 0; k < 4; ++k) {
            const uint32_t ra = mix(threadIdx.x * 64 + s * 8 + k + blockIdx.x * 77777), rb = mix(ra + 12345);
            a[s][k] = mode == 0 ? 0 : mode == 1 ? (int)(ra & 0x03030303u) : (int)ra;
            b[s][k] = mode == 0 ? 0 : mode == 1 ? (int)(rb & 0x03030303u) : (int)rb;
        }
    const long long t0 = clock64();
    const long long w0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int s = 0; s < 8; ++s) {
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                constexpr int AO[3][6] = {{0, 1, 2, 3, 4, 5}, {0, 1, 0, 1, 0, 1}, {0, 0, 0, 1, 1, 1}};
                constexpr int BO[3][6] = {{1, 3, 5, 7, 9, 11}, {0, 0, 1, 1, 2, 2}, {0, 1, 2, 2, 1, 0}};
                constexpr int AC[3][6] = {{0, 1, 2, 3, 4, 5}, {0, 1, 2, 3, 4, 5}, {0, 2, 4, 5, 3, 1}};  // accumulator of (a, b): the same tile either way
                acc[AC[ORDER][i]] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[(2 * s + AO[ORDER][i]) & 7], b[(3 * s + BO[ORDER][i]) & 7], acc[AC[ORDER][i]], 0, 0, 0);
            }
        }
    }
    const long long t1 = clock64();
    const long long w1 = wall_clock64();
    int sum = 0;
    for (int i = 0; i < 6; ++i)
        for (int r = 0; r < 16; ++r) sum += acc[i][r];
    if (sum == 0x7fffffff) out[0] = sum;
    if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = t1 - t0; clk[1] = w1 - w0; }
}

// the same stream with V_MFMA_I32_16X16X64_I8 (half the operations per instruction, a quarter of the accumulator registers)
__global__ __launch_bounds__(512) void mfma_stream_16(int iters, int mode, int *out, long long *clk)
{
    v4i acc[6];
    for (int i = 0; i < 6; ++i) acc[i] = v4i{0, 0, 0, 0};
    v4i a[8], b[8];
    for (int s = 0; s < 8; ++s)
        for (int k = 0; k < 4; ++k) {
            const uint32_t ra = mix(threadIdx.x * 64 + s * 8 + k + blockIdx.x * 77777), rb = mix(ra + 12345);
            a[s][k] = mode == 0 ? 0 : (int)ra;
            b[s][k] = mode == 0 ? 0 : (int)rb;
        }
    const long long t0 = clock64();
    const long long w0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int s = 0; s < 8; ++s) {
#pragma unroll
            for (int i = 0; i < 6; ++i) acc[i] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[(s + i) & 7], b[(s + 2 * i + 1) & 7], acc[i], 0, 0, 0);
        }
    }
    const long long t1 = clock64();
    const long long w1 = wall_clock64();
    int sum = 0;
    for (int i = 0; i < 6; ++i)
        for (int r = 0; r < 4; ++r) sum += acc[i][r];
    if (sum == 0x7fffffff) out[0] = sum;
    if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = t1 - t0; clk[1] = w1 - w0; }
}

int main()
{
    int *out; long long *clk;
    hipMalloc(&out, 4); hipMalloc(&clk, 16);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const char *names[3] = {"zero operands", "2-bit operands", "uniform random bytes"};
    const int iters = 20000;  // x 48 MFMAs per wave: ~20 ms per launch, long enough for the power management to settle
    for (int rep = 0; rep < 2; ++rep)
        for (int mode = 0; mode < 3; ++mode) {
            hipEventRecord(e0, 0);
            hipLaunchKernelGGL(mfma_stream<0>, dim3(256), dim3(512), 0, 0, iters, mode, out, clk);
            hipEventRecord(e1, 0);
            hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            long long h[2]; hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
            const double ops = 2.0 * 32 * 32 * 32 * 48.0 * iters * 8 * 256;
            printf("%-22s %8.2f ms  %6.0f TOP/s  shader clock %4.0f MHz\n", names[mode], ms, ops / (ms * 1e-3) / 1e12,
                   (double)h[0] / ((double)h[1] / 100.0));
        }
    for (int rep = 0; rep < 2; ++rep)
        for (int order = 1; order < 3; ++order) {  // random bytes, the K loop's register tile in two issue orders
            hipEventRecord(e0, 0);
            if (order == 1) hipLaunchKernelGGL(mfma_stream<1>, dim3(256), dim3(512), 0, 0, iters, 2, out, clk);
            else hipLaunchKernelGGL(mfma_stream<2>, dim3(256), dim3(512), 0, 0, iters, 2, out, clk);
            hipEventRecord(e1, 0);
            hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            long long h[2]; hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
            const double ops = 2.0 * 32 * 32 * 32 * 48.0 * iters * 8 * 256;
            printf("random bytes, %-26s %8.2f ms  %6.0f TOP/s  shader clock %4.0f MHz\n", order == 1 ? "2x3 tile, K-loop order" : "2x3 tile, snake order", ms,
                   ops / (ms * 1e-3) / 1e12, (double)h[0] / ((double)h[1] / 100.0));
        }
    for (int rep = 0; rep < 2; ++rep)
        for (int mode = 0; mode < 3; mode += 2) {
            hipEventRecord(e0, 0);
            hipLaunchKernelGGL(mfma_stream_16, dim3(256), dim3(512), 0, 0, 2 * iters, mode, out, clk);
            hipEventRecord(e1, 0);
            hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            long long h[2]; hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
            const double ops = 2.0 * 16 * 16 * 64 * 48.0 * (2 * iters) * 8 * 256;
            printf("16x16x64, %-30s %8.2f ms  %6.0f TOP/s  shader clock %4.0f MHz\n", mode == 0 ? "zero operands" : "uniform random bytes", ms,
                   ops / (ms * 1e-3) / 1e12, (double)h[0] / ((double)h[1] / 100.0));
        }
    return 0;
}
