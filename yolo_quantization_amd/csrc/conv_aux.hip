// conv_aux.hip -- the two non-MFMA convolution kernels:
//
//  * conv_first_u8_kernel: the network's first layer (3 input channels, K = 27).  HBM-bound (AI ~45 op/B,
//    SURVEY.md 8): one thread per output pixel, v_dot4_u32_u8 on (c0,c1,c2,0) dwords, weights broadcast from LDS.
//    Same math as ref src/convolutional_layer.c:718-721: acc = sum w_u8*x_u8 - zp_w * sum x_u8.
//  * conv_ref_f32_kernel: bit-faithful emulation of the Makefile-default reference accumulation
//    (ref src/gemm.c:279-299: `C += ALPHA*A*B` with float ALPHA on an int32 C, i.e. a sequential fp32 add per k,
//    pass 1 with the weights then pass 2 with -zp_w; k order (ci,ky,kx) of ref src/im2col.c:33-37).
//    One thread per output element; a verification kernel, never on the throughput path.
#include "kargs.h"


__global__ __launch_bounds__(256) void conv_first_u8_kernel(const AuxArgs a)
{
    extern __shared__ uint32_t wl[];  // [n][9] weights then [n] zp_w
    for (int i = threadIdx.x; i < a.n * 9; i += blockDim.x) wl[i] = a.wfirst[i];
    __syncthreads();
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= a.total_n) return;
    const int hw = a.H * a.W, W1 = a.W + 1;
    const int b = n / hw, rem = n - b * hw;
    const int y = rem / a.W, xx = rem - y * a.W;
    const int cell = a.in_lead + (b * (a.H + 1) + (y + 1)) * W1 + xx;
    const uint32_t *xc = reinterpret_cast<const uint32_t *>(a.x);
    uint32_t xin[9];
    uint32_t sumx = 0;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const int dy = t / 3 - 1, dx = t % 3 - 1;
        xin[t] = xc[cell + dy * W1 + dx];
        sumx = __builtin_amdgcn_udot4(xin[t], 0x00010101u, sumx, false);
    }
    const int ocell = a.out_lead + (b * (a.H + 1) + (y + 1)) * W1 + xx;
    for (int oc0 = 0; oc0 < a.n; oc0 += 4) {
        uint32_t packed = 0;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int oc = oc0 + r;
            if (oc >= a.n) break;
            uint32_t s1 = 0;
#pragma unroll
            for (int t = 0; t < 9; ++t) s1 = __builtin_amdgcn_udot4(wl[oc * 9 + t], xin[t], s1, false);
            const int zpw = 128 - a.dzp[oc];
            const int32_t accv = (int32_t)s1 - zpw * (int32_t)sumx;
            const uint32_t u8 = requant_u8(accv, a.bias[oc], a.mval[oc], a.sval[oc], a.zp_act, a.act, a.store_mode);
            packed |= (u8 ^ 0x80u) << (8 * r);
            const size_t ridx = ((size_t)b * a.n + oc) * hw + rem;
            if (a.acc_out) a.acc_out[ridx] = accv;
            if (a.y_f32) a.y_f32[ridx] = (float)((int)u8 - a.zp_act) * a.s_act;
        }
        if (a.y) *reinterpret_cast<uint32_t *>(a.y + (size_t)ocell * a.out_cs + oc0) = packed;
    }
}

__global__ __launch_bounds__(256) void conv_ref_f32_kernel(const AuxArgs a)
{
    const int hw = a.H * a.W;
    const long total = (long)a.B * a.n * hw;
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    // pixel fastest so that a wave reads neighbouring cells
    const int rem = (int)(idx % hw);
    const int oc = (int)((idx / hw) % a.n);
    const int b = (int)(idx / ((long)hw * a.n));
    const int y = rem / a.W, xx = rem - y * a.W;
    const int W1 = a.W + 1;
    const int K = a.c * a.ksize * a.ksize;
    const uint8_t *wrow = a.w_u8 + (size_t)oc * K;
    const float zpw = (float)a.zp_w[oc];
    const bool plain = a.in_cs == 4;
    int32_t C = 0;
    for (int pass = 0; pass < 2; ++pass) {
        int k = 0;
        for (int ci = 0; ci < a.c; ++ci)
            for (int ky = 0; ky < a.ksize; ++ky)
                for (int kx = 0; kx < a.ksize; ++kx, ++k) {
                    const int iy = y + ky - a.pad, ix = xx + kx - a.pad;
                    int xv;
                    if (iy < 0 || ix < 0 || iy >= a.H || ix >= a.W) {
                        xv = a.zp_in;  // ref src/im2col.c:10-11
                    } else {
                        const int cell = a.in_lead + (b * (a.H + 1) + (iy + 1)) * W1 + ix;
                        const uint8_t raw = a.x[(size_t)cell * a.in_cs + ci];
                        xv = plain ? raw : (raw ^ 0x80);
                    }
                    // ref src/gemm.c:295  C[i*ldc+j] += ALPHA*A[i*lda+k]*B[k*ldb+j]  (float ALPHA = +1 / -1)
                    const float av = pass == 0 ? (float)wrow[k] : -zpw;
                    const float p = av * (float)xv;
                    C = (int32_t)((float)C + p);
                }
    }
    const uint32_t u8 = requant_u8(C, a.bias[oc], a.mval[oc], a.sval[oc], a.zp_act, a.act, a.store_mode);
    const size_t ridx = ((size_t)b * a.n + oc) * hw + rem;
    if (a.acc_out) a.acc_out[ridx] = C;
    if (a.y_f32) a.y_f32[ridx] = (float)((int)u8 - a.zp_act) * a.s_act;
    if (a.y) {
        const int ocell = a.out_lead + (b * (a.H + 1) + (y + 1)) * W1 + xx;
        a.y[(size_t)ocell * a.out_cs + oc] = (uint8_t)(u8 ^ 0x80u);
    }
}

// First layer fused with the 2x2 / stride-2 maxpool that follows it in every yolo cfg: one thread per POOLED pixel
// computes the four pre-pool pixels (4x4 input window), requantises them (compile-time activation / store mode,
// folded multiplier) and writes the bytewise max.  Removes the 2.77 MB/image pre-pool write + re-read.  The pre-pool
// tensor is still written when `a.y` is given (parity runs keep the reference's per-layer tensors).
template <int ACT, bool SAT>
__global__ __launch_bounds__(256) void conv_first_pool_u8_kernel(const AuxArgs a)
{
    extern __shared__ uint32_t wl[];  // [n][9] weights (c0,c1,c2,0) per tap, then [n] lo, [n] hi wrap-safe ranges
    int32_t *slo = reinterpret_cast<int32_t *>(wl + a.n * 9), *shi = slo + a.n;
    for (int i = threadIdx.x; i < a.n * 9; i += blockDim.x) wl[i] = a.wfirst[i];
    // Max-pool commutes with the requantisation while no byte of the window wraps (see conv_small.hip): [lo, hi] is
    // the per-channel range of accumulators (bias included) that cannot wrap; pre-pool stores keep the plain path.
    const bool commute = !a.y && a.hdr->pow2 == 1;
    if (threadIdx.x < a.n) {
        int32_t lo = -2147483647 - 1, hi = 2147483647;
        if (!SAT) small_safe_range<ACT>(a.mprime[threadIdx.x], a.zp_act, lo, hi);
        slo[threadIdx.x] = lo;
        shi[threadIdx.x] = hi;
    }
    __syncthreads();
    const int OH = a.H >> 1, OW = a.W >> 1;
    const int total = a.B * OH * OW;
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int b = idx / (OH * OW), rem = idx - b * (OH * OW);
    const int oy = rem / OW, ox = rem - oy * OW;
    const int W1 = a.W + 1;
    const int cell00 = a.in_lead + (b * (a.H + 1) + (2 * oy + 1)) * W1 + 2 * ox;  // pre-pool pixel (2oy, 2ox)
    const uint32_t *xc = reinterpret_cast<const uint32_t *>(a.x);
    uint32_t xin[4][4];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c) xin[r][c] = xc[cell00 + (r - 1) * W1 + (c - 1)];
    int32_t sumx[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        uint32_t t = 0;
#pragma unroll
        for (int k = 0; k < 9; ++k) t = __builtin_amdgcn_udot4(xin[(p >> 1) + k / 3][(p & 1) + k % 3], 0x00010101u, t, false);
        sumx[p] = (int32_t)t;
    }
    const size_t pcell = a.pool_lead + ((size_t)b * (OH + 1) + (oy + 1)) * (OW + 1) + ox;
    for (int oc0 = 0; oc0 < a.n; oc0 += 4) {  // n % 4 == 0 (checked by the launcher)
        int32_t accb[4][4];
        int32_t amax[4][1];
        double mp[4];
        bool bad = false;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int oc = oc0 + r;
            const int zpw = 128 - a.dzp[oc];
            const int bias = a.bias[oc];
            mp[r] = a.mprime[oc];
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                uint32_t s1 = 0;
#pragma unroll
                for (int k = 0; k < 9; ++k)
                    s1 = __builtin_amdgcn_udot4(wl[oc * 9 + k], xin[(p >> 1) + k / 3][(p & 1) + k % 3], s1, false);
                accb[r][p] = (int32_t)s1 - zpw * sumx[p] + bias;
            }
            const int32_t mx = max(max(accb[r][0], accb[r][1]), max(accb[r][2], accb[r][3]));
            const int32_t mn = min(min(accb[r][0], accb[r][1]), min(accb[r][2], accb[r][3]));
            bad |= (mx > shi[oc]) | (mn < slo[oc]);
            amax[r][0] = mx;
        }
        int32_t m[4];
        if (commute && __builtin_amdgcn_ballot_w64(bad) == 0) {  // no window of this wave wraps: requantise the maxima only
            int32_t v1[4][1];
            requant_values<ACT, SAT, 1>(amax, mp, a.zp_act, v1);
#pragma unroll
            for (int r = 0; r < 4; ++r) m[r] = v1[r][0];
        } else {
            int32_t v[4][4];
            if (a.hdr->pow2 == 1) {
                requant_values<ACT, SAT, 4>(accb, mp, a.zp_act, v);
            } else {  // shift_value not a power of two: the reference's two-step form (never produced by its own prep)
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int p = 0; p < 4; ++p)
                        v[r][p] = (int32_t)requant_u8(accb[r][p], 0, a.mval[oc0 + r], a.sval[oc0 + r], a.zp_act, ACT,
                                                      SAT ? MI355_STORE_SATURATE : MI355_STORE_WRAP);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r)  // uint8 wrap first, then the unsigned max of the window
                m[r] = max(max(v[r][0] & 0xFF, v[r][1] & 0xFF), max(v[r][2] & 0xFF, v[r][3] & 0xFF));
            if (a.y) {
#pragma unroll
                for (int p = 0; p < 4; ++p) {
                    const size_t ocell = a.out_lead + ((size_t)b * (a.H + 1) + (2 * oy + (p >> 1) + 1)) * W1 + 2 * ox + (p & 1);
                    *reinterpret_cast<uint32_t *>(a.y + ocell * a.out_cs + oc0) = pack4_biased(v[0][p], v[1][p], v[2][p], v[3][p]);
                }
            }
        }
        *reinterpret_cast<uint32_t *>(a.ypool + pcell * a.pool_cs + oc0) = pack4_biased(m[0], m[1], m[2], m[3]);
    }
}

template <int ACT>
static int conv_first_pool_launch_act(AuxArgs &a, hipStream_t st)
{
    const int bs = 256;
    const int total = a.B * (a.H / 2) * (a.W / 2);
    const int grid = (total + bs - 1) / bs;
    const size_t lds = a.n * 11 * sizeof(uint32_t);
    if (a.store_mode == MI355_STORE_SATURATE)
        hipLaunchKernelGGL((conv_first_pool_u8_kernel<ACT, true>), dim3(grid), dim3(bs), lds, st, a);
    else
        hipLaunchKernelGGL((conv_first_pool_u8_kernel<ACT, false>), dim3(grid), dim3(bs), lds, st, a);
    return hipGetLastError() == hipSuccess ? MI355_OK : MI355_EHIP;
}

// returns MI355_EINVAL when the fused form does not apply (caller runs the two layers separately)
int conv_first_pool_launch(AuxArgs &a, hipStream_t st)
{
    if ((a.H & 1) || (a.W & 1) || (a.n & 3) || a.acc_out || a.y_f32) return MI355_EINVAL;
    if (a.act == MI355_ACT_LEAKY) return conv_first_pool_launch_act<MI355_ACT_LEAKY>(a, st);
    if (a.act == MI355_ACT_RELU6) return conv_first_pool_launch_act<MI355_ACT_RELU6>(a, st);
    return conv_first_pool_launch_act<MI355_ACT_LINEAR>(a, st);
}

int conv_first_launch(AuxArgs &a, hipStream_t st)
{
    const int bs = 256;
    const int grid = (a.total_n + bs - 1) / bs;
    hipLaunchKernelGGL(conv_first_u8_kernel, dim3(grid), dim3(bs), a.n * 9 * sizeof(uint32_t), st, a);
    return hipGetLastError() == hipSuccess ? MI355_OK : MI355_EHIP;
}

int conv_ref_f32_launch(AuxArgs &a, hipStream_t st)
{
    const int bs = 256;
    const long total = (long)a.B * a.n * a.H * a.W;
    const long grid = (total + bs - 1) / bs;
    hipLaunchKernelGGL(conv_ref_f32_kernel, dim3((unsigned)grid), dim3(bs), 0, st, a);
    return hipGetLastError() == hipSuccess ? MI355_OK : MI355_EHIP;
}
