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
#include <type_traits>


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

// ---------------------------------------------------------------------------------------------------------------
// First layer (3 input channels, 4-byte cells) fused with its 2x2 / stride-2 maxpool on the matrix pipe.
//
// One V_MFMA_I32_16X16X64_I8 covers a whole 3x3x3 receptive field: k-group g = lane >> 4 (16 bytes) is image row
// dy = g of the window -- the four cells x-1 .. x+2, i.e. 16 CONTIGUOUS bytes of the row image; the fourth cell and
// every cell's pad byte meet zero weights, k-group 3 is all zero weights.  Column lane & 15 of the B operand is a
// pooled pixel, the four MFMAs of a set are the four positions of its 2x2 window, so the window of every
// (pixel, channel) ends up in one lane (see conv_small.hip for the pooling / requantisation argument: the maximum
// accumulator is requantised once unless a byte of the window could wrap).  The signed-operand correction
// (128 - zp_w) * sum(x') is a second MFMA with the constant dz in every real k slot -- two when dz = 128 does not fit
// an int8 -- instead of per-pixel sums on the VALU.  Weights, corrections and per-channel constants live in registers;
// a workgroup walks 8 x 16 pooled patches persistently, staging the next 18 x 34 cell image (biased to signed bytes)
// through registers into a double-buffered LDS plane.  Result bytes are identical to conv_first_pool_u8_kernel's.
// ---------------------------------------------------------------------------------------------------------------
// cells per LDS image row of the first-layer MFMA kernels: 34 (x = 32 tx - 1 .. + 32) from the 4-byte-cell tensor; 42 when
// the image is read from the reference's colour planes in place: whole 4-pixel groups x = 32 tx - 4 .. + 35 stored from
// column 1, so that x = 32 tx - 1 sits at the EVEN column 4 and a lane's five-cell reads stay 8-byte aligned (ds_read_b64)
// (round 5, measured and not kept: a row pitch of 96 dwords for the planar form.  The B reads are ds_read_b64 -- sixteen lanes x 8 contiguous bytes per
// cycle = all 32 banks once, whatever the pitch; the kernel's bank conflicts (SQ_LDS_BANK_CONFLICT 1.15 M of 5.9 M LDS cycles per launch) are the
// staging's ds_write2_b32 pairs (lanes 16 bytes apart, two dwords each: sixteen of the 32 banks, twice) and the byte-table reads, and with a pitch
// that is a multiple of 32 dwords the staged rows all start in the same bank as well: 2.8 M conflict cycles, 32.9 -> 36 us.)
__host__ __device__ constexpr int first_stage_rowc(bool planar) { return planar ? 42 : 34; }

#ifdef MI355_ABLATE
// per-wave shader-clock sums of the first-layer kernel's phases (tools/l0_phases.py): [0] barrier, [1] deferred stores + prefetch issue,
// [2] / [4] B reads + MFMA chain of pooled row 0 / 1, [3] / [5] their epilogues, [6] staging wait + LDS writes, [7] tiles
__device__ long long g_l0_ph[4096][4][12];
#define L0P_ENTRY const long long l0t_entry = __builtin_readcyclecounter()
#define L0P_DECL long long l0p[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}; long long l0t = __builtin_readcyclecounter(); l0p[10] = l0t - l0t_entry
#define L0P_MARK(k)                                                  \
    do {                                                             \
        asm volatile("" ::: "memory");                               \
        const long long n_ = __builtin_readcyclecounter();           \
        l0p[k] += n_ - l0t;                                          \
        l0t = n_;                                                    \
    } while (0)
#define L0P_MARK_V(k, v)                                             \
    do {                                                             \
        asm volatile("s_nop 0" ::"v"(v));                            \
        L0P_MARK(k);                                                 \
    } while (0)
#define L0P_STORE()                                                                                  \
    do {                                                                                             \
        if ((threadIdx.x & 63) == 0 && blockIdx.x < 4096)                                            \
            for (int k = 0; k < 12; ++k) g_l0_ph[blockIdx.x][threadIdx.x >> 6][k] = k == 11 ? (long long)__builtin_readcyclecounter() - l0t_entry : l0p[k]; \
    } while (0)
extern "C" int mi355_debug_read_l0ph(long long *host)
{
    return hipMemcpyFromSymbol(host, HIP_SYMBOL(g_l0_ph), sizeof(long long) * 4096 * 4 * 12) == hipSuccess ? 0 : -5;
}
#else
#define L0P_DECL do { } while (0)
#define L0P_ENTRY do { } while (0)
#define L0P_MARK(k) do { } while (0)
#define L0P_MARK_V(k, v) do { } while (0)
#define L0P_STORE() do { } while (0)
#endif

template <int ACT, bool SAT, int NM, bool PLANAR>
__global__ __launch_bounds__(256, 4) void conv_first_mfma_pool_kernel(const AuxArgs a)
{
    L0P_ENTRY;
    constexpr int ROWC = first_stage_rowc(PLANAR), XO = PLANAR ? 4 : 0;
    __shared__ __attribute__((aligned(16))) uint32_t img[2][18 * ROWC];
    // LEAKY, wrapping store: windows inside the safe range take activation + zero point + bias from a byte table (common.h)
    constexpr bool LUT = ACT == MI355_ACT_LEAKY && !SAT;
    __shared__ __attribute__((aligned(16))) uint8_t lut[LUT ? LUTQ_N : 16];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave-uniform: tile rows and output rows stay on the scalar unit
    const int pc = lane & 15, g = lane >> 4;
    const unsigned pc_off = (unsigned)pc * (unsigned)a.pool_cs;  // byte offset of the lane's pooled column inside an output row
    const int W1 = a.W + 1;
    const int OH = a.H >> 1, OW = a.W >> 1;
    const int tiles_x = (OW + 15) >> 4, tiles_y = (OH + 7) >> 3, tpi = tiles_x * tiles_y;
    const int ntiles = a.B * tpi;
    // ---- per-lane constants: A fragments (row = channel 16*mt + pc, k-group g), channel parameters of the lane's four
    //      accumulator rows 16*mt + 4*g + r
    // Two A fragments per m-tile: window column jx = 0 reads image cells x-1 .. x+1 = cells 0 .. 2 of the lane's four-cell group, column
    // jx = 1 cells 1 .. 3 of the SAME group -- so the weights, not the B operand, are shifted by one cell: both window columns use one
    // aligned four-register B operand per image row (round 4; shifting the operand cost five-cell reads and seven v_mov per unit).
    v4i wa[NM][2], wd1[NM][2], wd2[NM][2], cb[NM], lo[NM], hi[NM];
    int chq[NM];
    double mp[NM][4];
    // integer requantisation of the window maxima (common.h intrq_make): multiplier M0 and shift s - 1 per channel, valid for the
    // whole launch only if EVERY channel's wrap-safe range passes the exactness conditions (workgroup-uniform flag)
    constexpr bool INTRQ = LUT || ACT == MI355_ACT_RELU6;
    int32_t qm0[NM][4], qsh[NM][4];
    int64_t qc[NM][4];  // lo * M0: the biased accumulator goes straight into one 64-bit multiply-add (u * M0 + lo * M0 = a * M0)
#pragma unroll
    for (int mt = 0; mt < NM; ++mt) chq[mt] = 16 * mt + 4 * g;
    // Round 5: the kernel starts 1 024 workgroups at once and nothing runs beside their prologues -- deriving the wrap-safe ranges and the
    // integer multipliers here (FP64 divisions, verification loops), building the byte table and gathering weights / zero points / multipliers
    // with dependent loads was 26-36 % of the kernel (tools/l0_phases.py).  mi355_conv_pack_epilogue leaves the whole per-lane state in the
    // blob (common.h L0Lane, EptHeader): fifteen independent 16-byte loads per m-tile + four table dwords behind the key test (issued with the
    // first tile's image already on its way); a blob without the table (or packed for another activation / zero point, or a saturating store) takes the derivation below.
    const bool have_ept = !SAT && a.ept != nullptr;  // launch-uniform (kernel argument)
    const int mpad4 = (a.n + 3) & ~3;                // the first-layer blob's mpad (shim.hip blob_layout)
    uint32_t ekey = 0, eflags = 0;
    if (have_ept) {
        ekey = a.ept->key;
        eflags = a.ept->flags;
    }
    // ---- staging: thread t owns image dwords t, t + 256, t + 512 (< 612): their cell offsets from the tile origin
    int soff[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int slot = min(tid + 256 * k, 18 * 34 - 1);
        const int r = slot / 34, c = slot - r * 34;
        soff[k] = r * W1 + c;
    }
    const uint32_t *xc = reinterpret_cast<const uint32_t *>(a.x);
    // PLANAR (the reference's [B][3][H][W] uint8 planes read in place: no layout conversion pass): thread t < 180 owns image
    // row t / 10 and the 4-pixel group t % 10 of the 40-cell LDS row (image x = 32 tx - 4 + 4 q .. + 3): one aligned dword
    // from each colour plane, interleaved into four (c0, c1, c2, 0) cells on the way into LDS.  W % 4 == 0, so a group is
    // either inside the image or entirely pad (cells of the input zero point, ref: src/convolutional_layer.c:703-705).
    const int prow_ix = tid / 10, pq = tid - prow_ix * 10;
    const uint32_t zsplat = (uint32_t)a.zp_in * 0x01010101u;
    const size_t plane_sz = (size_t)a.H * a.W;
    const int planar_off = (prow_ix - 1) * a.W + 4 * pq - 4;  // byte offset of this thread's group from the tile's (16 ty, 32 tx)
    // the thread's group of colour plane k from (tile origin - one plane): never negative, 32 bits (the launcher refuses inputs of 2 GiB and more)
    unsigned offk[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) offk[k] = (unsigned)(planar_off + (int)plane_sz) + (unsigned)k * (unsigned)plane_sz;
    const uint8_t *const xm1 = a.x - plane_sz;  // (only ever dereferenced one plane or more further on)
    // raw buffer descriptor over the planes (stride 0, no range to speak of: the launcher refuses inputs of 2 GiB and more; word 3 = 32-bit data format)
    const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(xm1), 0, (int)0xFFFFFFFFu, 0x00020000);

    // ---- The workgroup's tiles, ONE LANE PER TILE (round 5).  Rounds 1-4 carried the tile position (b, ty, tx) in scalar registers and advanced it
    // with carries, then derived the input origin, the "interior tile" test, the output offset and the "whole patch inside the map" test from it:
    // ~60 scalar instructions per tile, and this kernel runs at one instruction of ANY kind per issue slot -- its scalar stream
    // was as long as its vector stream (SQ_INSTS_SALU 10.7 M = SQ_INSTS_VALU per launch).  A persistent workgroup walks at most 64 tiles (the
    // launcher sizes the grid so), so lane l of every wave computes tile l's three words ONCE, in parallel, and the loop fetches them with
    // v_readlane: input origin, output offset, flags (bit 0: no cell of the 18 x 40 input patch lies outside the image, bit 1: the whole 8 x 16
    // pooled patch lies inside the pooled map, bits 2-11 tx, bits 12-21 ty for the border cases).
    // XCD-aware order as before: workgroup id w runs on XCD w % 8 (each with its own L2); neighbouring tiles share the cache lines of their halo
    // columns / rows, so every XCD takes one CONTIGUOUS eighth of the tiles and its workgroups walk it side by side.
    const bool xcd_walk = (gridDim.x & 7) == 0 && !(a.debug_flags & 2048);
    const int per_x = xcd_walk ? (ntiles + 7) >> 3 : ntiles;
    const int tstride = xcd_walk ? (int)(gridDim.x >> 3) : (int)gridDim.x;
    const int tbase_x = xcd_walk ? (int)(blockIdx.x & 7) * per_x : 0;
    const int tend = min(tbase_x + per_x, ntiles);
    const int tile0 = tbase_x + (xcd_walk ? (int)(blockIdx.x >> 3) : (int)blockIdx.x);
    unsigned T_in, T_out, T_fl;
    int nt;
    {
        const int t = tile0 + lane * tstride;
        nt = __builtin_popcountll(__builtin_amdgcn_ballot_w64(t < tend));  // this workgroup's tiles (<= 64)
        const int tc = min(t, ntiles - 1);
        const int b = fd_div(tc, a.fd_tpi);  // (divisions by launch constants: common.h FastDiv, set by the launcher)
        const int r = tc - b * tpi;
        const int ty = fd_div(r, a.fd_tx), tx = r - ty * tiles_x;
        if constexpr (PLANAR) T_in = ((unsigned)b * 3u) * (unsigned)plane_sz + (unsigned)(16 * ty) * (unsigned)a.W + 32u * (unsigned)tx;
        else T_in = (unsigned)(a.in_lead + (b * (a.H + 1) + 16 * ty) * W1 + 32 * tx - 1);  // image cell (0, 0) of the patch (fits an int: the launcher checks in_cells)
        T_out = (unsigned)(a.pool_lead + (b * (OH + 1) + 8 * ty + 1) * (OW + 1) + 16 * tx) * (unsigned)a.pool_cs;
        // rows 16 ty - 1 .. 16 ty + 16, columns 32 tx - 4 .. 32 tx + 35 all inside the image: no lane needs the pad value
        const bool tin = ty >= 1 && 16 * ty + 16 < a.H && tx >= 1 && 32 * tx + 32 < a.W;
        const bool tall = 16 * tx + 16 <= OW && 8 * ty + 8 <= OH;
        T_fl = (tin ? 1u : 0u) | (tall ? 2u : 0u) | ((unsigned)tx << 2) | ((unsigned)ty << 12);
    }
    auto tile_word = [&](unsigned v, int k) { return (unsigned)__builtin_amdgcn_readlane((int)v, k); };  // wave-uniform k

    auto fetch = [&](unsigned org, unsigned fl, uint32_t(&v)[3]) {
        if constexpr (PLANAR) {
            // wave-uniform tile origin in the instruction's SCALAR offset + the thread's loop-invariant 32-bit offsets in its vector offset: a raw
            // buffer load (descriptor = the input one plane early, built once per kernel), so no 64-bit address arithmetic per tile (the compiler
            // widens a pointer + loop-invariant offset to 64 bits outside the loop and adds base and offset on the VALU: three v_lshl_add_u64 per
            // tile, six registers).  Round 5 wrote these loads as inline asm (`global_load_dword v, voff, s[base]`) with one hand-written wait in
            // front of the staging; a load the compiler cannot see is a load whose destination it may spill or copy BEFORE the wait -- the 32-filter
            // instantiations did exactly that (`global_load_dword v6 ...` followed by `scratch_store_dword v6`), silently wrong bytes.
            // The builtin is visible to the wait-count model: a spill waits first.  MI355_L0_ASM_PREFETCH keeps the old form for A/B runs only.
#ifdef MI355_L0_ASM_PREFETCH
            const uint64_t tb = reinterpret_cast<uint64_t>(xm1) + (uint64_t)org;
            const uint64_t tbs = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(tb >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)tb);
            auto load3 = [&]() {
                asm volatile("global_load_dword %0, %3, %6\n\tglobal_load_dword %1, %4, %6\n\tglobal_load_dword %2, %5, %6"
                             : "+v"(v[0]), "+v"(v[1]), "+v"(v[2])
                             : "v"(offk[0]), "v"(offk[1]), "v"(offk[2]), "s"(tbs)
                             : "memory");
            };
#else
            const int so = __builtin_amdgcn_readfirstlane((int)org);
            auto load3 = [&]() {
#pragma unroll
                for (int k = 0; k < 3; ++k) v[k] = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(xrsrc, (int)offk[k], so, 0);
            };
#endif
            if (fl & 1u) {
                if (tid < 180) load3();
            } else {  // out-of-image groups are the input zero point
                const int tx = (int)((fl >> 2) & 1023u), ty = (int)((fl >> 12) & 1023u);
                const int y = 16 * ty - 1 + prow_ix, x0 = 32 * tx - 4 + 4 * pq;
                const bool inside = (unsigned)y < (unsigned)a.H && (unsigned)x0 < (unsigned)a.W && tid < 180;
#pragma unroll
                for (int k = 0; k < 3; ++k) v[k] = zsplat;
                if (inside) load3();
            }
        } else {
#pragma unroll
            for (int k = 0; k < 3; ++k) v[k] = xc[min(max((int)org + soff[k], 0), a.in_cells - 1)];
        }
    };
    auto stash = [&](int buf, const uint32_t(&v)[3]) {
        if constexpr (PLANAR) {
            if (tid < 180) {
                uint4 c;
                // byte i of the three plane dwords -> cell i = (c0, c1, c2, 0), then x' = x - 128 (pad byte: weight 0)
                // (the bias is applied to the three plane dwords, not to the four cells: the pad byte stays 0, its weight is 0)
                const uint32_t v0 = v[0] ^ 0x80808080u, v1 = v[1] ^ 0x80808080u, v2 = v[2] ^ 0x80808080u;
                const uint32_t lo01 = __builtin_amdgcn_perm(v1, v0, 0x05010400u);  // (p0.b0, p1.b0, p0.b1, p1.b1)
                const uint32_t hi01 = __builtin_amdgcn_perm(v1, v0, 0x07030602u);  // (p0.b2, p1.b2, p0.b3, p1.b3)
                c.x = __builtin_amdgcn_perm(v2, lo01, 0x0c040100u);                // (lo01.b0, lo01.b1, p2.b0, 0)
                c.y = __builtin_amdgcn_perm(v2, lo01, 0x0c050302u);
                c.z = __builtin_amdgcn_perm(v2, hi01, 0x0c060100u);
                c.w = __builtin_amdgcn_perm(v2, hi01, 0x0c070302u);
                uint32_t *dst = &img[buf][prow_ix * ROWC + 4 * pq + 1];
                dst[0] = c.x; dst[1] = c.y; dst[2] = c.z; dst[3] = c.w;
            }
        } else {
#pragma unroll
            for (int k = 0; k < 3; ++k)
                if (tid + 256 * k < 18 * 34) img[buf][tid + 256 * k] = v[k] ^ 0x80808080u;  // x' = x - 128 (pad byte: weight 0)
        }
    };

    // the prefetched dwords have landed: the compiler places that wait itself in front of the staging (A/B build with hand-written loads: this is the wait)
    auto land = [&](uint32_t(&v)[3]) {
#ifdef MI355_L0_ASM_PREFETCH
        if constexpr (PLANAR) asm volatile("s_waitcnt vmcnt(0)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2])::"memory");
#endif
    };
    uint32_t nxt[3] = {0, 0, 0};
    unsigned fl_cur = tile_word(T_fl, 0);
    if (nt > 0) fetch(tile_word(T_in, 0), fl_cur, nxt);
    // (the first tile's image is on its way: the per-lane records below share its latency)
    const bool ept_ok = have_ept && ekey == ept_key(ACT, a.zp_act);  // workgroup-uniform
    bool pow2, never, use_int, need_d2;
    if (ept_ok) {
        pow2 = (eflags & EPT_POW2) != 0;
        never = (eflags & EPT_NEVER) != 0;
        use_int = INTRQ && (eflags & EPT_NOINT) == 0;
        need_d2 = (eflags & EPT_D2) != 0;
        const char *after = reinterpret_cast<const char *>(a.ept + 1) + (size_t)mpad4 * sizeof(EptEntry);
        const L0Lane *ll = reinterpret_cast<const L0Lane *>(after + LUTQ_N);
#pragma unroll
        for (int mt = 0; mt < NM; ++mt) {
            const v4i *rec = reinterpret_cast<const v4i *>(ll + mt * 64 + lane);
            wa[mt][0] = rec[0]; wa[mt][1] = rec[1];
            wd1[mt][0] = rec[2]; wd1[mt][1] = rec[3];
            wd2[mt][0] = rec[4]; wd2[mt][1] = rec[5];
            cb[mt] = rec[6]; lo[mt] = rec[7]; hi[mt] = rec[8];
            const v4i m0v = rec[9], shv = rec[10], qc01 = rec[11], qc23 = rec[12], mp01 = rec[13], mp23 = rec[14];
#pragma unroll
            for (int r = 0; r < 4; ++r) { qm0[mt][r] = m0v[r]; qsh[mt][r] = shv[r]; }
            qc[mt][0] = (int64_t)(((uint64_t)(uint32_t)qc01[1] << 32) | (uint32_t)qc01[0]);
            qc[mt][1] = (int64_t)(((uint64_t)(uint32_t)qc01[3] << 32) | (uint32_t)qc01[2]);
            qc[mt][2] = (int64_t)(((uint64_t)(uint32_t)qc23[1] << 32) | (uint32_t)qc23[0]);
            qc[mt][3] = (int64_t)(((uint64_t)(uint32_t)qc23[3] << 32) | (uint32_t)qc23[2]);
            mp[mt][0] = __hiloint2double(mp01[1], mp01[0]); mp[mt][1] = __hiloint2double(mp01[3], mp01[2]);
            mp[mt][2] = __hiloint2double(mp23[1], mp23[0]); mp[mt][3] = __hiloint2double(mp23[3], mp23[2]);
        }
        if constexpr (LUT) {  // visible after the first __syncthreads of the tile loop
            uint32_t lutw[LUTQ_N / 4 / 256];
#pragma unroll
            for (int k = 0; k < LUTQ_N / 4 / 256; ++k) lutw[k] = reinterpret_cast<const uint32_t *>(after)[tid + 256 * k];
#pragma unroll
            for (int k = 0; k < LUTQ_N / 4 / 256; ++k) reinterpret_cast<uint32_t *>(lut)[tid + 256 * k] = lutw[k];
        }
    } else {
        pow2 = a.hdr->pow2 == 1;
        const int32_t *shiftp = reinterpret_cast<const int32_t *>(reinterpret_cast<const char *>(a.hdr) + a.hdr->off_shift);
        int never_l = 0, noint_l = 0;
#pragma unroll
        for (int mt = 0; mt < NM; ++mt) {
            const int ch = 16 * mt + pc;
            const int dz = a.dzp[ch];
            const int d1 = dz > 127 ? 127 : dz, d2 = dz - d1;  // dz in [-127, 128]
            const uint32_t m1 = (uint32_t)(d1 & 0xFF) * 0x00010101u, m2 = (uint32_t)(d2 & 0xFF) * 0x00010101u;
#pragma unroll
            for (int jx = 0; jx < 2; ++jx)
#pragma unroll
                for (int dx = 0; dx < 4; ++dx) {
                    const int t = dx - jx;  // tap column of cell dx for window column jx
                    const bool real = g < 3 && t >= 0 && t < 3;
                    wa[mt][jx][dx] = real ? (int)(a.wfirst[ch * 9 + 3 * g + t] ^ 0x00808080u) : 0;  // w' = w - 128 on the three channels
                    wd1[mt][jx][dx] = real ? (int)m1 : 0;
                    wd2[mt][jx][dx] = real ? (int)m2 : 0;
                }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int c2 = chq[mt] + r;
                const double m = a.mprime[c2];
                mp[mt][r] = m;
                int32_t l = -2147483647 - 1, h = 2147483647;
                if (!SAT) small_safe_range<ACT>(m, a.zp_act, l, h);
                int32_t lb = 0; uint32_t rg = 0;
                if (!biased_safe_range(l, h, lb, rg)) never_l = 1;
                qm0[mt][r] = qsh[mt][r] = 0;
                if (INTRQ && (!pow2 || !intrq_make(a.mval[c2], shiftp[c2], lb, (int32_t)((uint32_t)lb + rg), qm0[mt][r], qsh[mt][r], ACT == MI355_ACT_RELU6))) noint_l = 1;
                qc[mt][r] = (int64_t)lb * (int64_t)qm0[mt][r];
                cb[mt][r] = (int32_t)((uint32_t)a.cwb[c2] - (uint32_t)lb);  // accumulators biased by the safe range's lower end (common.h)
                lo[mt][r] = lb;
                hi[mt][r] = (int32_t)rg;  // hi - lo
            }
        }
        never = __syncthreads_or(never_l) != 0;
        use_int = INTRQ && __syncthreads_or(noint_l) == 0;
        if constexpr (LUT) {  // visible after the first __syncthreads of the tile loop
            if (use_int) leaky_lutf_build<false>(lut, a.zp_act, tid, 256);
            else leaky_lut_build<false>(lut, a.zp_act, tid, 256);
        }
        need_d2 = false;
#pragma unroll
        for (int mt = 0; mt < NM; ++mt) need_d2 |= __builtin_amdgcn_ballot_w64(wd2[mt][0][0] != 0) != 0;
    }
#pragma unroll
    for (int mt = 0; mt < NM; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) asm volatile("" : "+v"(qc[mt][r]));  // opaque: the compiler otherwise factors u * M0 + lo * M0 back into (u + lo) * M0 as a 64 x 32 multiply

    if (nt > 0) {
        land(nxt);
        stash(0, nxt);
    }
    // Every register loaded so far (weights, per-channel constants) is in: without this the compiler has to keep an
    // s_waitcnt vmcnt(0) in front of the first MFMA of the (shared) loop body, which then also waits for the image
    // prefetch issued a few instructions earlier and for the previous tile's stores -- a memory round trip per tile.
    __builtin_amdgcn_s_waitcnt(0x0F70);
    // A tile's packed bytes are stored one tile LATE, right after the next tile's prefetch: vmcnt counts loads and stores alike, and issued
    // in place they put a store round trip to HBM in front of every tile's staging wait.  Issued behind the prefetch they have a whole tile
    // to drain, like the prefetch.
    // Written by hand in the scalar-base form (wave-uniform patch pointer + the lane's loop-invariant 32-bit offset: no 64-bit address add per
    // store) and OPAQUE to the compiler's wait-count model (see the comment at the top of `run`).  The compiler's own vmcnt counts for the
    // prefetch stay valid: the stores are YOUNGER than the loads they follow, so a wait for "all but the k youngest loads" can only wait
    // longer, never too short.
    const unsigned rowpitch = (unsigned)(OW + 1) * (unsigned)a.pool_cs;
    unsigned st_off[2][NM];  // pooled rows 2 wave, 2 wave + 1 of the patch; the lane's pooled column; channels 16 mt + 4 g ..
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int mt = 0; mt < NM; ++mt) st_off[s][mt] = (unsigned)(2 * wave + s) * rowpitch + pc_off + (unsigned)chq[mt];
    uint32_t dpk[2][NM];
    unsigned d_out = 0;  // wave-uniform: byte offset of the deferred patch's first pooled cell
    bool dvalid[2] = {false, false};
    bool dall = false;   // wave-uniform: every lane of the deferred tile stores
    auto store_patch = [&](unsigned patch_off, unsigned off, uint32_t data) {
        // (the patch pointer is wave-uniform by construction; readfirstlane puts it into scalar registers where the compiler cannot prove it)
        const uint64_t rp = reinterpret_cast<uint64_t>(a.ypool) + (uint64_t)patch_off;
        const uint64_t rs = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(rp >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)rp);
        asm volatile("global_store_dword %0, %1, %2" ::"v"(off), "v"(data), "s"(rs) : "memory");
    };
    auto flush_stores = [&]() {
        if (dall) {
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int mt = 0; mt < NM; ++mt) store_patch(d_out, st_off[s][mt], dpk[s][mt]);
        } else {
#pragma unroll
            for (int s = 0; s < 2; ++s)
                if (dvalid[s]) {
#pragma unroll
                    for (int mt = 0; mt < NM; ++mt) store_patch(d_out, st_off[s][mt], dpk[s][mt]);
                }
        }
    };
    const int lane_b = ((g < 3 ? g : 2) * ROWC + 2 * pc + XO) * 4 + 2 * wave * 2 * ROWC * 4;  // the lane's operand bytes inside an image buffer
    L0P_DECL;
    // FASTC: power-of-two shifts, no channel without a wrap-safe range and (where the activation has an integer requantisation) every
    // channel passed its exactness conditions -- decided once per launch, so that the tile loop carries none of these tests; D2C: some
    // channel's zero-point correction is 128 (a third MFMA round).  Four instantiations of the loop, one runs.
    auto run = [&](auto fast_c, auto d2_c) {
        constexpr bool FASTC = decltype(fast_c)::value, D2C = decltype(d2_c)::value;
        int buf = 0;
        // Every instantiation starts (and ends, below) with no memory operation in flight AS FAR AS THE COMPILER'S WAIT-COUNT MODEL GOES.  The
        // four loops are exclusive, but jump threading chains them (the exit of one falls into the guard of the next, which then fails), and
        // along that infeasible path the next loop inherits the previous loop's prefetch as "still pending": when the register allocator
        // reuses one of THOSE destination registers for an MFMA accumulator the model demands s_waitcnt vmcnt(0) in front of the tile's third
        // MFMA -- a memory round trip per tile that no executed path needs (round 5: 3 150 -> 4 120 clocks per tile after an unrelated prologue
        // change moved the allocation; rounds 2-4 met the same effect as "allocation luck").  Executed once per kernel: free.  (The caller
        // also ends every instantiation with its own copy of the kernel's tail and a return.)
        __builtin_amdgcn_s_waitcnt(0x0F70);
        // the two image rows x four cells (x = 2 pcol - 1 .. 2 pcol + 2) that feed the four window positions of this lane's k-group for pooled
        // row s of the wave: one aligned 16-byte operand per row (8-byte aligned in LDS: two dwords each from a ds_read2_b64)
        auto read_b = [&](const char *ib, int s, v4i (&bf)[2]) {
            const uint32_t *p0 = reinterpret_cast<const uint32_t *>(ib + s * 2 * ROWC * 4);
#pragma unroll
            for (int jy = 0; jy < 2; ++jy) {
                const uint2 q0 = *reinterpret_cast<const uint2 *>(p0 + jy * ROWC), q1 = *reinterpret_cast<const uint2 *>(p0 + jy * ROWC + 2);
                bf[jy] = v4i{(int)q0.x, (int)q0.y, (int)q1.x, (int)q1.y};
            }
        };
        auto chain = [&](int mt, const v4i (&bf)[2], v4i (&acc)[4]) {
            __builtin_amdgcn_s_setprio(3);  // the MFMA chain outranks the other waves' requantisation
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(wa[mt][j & 1], bf[j >> 1], cb[mt], 0, 0, 0);
            // the correction passes as their own rounds over the four (independent) window positions
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(wd1[mt][j & 1], bf[j >> 1], acc[j], 0, 0, 0);
            if constexpr (D2C) {
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(wd2[mt][j & 1], bf[j >> 1], acc[j], 0, 0, 0);
            }
            __builtin_amdgcn_s_setprio(0);
        };
        for (int k = 0; k < nt; ++k, buf ^= 1) {
            L0P_MARK(6);
            __syncthreads();  // this tile's image is complete; every wave is past the previous tile
            L0P_MARK(0);
            const bool more = k + 1 < nt;
            unsigned fl_nxt = 0;
            if (more) {
                fl_nxt = tile_word(T_fl, k + 1);
                fetch(tile_word(T_in, k + 1), fl_nxt, nxt);
            }
            L0P_MARK(1);
#ifndef MI355_L0_STORES_IN_PLACE
            flush_stores();
#endif
            L0P_MARK(2);
            const char *const ib = reinterpret_cast<const char *>(img[buf]) + lane_b;
            if constexpr (FASTC) {
                // ---- the common launch: every window's maximum is requantised in its integer / FP64-of-maximum form with NO test in the way;
                // the range test is carried along as a per-lane minimum (rg -sat- umax == 0 <=> some accumulator of the window lies at or
                // beyond the end of the wrap-safe range) and looked at ONCE per tile: a wave that sees one redoes its two pooled rows in the
                // reference's order (bytes first, then the maximum) -- rare, and then exact.
                uint32_t margin = 0xFFFFFFFFu;
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    v4i bf[2];
                    read_b(ib, s, bf);
#pragma unroll
                    for (int mt = 0; mt < NM; ++mt) {
                        v4i acc[4];
                        chain(mt, bf, acc);
                        L0P_MARK_V(2 + 2 * s, acc[3][0]);
                        uint32_t umax[4];
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            umax[r] = max(max((uint32_t)acc[0][r], (uint32_t)acc[1][r]), max((uint32_t)acc[2][r], (uint32_t)acc[3][r]));
                            margin = min(margin, __builtin_elementwise_sub_sat((uint32_t)hi[mt][r], umax[r]));
                        }
                        uint32_t packed;
                        if constexpr (INTRQ) {  // f = floor(a * M0 / 2^(32 + sh)), a = u + lo: v_mad_u64_u32 (u * M0 + lo * M0, exact mod 2^64), then a shift of the high dword
                            int32_t f[4];
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                const uint64_t p = (uint64_t)umax[r] * (uint64_t)(uint32_t)qm0[mt][r] + (uint64_t)qc[mt][r];
                                f[r] = (int32_t)(uint32_t)(p >> 32) >> qsh[mt][r];
                            }
                            if constexpr (LUT) {
#ifndef MI355_L0_LEAKY_ARITH
                                uint32_t bt[4];
#pragma unroll
                                for (int r = 0; r < 4; ++r) bt[r] = lut[lutq_index(f[r])];  // (a window beyond the range may index anywhere: an LDS read outside the
                                                                                              // workgroup's allocation returns 0, and the tile is redone below)
                                packed = pack4_bytes(bt[0], bt[1], bt[2], bt[3]);
#else  // A/B build: LEAKY in four VALU instructions on the floor form instead of the table's LDS round trip
                                packed = pack4_biased(leaky_of_floor(f[0], a.zp_act), leaky_of_floor(f[1], a.zp_act), leaky_of_floor(f[2], a.zp_act),
                                                      leaky_of_floor(f[3], a.zp_act));
#endif
                            } else {  // RELU6: zp + max(q, 0) == zp + max(f, 0); SAT clamps
                                int32_t v[4];
#pragma unroll
                                for (int r = 0; r < 4; ++r) {
                                    v[r] = a.zp_act + max(f[r], 0);
                                    if (SAT) v[r] = min(v[r], 255);
                                }
                                packed = pack4_biased(v[0], v[1], v[2], v[3]);
                            }
                        } else {
                            int32_t amax[4][1], v1[4][1];
#pragma unroll
                            for (int r = 0; r < 4; ++r) amax[r][0] = (int32_t)(umax[r] + (uint32_t)lo[mt][r]);
                            requant_values<ACT, SAT, 1>(amax, mp[mt], a.zp_act, v1);
                            packed = pack4_biased(v1[0][0], v1[1][0], v1[2][0], v1[3][0]);
                        }
                        dpk[s][mt] = packed;
                        L0P_MARK_V(3 + 2 * s, packed);
                    }
                }
                if (__builtin_amdgcn_ballot_w64(margin == 0u) != 0) {  // some window of this wave may wrap: the reference's order for both pooled rows
#pragma unroll 1
                    for (int s = 0; s < 2; ++s) {
                        v4i bf[2];
                        read_b(ib, s, bf);
#pragma unroll
                        for (int mt = 0; mt < NM; ++mt) {
                            v4i acc[4];
                            chain(mt, bf, acc);
                            uint32_t ex;
                            if constexpr (INTRQ) {
                                // (the integer form's hot loop needs neither the FP64 multipliers nor the range's lower end: this cold path fetches
                                // them again -- lo = (cw + bias) - seed -- instead of holding twelve registers across the loop; the kernel sits at
                                // the 128-register edge of four waves per SIMD)
                                double mpr[4];
                                v4i lor;
#pragma unroll
                                for (int r = 0; r < 4; ++r) {
                                    mpr[r] = a.mprime[chq[mt] + r];
                                    lor[r] = (int32_t)((uint32_t)a.cwb[chq[mt] + r] - (uint32_t)cb[mt][r]);
                                }
                                ex = first_pool_exact_path<ACT, SAT>(acc, lor, mpr, a.mval + chq[mt], a.sval + chq[mt], a.zp_act, true);
                            } else {
                                ex = first_pool_exact_path<ACT, SAT>(acc, lo[mt], mp[mt], a.mval + chq[mt], a.sval + chq[mt], a.zp_act, true);
                            }
                            if (s == 0) dpk[0][mt] = ex; else dpk[1][mt] = ex;
                        }
                    }
                }
            } else {
                // ---- the general launch (shifts that are not powers of two, a channel without a wrap-safe range or outside the integer form's
                // conditions): tested per pooled row, as rounds 2-4 did for every launch
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    v4i bf[2];
                    read_b(ib, s, bf);
#pragma unroll
                    for (int mt = 0; mt < NM; ++mt) {
                        v4i acc[4];
                        chain(mt, bf, acc);
                        L0P_MARK_V(2 + 2 * s, acc[3][0]);
                        // accumulators are biased by lo: one unsigned maximum gives the range test and the window maximum (common.h)
                        uint32_t umax[4];
                        uint64_t badm = never ? ~0ull : 0ull;
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            umax[r] = max(max((uint32_t)acc[0][r], (uint32_t)acc[1][r]), max((uint32_t)acc[2][r], (uint32_t)acc[3][r]));
                            badm |= __builtin_amdgcn_ballot_w64(umax[r] > (uint32_t)hi[mt][r]);
                        }
                        uint32_t packed;
                        if (badm == 0 && pow2) {
                            if (INTRQ && use_int) {
                                int32_t f[4];
#pragma unroll
                                for (int r = 0; r < 4; ++r) {
                                    const uint64_t p = (uint64_t)umax[r] * (uint64_t)(uint32_t)qm0[mt][r] + (uint64_t)qc[mt][r];
                                    f[r] = (int32_t)(uint32_t)(p >> 32) >> qsh[mt][r];
                                }
                                if constexpr (LUT) {
                                    uint32_t bt[4];
#pragma unroll
                                    for (int r = 0; r < 4; ++r) bt[r] = lut[lutq_index(f[r])];
                                    packed = pack4_bytes(bt[0], bt[1], bt[2], bt[3]);
                                } else {  // RELU6: zp + max(q, 0) == zp + max(f, 0); SAT clamps
                                    int32_t v[4];
#pragma unroll
                                    for (int r = 0; r < 4; ++r) {
                                        v[r] = a.zp_act + max(f[r], 0);
                                        if (SAT) v[r] = min(v[r], 255);
                                    }
                                    packed = pack4_biased(v[0], v[1], v[2], v[3]);
                                }
                            } else if constexpr (LUT) {  // q of a window inside the safe range lies inside the table (common.h)
                                int32_t amax[4][1];
#pragma unroll
                                for (int r = 0; r < 4; ++r) amax[r][0] = (int32_t)(umax[r] + (uint32_t)lo[mt][r]);
                                // four independent chains, issued pass by pass: left alone the compiler threads all four conversions through
                                // one register pair and every FP64 instruction waits out the latency of the one before it
                                uint32_t bt[4];
                                double dd[4];
                                int32_t qq[4];
#pragma unroll
                                for (int r = 0; r < 4; ++r) dd[r] = (double)amax[r][0];
                                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                                for (int r = 0; r < 4; ++r) dd[r] = dd[r] * mp[mt][r];
                                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                                for (int r = 0; r < 4; ++r) qq[r] = (int32_t)dd[r];
                                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                                for (int r = 0; r < 4; ++r) bt[r] = lut[lutq_index(qq[r])];
                                packed = pack4_bytes(bt[0], bt[1], bt[2], bt[3]);
                            } else {
                                int32_t amax[4][1], v1[4][1];
#pragma unroll
                                for (int r = 0; r < 4; ++r) amax[r][0] = (int32_t)(umax[r] + (uint32_t)lo[mt][r]);
                                requant_values<ACT, SAT, 1>(amax, mp[mt], a.zp_act, v1);
                                packed = pack4_biased(v1[0][0], v1[1][0], v1[2][0], v1[3][0]);
                            }
                        } else {  // some window of this wave may wrap (or odd shifts): the reference's order, bytes first, then the maximum
                            packed = first_pool_exact_path<ACT, SAT>(acc, lo[mt], mp[mt], a.mval + chq[mt], a.sval + chq[mt], a.zp_act, pow2);
                        }
                        dpk[s][mt] = packed;
                        L0P_MARK_V(3 + 2 * s, packed);
                    }
                }
            }
            // the deferred stores of this tile: the patch's first pooled cell (tile word) + per-lane offsets; lanes outside the pooled map
            // (ragged right / lower patches) are masked
            d_out = tile_word(T_out, k);
            dall = (fl_cur & 2u) != 0;
            if (!dall) {
                const int tx = (int)((fl_cur >> 2) & 1023u), ty = (int)((fl_cur >> 12) & 1023u);
                const bool colvalid = 16 * tx + pc < OW;
                dvalid[0] = colvalid && 8 * ty + 2 * wave < OH;
                dvalid[1] = colvalid && 8 * ty + 2 * wave + 1 < OH;
            }
            fl_cur = fl_nxt;
#ifdef MI355_L0_STORES_IN_PLACE  // A/B builds: the stores in place, in front of the staging wait
            flush_stores();
#endif
            L0P_MARK(6);
            if (more) land(nxt);
            L0P_MARK(8);  // the prefetch (and the deferred stores) landed
            if (more) stash(buf ^ 1, nxt);
#ifdef MI355_ABLATE
            L0P_MARK(9);  // staging: permutes + LDS writes
            l0p[7] += 1;
#endif
        }
        __builtin_amdgcn_s_waitcnt(0x0F70);
    };
    const bool fastc = pow2 && !never && (!INTRQ || use_int);
    // each instantiation is followed by its OWN copy of the kernel's tail and a return: no control-flow edge leads from one loop to another
    // (see the comment at the top of `run`)
    auto finish = [&]() {
        flush_stores();
        L0P_MARK(6);
        L0P_STORE();
    };
    if (fastc) {
        if (need_d2) { run(std::true_type{}, std::true_type{}); finish(); return; }
        run(std::true_type{}, std::false_type{}); finish(); return;
    }
    if (need_d2) { run(std::false_type{}, std::true_type{}); finish(); return; }
    run(std::false_type{}, std::false_type{});
    finish();
}

// The same kernel without the pool: the first layer of the non-tiny networks (YOLOv3's 3 -> 32 at full resolution) stores
// every conv pixel.  Same tiling (a lane's four MFMAs are the 2x2 block of conv pixels at (2 prow + jy, 2 pcol + jx)), all
// sixteen values of a lane requantised, four 4-byte stores per m-tile.
template <int ACT, bool SAT, int NM, bool PLANAR>
__global__ __launch_bounds__(256, 4) void conv_first_mfma_kernel(const AuxArgs a)
{
    constexpr int ROWC = first_stage_rowc(PLANAR), XO = PLANAR ? 4 : 0;
    __shared__ __attribute__((aligned(16))) uint32_t img[2][18 * ROWC];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave-uniform: tile rows and output rows stay on the scalar unit
    const int pc = lane & 15, g = lane >> 4;
    const unsigned pc_off = 2u * (unsigned)pc * (unsigned)a.out_cs;  // byte offset of the lane's left conv column inside an output row
    const int W1 = a.W + 1;
    const int OH = a.H >> 1, OW = a.W >> 1;
    const int tiles_x = (OW + 15) >> 4, tiles_y = (OH + 7) >> 3, tpi = tiles_x * tiles_y;
    const int ntiles = a.B * tpi;
    const bool pow2 = a.hdr->pow2 == 1;

    // ---- per-lane constants: A fragments (row = channel 16*mt + pc, k-group g), channel parameters of the lane's four
    //      accumulator rows 16*mt + 4*g + r
    v4i wa[NM], wd1[NM], wd2[NM], cb[NM];
    int chq[NM];
    double mp[NM][4];
#pragma unroll
    for (int mt = 0; mt < NM; ++mt) {
        // With two m-tiles (32 filters) the A rows are dealt so that a lane's accumulator rows 4 g + r of BOTH tiles are eight consecutive
        // filters 8 g + 4 mt + r: one 8-byte store per output pixel instead of two 4-byte ones (the kernel's stores are its bottleneck, and
        // what they cost is their issue: DESIGN.md 4.5).  The weights are read by filter index here, so nothing changes at pack time.
        const int ch = NM == 2 ? 8 * (pc >> 2) + 4 * mt + (pc & 3) : 16 * mt + pc;
        const int dz = a.dzp[ch];
        const int d1 = dz > 127 ? 127 : dz, d2 = dz - d1;  // dz in [-127, 128]
        const uint32_t m1 = (uint32_t)(d1 & 0xFF) * 0x00010101u, m2 = (uint32_t)(d2 & 0xFF) * 0x00010101u;
#pragma unroll
        for (int dx = 0; dx < 4; ++dx) {
            const bool real = g < 3 && dx < 3;
            wa[mt][dx] = real ? (int)(a.wfirst[ch * 9 + 3 * g + dx] ^ 0x00808080u) : 0;  // w' = w - 128 on the three channels
            wd1[mt][dx] = real ? (int)m1 : 0;
            wd2[mt][dx] = real ? (int)m2 : 0;
        }
        chq[mt] = NM == 2 ? 8 * g + 4 * mt : 16 * mt + 4 * g;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int c2 = chq[mt] + r;
            const double m = a.mprime[c2];
            mp[mt][r] = m;
            cb[mt][r] = a.cwb[c2];
        }
    }
    bool need_d2 = false;
#pragma unroll
    for (int mt = 0; mt < NM; ++mt) need_d2 |= __builtin_amdgcn_ballot_w64(wd2[mt][0] != 0) != 0;

    // ---- staging: thread t owns image dwords t, t + 256, t + 512 (< 612): their cell offsets from the tile origin
    int soff[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int slot = min(tid + 256 * k, 18 * 34 - 1);
        const int r = slot / 34, c = slot - r * 34;
        soff[k] = r * W1 + c;
    }
    const uint32_t *xc = reinterpret_cast<const uint32_t *>(a.x);
    // tile walk without per-tile divisions: (b, ty, tx) advances by the decomposition of gridDim.x with carries
    struct Pos { int b, ty, tx; };
    auto pos_of = [&](int t) {  // (divisions by launch constants: common.h FastDiv, set by the launcher)
        Pos p;
        p.b = fd_div(t, a.fd_tpi);
        const int r = t - p.b * tpi;
        p.ty = fd_div(r, a.fd_tx);
        p.tx = r - p.ty * tiles_x;
        return p;
    };
    const Pos step = pos_of(((gridDim.x & 7) == 0 && !(a.debug_flags & 2048)) ? (int)(gridDim.x >> 3) : (int)gridDim.x);
    auto advance = [&](Pos &p) {
        p.tx += step.tx;
        p.ty += step.ty;
        p.b += step.b;
        if (p.tx >= tiles_x) { p.tx -= tiles_x; ++p.ty; }
        if (p.ty >= tiles_y) { p.ty -= tiles_y; ++p.b; }
    };
    // PLANAR (the reference's [B][3][H][W] uint8 planes read in place: no layout conversion pass): thread t < 180 owns image
    // row t / 10 and the 4-pixel group t % 10 of the 40-cell LDS row (image x = 32 tx - 4 + 4 q .. + 3): one aligned dword
    // from each colour plane, interleaved into four (c0, c1, c2, 0) cells on the way into LDS.  W % 4 == 0, so a group is
    // either inside the image or entirely pad (cells of the input zero point, ref: src/convolutional_layer.c:703-705).
    const int prow_ix = tid / 10, pq = tid - prow_ix * 10;
    const uint32_t zsplat = (uint32_t)a.zp_in * 0x01010101u;
    const size_t plane_sz = (size_t)a.H * a.W;
    const int planar_off = (prow_ix - 1) * a.W + 4 * pq - 4;  // byte offset of this thread's group from the tile's (16 ty, 32 tx)
    auto fetch = [&](const Pos &p, uint32_t(&v)[3]) {  // cell indices fit an int (the launcher checks in_cells)
        if constexpr (PLANAR) {
            // wave-uniform tile base on the scalar unit + the thread's loop-invariant 32-bit offset (scalar-base loads);
            // out-of-image groups read a clamped in-image address and are replaced by the pad value
            const int y = 16 * p.ty - 1 + prow_ix, x0 = 32 * p.tx - 4 + 4 * pq;
            const bool inside = (unsigned)y < (unsigned)a.H && (unsigned)x0 < (unsigned)a.W && tid < 180;
            const uint8_t *tbase = a.x + (size_t)p.b * 3 * plane_sz + (size_t)(16 * p.ty) * a.W + 32 * p.tx;
            const unsigned off = inside ? (unsigned)(planar_off + (int)plane_sz) : (unsigned)plane_sz;  // biased by one plane: never negative
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const uint32_t t = *reinterpret_cast<const uint32_t *>(tbase - plane_sz + off + (size_t)k * plane_sz);
                v[k] = inside ? t : zsplat;
            }
        } else {
            const int org = a.in_lead + (p.b * (a.H + 1) + 16 * p.ty) * W1 + 32 * p.tx - 1;  // image cell (0, 0)
#pragma unroll
            for (int k = 0; k < 3; ++k) v[k] = xc[min(max(org + soff[k], 0), a.in_cells - 1)];
        }
    };
    auto stash = [&](int buf, const uint32_t(&v)[3]) {
        if constexpr (PLANAR) {
            if (tid < 180) {
                uint4 c;
                // byte i of the three plane dwords -> cell i = (c0, c1, c2, 0), then x' = x - 128 (pad byte: weight 0)
                const uint32_t lo01 = __builtin_amdgcn_perm(v[1], v[0], 0x05010400u);  // (p0.b0, p1.b0, p0.b1, p1.b1)
                const uint32_t hi01 = __builtin_amdgcn_perm(v[1], v[0], 0x07030602u);  // (p0.b2, p1.b2, p0.b3, p1.b3)
                c.x = __builtin_amdgcn_perm(v[2], lo01, 0x0c040100u) ^ 0x80808080u;    // (lo01.b0, lo01.b1, p2.b0, 0)
                c.y = __builtin_amdgcn_perm(v[2], lo01, 0x0c050302u) ^ 0x80808080u;
                c.z = __builtin_amdgcn_perm(v[2], hi01, 0x0c060100u) ^ 0x80808080u;
                c.w = __builtin_amdgcn_perm(v[2], hi01, 0x0c070302u) ^ 0x80808080u;
                uint32_t *dst = &img[buf][prow_ix * ROWC + 4 * pq + 1];
                dst[0] = c.x; dst[1] = c.y; dst[2] = c.z; dst[3] = c.w;
            }
        } else {
#pragma unroll
            for (int k = 0; k < 3; ++k)
                if (tid + 256 * k < 18 * 34) img[buf][tid + 256 * k] = v[k] ^ 0x80808080u;  // x' = x - 128 (pad byte: weight 0)
        }
    };

    // XCD-aware tile walk: workgroup id b runs on XCD b % 8 (each with its own L2).  Neighbouring tiles share the cache lines
    // of their halo columns / rows, so every XCD takes one CONTIGUOUS eighth of the tiles and its workgroups walk it side by
    // side: tile = xcd * per + idx, idx = b / 8 + k * (gridDim / 8).  (Plain b + k * gridDim put neighbours on different
    // XCDs: 2.3x - 3x the input bytes fetched, profiles/r01_v6_pmc_traffic.json, r02_v1_pmc_traffic.json.)
    const bool xcd_walk = (gridDim.x & 7) == 0 && !(a.debug_flags & 2048);
    const int per_x = xcd_walk ? (ntiles + 7) >> 3 : ntiles;
    const int tstride = xcd_walk ? (int)(gridDim.x >> 3) : (int)gridDim.x;
    const int tbase_x = xcd_walk ? (int)(blockIdx.x & 7) * per_x : 0;
    const int tend = min(tbase_x + per_x, ntiles);
    int tile = tbase_x + (xcd_walk ? (int)(blockIdx.x >> 3) : (int)blockIdx.x);
    Pos cur = pos_of(tile), nxp = cur;
    uint32_t nxt[3];
    if (tile < tend) {
        fetch(cur, nxt);
        stash(0, nxt);
    }
    // Every register loaded so far (weights, per-channel constants) is in: without this the compiler has to keep an
    // s_waitcnt vmcnt(0) in front of the first MFMA of the (shared) loop body, which then also waits for the image
    // prefetch issued a few instructions earlier and for the previous tile's stores -- a memory round trip per tile.
    __builtin_amdgcn_s_waitcnt(0x0F70);
    int buf = 0;
    for (; tile < tend; tile += tstride, buf ^= 1, cur = nxp) {
        __syncthreads();  // this tile's image is complete; every wave is past the previous tile
        const bool more = tile + tstride < tend;
        advance(nxp);
        if (more) fetch(nxp, nxt);
        const int b = cur.b, ty = cur.ty, tx = cur.tx;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int pr = 2 * wave + s;  // pooled row inside the patch
            const int prow = 8 * ty + pr, pcol = 16 * tx + pc;
            const bool valid = prow < OH && pcol < OW;
            // two image rows x five cells feed the four window positions of this lane's k-group
            const uint32_t *p0 = img[buf] + (2 * pr + (g < 3 ? g : 2)) * ROWC + 2 * pc + XO;
            uint32_t rw[2][5];
#pragma unroll
            for (int i = 0; i < 5; ++i) {
                rw[0][i] = p0[i];
                rw[1][i] = p0[ROWC + i];
            }
            // wave-uniform part of the output cells (scalar arithmetic) + the lane's column (precomputed byte offset)
            const long rowcell = (long)a.out_lead + ((long)b * (a.H + 1) + (2 * prow + 1)) * W1 + 32 * tx;
            uint8_t *outp = a.y + rowcell * a.out_cs + pc_off;
            uint32_t pkj[NM][4];
#pragma unroll
            for (int mt = 0; mt < NM; ++mt) {
                v4i acc[4];
                if (!(a.debug_flags & 131072)) __builtin_amdgcn_s_setprio(3);  // the MFMA chain outranks the other waves' requantisation
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int jy = j >> 1, jx = j & 1;
                    const v4i bf = {(int)rw[jy][jx], (int)rw[jy][jx + 1], (int)rw[jy][jx + 2], (int)rw[jy][jx + 3]};
                    acc[j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(wa[mt], bf, cb[mt], 0, 0, 0);
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int jy = j >> 1, jx = j & 1;
                    const v4i bf = {(int)rw[jy][jx], (int)rw[jy][jx + 1], (int)rw[jy][jx + 2], (int)rw[jy][jx + 3]};
                    acc[j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(wd1[mt], bf, acc[j], 0, 0, 0);
                }
                if (need_d2) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int jy = j >> 1, jx = j & 1;
                        const v4i bf = {(int)rw[jy][jx], (int)rw[jy][jx + 1], (int)rw[jy][jx + 2], (int)rw[jy][jx + 3]};
                        acc[j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(wd2[mt], bf, acc[j], 0, 0, 0);
                    }
                }
                __builtin_amdgcn_s_setprio(0);
                // every window position is an output pixel of its own: requantise all sixteen values of the lane
                int32_t accb[4][4], v[4][4];
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int j = 0; j < 4; ++j) accb[r][j] = acc[j][r];
                if (pow2) {
                    requant_values<ACT, SAT, 4>(accb, mp[mt], a.zp_act, v);
                } else {  // (rolled: unrolled, this cold two-step form takes part in sizing the kernel's registers)
                    int32_t tmp[16];
#pragma unroll
                    for (int r = 0; r < 4; ++r)
#pragma unroll
                        for (int j = 0; j < 4; ++j) tmp[4 * r + j] = accb[r][j];
#pragma unroll 1
                    for (int idx = 0; idx < 16; ++idx)
                        tmp[idx] = (int32_t)requant_u8(tmp[idx], 0, a.mval[chq[mt] + (idx >> 2)], a.sval[chq[mt] + (idx >> 2)], a.zp_act, ACT,
                                                       SAT ? MI355_STORE_SATURATE : MI355_STORE_WRAP);
#pragma unroll
                    for (int r = 0; r < 4; ++r)
#pragma unroll
                        for (int j = 0; j < 4; ++j) v[r][j] = tmp[4 * r + j];
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) pkj[mt][j] = pack4_biased(v[0][j], v[1][j], v[2][j], v[3][j]);
            }
            if (valid) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    uint8_t *o = outp + ((size_t)(j >> 1) * W1 + (j & 1)) * a.out_cs;
                    if constexpr (NM == 2) {
                        *reinterpret_cast<uint2 *>(o + 8 * g) = uint2{pkj[0][j], pkj[1][j]};
                    } else {
#pragma unroll
                        for (int mt = 0; mt < NM; ++mt) *reinterpret_cast<uint32_t *>(o + chq[mt]) = pkj[mt][j];
                    }
                }
            }
        }
        if (more) stash(buf ^ 1, nxt);
    }
}

template <int ACT, int NM>
static int first_mfma_launch_sat(AuxArgs &a, hipStream_t st, int grid)
{
    if (a.planar) {
        if (a.store_mode == MI355_STORE_SATURATE)
            hipLaunchKernelGGL((conv_first_mfma_pool_kernel<ACT, true, NM, true>), dim3(grid), dim3(256), 0, st, a);
        else
            hipLaunchKernelGGL((conv_first_mfma_pool_kernel<ACT, false, NM, true>), dim3(grid), dim3(256), 0, st, a);
    } else if (a.store_mode == MI355_STORE_SATURATE)
        hipLaunchKernelGGL((conv_first_mfma_pool_kernel<ACT, true, NM, false>), dim3(grid), dim3(256), 0, st, a);
    else
        hipLaunchKernelGGL((conv_first_mfma_pool_kernel<ACT, false, NM, false>), dim3(grid), dim3(256), 0, st, a);
    return hipGetLastError() == hipSuccess ? MI355_OK : MI355_EHIP;
}

template <int ACT, int NM>
static int first_mfma_nopool_launch_sat(AuxArgs &a, hipStream_t st, int grid)
{
    if (a.planar) {
        if (a.store_mode == MI355_STORE_SATURATE)
            hipLaunchKernelGGL((conv_first_mfma_kernel<ACT, true, NM, true>), dim3(grid), dim3(256), 0, st, a);
        else
            hipLaunchKernelGGL((conv_first_mfma_kernel<ACT, false, NM, true>), dim3(grid), dim3(256), 0, st, a);
    } else if (a.store_mode == MI355_STORE_SATURATE)
        hipLaunchKernelGGL((conv_first_mfma_kernel<ACT, true, NM, false>), dim3(grid), dim3(256), 0, st, a);
    else
        hipLaunchKernelGGL((conv_first_mfma_kernel<ACT, false, NM, false>), dim3(grid), dim3(256), 0, st, a);
    return hipGetLastError() == hipSuccess ? MI355_OK : MI355_EHIP;
}

// first layer without a pool on the matrix pipe; MI355_EINVAL outside its domain (the caller uses the VALU kernel)
int conv_first_mfma_launch(AuxArgs &a, hipStream_t st)
{
    if ((a.n != 16 && a.n != 32) || !a.y || a.ypool || a.acc_out || a.y_f32 || (a.H & 1) || (a.W & 1) || !a.cwb) return MI355_EINVAL;
    if (a.planar ? ((a.W & 3) || (reinterpret_cast<size_t>(a.x) & 3)) : a.in_cs != 4) return MI355_EINVAL;
    if ((long)a.in_cells + 64L * (a.W + 1) >= (1L << 31)) return MI355_EINVAL;  // 32-bit cell arithmetic in the kernel
    const int OH = a.H / 2, OW = a.W / 2;
    const long ntiles = (long)a.B * ((OW + 15) / 16) * ((OH + 7) / 8);
    if (ntiles >= (1L << 31)) return MI355_EINVAL;
    a.fd_tx = fastdiv_make((uint32_t)((OW + 15) / 16));
    a.fd_tpi = fastdiv_make((uint32_t)(((OW + 15) / 16) * ((OH + 7) / 8)));
    const int grid = (int)(ntiles < 1024 ? ntiles : 1024);
    if (a.n == 16) {
        if (a.act == MI355_ACT_LEAKY) return first_mfma_nopool_launch_sat<MI355_ACT_LEAKY, 1>(a, st, grid);
        if (a.act == MI355_ACT_RELU6) return first_mfma_nopool_launch_sat<MI355_ACT_RELU6, 1>(a, st, grid);
        return first_mfma_nopool_launch_sat<MI355_ACT_LINEAR, 1>(a, st, grid);
    }
    if (a.act == MI355_ACT_LEAKY) return first_mfma_nopool_launch_sat<MI355_ACT_LEAKY, 2>(a, st, grid);
    if (a.act == MI355_ACT_RELU6) return first_mfma_nopool_launch_sat<MI355_ACT_RELU6, 2>(a, st, grid);
    return first_mfma_nopool_launch_sat<MI355_ACT_LINEAR, 2>(a, st, grid);
}

// returns MI355_EINVAL when the shape is outside the MFMA kernel's domain (the caller uses the VALU kernel)
int conv_first_mfma_pool_launch(AuxArgs &a, hipStream_t st)
{
    if ((a.n != 16 && a.n != 32) || a.y || a.acc_out || a.y_f32 || (a.H & 1) || (a.W & 1) || !a.cwb) return MI355_EINVAL;
    if (a.planar ? ((a.W & 3) || (reinterpret_cast<size_t>(a.x) & 3)) : a.in_cs != 4) return MI355_EINVAL;
    if ((long)a.in_cells + 64L * (a.W + 1) >= (1L << 31)) return MI355_EINVAL;  // 32-bit cell arithmetic in the kernel
    const int OH = a.H / 2, OW = a.W / 2;
    // 32-bit byte offsets in the kernel: the pooled tensor and (planar) the input planes
    if (((long)a.pool_lead + (long)a.B * (OH + 1) * (OW + 1) + OW + 2) * a.pool_cs >= (1L << 32)) return MI355_EINVAL;
    if (a.planar && (long)a.B * 3 * a.H * a.W >= (1L << 31)) return MI355_EINVAL;
    const long ntiles = (long)a.B * ((OW + 15) / 16) * ((OH + 7) / 8);
    if (ntiles >= (1L << 31)) return MI355_EINVAL;
    a.fd_tx = fastdiv_make((uint32_t)((OW + 15) / 16));
    a.fd_tpi = fastdiv_make((uint32_t)(((OW + 15) / 16) * ((OH + 7) / 8)));
    if ((OW + 15) / 16 > 1023 || (OH + 7) / 8 > 1023) return MI355_EINVAL;  // the kernel's tile words hold tx, ty in ten bits each
    // persistent: four workgroups per CU (five measured slower) -- and never more than 64 tiles per workgroup (one lane per tile in the kernel's
    // tile table), whichever eighth of the tiles an XCD takes: (grid / 8) * 64 >= ceil(ntiles / 8)
    static const int grid_cap = getenv("MI355_L0_GRID") ? atoi(getenv("MI355_L0_GRID")) : 1024;  // (A/B runs)
    long g = ntiles < grid_cap ? ntiles : grid_cap;
    const long need = (((ntiles + 7) / 8 + 63) / 64) * 8;
    if (g < need) g = need;
    if (g >= (1L << 31)) return MI355_EINVAL;
    const int grid = (int)g;
    if (a.n == 16) {
        if (a.act == MI355_ACT_LEAKY) return first_mfma_launch_sat<MI355_ACT_LEAKY, 1>(a, st, grid);
        if (a.act == MI355_ACT_RELU6) return first_mfma_launch_sat<MI355_ACT_RELU6, 1>(a, st, grid);
        return first_mfma_launch_sat<MI355_ACT_LINEAR, 1>(a, st, grid);
    }
    if (a.act == MI355_ACT_LEAKY) return first_mfma_launch_sat<MI355_ACT_LEAKY, 2>(a, st, grid);
    if (a.act == MI355_ACT_RELU6) return first_mfma_launch_sat<MI355_ACT_RELU6, 2>(a, st, grid);
    return first_mfma_launch_sat<MI355_ACT_LINEAR, 2>(a, st, grid);
}
