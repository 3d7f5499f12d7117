// conv_igemm.hip -- implicit-GEMM INT8 convolution on V_MFMA_I32_16X16X64_I8 (gfx950), fused requantise epilogue.
//
// Replaces, for a whole batch, the reference's
//     im2col_cpu_uint8 (ref: src/im2col.c:26-50)  ->  gemm_nn_uint8_int32_te x2 (ref: src/gemm.c:279-299,
//     src/convolutional_layer.c:718-721)  ->  requant / activation loop (ref: src/convolutional_layer.c:726-751)
// without materialising im2col: the GEMM is  acc[oc][p] = sum_k W[oc][k] * X[k][p]  with p = (image, y, x)
// flattened over the whole batch (so 13x13 layers still fill 128/256-wide N tiles) and k = (chunk, tap, channel).
//
// Signed x signed MFMA vs the reference's (u8 - zp_w) x u8 operands: with w' = w_u8 - 128 and
// x' = x_u8 - 128 (activations are *stored* biased, so no flip in the loop) and d = 128 - zp_w,
//     sum_k (w_u8 - zp_w) x_u8 = sum_k w'x'  +  d * sum_k x'  +  [128 * sum_k w' + 128 * K * d]
// The first term is the MFMA; sum_k x' (receptive-field sum, pad taps included) is accumulated on the VALU with
// v_dot4 against 0x01010101 from the very B fragments the MFMA consumes (MFMA and VALU pipes are separate); the
// bracket is a per-channel constant folded at pack time.  Everything is exact in int32.
//
// Data flow per workgroup (BM output channels x BN pixels):
//   * B operand: for each 64-byte channel chunk the *contiguous* cell range that covers the tile's pixels plus a
//     (W+2)-cell halo is staged once in LDS (coalesced 16-byte loads); all 9 taps read it at constant cell offsets
//     (the PHWC layout makes a tap a constant offset), i.e. 9x less global->LDS traffic than per-tap gathers.
//   * A operand: the packed weight slab of one K-step ([BM][64 B], contiguous in HBM) is double buffered in LDS.
//   * global loads for step g+1 are issued before the MFMAs of step g and written to the other LDS buffer after
//     them: one barrier per K-step.
#include "kargs.h"


template <int BM, int BN, int WMW, int WNW, int BPT>
__global__ __launch_bounds__(64 * WMW * WNW) void conv_igemm_i8_kernel(const ConvArgs a)
{
    constexpr int NT = 64 * WMW * WNW;
    constexpr int TM = BM / WMW, TN = BN / WNW;
    constexpr int MS = TM / 16, NS = TN / 16;
    constexpr int APT = (BM * 4 + NT - 1) / NT;  // 16-byte pieces of the A slab per thread
    static_assert(TM % 16 == 0 && TN % 16 == 0, "wave tile must be a multiple of the 16x16 MFMA tile");

    extern __shared__ __attribute__((aligned(16))) char smem[];
    char *ldsA = smem;                                   // [2][BM*64]
    char *ldsB = smem + 2 * BM * 64;                     // [2][ncell_cap*cb]
    const int bbytes = a.ncell_cap * a.cb;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave / WNW, wn = wave % WNW;
    const int kg = lane >> 4, lj = lane & 15;

    // ---- XCD-aware tile assignment: blocks b, b+8, b+16.. run on one XCD (observed dispatch), give each XCD a
    //      contiguous range of (mtile, ntile) so weight slabs and halos are shared in its private L2.
    const int nb = gridDim.x;
    int logical;
    {
        const int id = blockIdx.x;
        const int q = nb >> 3, r = nb & 7, xcd = id & 7, idx = id >> 3;
        logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int mtile = logical / a.ntiles_n;
    const int ntile = logical - mtile * a.ntiles_n;
    const int n0 = ntile * BN;
    const int nlast = min(n0 + BN, a.total_n) - 1;

    const int W1 = a.W + 1;
    const int halo = (a.ksize == 3) ? (a.W + 2) : 0;
    const int f0 = cell_of_pixel(n0, a.H, a.W, a.in_lead);
    const int f1 = cell_of_pixel(nlast, a.H, a.W, a.in_lead);
    const int fstart = f0 - halo;
    const int ncell = f1 - f0 + 1 + 2 * halo;
    const int bpc = a.cb >> 4;                 // 16-byte blocks per cell chunk: 1, 2 or 4
    const int bpc_sh = (bpc == 4) ? 2 : (bpc == 2 ? 1 : 0);
    const int npieces = ncell * bpc;

    // per-lane LDS byte offset of its B columns (pixel -> cell), one per 16-column sub-tile
    int bbase[NS];
#pragma unroll
    for (int ns = 0; ns < NS; ++ns) {
        int n = min(n0 + wn * TN + ns * 16 + lj, a.total_n - 1);
        bbase[ns] = (cell_of_pixel(n, a.H, a.W, a.in_lead) - fstart) * a.cb;
    }

    v4i acc[MS][NS];
#pragma unroll
    for (int ms = 0; ms < MS; ++ms)
#pragma unroll
        for (int ns = 0; ns < NS; ++ns) acc[ms][ns] = (v4i){0, 0, 0, 0};
    int sx[NS];
#pragma unroll
    for (int ns = 0; ns < NS; ++ns) sx[ns] = 0;

    int4 areg[APT];
    int4 breg[BPT];

    auto gloadA = [&](int g) {
#pragma unroll
        for (int i = 0; i < APT; ++i) {
            int p = tid + i * NT;
            if (p < BM * 4) {
                int sub = p >> 6, within = p & 63;
                const int8_t *src = a.wp + ((size_t)(mtile * (BM / 16) + sub) * a.ksteps + g) * 1024 + within * 16;
                areg[i] = *reinterpret_cast<const int4 *>(src);
            }
        }
    };
    auto sstoreA = [&](int buf) {
#pragma unroll
        for (int i = 0; i < APT; ++i) {
            int p = tid + i * NT;
            if (p < BM * 4) *reinterpret_cast<int4 *>(ldsA + buf * (BM * 64) + p * 16) = areg[i];
        }
    };
    auto gloadB = [&](int chunk) {
#pragma unroll
        for (int i = 0; i < BPT; ++i) {
            int p = tid + i * NT;
            if (p < npieces) {
                int cell = p >> bpc_sh, qq = p & (bpc - 1);
                int f = min(max(fstart + cell, 0), a.in_cells - 1);
                const int8_t *src = a.x + (size_t)f * a.in_cs + chunk * a.cb + qq * 16;
                breg[i] = *reinterpret_cast<const int4 *>(src);
            }
        }
    };
    auto sstoreB = [&](int buf) {
#pragma unroll
        for (int i = 0; i < BPT; ++i) {
            int p = tid + i * NT;
            if (p < npieces) *reinterpret_cast<int4 *>(ldsB + buf * bbytes + p * 16) = breg[i];
        }
    };

    // ---- prologue
    gloadB(0);
    gloadA(0);
    sstoreB(0);
    sstoreA(0);
    __syncthreads();

    int chunk = 0, s = 0;
    for (int g = 0; g < a.ksteps; ++g) {
        const int next = g + 1;
        const bool has_next = next < a.ksteps;
        const bool new_chunk = has_next && (s + 1 == a.spc);
        if (has_next) {
            gloadA(next);
            if (new_chunk) gloadB(chunk + 1);
        }
        // ---- compute K-step g
        {
            const int u = 4 * s + kg;
            const bool valid = u < a.upc;
            const int uc = valid ? u : 0;
            const int tap = uc >> bpc_sh, blk = uc & (bpc - 1);
            int dcell = 0;
            if (a.ksize == 3) {
                const int ty = tap / 3, tx = tap - ty * 3;
                dcell = (ty - 1) * W1 + (tx - 1);
            }
            const int koff = dcell * a.cb + blk * 16;
            const int ones = valid ? 0x01010101 : 0;
            const char *A = ldsA + (g & 1) * (BM * 64);
            const char *Bt = ldsB + (chunk & 1) * bbytes;
            v4i af[MS];
#pragma unroll
            for (int ms = 0; ms < MS; ++ms)
                af[ms] = *reinterpret_cast<const v4i *>(A + (wm * TM + ms * 16 + lj) * 64 + kg * 16);
#pragma unroll
            for (int ns = 0; ns < NS; ++ns) {
                const v4i bf = *reinterpret_cast<const v4i *>(Bt + bbase[ns] + koff);
                int t = sx[ns];
                t = __builtin_amdgcn_sdot4(bf[0], ones, t, false);
                t = __builtin_amdgcn_sdot4(bf[1], ones, t, false);
                t = __builtin_amdgcn_sdot4(bf[2], ones, t, false);
                t = __builtin_amdgcn_sdot4(bf[3], ones, t, false);
                sx[ns] = t;
#pragma unroll
                for (int ms = 0; ms < MS; ++ms)
                    acc[ms][ns] = __builtin_amdgcn_mfma_i32_16x16x64_i8(af[ms], bf, acc[ms][ns], 0, 0, 0);
            }
        }
        if (has_next) {
            sstoreA(next & 1);
            if (new_chunk) sstoreB((chunk + 1) & 1);
        }
        __syncthreads();
        if (++s == a.spc) { s = 0; ++chunk; }
    }

    // ---- epilogue: receptive-field sums across the 4 k-groups, per-channel corrections, requantise, store
#pragma unroll
    for (int ns = 0; ns < NS; ++ns) {
        int t = sx[ns];
        t += __shfl_xor(t, 16);
        t += __shfl_xor(t, 32);
        sx[ns] = t;
    }
    const int hw = a.H * a.W;
#pragma unroll
    for (int ns = 0; ns < NS; ++ns) {
        const int n = n0 + wn * TN + ns * 16 + lj;
        const bool nvalid = n < a.total_n;
        const int nn = nvalid ? n : 0;
        const int b = nn / hw, rem = nn - b * hw;
        const int y = rem / a.W, xx = rem - y * a.W;
        const int ocell = a.out_lead + (b * (a.H + 1) + (y + 1)) * W1 + xx;
#pragma unroll
        for (int ms = 0; ms < MS; ++ms) {
            const int oc0 = mtile * BM + wm * TM + ms * 16 + kg * 4;  // 4 consecutive channels held by this lane
            if (!nvalid || oc0 >= a.n) continue;
            const int4 cw4 = *reinterpret_cast<const int4 *>(a.cw + oc0);
            const int4 dz4 = *reinterpret_cast<const int4 *>(a.dzp + oc0);
            const int4 bi4 = *reinterpret_cast<const int4 *>(a.bias + oc0);
            const int cwv[4] = {cw4.x, cw4.y, cw4.z, cw4.w};
            const int dzv[4] = {dz4.x, dz4.y, dz4.z, dz4.w};
            const int biv[4] = {bi4.x, bi4.y, bi4.z, bi4.w};
            uint32_t packed = 0;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int oc = oc0 + r;
                const int32_t accv = acc[ms][ns][r] + cwv[r] + dzv[r] * sx[ns];
                if (oc < a.n) {
                    const uint32_t u8 = requant_u8(accv, biv[r], a.mval[oc], a.sval[oc], a.zp_act, a.act, a.store_mode);
                    packed |= (u8 ^ 0x80u) << (8 * r);
                    const size_t ridx = ((size_t)b * a.n + oc) * hw + rem;
                    if (a.acc_out) a.acc_out[ridx] = accv;
                    if (a.y_f32) a.y_f32[ridx] = (float)((int)u8 - a.zp_act) * a.s_act;  // ref :757
                }
            }
            if (a.y) *reinterpret_cast<uint32_t *>(a.y + (size_t)ocell * a.out_cs + oc0) = packed;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// host-side launcher
// ---------------------------------------------------------------------------------------------------------------
static int g_force_bm = 0, g_force_bn = 0;
extern "C" int mi355_conv_set_tile(int bm, int bn)
{
    g_force_bm = bm;
    g_force_bn = bn;
    return MI355_OK;
}

template <int BM, int BN, int WMW, int WNW, int BPT>
static int launch_cfg(ConvArgs &a, hipStream_t st)
{
    constexpr int NT = 64 * WMW * WNW;
    a.ntiles_n = (a.total_n + BN - 1) / BN;
    a.mtiles = (a.n + BM - 1) / BM;
    // B-tile capacity in cells: pixels + row pads + image-boundary pad rows + halo both sides
    const int halo = (a.ksize == 3) ? (a.W + 2) : 0;
    int span = BN + (BN + a.W - 1) / a.W + 1 + ((BN + a.H * a.W - 1) / (a.H * a.W) + 1) * (a.W + 1);
    int ncell = span + 2 * halo;
    if ((size_t)ncell * (a.cb / 16) > (size_t)BPT * NT) return MI355_EINVAL;  // staging registers exhausted
    a.ncell_cap = ncell;
    size_t lds = 2 * (size_t)BM * 64 + 2 * (size_t)ncell * a.cb;
    if (lds > 160 * 1024) return MI355_EINVAL;
    auto kern = conv_igemm_i8_kernel<BM, BN, WMW, WNW, BPT>;
    if (lds > 64 * 1024) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)lds) != hipSuccess)
            return MI355_EHIP;
    }
    dim3 grid(a.ntiles_n * a.mtiles), block(NT);
    hipLaunchKernelGGL(kern, grid, block, lds, st, a);
    return hipGetLastError() == hipSuccess ? MI355_OK : MI355_EHIP;
}

int conv_igemm_launch(ConvArgs &a, hipStream_t st)
{
    int bm = g_force_bm, bn = g_force_bn;
    if (!bm) bm = a.n >= 128 ? 128 : (a.n > 32 ? 64 : 32);
    if (!bn) {
        bn = 256;
        // keep at least ~1.5 workgroups per CU on the small 13x13 / 26x26 layers
        long tiles = (long)((a.total_n + 255) / 256) * ((a.n + bm - 1) / bm);
        if (tiles < 384) bn = 128;
    }
    int rc = MI355_EINVAL;
    if (bm == 128 && bn == 256) rc = launch_cfg<128, 256, 2, 4, 6>(a, st);
    else if (bm == 128 && bn == 128) rc = launch_cfg<128, 128, 2, 2, 8>(a, st);
    else if (bm == 64 && bn == 256) rc = launch_cfg<64, 256, 1, 4, 12>(a, st);
    else if (bm == 64 && bn == 128) rc = launch_cfg<64, 128, 1, 2, 12>(a, st);
    else if (bm == 32 && bn == 256) rc = launch_cfg<32, 256, 1, 4, 12>(a, st);
    else if (bm == 32 && bn == 128) rc = launch_cfg<32, 128, 1, 2, 12>(a, st);
    if (rc == MI355_EINVAL && !(g_force_bm || g_force_bn)) {
        // large-W early layers: the halo dominates the staging budget -> narrower wave layout with more registers
        if (bm == 64) rc = launch_cfg<64, 128, 1, 2, 12>(a, st);
        else if (bm == 32) rc = launch_cfg<32, 128, 1, 2, 12>(a, st);
        else rc = launch_cfg<128, 128, 2, 2, 8>(a, st);
    }
    return rc;
}
