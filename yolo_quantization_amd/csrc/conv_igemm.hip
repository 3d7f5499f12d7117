// conv_igemm.hip -- implicit-GEMM INT8 convolution on V_MFMA_I32_32X32X32_I8 (gfx950), fused requantise epilogue.
//
// Replaces, for a whole batch, the reference's
//     im2col_cpu_uint8 (ref: src/im2col.c:26-50)  ->  gemm_nn_uint8_int32_te x2 (ref: src/gemm.c:279-299,
//     src/convolutional_layer.c:718-721)  ->  requant / activation loop (ref: src/convolutional_layer.c:726-751)
// without materialising im2col: the GEMM is  acc[oc][p] = sum_k W[oc][k] * X[k][p]  with p = (image, y, x) over the
// whole batch and k = (channel chunk, tap, channel).
//
// Signed x signed MFMA vs the reference's (u8 - zp_w) x u8 operands: with w' = w_u8 - 128,
// x' = x_u8 - 128 (activations are *stored* biased, so there is no flip in the loop) and d = 128 - zp_w,
//     sum_k (w_u8 - zp_w) x_u8 = sum_k w'x'  +  d * sum_k x'  +  [128 * sum_k w' + 128 * K * d]
// The first term is the MFMA; sum_k x' (receptive-field sum, pad taps included) is accumulated on the VALU with
// v_dot4 against 0x01010101 from the very B fragments the MFMA consumes (MFMA and VALU are separate pipes); the
// bracket is a per-channel constant folded at pack time.  Everything is exact in int32.
//
// Structure of one workgroup (BM output channels x BN pixels):
//   * N tile = BN consecutive valid pixels of the flattened batch (FLAT, small feature maps: 13x13 layers still
//     fill 256-wide tiles) or a TH x 16 pixel patch of one image (PATCH, large feature maps: small halo).
//   * B operand: per 64-byte channel chunk the tile's cells + halo are DMA'd ONCE into LDS
//     (global_load_lds_dwordx4); all 9 taps read them at constant cell offsets (the PHWC layout makes a tap a
//     constant offset): 9x less global->LDS traffic than per-tap gathers.  Double buffered per chunk.
//   * A operand: the packed weight slab of one K-step ([BM][64 B], 1 KiB per 16 rows, stored in HBM in LDS image
//     order) is DMA'd into a 3-stage LDS ring, two K-steps ahead of its use.
//   * LDS images are "piece-major" ([16-byte piece][16 rows|cells][16 B] per KiB): every ds_read_b128 lane group
//     of an MFMA fragment read then covers 16 distinct 16-byte bank slots (conflict-free); with DMA staging the
//     permutation is applied on the per-lane *source* address, the LDS destination stays lane-linear.
//   * one s_barrier per K-step; DMA stays in flight across it (counted s_waitcnt vmcnt(N), never 0 in steady state).
//   * epilogue: corrections + FP64 requantise (exact reference op order) in registers, uint8 tile transposed through
//     LDS so that every pixel's channel run is written with 16-byte stores.
#include "kargs.h"
#include <type_traits>

#define DMA16(gsrc, ldst)                                                                               \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(gsrc),           \
                                     (__attribute__((address_space(3))) void *)(ldst), 16, 0, 0)

__device__ __forceinline__ void wait_vmcnt(int n)
{
    // counted wait; n is wave-uniform.  (s_waitcnt needs an immediate.)
    switch (n) {
    case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
    case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
    case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
    case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
    case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
    case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
    case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
    case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
    case 9: asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); break;
    case 10: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;
    case 11: asm volatile("s_waitcnt vmcnt(11)" ::: "memory"); break;
    case 12: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
    case 13: asm volatile("s_waitcnt vmcnt(13)" ::: "memory"); break;
    case 14: asm volatile("s_waitcnt vmcnt(14)" ::: "memory"); break;
    case 15: asm volatile("s_waitcnt vmcnt(15)" ::: "memory"); break;
    case 16: asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); break;
    case 17: asm volatile("s_waitcnt vmcnt(17)" ::: "memory"); break;
    case 18: asm volatile("s_waitcnt vmcnt(18)" ::: "memory"); break;
    default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
}

constexpr int A_STAGES = 3;
constexpr int BPT_MAX = 12;  // B DMA instructions per wave per chunk load (upper bound; stride-2 patches need 9)
#define APT_OF(BM, NW) (((BM) / 16 + (NW) - 1) / (NW))

// KMODE: 0 = generic K loop (any cb, any ksize); 3 = cb==64 && ksize==3 (taps unrolled, address tables);
//        1 = cb==64 && ksize==1
template <int BM, int BN, int WMW, int WNW, bool PATCH, int KMODE>
__global__ __launch_bounds__(64 * WMW * WNW, (WMW * WNW == 8) ? 4 : 2) void conv_igemm_i8_kernel(const ConvArgs a)
{
    constexpr int NW = WMW * WNW, NT = 64 * NW;
    constexpr int TM = BM / WMW, TN = BN / WNW;
    constexpr int MS = TM / 32, NS = TN / 32;
    constexpr int ACH = BM / 16;                 // 1 KiB chunks per A stage
    constexpr int APT = (ACH + NW - 1) / NW;     // A DMA instructions per wave per stage
    constexpr int TW = 16, TH = BN / 16;         // PATCH geometry
    constexpr int OSTR = BM + 4;                 // epilogue LDS row stride (bytes): odd dword count -> conflict-free b32
    static_assert(TM % 32 == 0 && TN % 32 == 0, "wave tile must be a multiple of the 32x32 MFMA tile");

    extern __shared__ __attribute__((aligned(16))) char smem[];
    char *ldsA = smem;                                 // [A_STAGES][BM*64]
    char *ldsB = smem + A_STAGES * BM * 64;            // [2][bchunks KiB]
    const int bbytes = a.bchunks << 10;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WNW, wn = wave % WNW;
    const int kh = lane >> 5, lj = lane & 31;

    // ---- XCD-aware tile assignment: blocks b, b+8, b+16.. run on one XCD (observed dispatch); give each XCD a
    //      contiguous range of (mtile, ntile) so weight slabs and halos are shared in its private L2.
    int logical;
    {
        const int nb = gridDim.x, id = blockIdx.x;
        const int q = nb >> 3, r = nb & 7, xcd = id & 7, idx = id >> 3;
        logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int mtile = logical / a.ntiles_n;
    const int ntile = logical - mtile * a.ntiles_n;

    const int W1 = a.W + 1;                    // input row pitch in cells
    const int OW1 = a.OW + 1, hw = a.OH * a.OW;  // output geometry (== input for stride 1)
    const int cs = a.stride;                   // PATCH mode only: 1 or 2
    const int cb = a.cb, bpc = cb >> 4;
    const int bpc_sh = (bpc == 4) ? 2 : (bpc == 2 ? 1 : 0);
    const int cpc_sh = 6 - bpc_sh, cpc = 1 << cpc_sh;     // cells per 1 KiB LDS chunk
    const int pstride = cpc << 4;                          // bytes between pieces inside a chunk

    // ---- tile geometry
    int rs;            // LDS row stride in cells (tap (dy,dx) = dy*rs + dx)
    int fstart = 0;    // FLAT: first global cell held in LDS
    int pb = 0, py0 = 0, px0 = 0;  // PATCH: image, patch origin
    int n0 = 0;
    if constexpr (PATCH) {
        const int tpi = a.tiles_x * a.tiles_y;
        pb = ntile / tpi;
        const int t = ntile - pb * tpi;
        const int ty = t / a.tiles_x;
        py0 = ty * TH;                       // patch origin in OUTPUT pixels
        px0 = (t - ty * a.tiles_x) * TW;
        rs = (TW - 1) * cs + 3;              // input cells per LDS row: the patch's columns at stride cs + one halo cell each side
    } else {
        n0 = ntile * BN;
        const int nlast = min(n0 + BN, a.total_n) - 1;
        const int halo = (a.ksize == 3) ? (a.W + 2) : 0;
        fstart = cell_of_pixel(n0, a.H, a.W, a.in_lead) - halo;
        (void)nlast;
        rs = W1;
    }

    // ---- per-lane B columns: LDS cell of each 32-column sub-tile's pixel (tap offset added per K-step)
    int bcell[NS];
#pragma unroll
    for (int ns = 0; ns < NS; ++ns) {
        const int nl = wn * TN + ns * 32 + lj;
        if constexpr (PATCH) {
            bcell[ns] = (cs * (nl >> 4) + 1) * rs + cs * (nl & 15) + 1;  // centre tap of output pixel (nl/16, nl%16)
        } else {
            const int n = min(n0 + nl, a.total_n - 1);
            bcell[ns] = cell_of_pixel(n, a.H, a.W, a.in_lead) - fstart;
        }
    }

    // ---- DMA source bookkeeping.  Every wave issues exactly APT (A) / a.bpt (B) instructions per load so that
    //      the counted vmcnt waits are exact; surplus slots re-load the last chunk (same bytes, same place).
    const int8_t *asrc[APT];
    int adst[APT];
#pragma unroll
    for (int i = 0; i < APT; ++i) {
        const int ch = min(wave + i * NW, ACH - 1);
        asrc[i] = a.wp + ((size_t)(mtile * ACH + ch) * a.ksteps) * 1024 + lane * 16;
        adst[i] = ch << 10;
    }
    // B: chunk q of this wave's i-th instruction; lane -> (cell in chunk, piece)
    const int lcell = lane & (cpc - 1), lpiece = lane >> cpc_sh;
    auto bsrc_of = [&](int q) -> unsigned {  // byte offset of this lane's source cell (+ piece) for LDS chunk q
        const int lc = (q << cpc_sh) + lcell;  // logical LDS cell index
        long f;
        if constexpr (PATCH) {
            const int rr = lc / rs, cc = lc - rr * rs;
            f = (long)a.in_lead + ((long)pb * (a.H + 1) + (cs * py0 + rr)) * W1 + (cs * px0 - 1 + cc);
        } else {
            f = (long)fstart + lc;
        }
        f = f < 0 ? 0 : (f > a.in_cells - 1 ? a.in_cells - 1 : f);
        return (unsigned)(f * a.in_cs + lpiece * 16);
    };
    unsigned bsrc_off[BPT_MAX];  // PATCH only: the div/mod above is hoisted out of the K loop
    if constexpr (PATCH) {
#pragma unroll
        for (int i = 0; i < BPT_MAX; ++i) bsrc_off[i] = (i < a.bpt) ? bsrc_of(min(wave + i * NW, a.bchunks - 1)) : 0u;
    }

    auto issueA = [&](int g) {
        char *stage = ldsA + (g % A_STAGES) * (BM * 64);
#pragma unroll
        for (int i = 0; i < APT; ++i) DMA16(asrc[i] + (size_t)g * 1024, stage + adst[i]);
    };
    auto issueB = [&](int chunk) {
        char *buf = ldsB + (chunk & 1) * bbytes;
        const int8_t *base = a.x + (size_t)chunk * cb;
#pragma unroll
        for (int i = 0; i < BPT_MAX; ++i)
            if (i < a.bpt) {
                const int q = min(wave + i * NW, a.bchunks - 1);
                unsigned off;
                if constexpr (PATCH) off = bsrc_off[i];
                else off = bsrc_of(q);
                DMA16(base + off, buf + (q << 10));
            }
    };

    v16i acc[MS][NS];
#pragma unroll
    for (int ms = 0; ms < MS; ++ms)
#pragma unroll
        for (int ns = 0; ns < NS; ++ns)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[ms][ns][r] = 0;
    int sx[NS];
#pragma unroll
    for (int ns = 0; ns < NS; ++ns) sx[ns] = 0;

    // A fragment LDS offsets (row part): lane row i = lj of 32-row sub-tile ms
    int arow[MS];
#pragma unroll
    for (int ms = 0; ms < MS; ++ms) {
        const int row = wm * TM + ms * 32 + lj;
        arow[ms] = ((row >> 4) << 10) + ((row & 15) << 4);
    }

    if constexpr (KMODE == 0) {
        // ---- prologue: B(0), A(0), A(1)
        issueB(0);
        issueA(0);
        if (a.ksteps > 1) issueA(1);

        int chunk = 0, s = 0;
        bool b_prev = false;  // a B load was issued in the previous iteration (after A(g), before A(g+1) in the queue)
        for (int g = 0; g < a.ksteps; ++g) {
            // ---- retire A(g) (and B(chunk) when this step opens a chunk), leave younger DMA in flight
            {
                const int younger_a = (g + 1 < a.ksteps) ? APT : 0;
                int n = younger_a;
                if (b_prev && s != 0) n += a.bpt;  // B(chunk+1) sits between A(g) and A(g+1): not needed yet
                if (g == 0) n = (a.ksteps > 1) ? APT : 0;  // queue: B(0), A(0), A(1)
                if (n == APT) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(APT) : "memory");  // steady state
                else wait_vmcnt(n);
            }
            __builtin_amdgcn_s_barrier();
            // ---- issue DMA two steps ahead (stage (g+2)%3 was last read in step g-1: every wave is past it)
            b_prev = false;
            if (s == 0 && chunk + 1 < a.nchunks) {
                issueB(chunk + 1);
                b_prev = true;
            }
            if (g + 2 < a.ksteps) issueA(g + 2);

            // ---- compute K-step g: 2 halves of 32 k each
            {
                const char *A = ldsA + (g % A_STAGES) * (BM * 64);
                const char *Bt = ldsB + (chunk & 1) * bbytes;
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int pk = 2 * h + kh;           // 16-byte piece of this K-step held by this lane
                    const int u = 4 * s + pk;
                    const bool valid = u < a.upc;
                    const int uc = valid ? u : 0;
                    const int tap = uc >> bpc_sh, blk = uc & (bpc - 1);
                    int dcell = 0;
                    if (a.ksize == 3) {
                        const int ty = tap / 3, tx = tap - ty * 3;
                        dcell = (ty - 1) * rs + (tx - 1);
                    }
                    const int ones = valid ? 0x01010101 : 0;
                    v4i af[MS];
#pragma unroll
                    for (int ms = 0; ms < MS; ++ms)
                        af[ms] = *reinterpret_cast<const v4i *>(A + arow[ms] + pk * 256);
#pragma unroll
                    for (int ns = 0; ns < NS; ++ns) {
                        const int ci = bcell[ns] + dcell;
                        const int boff = ((ci >> cpc_sh) << 10) + blk * pstride + ((ci & (cpc - 1)) << 4);
                        const v4i bf = *reinterpret_cast<const v4i *>(Bt + boff);
                        int t = sx[ns];
                        t = __builtin_amdgcn_sdot4(bf[0], ones, t, false);
                        t = __builtin_amdgcn_sdot4(bf[1], ones, t, false);
                        t = __builtin_amdgcn_sdot4(bf[2], ones, t, false);
                        t = __builtin_amdgcn_sdot4(bf[3], ones, t, false);
                        sx[ns] = t;
#pragma unroll
                        for (int ms = 0; ms < MS; ++ms)
                            acc[ms][ns] = __builtin_amdgcn_mfma_i32_32x32x32_i8(af[ms], bf, acc[ms][ns], 0, 0, 0);
                    }
                }
            }
            if (++s == a.spc) { s = 0; ++chunk; }
        }
    } else {
        // ---- fast paths (cb == 64: one K-step = one tap of one 64-channel chunk).  All per-lane LDS addresses are
        //      tabulated once per tile, so a K-step is 8 ds_read_b128 + 8 MFMA + 16 v_dot4 + a handful of VALU.
        // per-lane cell -> LDS byte offset (piece-major chunks of 16 cells), incl. this lane's k-half
        auto boff_of = [&](int ci) { return ((ci & ~15) << 6) + ((ci & 15) << 4) + kh * 256; };
        int atab[MS];
#pragma unroll
        for (int ms = 0; ms < MS; ++ms) atab[ms] = arow[ms] + kh * 256;

        auto compute = [&](const char *A, const char *Bt, int dcell) {
            int bt[NS];
#pragma unroll
            for (int ns = 0; ns < NS; ++ns) bt[ns] = boff_of(bcell[ns] + dcell);
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                v4i af[MS];
#pragma unroll
                for (int ms = 0; ms < MS; ++ms) af[ms] = *reinterpret_cast<const v4i *>(A + atab[ms] + h * 512);
#pragma unroll
                for (int ns = 0; ns < NS; ++ns) {
                    const v4i bf = *reinterpret_cast<const v4i *>(Bt + bt[ns] + h * 512);
                    int t = sx[ns];
                    t = __builtin_amdgcn_sdot4(bf[0], 0x01010101, t, false);
                    t = __builtin_amdgcn_sdot4(bf[1], 0x01010101, t, false);
                    t = __builtin_amdgcn_sdot4(bf[2], 0x01010101, t, false);
                    t = __builtin_amdgcn_sdot4(bf[3], 0x01010101, t, false);
                    sx[ns] = t;
#pragma unroll
                    for (int ms = 0; ms < MS; ++ms)
                        acc[ms][ns] = __builtin_amdgcn_mfma_i32_32x32x32_i8(af[ms], bf, acc[ms][ns], 0, 0, 0);
                }
            }
        };

        issueB(0);
        issueA(0);
        if (a.ksteps > 1) issueA(1);
        if constexpr (KMODE == 3) {
            for (int chunk = 0; chunk < a.nchunks; ++chunk) {
                const bool more_chunks = chunk + 1 < a.nchunks;
                const char *Bt = ldsB + (chunk & 1) * bbytes;
                const int g0 = chunk * 9;
#pragma unroll 1
                for (int ty = 0; ty < 3; ++ty) {
                    const int drow = (ty - 1) * rs - 1;
#pragma unroll
                    for (int tx = 0; tx < 3; ++tx) {
                        const int t = ty * 3 + tx;
                        const int g = g0 + t;
                        // queue (old -> young): A(g) [B(chunk+1) when it was issued at t==0 and this is t==1] A(g+1)
                        if (tx == 1 && ty == 0 && more_chunks) wait_vmcnt(APT + a.bpt);
                        else if (t < 8 || more_chunks) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(APT) : "memory");
                        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // very last K-step: nothing younger
                        __builtin_amdgcn_s_barrier();
                        if (tx == 0 && ty == 0 && more_chunks) issueB(chunk + 1);
                        if (t < 7 || more_chunks) issueA(g + 2);  // ring stage (g+2)%3 == (tx+2)%3: 9 taps/chunk
                        compute(ldsA + tx * (BM * 64), Bt, drow + tx);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            }
        } else {
            // 1x1: every K-step opens a new 64-channel chunk: B(g) double buffered, A ring phase = g % 3
            for (int g0 = 0; g0 < a.ksteps; g0 += 3) {
#pragma unroll
                for (int u = 0; u < 3; ++u) {
                    const int g = g0 + u;
                    if (g < a.ksteps) {
                        // queue: A(g) B(g) A(g+1)  (B(g) was issued in step g-1 before A(g+1))
                        if (g + 1 < a.ksteps) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(APT) : "memory");
                        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                        __builtin_amdgcn_s_barrier();
                        if (g + 1 < a.ksteps) issueB(g + 1);
                        if (g + 2 < a.ksteps) issueA(g + 2);
                        compute(ldsA + u * (BM * 64), ldsB + (g & 1) * bbytes, 0);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            }
        }
    }

    // ---- epilogue
#pragma unroll
    for (int ns = 0; ns < NS; ++ns) sx[ns] += __shfl_xor(sx[ns], 32);  // the two 16-byte k-halves

    __syncthreads();  // all MFMA reads of LDS are done: reuse it as the [BN][BM+16] uint8 output tile
    char *otile = smem;
    int *celltab = reinterpret_cast<int *>(smem + BN * OSTR);
    const int m0 = mtile * BM;

    bool nvalid[NS];
    int pb_[NS], rem[NS], nl_[NS];
#pragma unroll
    for (int ns = 0; ns < NS; ++ns) {
        const int nl = wn * TN + ns * 32 + lj;
        nl_[ns] = nl;
        int b, y, xx;
        if constexpr (PATCH) {
            b = pb; y = py0 + (nl >> 4); xx = px0 + (nl & 15);
            nvalid[ns] = y < a.OH && xx < a.OW;
        } else {
            const int n = n0 + nl;
            nvalid[ns] = n < a.total_n;
            const int nn = nvalid[ns] ? n : 0;
            b = nn / hw;
            const int rem0 = nn - b * hw;
            y = rem0 / a.OW; xx = rem0 - y * a.OW;
        }
        pb_[ns] = b;
        rem[ns] = y * a.OW + xx;
        if (wm == 0 && kh == 0) celltab[nl] = nvalid[ns] ? a.out_lead + (b * (a.OH + 1) + (y + 1)) * OW1 + xx : -1;
    }
    // Fast path: whole M tile inside n, no parity dumps, power-of-two shifts (always true for the reference's prep):
    // compile-time activation / store mode, folded single-multiply requantise (csrc/common.h requant_group).
    const bool fast = !a.acc_out && !a.y_f32 && (m0 + BM <= a.n) && a.hdr->pow2 == 1;
    auto epi_fast = [&](auto act_c, auto sat_c) {
        constexpr int ACT = decltype(act_c)::value;
        constexpr bool SAT = decltype(sat_c)::value != 0;
#pragma unroll
        for (int ms = 0; ms < MS; ++ms) {
#pragma unroll
            for (int grp = 0; grp < 4; ++grp) {
                const int ocl = wm * TM + ms * 32 + 8 * grp + 4 * kh;
                const int oc0 = m0 + ocl;
                const int4 dz4 = *reinterpret_cast<const int4 *>(a.dzp + oc0);
                const int4 cb4 = *reinterpret_cast<const int4 *>(a.cwb + oc0);
                const int dzv[4] = {dz4.x, dz4.y, dz4.z, dz4.w}, cbv[4] = {cb4.x, cb4.y, cb4.z, cb4.w};
                int32_t accb[4][NS];
                double mp[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    mp[r] = a.mprime[oc0 + r];
#pragma unroll
                    for (int ns = 0; ns < NS; ++ns)
                        accb[r][ns] = acc[ms][ns][grp * 4 + r] + cbv[r] + __mul24(dzv[r], sx[ns]);
                }
                uint32_t packed[NS];
                requant_group<ACT, SAT, NS>(accb, mp, a.zp_act, packed);
#pragma unroll
                for (int ns = 0; ns < NS; ++ns) *reinterpret_cast<uint32_t *>(otile + nl_[ns] * OSTR + ocl) = packed[ns];
            }
        }
    };
    using std::integral_constant;
    if (fast) {
        const bool sat = a.store_mode == MI355_STORE_SATURATE;
        if (a.act == MI355_ACT_LEAKY) {
            if (sat) epi_fast(integral_constant<int, MI355_ACT_LEAKY>{}, integral_constant<int, 1>{});
            else epi_fast(integral_constant<int, MI355_ACT_LEAKY>{}, integral_constant<int, 0>{});
        } else if (a.act == MI355_ACT_RELU6) {
            if (sat) epi_fast(integral_constant<int, MI355_ACT_RELU6>{}, integral_constant<int, 1>{});
            else epi_fast(integral_constant<int, MI355_ACT_RELU6>{}, integral_constant<int, 0>{});
        } else {
            if (sat) epi_fast(integral_constant<int, MI355_ACT_LINEAR>{}, integral_constant<int, 1>{});
            else epi_fast(integral_constant<int, MI355_ACT_LINEAR>{}, integral_constant<int, 0>{});
        }
    } else {
    #pragma unroll
        for (int ms = 0; ms < MS; ++ms) {
    #pragma unroll
            for (int grp = 0; grp < 4; ++grp) {
                const int ocl = wm * TM + ms * 32 + 8 * grp + 4 * kh;  // 4 consecutive channels held by this lane
                const int oc0 = m0 + ocl;
                if (oc0 >= a.n) {
    #pragma unroll
                    for (int ns = 0; ns < NS; ++ns) *reinterpret_cast<uint32_t *>(otile + nl_[ns] * OSTR + ocl) = 0x80808080u;
                    continue;
                }
                uint32_t packed[NS];
    #pragma unroll
                for (int ns = 0; ns < NS; ++ns) packed[ns] = 0;
    #pragma unroll
                for (int r = 0; r < 4; ++r) {  // per-channel parameters (arrays are padded to mpad: in-bounds for oc >= n)
                    const int oc = oc0 + r;
                    const int cwv = a.cw[oc], dzv = a.dzp[oc], biv = a.bias[oc];
                    const double mv = a.mval[oc], sv = a.sval[oc];
    #pragma unroll
                    for (int ns = 0; ns < NS; ++ns) {
                        const int32_t accv = acc[ms][ns][grp * 4 + r] + cwv + dzv * sx[ns];
                        uint32_t u8 = 0;
                        if (oc < a.n) {
                            u8 = requant_u8(accv, biv, mv, sv, a.zp_act, a.act, a.store_mode);
                            if (nvalid[ns] && (a.acc_out || a.y_f32)) {
                                const size_t ridx = ((size_t)pb_[ns] * a.n + oc) * hw + rem[ns];
                                if (a.acc_out) a.acc_out[ridx] = accv;
                                if (a.y_f32) {
                                    const float f = (float)((int)u8 - a.zp_act) * a.s_act;  // ref :757
                                    a.y_f32[ridx] = f;
                                    if (a.yolo_out) a.yolo_out[ridx] = yolo_entry_act(f, oc % a.yolo_per);
                                }
                            }
                        }
                        packed[ns] |= (u8 ^ 0x80u) << (8 * r);
                    }
                }
    #pragma unroll
                for (int ns = 0; ns < NS; ++ns) *reinterpret_cast<uint32_t *>(otile + nl_[ns] * OSTR + ocl) = packed[ns];
            }
        }
    }
    __syncthreads();
    const int dwords = min(BM, a.out_w - m0) >> 2;  // dwords of this M tile inside the output cell
    if (a.y) {
        // dword-granular copy-out: consecutive lanes -> consecutive dwords of a pixel's channel run (coalesced,
        // conflict-free LDS reads)
        if (dwords == BM / 4) {
            for (int p = tid; p < BN * (BM / 4); p += NT) {
                const int pix = p / (BM / 4), d = p % (BM / 4);
                const int cell = celltab[pix];
                if (cell >= 0)
                    *reinterpret_cast<uint32_t *>(a.y + (size_t)cell * a.out_cs + m0 + d * 4) =
                        *reinterpret_cast<const uint32_t *>(otile + pix * OSTR + d * 4);
            }
        } else {
            const int total = BN * dwords;
            for (int p = tid; p < total; p += NT) {
                const int pix = p / dwords, d = p - pix * dwords;
                const int cell = celltab[pix];
                if (cell >= 0)
                    *reinterpret_cast<uint32_t *>(a.y + (size_t)cell * a.out_cs + m0 + d * 4) =
                        *reinterpret_cast<const uint32_t *>(otile + pix * OSTR + d * 4);
            }
        }
    }
    if constexpr (PATCH) {
        // fused forward_maxpool_layer_quant (ref: src/maxpool_layer.c:109-172) for size 2 / stride 2 / offset 0 on even
        // maps: the TH x 16 patch starts on even coordinates, so it holds complete 2x2 windows; bytewise max of the
        // four (biased) uint8 values, written straight into the pooled PHWC tensor.
        if (a.ypool) {
            const int OH = a.H >> 1, OW = a.W >> 1;
            const int pdw = min(BM, a.pool_w - m0) >> 2;
            const int total = (TH / 2) * 8 * pdw;
            for (int p = tid; p < total; p += NT) {
                const int pp = p / pdw, d = p - pp * pdw;
                const int pr = pp >> 3, pc = pp & 7;
                const int oy = (py0 >> 1) + pr, ox = (px0 >> 1) + pc;
                if (oy < OH && ox < OW) {
                    const char *src = otile + ((2 * pr) * 16 + 2 * pc) * OSTR + d * 4;
                    uint32_t m = max_s8x4(*reinterpret_cast<const uint32_t *>(src),
                                          *reinterpret_cast<const uint32_t *>(src + OSTR));
                    m = max_s8x4(m, *reinterpret_cast<const uint32_t *>(src + 16 * OSTR));
                    m = max_s8x4(m, *reinterpret_cast<const uint32_t *>(src + 17 * OSTR));
                    const size_t cell = a.pool_lead + ((size_t)pb * (OH + 1) + (oy + 1)) * (OW + 1) + ox;
                    *reinterpret_cast<uint32_t *>(a.ypool + cell * a.pool_cs + m0 + d * 4) = m;
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// host-side launcher
// ---------------------------------------------------------------------------------------------------------------
static int g_force_nt = 0, g_force_bm = 0, g_force_bn = 0, g_force_patch = -1, g_force_generic = 0, g_no_rows = 0;
static int g_debug = 0;
int mi355_debug_flags_get() { return g_debug; }
extern "C" int mi355_debug_flags(int flags)
{
    g_debug = flags;
    return MI355_OK;
}
extern "C" int mi355_conv_set_tile(int bm, int bn)
{
    // bn > 0: force tile; bn encodes the N-tile mode in bit 30 (patch) / bit 29 (flat) for benchmarking
    g_force_patch = (bn & (1 << 30)) ? 1 : ((bn & (1 << 29)) ? 0 : -1);
    g_force_generic = (bn & (1 << 28)) ? 1 : 0;  // bit 28: force the generic K loop (tests)
    g_no_rows = (bn & (1 << 27)) ? 1 : 0;        // bit 27: do not use conv_rows.hip
    g_force_bm = bm & 0xFFFF;
    g_force_nt = (bm >> 16) & 0x7FFF;            // bm bits 16..30: force the N-tile count of conv_rows.hip
    g_force_bn = bn & 0xFFFF;
    return MI355_OK;
}

template <int BM, int BN, int WMW, int WNW, bool PATCH, int KMODE>
static int launch_cfg(ConvArgs &a, hipStream_t st)
{
    constexpr int NW = WMW * WNW, NT = 64 * NW;
    if (a.mpad % BM) return MI355_EINVAL;
    a.mtiles = a.mpad / BM;
    const int cpc = 1024 / a.cb;
    int ncell;
    if (PATCH) {
        constexpr int TH = BN / 16, TW = 16;
        a.tiles_x = (a.OW + TW - 1) / TW;
        a.tiles_y = (a.OH + TH - 1) / TH;
        a.ntiles_n = a.B * a.tiles_x * a.tiles_y;
        ncell = ((TH - 1) * a.stride + 3) * ((TW - 1) * a.stride + 3);  // input patch incl. the 3x3 halo
    } else {
        if (a.stride != 1) return MI355_EINVAL;
        a.ntiles_n = (a.total_n + BN - 1) / BN;
        const int halo = (a.ksize == 3) ? (a.W + 2) : 0;
        // pixels + row pads + image-boundary pad rows, + halo both sides
        int span = BN + (BN + a.W - 1) / a.W + 1 + ((BN + a.H * a.W - 1) / (a.H * a.W) + 1) * (a.W + 1);
        ncell = span + 2 * halo;
    }
    a.bchunks = (ncell + cpc - 1) / cpc;
    a.bpt = (a.bchunks + NW - 1) / NW;
    if (a.bpt > BPT_MAX || APT_OF(BM, NW) + a.bpt > 20) return MI355_EINVAL;
    size_t lds = (size_t)A_STAGES * BM * 64 + 2 * ((size_t)a.bchunks << 10);
    const size_t lds_epi = (size_t)BN * (BM + 4) + (size_t)BN * 4;
    if (lds_epi > lds) lds = lds_epi;
    if (lds > 160 * 1024) return MI355_EINVAL;
    auto kern = conv_igemm_i8_kernel<BM, BN, WMW, WNW, PATCH, KMODE>;
    // per kernel instantiation AND per device (function attributes are per device; `darknet -gpus` drives several devices
    // from one process): raise the dynamic-LDS limit once, not per launch
    static size_t lds_attr_dev[64] = {0};
    int dev_ix = 0;
    (void)hipGetDevice(&dev_ix);
    size_t &lds_attr = lds_attr_dev[dev_ix & 63];
    if (lds > 64 * 1024 && lds > lds_attr) {
        lds_attr = lds;
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)lds) != hipSuccess)
            return MI355_EHIP;
    }
    dim3 grid(a.ntiles_n * a.mtiles), block(NT);
    hipLaunchKernelGGL(kern, grid, block, lds, st, a);
    return hipGetLastError() == hipSuccess ? MI355_OK : MI355_EHIP;
}

template <bool PATCH, int KMODE>
static int launch_mode(ConvArgs &a, hipStream_t st, int bm, int bn)
{
    if (bm == 128 && bn == 256) return launch_cfg<128, 256, 2, 4, PATCH, KMODE>(a, st);
    if (bm == 128 && bn == 128) return launch_cfg<128, 128, 2, 2, PATCH, KMODE>(a, st);
    if (bm == 64 && bn == 256) return launch_cfg<64, 256, 1, 4, PATCH, KMODE>(a, st);
    if (bm == 64 && bn == 128) return launch_cfg<64, 128, 1, 4, PATCH, KMODE>(a, st);
    if (bm == 32 && bn == 256) return launch_cfg<32, 256, 1, 4, PATCH, KMODE>(a, st);
    if (bm == 32 && bn == 128) return launch_cfg<32, 128, 1, 4, PATCH, KMODE>(a, st);
    return MI355_EINVAL;
}

static int launch_any(ConvArgs &a, hipStream_t st, int bm, int bn, bool patch)
{
    const int kmode = (a.cb == 64 && !g_force_generic) ? (a.ksize == 3 ? 3 : 1) : 0;
    if (patch) {  // PATCH is only meaningful for 3x3
        if (kmode == 3) return launch_mode<true, 3>(a, st, bm, bn);
        return launch_mode<true, 0>(a, st, bm, bn);
    }
    if (kmode == 3) return launch_mode<false, 3>(a, st, bm, bn);
    if (kmode == 1) return launch_mode<false, 1>(a, st, bm, bn);
    return launch_mode<false, 0>(a, st, bm, bn);
}

int conv_igemm_launch(ConvArgs &a, hipStream_t st)
{
    if (a.res) return MI355_EINVAL;  // no fused residual add in the implicit-GEMM / row-image kernels: the caller runs the layers separately
    int bm = g_force_bm, bn = g_force_bn;
    if (!bm) bm = a.n >= 128 ? 128 : (a.n > 32 ? 64 : 32);
    a.debug = g_debug;
    if (a.cb == 64 && !g_force_generic && !g_no_rows && g_force_patch < 0 && !a.ypool && a.stride == 1) {
        // row-image kernel (conv_rows.hip).  Tile plan: N tiles split the pixel range evenly; pick the tile capacity
        // (384 / 256 / 128 columns) and the tile count that minimise  rounds x (fixed + K-steps x step time)  where a
        // round is one workgroup per CU for the 8-wave configurations and two for the 4-wave one (measured model,
        // profiles/r01_ablation.md: ~10 us fixed per workgroup, ~0.2 us + 0.04 us per MFMA of the busiest wave per step).
        int best_bn = 0, best_nt = 0;
        if (bn) {
            best_bn = bn; best_nt = g_force_nt;
        } else {
            double best = 1e30;
            const int mt = (a.mpad + bm - 1) / bm;
            const int cands[3] = {384, 256, 128};
            // throughput plan (several batches in flight): 128-column tiles only -- 4 waves, <= 80 KB of LDS, two workgroups
            // (of any two launches) per CU; the wide tiles remain the fallback for shapes the narrow one cannot serve
            const bool thr = plan_one_round(a) && a.ksize == 3 && !(g_debug & (1 << 28));
            for (int ci = thr ? 2 : 0; ci < 3; ++ci) {
                const int cbn = cands[ci];
                if (cbn == 384 && bm != 128) continue;
                const int nt0 = (a.total_n + cbn - 1) / cbn;
                const long blocks0 = (long)mt * nt0;
                // the 128-column configurations fit two workgroups per CU, but one per CU is faster while that is enough
                const int slots = (cbn == 128 && blocks0 > 256) ? 512 : 256;
                const int rounds = (int)((blocks0 + slots - 1) / slots);
                int nt = (int)(((long)rounds * slots) / mt);  // fill the last round with more, narrower tiles
                if (nt < nt0) nt = nt0;
                const int waves_n = (cbn == 128 && bm == 128) ? 2 : 4;
                int px = (a.total_n + nt - 1) / nt;
                int sub = ((px + 31) / 32 + waves_n - 1) / waves_n;   // sub-tiles of the busiest wave
                {   // ... unless the narrower tiles keep the busiest wave as busy: then more of them is only more work
                    const int px0 = (a.total_n + nt0 - 1) / nt0;
                    const int sub0 = ((px0 + 31) / 32 + waves_n - 1) / waves_n;
                    if (sub0 == sub && slots == 512) { nt = nt0; px = px0; }
                }
                const int ms = (bm >= 64) ? 2 : 1;
                const double step = 0.2 + 0.04 * (2.0 * ms * sub);
                // narrow tiles re-load the weights and the halo rows more often (3x3); for 1x1 they are the cheapest
                const double shape = (cbn == 128) ? (a.ksize == 3 ? 1.15 : 0.85) : 1.0;
                // maps that fill their LDS row image badly (19 + 2 cells in a 32-slot row, 38 + 2 in 64) make the wide
                // tiles pay for the empty slots in every B-slab DMA and LDS byte: measured 173 / 114 us with 256-column
                // tiles against 116 / 88 us with two 128-column workgroups per CU (256->512 @38, 512->1024 @19, batch 32)
                const int rs = a.W + 2 <= 16 ? 16 : (a.W + 2 <= 32 ? 32 : 64);
                // (128 x 384 tiles on 32-slot rows have a narrow-map variant in conv_rows16.hip: exact LDS rows, extra DMA slots)
                const bool narrow_ok = bm == 128 && cbn == 384 && rs == 32 && !(g_debug & (1 << 20));
                const double rowpen = (cbn != 128 && a.ksize == 3 && (a.W + 2) < 0.7 * rs && !narrow_ok) ? 1.5 : 1.0;
                // two 4-wave workgroups per CU are scheduled as they finish: count fractional rounds for them
                const double nrounds = (slots == 512) ? (double)((long)mt * nt) / slots : (double)rounds;
                const double cost = (nrounds < 1.0 ? 1.0 : nrounds) * (10.0 + a.ksteps * step) * shape * rowpen + 0.005 * cbn;
                if (g_debug & (1 << 25)) fprintf(stderr, "[plan] bn %d nt0 %d nt %d px %d sub %d rounds %.2f cost %.1f\n", cbn, nt0, nt, px, sub, nrounds, cost);
                if (cost < best) { best = cost; best_bn = cbn; best_nt = nt; }
            }
        }
        a.ntiles_n = best_nt;
        int rc = conv_rows_launch(a, st, bm, best_bn);
        if (rc == MI355_EINVAL && !bn) { a.ntiles_n = 0; rc = conv_rows_launch(a, st, bm, 128); }
        if (rc != MI355_EINVAL) return rc;
        a.ntiles_n = 0;
    }
    if (a.up != 1) return MI355_EINVAL;  // the fused upsample store exists in conv_rows.hip only
    // N-tile mode: PATCH when a 16-wide patch wastes little (W >= 24) -- its halo is (TH+2)x18 cells instead of two
    // full image rows; FLAT for the small maps where a patch would be mostly padding.
    bool patch = a.ksize == 3 && a.W >= 24 && a.H >= 8;
    if (g_force_patch >= 0) patch = g_force_patch == 1;
    if (a.ypool) patch = true;  // the fused 2x2 maxpool needs 2-D patches (complete windows)
    if (!bn) {
        bn = 256;
        long tiles;
        if (patch) tiles = (long)a.B * ((a.OW + 15) / 16) * ((a.OH + 15) / 16);
        else tiles = (a.total_n + 255) / 256;
        tiles *= (a.n + bm - 1) / bm;
        if (tiles < 200) bn = 128;  // measured: 256-wide tiles win down to ~0.8 workgroups per CU (r01 sweep)
        // ... on few-channel stride-1 layers.  With 64-channel chunks (the layers the row-image kernel refuses: maps wider
        // than 62 pixels, row images it cannot tile) and for stride 2, two 128-column workgroups per CU win: the YOLOv3
        // shapes at 608 x 608, batch 32, run 470 -> 278 us (3x3 s2 32->64), 163 -> 103 us (256->512 @38), 106 -> 80 us
        // (512->1024 @19) -- profiles/r01_v6_chain608_layers.log
        if (a.cb == 64 || a.stride != 1) bn = 128;
    }
    if (a.ksize == 1) patch = false;
    if (a.stride != 1) patch = true;  // strided convs exist as 2-D patches only (input patch = stride x the output patch)
    int rc = launch_any(a, st, bm, bn, patch);
    if (rc == MI355_EINVAL && !(g_force_bm || g_force_bn)) {
        // staging budget exceeded (very wide rows in FLAT mode): fall back to the other mode / narrower tile
        rc = launch_any(a, st, bm, 128, (a.ypool || a.stride != 1) ? true : (a.ksize == 3 ? !patch : false));
    }
    return rc;
}
