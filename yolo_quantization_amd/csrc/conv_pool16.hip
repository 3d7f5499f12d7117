// conv_pool16.hip -- 3x3 s1 p1 INT8 convolution 16 -> 32 channels fused with the 2x2 / stride-2 maxpool behind it (layer 2 of yolov3-tiny), on
// V_MFMA_I32_16X16X64_I8 in the first-layer kernel's shape (conv_aux.hip) -- round 5.
//
// conv_small.hip serves these layers on 32 x 32 x 32 tiles: a lane holds SIXTEEN channels of one pooled pixel, its epilogue costs ~61 VALU
// instructions per four pooled outputs (the zero-point correction as a multiply-add per accumulator, LEAKY in arithmetic because its LDS -- one
// 1 KiB B-fragment read per MFMA -- has no room for a byte table), and MFMA-busy + VALU-busy + LDS-busy clocks add up to its run time.  Here:
//   * one MFMA = 16 pooled pixels x 16 filters x 64 K-slots (four taps x 16 channels); lane (pc = lane & 15, g = lane >> 4) holds FOUR channels
//     of pooled pixel pc: the four MFMAs of a set are the four positions of its 2x2 window, as in the first layer;
//   * a B fragment (16 bytes per lane: one image cell) is read ONCE per unit and used by both 16-filter m-tiles and by the "ones" tile (A = 1 in
//     every real K-slot) that yields the receptive-field sum of x' for the signed-operand correction: three MFMAs per 16-byte read instead of
//     one -- which leaves LDS bandwidth for the per-channel constants and the LEAKY byte table;
//   * the epilogue is the first layer's: biased accumulators (seed = cw + bias - lo as the MFMA's C operand), one unsigned window maximum, the range
//     test carried as a per-lane margin and looked at once per tile (a wave that sees a zero redoes its two pooled rows in the reference's order),
//     integer requantisation, byte table: ~45 VALU per four pooled outputs x 4 channels;
//   * A fragments come straight from the blob's K-ordered weight plane (`wp`: [16 filters][K-step][piece][row][16 B] IS the A operand's lane
//     order); per-channel constants from the host-derived epilogue table (mi355_conv_pack_epilogue); the image of an 8 x 16 pooled patch is DMAed
//     (global_load_lds) into a double-buffered, column-parity de-interleaved LDS plane per 16 channels, one tile ahead.
// Domain: c = 16, n = 32, even maps, pooled output only, the blob finished with mi355_conv_pack_epilogue for this activation / zero point
// (mi355_conv_desc.epilogue_packed).  A blob whose key does not match still gives the right bytes: every window then takes the exact path (slow).
// Same mathematics and the same bytes as conv_small.hip (tests/test_gpu_parity.py runs both on the same calls).
// Measured (profiles/r05_pool16_*): layer 2 alone 35 -> 31 us, in flight 30.0 -> 28.9 us.  The counters say where the time is: per SIMD 26 K
// clocks of MFMA + 21 K of VALU in a 59 K-clock launch (they do not overlap) and a wave waits 39 % of its life for an operand -- hence the
// hand-ordered phases of the fast path below.  Built and NOT kept: the 32 -> 64 form (five K-steps, four m-tiles: 256 registers with spills,
// 22.5 us against conv_small.hip's 17.5), and the transposed form (image as the A operand, so that a lane holds ONE channel and its constants
// are six registers instead of LDS reads, + a 4 x 4 byte transpose per quad before the store: bit-identical, 29.8-30.0 us at two waves per
// SIMD, 40 us at three with spills).
#include "kargs.h"
#include <cstdlib>
#include <type_traits>

constexpr int P16_PITCH = 40;              // 16-byte slots per LDS image row: even cells 0, 2, .. in slots 0 .., odd cells in slots P16_HC ..
constexpr int P16_HC = 20;
constexpr int P16_NDMA = 12;               // DMA instructions (64 slots each) per 16-channel plane: 18 rows x 40 slots = 720 <= 768
constexpr int P16_PLANEB = P16_NDMA * 1024;

#define P16_DMA(ldsdst_u32, sbase_ptr, voff_u32)                                                                  \
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(ldsdst_u32), "v"(voff_u32), \
                 "s"(sbase_ptr)                                                                                  \
                 : "memory", "m0")

template <int ACT, bool SAT>
__global__ __launch_bounds__(256, 3) void conv_pool16_kernel(const ConvArgs a)
{
    constexpr int C = 16, NM = 2, KS = 3, N = 16 * NM;  // K-step ks, k-group g: tap 4 ks + g (taps 9 .. 11: zero weights)
    constexpr int BUFB = P16_PLANEB;
    constexpr bool LUT = ACT == MI355_ACT_LEAKY && !SAT;
    constexpr bool INTRQ = LUT || ACT == MI355_ACT_RELU6;
    __shared__ __attribute__((aligned(16))) char img[2 * BUFB];
    __shared__ __attribute__((aligned(16))) uint8_t lut[LUT ? LUTQ_N : 16];
    __shared__ __attribute__((aligned(16))) int ldsCB[N], ldsHI[N], ldsLO[N], ldsM0[N], ldsSH[N], ldsDZ[N];
    __shared__ __attribute__((aligned(16))) double ldsMP[INTRQ ? 2 : N];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pc = lane & 15, g = lane >> 4;
    const int W1 = a.W + 1;
    const int OH = a.H >> 1, OW = a.W >> 1;
    const int tiles_x = (OW + 15) >> 4, tiles_y = (OH + 7) >> 3, tpi = tiles_x * tiles_y;
    const int ntiles = a.B * tpi;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char *)img;

    // ---- image DMA: instruction k of a plane fills slots [64 k, 64 k + 64); wave w issues k = w, w + 4, w + 8.  The lane's cell offset from the patch
    //      origin (image row 16 ty - 1, column 32 tx - 1) does not depend on the tile
    int relc[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int q = 64 * (wave + 4 * i) + lane;
        const int r = min(q / P16_PITCH, 17), sl = q % P16_PITCH;
        const int half = sl >= P16_HC ? 1 : 0, ci = min(2 * (sl - P16_HC * half) + half, 33);
        relc[i] = r * W1 + ci;
    }
    const int maxcell = a.in_cells - 1;
    auto issue_tile = [&](unsigned org, int buf) {
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int f = min(max((int)org + relc[i], 0), maxcell);
            const unsigned voff = (unsigned)f * (unsigned)C;  // in_cs == C (launcher); < 2^32 (launcher)
            const unsigned d = lds0 + (unsigned)buf * BUFB + (unsigned)(wave + 4 * i) * 1024u;
            P16_DMA(d, a.x, voff);
        }
    };

    // ---- the workgroup's tiles, one lane per tile (see conv_first_mfma_pool_kernel): input origin (cell), output offset (bytes), flags
    const bool xcd_walk = (gridDim.x & 7) == 0 && !(a.debug & 2048);
    const int per_x = xcd_walk ? (ntiles + 7) >> 3 : ntiles;
    const int tstride = xcd_walk ? (int)(gridDim.x >> 3) : (int)gridDim.x;
    const int tbase_x = xcd_walk ? (int)(blockIdx.x & 7) * per_x : 0;
    const int tend = min(tbase_x + per_x, ntiles);
    const int tile0 = tbase_x + (xcd_walk ? (int)(blockIdx.x >> 3) : (int)blockIdx.x);
    unsigned T_in, T_out, T_fl;
    int nt;
    {
        const int t = tile0 + lane * tstride;
        nt = __builtin_popcountll(__builtin_amdgcn_ballot_w64(t < tend));  // <= 64 (launcher)
        const int tc = min(t, ntiles - 1);
        const int b = fd_div(tc, a.fd_hw);  // (fd_hw / fd_w: the launcher's divisions by tiles per image / tiles per row)
        const int r = tc - b * tpi;
        const int ty = fd_div(r, a.fd_w), tx = r - ty * tiles_x;
        T_in = (unsigned)(a.in_lead + (b * (a.H + 1) + 16 * ty) * W1 + 32 * tx - 1);
        T_out = (unsigned)(a.pool_lead + (b * (OH + 1) + 8 * ty + 1) * (OW + 1) + 16 * tx) * (unsigned)a.pool_cs;
        const bool tall = 16 * tx + 16 <= OW && 8 * ty + 8 <= OH;
        T_fl = (tall ? 2u : 0u) | ((unsigned)tx << 2) | ((unsigned)ty << 12);
    }
    auto tile_word = [&](unsigned v, int k) { return (unsigned)__builtin_amdgcn_readlane((int)v, k); };

    // (the first tile's image is requested BEFORE the constants, the byte table and the A fragments are fetched: their latencies overlap)
    if (nt > 0) issue_tile(tile_word(T_in, 0), 0);

    // ---- per-channel constants (LDS), byte table, A fragments (registers)
    const bool ept_ok = a.ept != nullptr && a.ept->key == ept_key(ACT, a.zp_act);  // workgroup-uniform
    const uint32_t eflags = ept_ok ? a.ept->flags : (EPT_NEVER | EPT_NOINT);
    const bool pow2 = a.hdr->pow2 == 1;
    // fast: every window maximum may be requantised in its integer / FP64-of-maximum form unless its margin says otherwise
    const bool fast = ept_ok && pow2 && !(eflags & EPT_NEVER) && (!INTRQ || !(eflags & EPT_NOINT));
    if (tid < N) {
        int cbv = a.cwb[tid], hiv = 0, lov = 0, m0v = 0, shv = 0;
        if (ept_ok) {
            const EptEntry e = reinterpret_cast<const EptEntry *>(a.ept + 1)[tid];
            cbv = e.cbl; hiv = (int)e.rg; lov = e.lb; m0v = e.m0; shv = e.sh;
        }
        ldsCB[tid] = cbv; ldsHI[tid] = hiv; ldsLO[tid] = lov; ldsM0[tid] = m0v; ldsSH[tid] = shv;
        ldsDZ[tid] = a.dzp[tid];
        if (!INTRQ) ldsMP[tid] = a.mprime[tid];
    }
    if constexpr (LUT) {
        if (ept_ok) {  // the host's table for this (activation, zero point): behind the entries (shim.hip ept_fill)
            const uint32_t *src = reinterpret_cast<const uint32_t *>(reinterpret_cast<const EptEntry *>(a.ept + 1) + a.hdr->mpad);
#pragma unroll
            for (int k = 0; k < LUTQ_N / 4 / 256; ++k) reinterpret_cast<uint32_t *>(lut)[tid + 256 * k] = src[tid + 256 * k];
        }
    }
    v4i wa[NM][KS], ones[KS];
    int tapk[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        const int t = 4 * ks + g;  // the blob's unit order
        tapk[ks] = t > 8 ? 8 : t;  // slots past tap 8 carry zero weights; their B operand re-reads tap 8's cell
        const int o = t <= 8 ? 0x01010101 : 0;
        ones[ks] = v4i{o, o, o, o};
#pragma unroll
        for (int mt = 0; mt < NM; ++mt)
            wa[mt][ks] = *reinterpret_cast<const v4i *>(a.wp + ((size_t)(mt * KS + ks) * 1024) + g * 256 + pc * 16);
    }

    // ---- B operand addresses: LDS row 4 wave + dy, slot pc + f(jx + dx), f(e) = (e & 1) HC + (e >> 1); (2 s + jy) rows further down is an immediate
    unsigned baddr[KS][2];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        const int dy = tapk[ks] / 3, dx = tapk[ks] - 3 * dy;
#pragma unroll
        for (int jx = 0; jx < 2; ++jx) {
            const int e = jx + dx;
            baddr[ks][jx] = (unsigned)(((4 * wave + dy) * P16_PITCH + pc + (e & 1) * P16_HC + (e >> 1)) * 16);
        }
    }

    // ---- deferred stores (one tile late, behind the next tile's DMA): hand-written, scalar base + lane offset
    const unsigned rowpitch = (unsigned)(OW + 1) * (unsigned)a.pool_cs;
    unsigned st_off[2];  // pooled rows 2 wave, 2 wave + 1 of the patch, the lane's pooled column, channels 4 g .. (+ 16 mt: an immediate)
#pragma unroll
    for (int s = 0; s < 2; ++s) st_off[s] = (unsigned)(2 * wave + s) * rowpitch + (unsigned)pc * (unsigned)a.pool_cs + 4u * g;
    uint32_t dpk[2][NM];
    unsigned d_out = 0;
    bool dvalid[2] = {false, false};
    auto store_patch = [&](unsigned patch_off, unsigned off, uint32_t data, auto mt_c) {
        constexpr int MT = decltype(mt_c)::value;
        const uint64_t rp = reinterpret_cast<uint64_t>(a.ypool) + (uint64_t)patch_off;
        const uint64_t rs = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(rp >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)rp);
        asm volatile("global_store_dword %0, %1, %2 offset:%3" ::"v"(off), "v"(data), "s"(rs), "n"(16 * MT) : "memory");
    };
    auto flush_stores = [&]() {
#pragma unroll
        for (int s = 0; s < 2; ++s)
            if (dvalid[s]) {
                store_patch(d_out, st_off[s], dpk[s][0], std::integral_constant<int, 0>{});
                store_patch(d_out, st_off[s], dpk[s][1], std::integral_constant<int, 1>{});
            }
    };

    unsigned fl_cur = tile_word(T_fl, 0);
    int buf = 0;
    for (int k = 0; k < nt; ++k, buf ^= 1) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's share of the tile's image (and the previous tile's stores) landed
        __syncthreads();                                   // ... everyone's; every wave is past the previous tile (whose buffer the next DMA overwrites)
        const bool more = k + 1 < nt;
        unsigned fl_nxt = 0;
        if (more) {
            fl_nxt = tile_word(T_fl, k + 1);
            issue_tile(tile_word(T_in, k + 1), buf ^ 1);
        }
        flush_stores();
        const char *const ibp = img + (size_t)buf * BUFB;
        uint32_t margin = 0xFFFFFFFFu;
        auto load_b = [&](int s, v4i (&bf)[KS][4]) {  // the B fragments of one pooled row of the wave: 3 K-steps x 4 window positions
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    bf[ks][j] = *reinterpret_cast<const v4i *>(ibp + baddr[ks][j & 1] + (unsigned)((2 * s + (j >> 1)) * P16_PITCH * 16));
        };
        auto chain_ones = [&](const v4i (&bf)[KS][4], v4i (&sxa)[4]) {  // every row of the ones tile: the sum of x' under the filter
            __builtin_amdgcn_s_setprio(3);
#pragma unroll
            for (int j = 0; j < 4; ++j) sxa[j] = v4i{0, 0, 0, 0};
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                for (int j = 0; j < 4; ++j) sxa[j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(ones[ks], bf[ks][j], sxa[j], 0, 0, 0);
            __builtin_amdgcn_s_setprio(0);
        };
        auto chain_mt = [&](int mt, const v4i (&bf)[KS][4], const v4i &cb, v4i (&acc)[4]) {
            __builtin_amdgcn_s_setprio(3);
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(wa[mt][0], bf[0][j], cb, 0, 0, 0);
#pragma unroll
            for (int ks = 1; ks < KS; ++ks)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(wa[mt][ks], bf[ks][j], acc[j], 0, 0, 0);
            __builtin_amdgcn_s_setprio(0);
        };
        auto correct = [&](v4i (&acc)[4], const v4i &dz, const v4i (&sxa)[4]) {  // + (128 - zp_w) * sum(x')
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[j][r] += __mul24(dz[r], sxa[j][0]);
        };
        if (fast) {
            // Both pooled rows x both m-tiles as four phases in a hand-made order (the scheduler is fenced between them): a phase's constants are
            // read behind its MFMA chain's issue and land while the chain drains; its byte-table reads land during the NEXT phase's chain and are
            // packed behind it; the second row's B fragments are read as soon as the first row's last chain has issued.
            v4i bf[KS][4], sxa[4];
            uint32_t bt[2][4];
            load_b(0, bf);
            v4i cb = *reinterpret_cast<const v4i *>(ldsCB + 4 * g);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int s = q >> 1, mt = q & 1, c0 = 16 * mt + 4 * g;
                if (mt == 0) chain_ones(bf, sxa);
                v4i acc[4];
                chain_mt(mt, bf, cb, acc);
                const v4i dz = *reinterpret_cast<const v4i *>(ldsDZ + c0);
                const v4i hi = *reinterpret_cast<const v4i *>(ldsHI + c0);
                const v4i lo = *reinterpret_cast<const v4i *>(ldsLO + c0);
                v4i m0 = v4i{0, 0, 0, 0}, sh = v4i{0, 0, 0, 0};
                if constexpr (INTRQ) {
                    m0 = *reinterpret_cast<const v4i *>(ldsM0 + c0);
                    sh = *reinterpret_cast<const v4i *>(ldsSH + c0);
                }
                if (q < 3) cb = *reinterpret_cast<const v4i *>(ldsCB + (16 * (mt ^ 1) + 4 * g));
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (LUT) {
                    if (q > 0) {  // the previous phase's bytes
                        const uint32_t pk = pack4_bytes(bt[(q - 1) & 1][0], bt[(q - 1) & 1][1], bt[(q - 1) & 1][2], bt[(q - 1) & 1][3]);
                        dpk[(q - 1) >> 1][(q - 1) & 1] = pk;
                    }
                }
                correct(acc, dz, sxa);
                uint32_t umax[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    umax[r] = max(max((uint32_t)acc[0][r], (uint32_t)acc[1][r]), max((uint32_t)acc[2][r], (uint32_t)acc[3][r]));
                    margin = min(margin, __builtin_elementwise_sub_sat((uint32_t)hi[r], umax[r]));
                }
                if constexpr (INTRQ) {
                    int32_t f[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) f[r] = __mulhi((int32_t)(umax[r] + (uint32_t)lo[r]), m0[r]) >> sh[r];
                    if constexpr (LUT) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) bt[q & 1][r] = lut[lutq_index(f[r])];  // (beyond the range: anywhere, even outside the allocation -> 0; redone below)
                    } else {  // RELU6: zp + max(q, 0) == zp + max(f, 0); SAT clamps
                        int32_t v[4];
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            v[r] = a.zp_act + max(f[r], 0);
                            if (SAT) v[r] = min(v[r], 255);
                        }
                        dpk[s][mt] = pack4_biased(v[0], v[1], v[2], v[3]);
                    }
                } else {
                    int32_t amax[4][1], v1[4][1];
                    double mp4[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        amax[r][0] = (int32_t)(umax[r] + (uint32_t)lo[r]);
                        mp4[r] = ldsMP[c0 + r];
                    }
                    requant_values<ACT, SAT, 1>(amax, mp4, a.zp_act, v1);
                    dpk[s][mt] = pack4_biased(v1[0][0], v1[1][0], v1[2][0], v1[3][0]);
                }
                if (q == 1) load_b(1, bf);
                __builtin_amdgcn_sched_barrier(0);
            }
            if constexpr (LUT) dpk[1][1] = pack4_bytes(bt[1][0], bt[1][1], bt[1][2], bt[1][3]);
        }
        if (!fast || __builtin_amdgcn_ballot_w64(margin == 0u) != 0) {  // some window of this wave may wrap (or the launch has no fast form): the reference's order
#pragma unroll 1
            for (int s = 0; s < 2; ++s) {
                v4i bf[KS][4], sxa[4];
                load_b(s, bf);
                chain_ones(bf, sxa);
#pragma unroll
                for (int mt = 0; mt < NM; ++mt) {
                    const int c0 = 16 * mt + 4 * g;
                    v4i acc[4];
                    chain_mt(mt, bf, *reinterpret_cast<const v4i *>(ldsCB + c0), acc);
                    correct(acc, *reinterpret_cast<const v4i *>(ldsDZ + c0), sxa);
                    const v4i lo = *reinterpret_cast<const v4i *>(ldsLO + c0);
                    double mpr[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) mpr[r] = a.mprime[c0 + r];
                    const uint32_t packed = first_pool_exact_path<ACT, SAT>(acc, lo, mpr, a.mval + c0, a.sval + c0, a.zp_act, pow2);
                    if (s == 0) dpk[0][mt] = packed; else dpk[1][mt] = packed;
                }
            }
        }
        // this tile's deferred stores: lanes outside the pooled map (ragged right / lower patches) are masked
        d_out = tile_word(T_out, k);
        {
            const int tx = (int)((fl_cur >> 2) & 1023u), ty = (int)((fl_cur >> 12) & 1023u);
            const bool all = (fl_cur & 2u) != 0;
            const bool colvalid = all || 16 * tx + pc < OW;
            dvalid[0] = colvalid && (all || 8 * ty + 2 * wave < OH);
            dvalid[1] = colvalid && (all || 8 * ty + 2 * wave + 1 < OH);
        }
        fl_cur = fl_nxt;
    }
    flush_stores();
}

bool conv_pool16_eligible(int n, int c, int ksize) { return ksize == 3 && c == 16 && n == 32; }

template <int ACT>
static int p16_launch_sat(ConvArgs &a, hipStream_t st, int grid)
{
    if (a.store_mode == MI355_STORE_SATURATE) hipLaunchKernelGGL((conv_pool16_kernel<ACT, true>), dim3(grid), dim3(256), 0, st, a);
    else hipLaunchKernelGGL((conv_pool16_kernel<ACT, false>), dim3(grid), dim3(256), 0, st, a);
    return hipGetLastError() == hipSuccess ? MI355_OK : MI355_EHIP;
}

// returns MI355_EINVAL when the shape is outside this kernel's domain (the caller falls back to conv_small.hip)
int conv_pool16_launch(ConvArgs &a, hipStream_t st)
{
    const int c = a.cb * a.nchunks;
    if (!conv_pool16_eligible(a.n, c, a.ksize) || !a.ypool || a.y || a.acc_out || a.y_f32 || a.res || a.stride != 1 || !a.ept) return MI355_EINVAL;
    if ((a.H & 1) || (a.W & 1) || a.in_cs != c || a.in_lead < 1 || a.pool_w < a.n) return MI355_EINVAL;
    if ((size_t)a.in_cells * (size_t)a.in_cs >= ((size_t)1 << 31)) return MI355_EINVAL;  // 32-bit cell / byte arithmetic in the kernel
    const int OH = a.H / 2, OW = a.W / 2;
    if (((long)a.pool_lead + (long)a.B * (OH + 1) * (OW + 1) + OW + 2) * a.pool_cs >= (1L << 32)) return MI355_EINVAL;
    const int tx = (OW + 15) / 16, ty = (OH + 7) / 8;
    if (tx > 1023 || ty > 1023) return MI355_EINVAL;
    const long ntiles = (long)a.B * tx * ty;
    if (ntiles >= (1L << 31)) return MI355_EINVAL;
    a.fd_w = fastdiv_make((uint32_t)tx);
    a.fd_hw = fastdiv_make((uint32_t)(tx * ty));
    a.debug = mi355_debug_flags_get();
    // persistent: three workgroups per CU, never more than 64 tiles per workgroup (one lane per tile)
    // (measured: 384 / 512 workgroups are 1.0-1.5 us faster when the layer floods the chip with itself, 3-9 us slower alone, and the in-flight STEP is 1.5-2 us
    // slower with them: 768 stays)
    static const int grid_cap = getenv("MI355_P16_GRID") ? atoi(getenv("MI355_P16_GRID")) : 768;  // (A/B runs)
    long g = grid_cap;
    if (ntiles < g) g = ntiles;
    const long need = (((ntiles + 7) / 8 + 63) / 64) * 8;
    if (g < need) g = need;
    if (a.act == MI355_ACT_LEAKY) return p16_launch_sat<MI355_ACT_LEAKY>(a, st, (int)g);
    if (a.act == MI355_ACT_RELU6) return p16_launch_sat<MI355_ACT_RELU6>(a, st, (int)g);
    return p16_launch_sat<MI355_ACT_LINEAR>(a, st, (int)g);
}
