// conv_small32.hip -- 3x3 s1 p1 INT8 convolution 32 -> 64 channels fused with the 2x2 / stride-2 maxpool behind it (layer 4 of yolov3-tiny):
// conv_small.hip's kernel re-cut for OCCUPANCY (round 6).
//
// conv_small_pool_kernel<32, 2> gives a wave both 32-filter m-tiles of 32 pooled pixels: 72 registers of stationary A fragments + 64 accumulator
// registers (four window positions x 16) put it at 216-244 VGPRs = TWO waves per SIMD, and the counters of round 5 showed what that costs: per
// SIMD 17.5 K clocks of VALU + 12.2 K of MFMA + 9.7 K of LDS in a 54 K-clock launch -- every pipe under a third busy, a quarter of the time
// nothing at all (profiles/r05_v4_pmc_sq*_plan1.txt).  Two waves that leave the same barrier run the same phase at the same time.  Here:
//   * EIGHT waves per workgroup: wave w owns m-tile w >> 2 (32 filters: 36 registers of A fragments) of pixel group w & 3 -- the B fragments were
//     read once per m-tile before as well (the old kernel's m-tile loop is outermost), so no LDS traffic is added; the per-lane geometry and box
//     sums are computed twice;
//   * the 2x2 window is processed in two HALVES (its two rows): 32 accumulator registers instead of 64.  After the first half only the unsigned
//     maximum of the two biased accumulators is kept per channel (16 registers); the range test runs per half, so a half that fails is requantised
//     in the reference's order on the spot and leaves its byte maximum behind instead -- the maximum of BYTES is associative, which is all the
//     reference's order (src/convolutional_layer.c:737-749 then src/maxpool_layer.c:134-146) needs.  A clean window still costs ONE requantisation;
//   * <= 128 VGPRs -> four waves per SIMD, two workgroups per CU (LDS: the same double-buffered image as before, 81.6 KB for layer 4);
//   * `noint` and `never` are separate: a channel that fails the integer form's conditions sends the launch to the FP64-of-maximum
//     fast path, not to the exact path for every window.
// Same mathematics, tile geometry, LDS image (column-parity de-interleaved rows), DMA and deferred stores as conv_small.hip: read its header.
// MEASURED (profiles/r06_small32_ab_flood.log, same box, three rounds): layer 4 alone 23.3 -> 24.1 us, with four batches in flight 18.5 -> 19.2 us -- SLOWER than
// conv_small.hip; with ONE eight-wave workgroup per CU 28.0 / 18.1 us.  Twice the resident waves buy nothing here: kept for the record behind debug bit
// 4096, not in the default path.
// Domain: c = 32, n = 64, even maps, pooled output only.  Bytes identical to conv_small.hip (tests/test_gpu_parity.py runs both on the same calls).
#include "kargs.h"
#include <cstdlib>

#define DMA_S32(ldsdst_u32, sbase_ptr, voff_u32)                                                                 \
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(ldsdst_u32), "v"(voff_u32), \
                 "s"(sbase_ptr)                                                                                  \
                 : "memory", "m0")

constexpr int S32_PPB = 128;  // pooled pixels per workgroup tile (4 pixel groups x 32 lanes)
constexpr int S32_KDMA = 3;   // DMA instructions per wave and piece per tile image (image <= 8 * 3 * 64 cells)

// The reference's order for NS accumulators of one channel: every value requantised (src/convolutional_layer.c:732-749), then the maximum of
// the stored BYTES (src/maxpool_layer.c:134-146).  acc: true accumulators; mp: the folded multiplier (valid when pow2).
template <int ACT, bool SAT, int NS>
__device__ __forceinline__ int32_t s32_byte_max(const int32_t (&acc)[NS], double mp, const double *mval, const double *sval, int zp_act, bool pow2)
{
    int32_t m = 0;
    if (pow2) {
#pragma unroll
        for (int k = 0; k < NS; ++k) m = max(m, (int32_t)requant_finish<ACT, SAT>(requant_q_exact(acc[k], mp), zp_act));
    } else {  // shift_value not a power of two: the two-step form (never produced by the reference's own preparation)
        const double mv = *mval, sv = *sval;
#pragma unroll
        for (int k = 0; k < NS; ++k) m = max(m, (int32_t)requant_u8(acc[k], 0, mv, sv, zp_act, ACT, SAT ? MI355_STORE_SATURATE : MI355_STORE_WRAP));
    }
    return m;
}

template <int ACT, bool SAT>
__global__ __launch_bounds__(512, 4) void conv_small32_kernel(const ConvArgs a)
{
    constexpr int KST = 9, PIECES = 2, N = 64;
    constexpr bool INTRQC = (ACT == MI355_ACT_LEAKY || ACT == MI355_ACT_RELU6) && !SAT;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int ncell = a.sm_ncell, lcell = a.sm_lcell;
    const int hc = a.sm_hc, hcb = hc * 16;
    const int rowb = ncell * 16, pieceb = a.sm_pieceb, bbytes = PIECES * pieceb;
    const bool patch = a.tiles_x > 0;
    int *ldsS = reinterpret_cast<int *>(smem + 2 * bbytes);               // [rows_cap * ncell] per-cell channel sums
    double *ldsMP = reinterpret_cast<double *>(smem + a.lds_param_off);   // [N] folded multiplier
    int *ldsDZ = reinterpret_cast<int *>(ldsMP + N);                      // [N] 128 - zp_w
    int *ldsCB = ldsDZ + N;                                               // [N] cw + bias - lo (biased seed)
    int *ldsLO = ldsCB + N, *ldsHI = ldsLO + N;                           // [N] lower end and width of the wrap-safe range
    int *ldsM0 = ldsHI + N, *ldsSH = ldsM0 + N;                           // [N] integer requantisation
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char *)smem;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pw = wave & 3, mt = wave >> 2;  // pixel group, m-tile of this wave
    const int chb = 32 * mt;
    const int kh = lane >> 5, lj = lane & 31;
    const int W1 = a.W + 1;
    const int OH = a.H >> 1, OW = a.W >> 1, ohw = OH * OW;
    const int total_p = a.B * ohw;
    const int tpi = a.tiles_x * a.tiles_y;
    const int ntiles = patch ? a.B * tpi : (total_p + S32_PPB - 1) / S32_PPB;
    const bool pow2 = a.hdr->pow2 == 1;

    // ---- per-channel parameters (see conv_small_pool_kernel)
    int never_l = 0, noint_l = 0;
    const bool ept_ok = !SAT && a.ept != nullptr && a.ept->key == ept_key(ACT, a.zp_act);
    if (tid < N) {
        const double mp = a.mprime[tid];
        ldsMP[tid] = mp;
        ldsDZ[tid] = a.dzp[tid];
        int32_t m0 = 0, sh = 0;
        if (ept_ok) {
            const EptEntry e = reinterpret_cast<const EptEntry *>(a.ept + 1)[tid];
            ldsCB[tid] = e.cbl;
            ldsLO[tid] = e.lb;
            ldsHI[tid] = (int32_t)e.rg;
            m0 = e.m0; sh = e.sh;
        } else {
            int32_t lo = -2147483647 - 1, hi = 2147483647;
            if (!SAT) small_safe_range<ACT>(mp, a.zp_act, lo, hi);
            int32_t lb = 0; uint32_t rg = 0;
            if (!biased_safe_range(lo, hi, lb, rg)) never_l = 1;
            if (!(pow2 && intrq_make(a.mval[tid], a.shift[tid], lb, (int32_t)((uint32_t)lb + rg), m0, sh, ACT == MI355_ACT_RELU6))) noint_l = 1;
            ldsCB[tid] = (int32_t)((uint32_t)a.cwb[tid] - (uint32_t)lb);
            ldsLO[tid] = lb;
            ldsHI[tid] = (int32_t)rg;
        }
        ldsM0[tid] = m0;
        ldsSH[tid] = sh;
    }
    bool never, use_int;
    if (ept_ok) {
        never = (a.ept->flags & EPT_NEVER) != 0;
        use_int = INTRQC && (a.ept->flags & EPT_NOINT) == 0;
        __syncthreads();
    } else {
        never = __syncthreads_or(never_l) != 0;
        use_int = INTRQC && __syncthreads_or(noint_l) == 0;
    }
    never = never || !pow2;  // (shifts that are no powers of two: the reference's two-step form for every value; never produced by its own prep)

    // ---- stationary A fragments of this wave's m-tile
    v4i wf[KST];
#pragma unroll
    for (int s = 0; s < KST; ++s) wf[s] = *reinterpret_cast<const v4i *>(a.ws + ((size_t)(mt * KST + s) * 64 + lane) * 16);

    // ---- image DMA (see conv_small_pool_kernel): instruction k fills slots [64 k, 64 k + 64) of every piece plane; wave w issues k = w, w + 8, w + 16
    int doff[S32_KDMA], dstart[S32_KDMA];
#pragma unroll
    for (int i = 0; i < S32_KDMA; ++i) {
        const int k = wave + 8 * i;
        dstart[i] = min(k * 64, a.rows_cap * ncell - 64);
        const int slot = dstart[i] + lane;
        const int r = slot / ncell, cs = slot - r * ncell;
        const int c = min(cs < hc ? 2 * cs : 2 * (cs - hc) + 1, lcell - 1);
        doff[i] = r * W1 + c;
    }
    auto issue_tile = [&](int gr_first, int col0, int nrows, int parity) {
        const unsigned buf = lds0 + parity * bbytes;
        const int ncells = nrows * ncell;
        const long org = (long)a.in_lead + (long)(gr_first - 1) * W1 + col0;
#pragma unroll
        for (int i = 0; i < S32_KDMA; ++i)
            if ((wave + 8 * i) * 64 < ncells) {
                long f = org + doff[i];
                f = f < 0 ? 0 : (f > a.in_cells - 1 ? a.in_cells - 1 : f);
                const unsigned voff = (unsigned)(f * a.in_cs);
#pragma unroll
                for (int p = 0; p < PIECES; ++p) {
                    const unsigned dst = buf + p * pieceb + dstart[i] * 16;
                    const unsigned v = voff + p * 16;
                    DMA_S32(dst, a.x, v);
                }
            }
    };
    struct Pos { int b, ty, tx; };
    auto pos_of = [&](int t) {
        Pos p{0, 0, 0};
        if (patch) {
            p.b = t / tpi;
            const int r = t - p.b * tpi;
            p.ty = r / a.tiles_x;
            p.tx = r - p.ty * a.tiles_x;
        }
        return p;
    };
    // XCD-aware tile order: every XCD one contiguous share of the tiles (neighbouring tiles share halo rows through its L2)
    const bool xcd_walk = (gridDim.x & 7) == 0 && !(a.debug & 2048);
    const int tq = ntiles >> 3, tr = ntiles & 7, xcd = (int)(blockIdx.x & 7);
    const int tstride = xcd_walk ? (int)(gridDim.x >> 3) : (int)gridDim.x;
    const int tfirst = xcd_walk ? xcd * tq + min(xcd, tr) + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
    const int tend = xcd_walk ? (xcd + 1) * tq + min(xcd + 1, tr) : ntiles;
    const Pos pstep = pos_of(tstride);
    auto advance = [&](Pos &p) {
        p.tx += pstep.tx; p.ty += pstep.ty; p.b += pstep.b;
        if (p.tx >= a.tiles_x) { p.tx -= a.tiles_x; ++p.ty; }
        if (p.ty >= a.tiles_y) { p.ty -= a.tiles_y; ++p.b; }
    };
    auto tile_geom = [&](int tile, const Pos &p, int &gr_first, int &col0, int &nrows) {
        if (patch) {
            gr_first = p.b * (a.H + 1) + 16 * p.ty + 1;
            col0 = 32 * p.tx - 1;
            nrows = 18;
        } else {
            const int p0 = tile * S32_PPB;
            const int p1 = min(p0 + S32_PPB, total_p) - 1;
            const int b0 = fd_div(p0, a.fd_hw), r0 = fd_div(p0 - b0 * ohw, a.fd_w);
            const int b1 = fd_div(p1, a.fd_hw), r1 = fd_div(p1 - b1 * ohw, a.fd_w);
            gr_first = b0 * (a.H + 1) + 2 * r0 + 1;
            const int gr_last = b1 * (a.H + 1) + 2 * r1 + 2;
            col0 = -1;
            nrows = gr_last - gr_first + 3;
        }
    };

    int tile = tfirst;
    Pos cur = pos_of(tile), nxp = cur;
    int gr_first = 0, col0 = 0, nrows = 0;
    if (tile < tend) {
        tile_geom(tile, cur, gr_first, col0, nrows);
        issue_tile(gr_first, col0, nrows, 0);
    }
    int parity = 0;
    // the packed bytes of a tile are stored one tile late, behind the next tile's barrier and DMA issue (vmcnt counts stores too)
    uint32_t pk[4];
    uint8_t *pk_outp = a.ypool;
    bool pk_valid = false;
    auto flush_stores = [&]() {
        if (pk_valid) *reinterpret_cast<uint4 *>(pk_outp + chb + 16 * kh) = uint4{pk[0], pk[1], pk[2], pk[3]};  // filters chb + 16 kh .. + 15 (ws_row_filter)
    };
    for (; tile < tend; tile += tstride, parity ^= 1, cur = nxp) {
        tile_geom(tile, cur, gr_first, col0, nrows);
        const char *X = smem + parity * bbytes;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();  // the tile's image has landed; every wave is past the previous tile
        flush_stores();
        advance(nxp);
        if (tile + tstride < tend) {
            int g2, c2, n2;
            tile_geom(tile + tstride, nxp, g2, c2, n2);
            issue_tile(g2, c2, n2, parity ^ 1);
        }
        // ---- per-cell channel sums S (the receptive-field sum of x' is the 3x3 box sum of S)
        for (int id = tid; id < nrows * ncell; id += 512) {
            int t = 0;
#pragma unroll
            for (int p = 0; p < PIECES; ++p) {
                const v4i v = *reinterpret_cast<const v4i *>(X + p * pieceb + id * 16);
                t = __builtin_amdgcn_sdot4(v[0], 0x01010101, t, false);
                t = __builtin_amdgcn_sdot4(v[1], 0x01010101, t, false);
                t = __builtin_amdgcn_sdot4(v[2], 0x01010101, t, false);
                t = __builtin_amdgcn_sdot4(v[3], 0x01010101, t, false);
            }
            ldsS[id] = t;
        }
        __syncthreads();

        // ---- this lane's pooled pixel and its 2x2 window in the image
        int b, prow, pcol;
        bool valid;
        if (patch) {
            b = cur.b;
            prow = 8 * cur.ty + 2 * pw + (lj >> 4);
            pcol = 16 * cur.tx + (lj & 15);
            valid = prow < OH && pcol < OW;
        } else {
            const int pp = tile * S32_PPB + pw * 32 + lj;
            valid = pp < total_p;
            const int ppc = valid ? pp : total_p - 1;
            b = fd_div(ppc, a.fd_hw);
            const int prem = ppc - b * ohw;
            prow = fd_div(prem, a.fd_w);
            pcol = prem - prow * OW;
        }
        const int lrow = b * (a.H + 1) + 2 * prow + 1 - gr_first;  // image row of the window's top row's top tap
        const int lch = (2 * pcol - 1 - col0) >> 1;                  // half the (even) image cell of the window's left column's left tap
        int sx[4];
        {
            int rs[4][2];
#pragma unroll
            for (int row = 0; row < 4; ++row) {
                const int *p = ldsS + (lrow + row) * ncell + lch;
                const int c0 = p[0], c2 = p[1], c1 = p[hc], c3 = p[hc + 1];
                const int s12 = c1 + c2;
                rs[row][0] = c0 + s12;
                rs[row][1] = s12 + c3;
            }
#pragma unroll
            for (int jx = 0; jx < 2; ++jx) {
                const int m = rs[1][jx] + rs[2][jx];
                sx[jx] = rs[0][jx] + m;
                sx[2 + jx] = m + rs[3][jx];
            }
        }
        const size_t pcell = (size_t)a.pool_lead + ((size_t)b * (OH + 1) + (prow + 1)) * (OW + 1) + pcol;
        uint8_t *outp = a.ypool + pcell * a.pool_cs;

        // ---- the window's two rows, one after the other
        uint32_t um01[16];     // first half: per channel the unsigned maximum of its two biased accumulators -- or, for a channel group whose half
        unsigned exact0 = 0;   // failed the range test (bit grp, wave-uniform), the maximum of its two stored BYTES
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int base = (lrow + h) * rowb + lch * 16 + kh * pieceb;
            // accumulators start at cw + bias - lo + (128 - zp_w) * sum(x'): the signed-operand correction as the SEED (one multiply-add per
            // accumulator either way; seeded this way neither the sixteen-register C tuple nor the box sums stay live across the chain)
            v16i acc[2];
#pragma unroll
            for (int grp = 0; grp < 4; ++grp) {
                const int4 c4 = *reinterpret_cast<const int4 *>(ldsCB + chb + 16 * kh + 4 * grp);
                const int4 d4 = *reinterpret_cast<const int4 *>(ldsDZ + chb + 16 * kh + 4 * grp);
#pragma unroll
                for (int jx = 0; jx < 2; ++jx) {
                    const int sxj = sx[2 * h + jx];
                    acc[jx][grp * 4 + 0] = c4.x + __mul24(d4.x, sxj); acc[jx][grp * 4 + 1] = c4.y + __mul24(d4.y, sxj);
                    acc[jx][grp * 4 + 2] = c4.z + __mul24(d4.z, sxj); acc[jx][grp * 4 + 3] = c4.w + __mul24(d4.w, sxj);
                }
            }
            __builtin_amdgcn_s_setprio(3);
#pragma unroll
            for (int s = 0; s < KST; ++s)
#pragma unroll
                for (int jx = 0; jx < 2; ++jx) {
                    const int e = jx + s % 3;
                    const v4i bf = *reinterpret_cast<const v4i *>(X + base + (s / 3) * rowb + (e & 1) * hcb + (e >> 1) * 16);
                    acc[jx] = __builtin_amdgcn_mfma_i32_32x32x32_i8(wf[s], bf, acc[jx], 0, 0, 0);
                }
            __builtin_amdgcn_s_setprio(0);
#pragma unroll
            for (int grp = 0; grp < 4; ++grp) {
                const int ch0 = chb + 16 * kh + 4 * grp;  // accumulator rows 8 grp + 4 kh + r hold filters 16 kh + 4 grp + r of the m-tile (ws_row_filter)
                const int4 rg4 = *reinterpret_cast<const int4 *>(ldsHI + ch0);
                const uint32_t rgv[4] = {(uint32_t)rg4.x, (uint32_t)rg4.y, (uint32_t)rg4.z, (uint32_t)rg4.w};
                uint32_t u[4][2], m2[4];
                bool bad = never;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
#pragma unroll
                    for (int jx = 0; jx < 2; ++jx) u[r][jx] = (uint32_t)acc[jx][grp * 4 + r];
                    m2[r] = max(u[r][0], u[r][1]);
                    bad |= m2[r] > rgv[r];
                }
                const bool e0 = h == 1 && ((exact0 >> grp) & 1u) != 0;  // wave-uniform
                const bool anybad = __builtin_amdgcn_ballot_w64(bad) != 0;
                if (h == 0) {
                    if (!anybad) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) um01[grp * 4 + r] = m2[r];
                    } else {  // the reference's order for this half: bytes first, then their maximum
                        const int4 lo4 = *reinterpret_cast<const int4 *>(ldsLO + ch0);
                        const int lov[4] = {lo4.x, lo4.y, lo4.z, lo4.w};
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int32_t t2[2] = {(int32_t)(u[r][0] + (uint32_t)lov[r]), (int32_t)(u[r][1] + (uint32_t)lov[r])};
                            um01[grp * 4 + r] = (uint32_t)s32_byte_max<ACT, SAT, 2>(t2, ldsMP[ch0 + r], a.mval + ch0 + r, a.sval + ch0 + r, a.zp_act, pow2);
                            __builtin_amdgcn_sched_barrier(0);  // one channel at a time: this path must not size the kernel's registers
                        }
                        exact0 |= 1u << grp;
                    }
                } else {
                    const int4 lo4 = *reinterpret_cast<const int4 *>(ldsLO + ch0);
                    const int lov[4] = {lo4.x, lo4.y, lo4.z, lo4.w};
                    if (!e0 && !anybad) {  // no byte of this window can wrap: ONE requantisation, of its maximum
                        int32_t amax[4][1], v[4][1];
#pragma unroll
                        for (int r = 0; r < 4; ++r) amax[r][0] = (int32_t)(max(um01[grp * 4 + r], m2[r]) + (uint32_t)lov[r]);
                        if (INTRQC && use_int) {
                            const int4 m04 = *reinterpret_cast<const int4 *>(ldsM0 + ch0), sh4 = *reinterpret_cast<const int4 *>(ldsSH + ch0);
                            const int m0v[4] = {m04.x, m04.y, m04.z, m04.w}, shv[4] = {sh4.x, sh4.y, sh4.z, sh4.w};
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                const int32_t f = intrq_floor(amax[r][0], m0v[r], shv[r]);
                                v[r][0] = ACT == MI355_ACT_LEAKY ? leaky_of_floor(f, a.zp_act) : a.zp_act + max(f, 0);
                            }
                        } else {
                            const double mp[4] = {ldsMP[ch0], ldsMP[ch0 + 1], ldsMP[ch0 + 2], ldsMP[ch0 + 3]};
                            requant_values<ACT, SAT, 1>(amax, mp, a.zp_act, v);
                        }
                        pk[grp] = pack4_biased(v[0][0], v[1][0], v[2][0], v[3][0]);
                    } else {
                        // this half's two values -- and, unless the first half already left a byte, its maximum (that half was clean, so the
                        // maximum's byte is the maximum of its bytes) -- requantised one by one, then the maximum of the bytes
                        int32_t mb[4];
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int32_t t3[3] = {(int32_t)(u[r][0] + (uint32_t)lov[r]), (int32_t)(u[r][1] + (uint32_t)lov[r]),
                                                   e0 ? (int32_t)(u[r][0] + (uint32_t)lov[r]) : (int32_t)(um01[grp * 4 + r] + (uint32_t)lov[r])};
                            const int32_t b = s32_byte_max<ACT, SAT, 3>(t3, ldsMP[ch0 + r], a.mval + ch0 + r, a.sval + ch0 + r, a.zp_act, pow2);
                            mb[r] = e0 ? max(b, (int32_t)um01[grp * 4 + r]) : b;
                            __builtin_amdgcn_sched_barrier(0);
                        }
                        pk[grp] = pack4_biased(mb[0], mb[1], mb[2], mb[3]);
                    }
                }
            }
        }
        pk_outp = outp;
        pk_valid = valid;
    }
    flush_stores();
}

template <int ACT>
static int s32_launch_sat(ConvArgs &a, hipStream_t st, int grid, size_t lds)
{
    if (a.store_mode == MI355_STORE_SATURATE) return launch_big_lds<conv_small32_kernel<ACT, true>>(grid, 512, lds, st, a);
    return launch_big_lds<conv_small32_kernel<ACT, false>>(grid, 512, lds, st, a);
}

bool conv_small32_eligible(int n, int c, int ksize) { return ksize == 3 && c == 32 && n == 64; }

// returns MI355_EINVAL when the shape is outside this kernel's domain (the caller falls back to conv_small.hip)
int conv_small32_launch(ConvArgs &a, hipStream_t st)
{
    const int c = a.cb * a.nchunks;
    if (!conv_small32_eligible(a.n, c, a.ksize) || !a.ypool || a.y || a.acc_out || a.y_f32 || a.res || a.stride != 1 || !a.ws) return MI355_EINVAL;
    a.debug = mi355_debug_flags_get();
    if ((a.H & 1) || (a.W & 1) || a.in_cs != c) return MI355_EINVAL;
    if ((size_t)a.in_cells * (size_t)a.in_cs >= ((size_t)1 << 32)) return MI355_EINVAL;  // 32-bit DMA lane offsets
    const int OH = a.H / 2, OW = a.W / 2;
    const long total_p = (long)a.B * OH * OW;
    if (total_p + 256 >= (1L << 31)) return MI355_EINVAL;
    a.fd_hw = fastdiv_make((uint32_t)(OH * OW));
    a.fd_w = fastdiv_make((uint32_t)OW);
    int ntiles;
    if (OW >= 64) {  // wide map: 8 x 16 pooled patches
        a.tiles_x = (OW + 15) / 16;
        a.tiles_y = (OH + 7) / 8;
        a.sm_lcell = 34;
        a.sm_ncell = 40;
        a.rows_cap = 18;
        ntiles = a.B * a.tiles_x * a.tiles_y;
    } else {  // flat runs of 128 pooled pixels (conv_small_pool_launch has the geometry's derivation)
        a.tiles_x = a.tiles_y = 0;
        a.sm_lcell = a.W + 2;
        a.sm_ncell = a.sm_lcell + ((OW & 1) ? 0 : ((OW / 2 - a.sm_lcell) % 8 + 8) % 8);
        a.rows_cap = 2 * ((S32_PPB - 2 + OW) / OW + 1) + (S32_PPB - 2 + OH * OW) / (OH * OW) + 2;
        ntiles = (int)((total_p + S32_PPB - 1) / S32_PPB);
    }
    a.sm_hc = a.sm_lcell / 2;
    a.sm_pieceb = a.rows_cap * a.sm_ncell * 16;
    if (a.rows_cap * a.sm_ncell < 64 || a.rows_cap * a.sm_ncell > 8 * S32_KDMA * 64) return MI355_EINVAL;
    size_t lds = 2 * (size_t)(c / 16) * a.sm_pieceb + (size_t)a.rows_cap * a.sm_ncell * 4;
    lds = (lds + 15) & ~(size_t)15;
    a.lds_param_off = (int)lds;
    lds += (size_t)a.n * 32;
    if (2 * lds > 160 * 1024) return MI355_EINVAL;  // built for two workgroups per CU
    static const int per_cu = getenv("MI355_S32_PER_CU") ? atoi(getenv("MI355_S32_PER_CU")) : 2;  // (A/B runs)
    const int grid = ntiles < 256 * per_cu ? ntiles : 256 * per_cu;
    if (a.act == MI355_ACT_LEAKY) return s32_launch_sat<MI355_ACT_LEAKY>(a, st, grid, lds);
    if (a.act == MI355_ACT_RELU6) return s32_launch_sat<MI355_ACT_RELU6>(a, st, grid, lds);
    return s32_launch_sat<MI355_ACT_LINEAR>(a, st, grid, lds);
}
