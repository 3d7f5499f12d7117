// conv_small.hip -- 3x3 s1 p1 INT8 convolution with few input channels (c = 16 or 32, n = 32 or 64) fused with the
// 2x2 / stride-2 maxpool that follows it: layers 2 and 4 of yolov3-tiny, a quarter of the network's time when they ran
// through the generic K loop of conv_igemm.hip.  Same mathematics as conv_igemm.hip (read its header for the signed
// operand decomposition); built differently because K is tiny (144 / 288) and the epilogue dominates:
//
//   * weights stationary in registers: a wave keeps the A fragments of ALL K-steps (5 for c = 16: two taps per
//     V_MFMA_I32_32X32X32_I8, the k-half is the tap parity; 9 for c = 32: one tap per MFMA, the k-half is the 16-channel
//     piece) and loops over tiles persistently -- there is no A traffic and no K loop.
//   * a tile is 128 POOLED pixels: on narrow maps a run of consecutive ones (flattened over b, y, x: no ragged edge
//     tiles; its input is the run of whole PHWC rows covering the 2x2 windows), on wide maps (pooled width >= 64, where
//     such a run would drag in two mostly unused rows) an 8 x 16 patch.  Either way the input is a dense image of
//     nrows x ncell cells incl. one halo cell all round, DMAed (global_load_lds) into a double buffered LDS plane per
//     16-channel piece while the previous tile computes.
//   * an LDS image row is DE-INTERLEAVED: its even cells x = 0, 2, .. sit in slots 0 .., its odd cells in slots hc ..  The lanes of
//     a B read are consecutive POOLED pixels, i.e. image cells two apart: in a plain row image their 16-byte fragments are 32 B apart
//     and every ds_read_b128 is a two-way bank conflict (SQ_LDS_BANK_CONFLICT was 35-50 % of SQ_LDS_IDX_ACTIVE on layers 2 / 4 / 6,
//     profiles/r04_v2_pmc_sq2_plan1.txt); split by column parity they are consecutive slots.  The DMA's per-lane SOURCE address
//     carries the permutation (its LDS side is contiguous by construction), the tap offsets of the reads absorb it: window column jx
//     + tap column dx = e -> slot offset (e & 1) * hc + (e >> 1).  The row pitch is chosen so that a wave's second pooled row
//     (patches: 2 * pitch = 0 mod 16) or its wrap into the next pooled row (flat tiles: 2 * pitch = OW mod 16) continues the same
//     sequence of 16-byte bank groups.
//   * the A rows of a 32-filter m-tile are PERMUTED (ws_row_filter below, applied by mi355_conv_pack): the accumulator rows a lane holds,
//     8 grp + 4 kh + r, are then the sixteen CONSECUTIVE filters 16 kh + 4 grp + r -- one 16-byte store per lane and m-tile instead of four
//     4-byte ones (memory-instruction issue is 11-16 % of these kernels' time, tools/small_phases.py);
//   * lane l of a wave owns pooled pixel 32*wave + l; the four 32-column MFMA sub-tiles of the wave are the four window
//     positions, so the 2x2 window of every (pixel, channel) sits in ONE lane: the pool is three v_max_i32, no
//     cross-lane traffic and no pre-pool tensor.
//   * max-pool commutes with the requantisation: per channel the map accumulator -> stored byte is monotone as long as
//     the byte does not wrap, and the range of accumulators that cannot wrap, [tlo, thi], is found once per workgroup
//     (analytic guess, verified with the exact requantise function).  A window whose four accumulators lie inside it is
//     requantised ONCE (its maximum); only waves that see an accumulator outside the range (the reference's
//     wrap-on-store cases, src/convolutional_layer.c:737-749) take the exact four-requantisation path.  SATURATE stores
//     are monotone everywhere.  Result bytes are identical to conv + forward_maxpool_layer_quant
//     (ref: src/maxpool_layer.c:109-172) in every case.
#include "kargs.h"
#include <cstdlib>
#include <type_traits>

#define DMA_S(ldsdst_u32, sbase_ptr, voff_u32)                                                                   \
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(ldsdst_u32), "v"(voff_u32), \
                 "s"(sbase_ptr)                                                                                  \
                 : "memory", "m0")

#ifdef MI355_ABLATE
// per-wave shader-clock sums of conv_small_pool_kernel's phases (tools/small_phases.py): [0] image wait + barrier, [1] deferred stores + next
// tile's DMA issue, [2] cell sums + barrier (VALU correction form), [3] lane geometry + box sums, [4] MFMA chains, [5] epilogues, [6] rest, [7] tiles
__device__ long long g_sm_ph[4096][4][8];
#define SMP_DECL long long smp[8] = {0, 0, 0, 0, 0, 0, 0, 0}; long long smt = __builtin_readcyclecounter()
#define SMP_MARK(k)                                              \
    do {                                                         \
        asm volatile("" ::: "memory");                           \
        const long long n_ = __builtin_readcyclecounter();       \
        smp[k] += n_ - smt;                                      \
        smt = n_;                                                \
    } while (0)
#define SMP_MARK_V(k, v)                                         \
    do {                                                         \
        asm volatile("s_nop 0" ::"v"(v));                        \
        SMP_MARK(k);                                             \
    } while (0)
#define SMP_STORE()                                                                             \
    do {                                                                                        \
        if ((threadIdx.x & 63) == 0 && blockIdx.x < 4096)                                       \
            for (int k = 0; k < 8; ++k) g_sm_ph[blockIdx.x][threadIdx.x >> 6][k] = smp[k];      \
    } while (0)
extern "C" int mi355_debug_read_smph(long long *host)
{
    return hipMemcpyFromSymbol(host, HIP_SYMBOL(g_sm_ph), sizeof(long long) * 4096 * 4 * 8) == hipSuccess ? 0 : -5;
}
#else
#define SMP_DECL do { } while (0)
#define SMP_MARK(k) do { } while (0)
#define SMP_MARK_V(k, v) do { } while (0)
#define SMP_STORE() do { } while (0)
#endif

#ifdef MI355_ABLATE
// conv_mid_pool_kernel: phase timestamps (100 MHz wall clock) of thread 0 of every workgroup (tools/conv_microbench.py --timelinem)
__device__ long long g_mid_ts[6][4096];
#define TSM(k) do { if (threadIdx.x == 0 && blockIdx.x < 4096) g_mid_ts[k][blockIdx.x] = wall_clock64(); } while (0)
extern "C" int mi355_debug_read_tsm(long long *host)
{
    return hipMemcpyFromSymbol(host, HIP_SYMBOL(g_mid_ts), sizeof(long long) * 6 * 4096) == hipSuccess ? 0 : -5;
}
#else
#define TSM(k) do { } while (0)
#endif

constexpr int SM_PPB = 128;  // pooled pixels per workgroup tile (4 waves x 32 lanes)
constexpr int SM_GMAX = 8;    // conv_mid_pool_kernel: groups of 32 pooled pixels per tile (tile <= 256 pooled pixels)
constexpr int SM_KMAX = 6;    // DMA instructions per wave and piece per tile image (image <= 4 * 6 * 64 cells)


// The pooled store of four channels x one 2x2 window held by a lane (wave-uniform control flow inside: call it from uniform code).
// Rounds 1-3 tested the window's minimum and maximum against the wrap-safe range [lo, hi] and requantised the maximum in FP64;
// round 4 keeps the accumulators BIASED by the lower end of the wrap-safe range (the seed cw + bias - lo is the MFMA's C operand /
// the accumulator's start value: free), common.h biased_safe_range: one unsigned maximum over the window is the range test and the
// maximum (3 VALU per pooled output instead of 8), and the maximum is requantised with two integer instructions where the launch's
// channels allow it (common.h intrq_make) -- 11 VALU per pooled output instead of ~21 on the fast path.
// u[r][j] = biased accumulator of channel r at window position j; lo / rg = lower end and width of the safe range.
// `never` is the launch's "always take the exact path" flag: a channel without a safe range, shifts that are not powers of two, or (where
// the activation has the integer form) a channel that failed its exactness conditions -- folded into ONE wave-uniform flag by the caller, so
// that the common path tests nothing else per group of four channels (the kernels are close to instruction-issue bound, DESIGN.md 4.5;
// such launches requantise every window value: slower, same bytes).
template <int ACT, bool SAT>
// `use_int` (round 6): a channel that fails the integer form's conditions (intrq_make: multipliers below ~5e-4 widen the
// wrap-safe range beyond 2^53 / M0) used to be folded into `never` -- ONE such channel sent every window of the launch down the exact path.  It
// now only selects the FP64-of-maximum form of the fast path (one FP64 requantisation per pooled output instead of two integer instructions).
__device__ __forceinline__ uint32_t pool_requant_quad_biased(const uint32_t (&u)[4][4], const int (&lo)[4], const int (&rg)[4], bool never,
                                                             const int (&m0)[4], const int (&sh)[4], const double *ldsMP4,
                                                             int zp_act, bool pow2, const double *mval4, const double *sval4, bool use_int = true)
{
    uint32_t umax[4];
    bool bad = never;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        umax[r] = max(max(u[r][0], u[r][1]), max(u[r][2], u[r][3]));
        bad |= umax[r] > (uint32_t)rg[r];
    }
    if (__builtin_amdgcn_ballot_w64(bad) == 0) {  // no window of this wave can wrap (and never == false: power-of-two shifts, integer form valid)
        int32_t amax[4][1], v[4][1];
#pragma unroll
        for (int r = 0; r < 4; ++r) amax[r][0] = (int32_t)(umax[r] + (uint32_t)lo[r]);
        if ((ACT == MI355_ACT_LEAKY || ACT == MI355_ACT_RELU6) && !SAT && use_int) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int32_t f = intrq_floor(amax[r][0], m0[r], sh[r]);
                v[r][0] = ACT == MI355_ACT_LEAKY ? leaky_of_floor(f, zp_act) : zp_act + max(f, 0);
            }
        } else {
            const double mp[4] = {ldsMP4[0], ldsMP4[1], ldsMP4[2], ldsMP4[3]};
            requant_values<ACT, SAT, 1>(amax, mp, zp_act, v);
        }
        return pack4_biased(v[0][0], v[1][0], v[2][0], v[3][0]);
    }
    int32_t accb[4][4];  // true accumulators: the reference's order, bytes first, then the max
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int j = 0; j < 4; ++j) accb[r][j] = (int32_t)(u[r][j] + (uint32_t)lo[r]);
    int32_t m[4];
    if (pow2) {
        const double mp[4] = {ldsMP4[0], ldsMP4[1], ldsMP4[2], ldsMP4[3]};
        int32_t v[4][4];
        requant_values<ACT, SAT, 4>(accb, mp, zp_act, v);
#pragma unroll
        for (int r = 0; r < 4; ++r) m[r] = max(max(v[r][0] & 0xFF, v[r][1] & 0xFF), max(v[r][2] & 0xFF, v[r][3] & 0xFF));
    } else {  // shift_value not a power of two: the reference's two-step form (never produced by its own prep)
        // A ROLLED loop over a private array: unrolled, these sixteen two-step requantisations (global multiplier loads, two FP64 chains
        // each) were what the register allocator sized the whole kernel by -- 8 to 17 registers of the common path (the 16-channel kernel
        // sits at its three-workgroups-per-CU edge).  Slow, and only ever run by models the reference's preparation does not produce.
        int32_t tmp[16];
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int j = 0; j < 4; ++j) tmp[4 * r + j] = accb[r][j];
#pragma unroll 1
        for (int idx = 0; idx < 16; ++idx)
            tmp[idx] = (int32_t)requant_u8(tmp[idx], 0, mval4[idx >> 2], sval4[idx >> 2], zp_act, ACT, SAT ? MI355_STORE_SATURATE : MI355_STORE_WRAP);
#pragma unroll
        for (int r = 0; r < 4; ++r) m[r] = max(max(tmp[4 * r], tmp[4 * r + 1]), max(tmp[4 * r + 2], tmp[4 * r + 3]));
    }
    return pack4_biased(m[0], m[1], m[2], m[3]);
}

// sixteen two-step requantisations (shift_value not a power of two: never produced by the reference's own preparation) as a ROLLED loop
// over a private array -- unrolled, this cold path sizes the kernel's registers (see pool_requant_quad_biased)
template <int ACT, bool SAT>
__device__ __forceinline__ void requant16_two_step(const int32_t (&accb)[4][4], const double *mval4, const double *sval4, int zp_act, int32_t (&v)[4][4])
{
    int32_t tmp[16];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int j = 0; j < 4; ++j) tmp[4 * r + j] = accb[r][j];
#pragma unroll 1
    for (int idx = 0; idx < 16; ++idx)
        tmp[idx] = (int32_t)requant_u8(tmp[idx], 0, mval4[idx >> 2], sval4[idx >> 2], zp_act, ACT, SAT ? MI355_STORE_SATURATE : MI355_STORE_WRAP);
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int j = 0; j < 4; ++j) v[r][j] = tmp[4 * r + j];
}

// MODE 0: conv + 2x2/2 maxpool.  1: no pool, the four window positions of a lane are four output pixels.  2: stride-2
// convolution = the stride-1 output at the even positions = window position 0 only (a quarter of the MFMAs), stored on the
// pooled geometry (the output map of a stride-2 3x3 pad-1 convolution on an even map is the pooled map).
// (launch bounds: HIP's second number is waves per SIMD.  The launcher runs the 16-channel, 32-filter pooled kernel three workgroups per
// CU, which its LEAKY instantiations allow with 153-162 registers (<= 168).  Round 4 measured what happens beyond that: a requantise
// variant that needed 170 put the third workgroup of every CU into a second round -- L2 38 -> 53 us; forced back to 168 with three
// spilled dwords 39-40 us, and no gain where it fitted (profiles/r04_l2_occupancy.log): not kept.)
template <int C, int NM, int ACT, bool SAT, int MODE = 0, bool VDZ = false>
// (the throughput-plan form of the 16-channel, 32-filter kernel is launched three workgroups per CU: its RELU / RELU6 instantiations, left
// to a budget of 256 registers, take 184-186 and lose the third; bounded to three waves per SIMD they fit in 166-167 without scratch)
__global__ __launch_bounds__(256, (C == 16 && NM == 1 && VDZ) ? 3 : 2) void conv_small_pool_kernel(const ConvArgs a)
{
    constexpr bool POOL = MODE == 0;
    constexpr int NJ = MODE == 2 ? 1 : 4;  // window positions computed
    constexpr int KST = (C == 16) ? 5 : 9;
    constexpr int PIECES = C / 16;
    constexpr int N = 32 * NM;
    // C == 16: the signed-operand correction (128 - zp_w) * sum(x') comes from the matrix pipe -- a second pass over
    // the same B fragments with the constant dz in every real k slot (a third, almost always skipped, for dz = 128,
    // which does not fit an int8) -- instead of per-cell sums, a 3x3 box sum per pixel and a multiply-add per
    // accumulator on the VALU, which is the bottleneck of this kernel.  With 32 input channels the extra MFMAs (72 per
    // wave and tile) would cost more than they save.
    // VDZ (throughput plan): the correction on the VALU after all, one multiply-add per accumulator.  Alone on the device the
    // kernel is bound by its VALU stream and the MFMA form is faster (layer 2: 37.6 vs 38.7 us); with other batches in flight what
    // counts is the sum of both pipes' time, and the second MFMA pass costs 0.16 clk per output against 0.06 (flood 34.9 -> 33.1 us)
    constexpr bool DZM = (C == 16) && !VDZ;
#ifdef MI355_SMALL_BSHARE  // (A/B builds: see BSH below; measured slower, off)
    constexpr bool BSH = C == 32 && NM == 2 && MODE == 0;
#else
    constexpr bool BSH = false;
#endif

    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int ncell = a.sm_ncell;     // slots of an LDS image row = its pitch (>= the image cells: flat tiles W + 2, x = -1 .. W; patches 34)
    const int lcell = a.sm_lcell;     // image cells of a row
    const int hc = a.sm_hc, hcb = hc * 16;  // slot / byte offset of the row's first odd cell (rows are de-interleaved by column parity)
    const int rowb = ncell * 16;      // bytes between image rows inside a piece plane
    const int pieceb = a.sm_pieceb;   // bytes of one 16-channel piece plane (rows_cap * ncell cells)
    const int bbytes = PIECES * pieceb;
    const bool patch = a.tiles_x > 0;
    int *ldsS = reinterpret_cast<int *>(smem + 2 * bbytes);               // [rows_cap * ncell] per-cell channel sums
    double *ldsMP = reinterpret_cast<double *>(smem + a.lds_param_off);   // [N] folded multiplier
    int *ldsDZ = reinterpret_cast<int *>(ldsMP + N);                      // [N] 128 - zp_w
    int *ldsCB = ldsDZ + N;                                               // [N] cw + bias
    int *ldsLO = ldsCB + N, *ldsHI = ldsLO + N;                           // [N] wrap-safe accumulator range (POOL: lower end, width)
    int *ldsM0 = ldsHI + N, *ldsSH = ldsM0 + N;                           // [N] integer requantisation: M0, s - 1 (common.h intrq_make)
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char *)smem;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kh = lane >> 5, lj = lane & 31;
    const int W1 = a.W + 1;
    const int OH = a.H >> 1, OW = a.W >> 1, ohw = OH * OW;
    const int total_p = a.B * ohw;
    const int tpi = a.tiles_x * a.tiles_y;  // patches per image
    const int ntiles = patch ? a.B * tpi : (total_p + SM_PPB - 1) / SM_PPB;
    const bool pow2 = a.hdr->pow2 == 1;

    // ---- per-channel parameters and the wrap-safe ranges.  POOL: the accumulators are kept BIASED by the range's lower end (seed
    //      cw + bias - lo), ldsLO / ldsHI hold that end and the range's width (common.h biased_safe_range)
    int never_l = 0, noint_l = 0;
    // the host's epilogue table for this (activation, zero point), if the blob carries one (mi355_conv_pack_epilogue, common.h): the
    // ranges and integer multipliers are then 32 bytes per channel to load instead of FP64 divisions and verification loops per workgroup
    const bool ept_ok = POOL && !SAT && a.ept != nullptr && a.ept->key == ept_key(ACT, a.zp_act);
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
            if (!SAT && POOL) small_safe_range<ACT>(mp, a.zp_act, lo, hi);
            if constexpr (POOL) {
                int32_t lb = 0; uint32_t rg = 0;
                if (!biased_safe_range(lo, hi, lb, rg)) never_l = 1;
                if (!(pow2 && intrq_make(a.mval[tid], a.shift[tid], lb, (int32_t)((uint32_t)lb + rg), m0, sh, ACT == MI355_ACT_RELU6))) noint_l = 1;
                ldsCB[tid] = (int32_t)((uint32_t)a.cwb[tid] - (uint32_t)lb);
                ldsLO[tid] = lb;
                ldsHI[tid] = (int32_t)rg;
            } else {
                ldsCB[tid] = a.cwb[tid];
                ldsLO[tid] = lo;
                ldsHI[tid] = hi;
            }
        }
        ldsM0[tid] = m0;
        ldsSH[tid] = sh;
    }
    // one wave-uniform flag for "this launch requantises every window value" (see pool_requant_quad_biased)
    constexpr bool INTRQC = (ACT == MI355_ACT_LEAKY || ACT == MI355_ACT_RELU6) && !SAT;
    // (the 16-channel kernels sit at their three-workgroups-per-CU register edge and keep round 4's folding: a channel outside the integer form's
    // conditions counts as `never` there; layer 2 of the nets runs conv_pool16.hip anyway)
    constexpr bool SPLIT_NOINT = C != 16;
    bool never_any, use_int = true;
    if (ept_ok) {
        never_any = (a.ept->flags & (EPT_NEVER | ((INTRQC && !SPLIT_NOINT) ? EPT_NOINT : 0u))) != 0;
        if (SPLIT_NOINT) use_int = (a.ept->flags & EPT_NOINT) == 0;
        __syncthreads();
    } else {
        never_any = __syncthreads_or(never_l | ((INTRQC && !SPLIT_NOINT) ? noint_l : 0)) != 0;
        if (SPLIT_NOINT) use_int = __syncthreads_or(noint_l) == 0;
    }
    const bool never = POOL && (never_any || !pow2);

    // ---- stationary A fragments: plane ws = [m-tile][k-step][lane][16 B]
    v4i wf[NM][KST];
#pragma unroll
    for (int mt = 0; mt < NM; ++mt)
#pragma unroll
        for (int s = 0; s < KST; ++s)
            wf[mt][s] = *reinterpret_cast<const v4i *>(a.ws + ((size_t)(mt * KST + s) * 64 + lane) * 16);
    int wd1[NM], wd2[NM];  // DZM: dz of this lane's A row, replicated over the four bytes of a dword (split 127 + 1 for 128)
    bool need_d2 = false;
#pragma unroll
    for (int mt = 0; mt < NM; ++mt) {
        const int dz = a.dzp[32 * mt + ws_row_filter(lj)];  // (the lane's A row is filter ws_row_filter(lj) of the m-tile)
        const int d1 = dz > 127 ? 127 : dz, d2 = dz - d1;
        wd1[mt] = (int)((uint32_t)(d1 & 0xFF) * 0x01010101u);
        wd2[mt] = (int)((uint32_t)(d2 & 0xFF) * 0x01010101u);
        need_d2 |= __builtin_amdgcn_ballot_w64(d2 != 0) != 0;
    }

    // tap byte offsets inside the row image.  C == 16: lane-dependent (k-half = tap parity; tap 9 does not exist, its
    // weights are zero and the lane re-reads tap 8).  C == 32: uniform per step, the k-half selects the piece plane.
    // Per window column jx: cell lcol + jx + dx of the de-interleaved row = slot (lcol >> 1) + (e & 1) * hc + (e >> 1), e = jx + dx
    // (lcol, the cell of the left window column's left tap, is even for every lane).
    int toff[2][KST];
#pragma unroll
    for (int s = 0; s < KST; ++s) {
        int t = (C == 16) ? 2 * s + kh : s;
        if (t > 8) t = 8;
#pragma unroll
        for (int jx = 0; jx < 2; ++jx) {
            const int e = jx + t % 3;
            toff[jx][s] = (t / 3) * rowb + (e & 1) * hcb + (e >> 1) * 16;
        }
    }

    // DMA of a tile's image: instruction k fills cells [64k, 64k + 64) of every piece plane (the last one is shifted
    // back to end on the plane's last cell); wave w issues k = w, w+4, ..  The lane's cell offset from the tile origin
    // does not depend on the tile: computed once per slot.
    int doff[SM_KMAX], dstart[SM_KMAX];
#pragma unroll
    for (int i = 0; i < SM_KMAX; ++i) {
        const int k = wave + 4 * i;
        dstart[i] = min(k * 64, a.rows_cap * ncell - 64);
        const int slot = dstart[i] + lane;
        const int r = slot / ncell, cs = slot - r * ncell;
        const int c = min(cs < hc ? 2 * cs : 2 * (cs - hc) + 1, lcell - 1);  // slot -> image cell (pad slots repeat the last cell)
        doff[i] = r * W1 + c;
    }
    auto issue_tile = [&](int gr_first, int col0, int nrows, int parity) {
        const unsigned buf = lds0 + parity * bbytes;
        const int ncells = nrows * ncell;
        const long org = (long)a.in_lead + (long)(gr_first - 1) * W1 + col0;
#pragma unroll
        for (int i = 0; i < SM_KMAX; ++i)
            if ((wave + 4 * i) * 64 < ncells) {
                long f = org + doff[i];
                f = f < 0 ? 0 : (f > a.in_cells - 1 ? a.in_cells - 1 : f);
                const unsigned voff = (unsigned)(f * a.in_cs);
#pragma unroll
                for (int p = 0; p < PIECES; ++p) {
                    const unsigned dst = buf + p * pieceb + dstart[i] * 16;
                    const unsigned v = voff + p * 16;
                    DMA_S(dst, a.x, v);
                }
            }
    };
    // patch walk without per-tile divisions: (b, ty, tx) advances by the decomposition of gridDim.x with carries
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
    // XCD-aware tile order (round 5): workgroup id w runs on XCD w % 8, each with its own L2, and neighbouring tiles share their halo rows (flat
    // tiles: whole rows; patches: a row above / below and a cell left / right).  Dealt round-robin (tile = w + k * grid), neighbours sat on
    // different XCDs and every shared row came from HBM twice: PMC reads 1.53x (layer 2) / 2.0x (layer 4) the input.  Now every XCD takes one
    // CONTIGUOUS share of the tiles and its workgroups walk it side by side: tile = start_x + (w >> 3) + k * (grid >> 3).
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
    // image of a tile: global row (over all image blocks of the PHWC tensor, pad rows included) of its first pre-pool
    // row, x of image cell 0, rows
    auto tile_geom = [&](int tile, const Pos &p, int &gr_first, int &col0, int &nrows) {
        if (patch) {
            gr_first = p.b * (a.H + 1) + 16 * p.ty + 1;
            col0 = 32 * p.tx - 1;
            nrows = 18;
        } else {
            const int p0 = tile * SM_PPB;
            const int p1 = min(p0 + SM_PPB, total_p) - 1;
            const int b0 = fd_div(p0, a.fd_hw), r0 = fd_div(p0 - b0 * ohw, a.fd_w);  // (launch constants: common.h FastDiv, set by the launcher)
            const int b1 = fd_div(p1, a.fd_hw), r1 = fd_div(p1 - b1 * ohw, a.fd_w);
            gr_first = b0 * (a.H + 1) + 2 * r0 + 1;
            const int gr_last = b1 * (a.H + 1) + 2 * r1 + 2;
            col0 = -1;
            nrows = gr_last - gr_first + 3;  // + one halo row above and below
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
    // The packed bytes of a tile are stored one tile late, right after the next tile's barrier: vmcnt counts loads and
    // stores alike on gfx9-class hardware, so stores issued just before the `s_waitcnt vmcnt(0)` that guards the image
    // DMA would make every tile wait for a full store round trip; issued after it they have a whole tile to drain.
    uint32_t pk[NM][4];
    uint8_t *pk_outp = a.ypool;
    bool pk_valid = false;
    auto flush_stores = [&]() {
        if (pk_valid) {
#pragma unroll
            for (int mt = 0; mt < NM; ++mt)
                *reinterpret_cast<uint4 *>(pk_outp + 32 * mt + 16 * kh) = uint4{pk[mt][0], pk[mt][1], pk[mt][2], pk[mt][3]};  // filters 16 kh .. + 15
        }
    };
    SMP_DECL;
    for (; tile < tend; tile += tstride, parity ^= 1, cur = nxp) {
        SMP_MARK(6);
        tile_geom(tile, cur, gr_first, col0, nrows);
        const char *X = smem + parity * bbytes;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();  // the tile's image has landed (and the parameters, first time round); every wave is past the
                          // previous tile, so its buffer may be overwritten
        SMP_MARK(0);
        flush_stores();
        advance(nxp);
        if (tile + tstride < tend) {
            int g2, c2, n2;
            tile_geom(tile + tstride, nxp, g2, c2, n2);
            issue_tile(g2, c2, n2, parity ^ 1);
        }

        SMP_MARK(1);
        // ---- per-cell channel sums S (the receptive-field sum of x' is the 3x3 box sum of S)
        if (!DZM) for (int id = tid; id < nrows * ncell; id += 256) {
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
        if (!DZM) __syncthreads();
        SMP_MARK(2);

        // ---- this lane's pooled pixel and its 2x2 window in the image
        int b, prow, pcol;
        bool valid;
        if (patch) {
            b = cur.b;
            prow = 8 * cur.ty + 2 * wave + (lj >> 4);
            pcol = 16 * cur.tx + (lj & 15);
            valid = prow < OH && pcol < OW;
        } else {
            const int pp = tile * SM_PPB + wave * 32 + lj;
            valid = pp < total_p;
            const int ppc = valid ? pp : total_p - 1;
            b = fd_div(ppc, a.fd_hw);
            const int prem = ppc - b * ohw;
            prow = fd_div(prem, a.fd_w);
            pcol = prem - prow * OW;
        }
        const int lrow = b * (a.H + 1) + 2 * prow + 1 - gr_first;  // image row of the window's top row's top tap
        const int lch = (2 * pcol - 1 - col0) >> 1;                  // half the (even) image cell of the window's left column's left tap
        int base[2], sx[4];                                          // base: per window ROW (the window column is in the tap offsets)
#pragma unroll
        for (int jy = 0; jy < 2; ++jy) base[jy] = (lrow + jy) * rowb + lch * 16 + ((C == 32) ? kh * pieceb : 0);
        if (!DZM) {
            // the four 3 x 3 box sums of the lane's 2 x 2 window share a 4 x 4 block of cell sums: 16 reads (cells 0, 2 and 1, 3 of a row are
            // neighbours in the de-interleaved row: two pairs) and 18 additions instead of 36 reads and 32 additions (round 5)
            int rs[4][2];
#pragma unroll
            for (int row = 0; row < 4; ++row) {
                const int *p = ldsS + (lrow + row) * ncell + lch;
                const int c0 = p[0], c2 = p[1], c1 = p[hc], c3 = p[hc + 1];
                const int s12 = c1 + c2;
                rs[row][0] = c0 + s12;   // window column 0: cells 0 .. 2
                rs[row][1] = s12 + c3;   // window column 1: cells 1 .. 3
            }
#pragma unroll
            for (int jx = 0; jx < 2; ++jx) {
                const int m = rs[1][jx] + rs[2][jx];
                sx[jx] = rs[0][jx] + m;      // window row 0: image rows 0 .. 2
                sx[2 + jx] = m + rs[3][jx];  // window row 1: image rows 1 .. 3
            }
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) sx[j] = 0;
        }
        const size_t pcell = (size_t)a.pool_lead + ((size_t)b * (OH + 1) + (prow + 1)) * (OW + 1) + pcol;
        uint8_t *outp = a.ypool + pcell * a.pool_cs;

        SMP_MARK_V(3, sx[3]);
        // BSH (round 6 experiment, -DMI355_SMALL_BSHARE builds only): 32 -> 64 + maxpool keeps the accumulators of BOTH m-tiles (128 registers)
        // and issues the two MFMAs of a B fragment back to back -- one 1 KiB ds_read_b128 per two MFMAs instead of one per MFMA.  Bit-identical
        // and SLOWER: layer 4 in flight 18.8-19.2 -> 20.5-20.6 us, the whole in-flight step 0.2403-0.2409 -> 0.2428-0.2434 ms
        // (profiles/r06_bshare_ab_flood.log, r06_l4_variants_ab_bench.log): LDS reads are not what this kernel waits for.
        v16i accs[BSH ? NM : 1][4];
        if constexpr (BSH) {
#pragma unroll
            for (int mt = 0; mt < NM; ++mt)
#pragma unroll
                for (int grp = 0; grp < 4; ++grp) {
                    const int4 c4 = *reinterpret_cast<const int4 *>(ldsCB + 32 * mt + 16 * kh + 4 * grp);
#pragma unroll
                    for (int j = 0; j < NJ; ++j) {
                        accs[mt][j][grp * 4 + 0] = c4.x; accs[mt][j][grp * 4 + 1] = c4.y;
                        accs[mt][j][grp * 4 + 2] = c4.z; accs[mt][j][grp * 4 + 3] = c4.w;
                    }
                }
            __builtin_amdgcn_s_setprio(3);
#pragma unroll
            for (int s = 0; s < KST; ++s)
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    const v4i bf = *reinterpret_cast<const v4i *>(X + base[j >> 1] + toff[j & 1][s]);
#pragma unroll
                    for (int mt = 0; mt < NM; ++mt) accs[mt][j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(wf[mt][s], bf, accs[mt][j], 0, 0, 0);
                }
            __builtin_amdgcn_s_setprio(0);
        }
#pragma unroll
        for (int mt = 0; mt < NM; ++mt) {
            // accumulators start at cw + bias: register grp*4+r of a 32x32 tile is channel row 8*grp + 4*kh + r
            v16i (&acc)[4] = accs[BSH ? mt : 0];
            if constexpr (!BSH) {
#pragma unroll
            for (int grp = 0; grp < 4; ++grp) {
                const int4 c4 = *reinterpret_cast<const int4 *>(ldsCB + 32 * mt + 16 * kh + 4 * grp);
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    acc[j][grp * 4 + 0] = c4.x; acc[j][grp * 4 + 1] = c4.y;
                    acc[j][grp * 4 + 2] = c4.z; acc[j][grp * 4 + 3] = c4.w;
                }
            }
            // the wave inside its MFMA chain outranks the co-resident waves that requantise: its MFMAs get their issue slot at
            // once (one in eight) and the others' VALU work fills the rest, instead of the matrix pipe idling behind an older
            // wave's VALU stream (issue arbitration is by priority, then age)
#ifdef MI355_ABLATE
            if (!(a.debug & 131072))
#endif
                __builtin_amdgcn_s_setprio(3);
#pragma unroll
            for (int s = 0; s < KST; ++s)
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    const v4i bf = *reinterpret_cast<const v4i *>(X + base[j >> 1] + toff[j & 1][s]);
                    acc[j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(wf[mt][s], bf, acc[j], 0, 0, 0);
                    if (DZM) {
                        // tap 2s + kh: the k-half of the nonexistent tap 9 carries a zero constant
                        const int z1 = (s == KST - 1 && kh) ? 0 : wd1[mt], z2 = (s == KST - 1 && kh) ? 0 : wd2[mt];
                        const v4i d1 = {z1, z1, z1, z1};
                        acc[j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(d1, bf, acc[j], 0, 0, 0);
                        if (need_d2) {
                            const v4i d2 = {z2, z2, z2, z2};
                            acc[j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(d2, bf, acc[j], 0, 0, 0);
                        }
                    }
                }
            __builtin_amdgcn_s_setprio(0);
            }
            SMP_MARK_V(4, acc[NJ - 1][15]);
            // ---- epilogue: window max, one requantisation per (pixel, channel), biased packed store
            uint32_t po[POOL ? 1 : NJ][4];  // no-pool / stride-2 modes: the packed bytes of the four channel groups, stored together
            (void)po;
            // POOL: the per-channel constants of a group of four channels are read while the group before it is requantised (the first group's
            // behind the MFMA chain's issue) -- read where they are used, every group waited for three LDS round trips with one other wave on the
            // SIMD to cover them (conv_pool16.hip has the measurement)
            struct GroupConst { int4 dz, lo, hi, m0, sh; };
            auto group_const = [&](int grp) {
                const int ch0 = 32 * mt + 16 * kh + 4 * grp;
                GroupConst g;
                g.dz = *reinterpret_cast<const int4 *>(ldsDZ + ch0);
                g.lo = *reinterpret_cast<const int4 *>(ldsLO + ch0);
                g.hi = *reinterpret_cast<const int4 *>(ldsHI + ch0);
                g.m0 = *reinterpret_cast<const int4 *>(ldsM0 + ch0);
                g.sh = *reinterpret_cast<const int4 *>(ldsSH + ch0);
                return g;
            };
            constexpr bool AHEAD = POOL && C == 32 && !BSH;  // (c = 16 sits at its three-workgroups-per-CU register edge: the 20 registers of a group in flight would cost a workgroup)
            GroupConst gnext = {};
            if constexpr (AHEAD) {
                gnext = group_const(0);
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int grp = 0; grp < 4; ++grp) {
                const int ch0 = 32 * mt + 16 * kh + 4 * grp;  // accumulator rows 8 grp + 4 kh + r hold filters 16 kh + 4 grp + r (ws_row_filter)
                GroupConst gc = gnext;
                if constexpr (AHEAD) {
                    if (grp < 3) gnext = group_const(grp + 1);
                    __builtin_amdgcn_sched_barrier(0);
                } else {
                    gc.dz = *reinterpret_cast<const int4 *>(ldsDZ + ch0);
                    gc.lo = *reinterpret_cast<const int4 *>(ldsLO + ch0);
                    gc.hi = *reinterpret_cast<const int4 *>(ldsHI + ch0);
                    if constexpr (POOL) {
                        gc.m0 = *reinterpret_cast<const int4 *>(ldsM0 + ch0);
                        gc.sh = *reinterpret_cast<const int4 *>(ldsSH + ch0);
                    }
                }
                const int4 dz4 = gc.dz, lo4 = gc.lo, hi4 = gc.hi;
                const int dzv[4] = {dz4.x, dz4.y, dz4.z, dz4.w};
                const int lov[4] = {lo4.x, lo4.y, lo4.z, lo4.w}, hiv[4] = {hi4.x, hi4.y, hi4.z, hi4.w};
                int32_t accb[4][4];  // [channel r][window position j]
                double mp[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (!POOL) mp[r] = ldsMP[ch0 + r];
#pragma unroll
                    for (int j = 0; j < 4; ++j) accb[r][j] = j >= NJ ? 0 : (DZM ? acc[j][grp * 4 + r] : acc[j][grp * 4 + r] + __mul24(dzv[r], sx[j]));
                }
                if constexpr (POOL) {
                    const int4 m04 = gc.m0, sh4 = gc.sh;
                    const int m0v[4] = {m04.x, m04.y, m04.z, m04.w}, shv[4] = {sh4.x, sh4.y, sh4.z, sh4.w};
                    uint32_t ub[4][4];
#pragma unroll
                    for (int r = 0; r < 4; ++r)
#pragma unroll
                        for (int j = 0; j < 4; ++j) ub[r][j] = (uint32_t)accb[r][j];
                    pk[mt][grp] = pool_requant_quad_biased<ACT, SAT>(ub, lov, hiv, never, m0v, shv, ldsMP + ch0, a.zp_act, pow2,
                                                                     a.mval + ch0, a.sval + ch0, use_int);
                } else if constexpr (MODE == 2) {  // stride 2: one value per (pixel, channel), plain requantisation
                    int32_t a1[4][1], v1[4][1];
#pragma unroll
                    for (int r = 0; r < 4; ++r) a1[r][0] = accb[r][0];
                    if (pow2) {
                        requant_values<ACT, SAT, 1>(a1, mp, a.zp_act, v1);
                    } else {
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            v1[r][0] = (int32_t)requant_u8(a1[r][0], 0, a.mval[ch0 + r], a.sval[ch0 + r], a.zp_act, ACT,
                                                           SAT ? MI355_STORE_SATURATE : MI355_STORE_WRAP);
                    }
                    po[0][grp] = pack4_biased(v1[0][0], v1[1][0], v1[2][0], v1[3][0]);
                    if (grp == 3 && valid)  // the lane's sixteen consecutive filters (ws_row_filter): one 16-byte store
                        *reinterpret_cast<uint4 *>(a.y + ((size_t)a.out_lead + ((size_t)b * (OH + 1) + (prow + 1)) * (OW + 1) + pcol) * a.out_cs + 32 * mt + 16 * kh) =
                            uint4{po[0][0], po[0][1], po[0][2], po[0][3]};
                } else {
                    // no pool behind this layer (the 3x3 layers of the non-tiny nets' residual blocks): the four window
                    // positions of the lane are four output pixels, all sixteen values are requantised and stored
                    int32_t v[4][4];
                    if (pow2) {
                        requant_values<ACT, SAT, 4>(accb, mp, a.zp_act, v);
                    } else {
                        requant16_two_step<ACT, SAT>(accb, a.mval + ch0, a.sval + ch0, a.zp_act, v);
                    }
#pragma unroll
                    for (int j = 0; j < 4; ++j) po[j][grp] = pack4_biased(v[0][j], v[1][j], v[2][j], v[3][j]);
                    if (grp == 3 && valid) {  // four output pixels x the lane's sixteen consecutive filters (ws_row_filter): four 16-byte stores
                        uint8_t *oy = a.y + ((size_t)a.out_lead + ((size_t)b * (a.H + 1) + (2 * prow + 1)) * W1 + 2 * pcol) * a.out_cs + 32 * mt + 16 * kh;
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            *reinterpret_cast<uint4 *>(oy + ((size_t)(j >> 1) * W1 + (j & 1)) * a.out_cs) = uint4{po[j][0], po[j][1], po[j][2], po[j][3]};
                    }
                }
            }
        }
        SMP_MARK_V(5, pk[NM - 1][3]);
        pk_outp = outp;
#ifdef MI355_ABLATE
        smp[7] += 1;
#endif
        pk_valid = valid && POOL;
    }
    flush_stores();
    SMP_MARK(6);
    SMP_STORE();
}

// ---------------------------------------------------------------------------------------------------------------
// 64 input channels (layer 6 of yolov3-tiny, 64 -> 128 at 52x52, + its maxpool): the same idea with the output
// channels split over the waves.  A workgroup is n / 32 waves; wave w keeps the A fragments of channels [32w, 32w + 32)
// for all 18 K-steps (tap = s / 2, channel half = s % 2, the k-half selects the 16-byte piece); two such sets of waves
// share a workgroup and deal the tile's groups of 32 pooled pixels between them, so that every SIMD holds two waves whose
// MFMA and requantise phases overlap.  The tile size is chosen by the launcher so that the tiles fill whole rounds of CUs.  The image (four piece planes), its per-cell channel sums, the 3x3 box sums and
// the image address / output cell of every pooled pixel are prepared once per tile by all waves together.  One tile
// per workgroup, single-buffered.
// Against the row-image kernel + stand-alone maxpool this does a quarter of the requantisations, keeps the weights out
// of the loop and has no K-loop barriers: 36 + 5 us -> see profiles/.
// ---------------------------------------------------------------------------------------------------------------
template <int ACT, bool SAT, int MODE = 0>  // MODE as in conv_small_pool_kernel: 0 pool, 1 no pool, 2 stride 2
__global__ __launch_bounds__(512, 2) void conv_mid_pool_kernel(const ConvArgs a)
{
    constexpr bool POOL = MODE == 0;
    constexpr int NJ = MODE == 2 ? 1 : 4;
    constexpr int KST = 18, PIECES = 4;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int ncell = a.sm_ncell, rowb = ncell * 16, pieceb = a.sm_pieceb;  // ncell: slots per LDS row (pitch)
    const int lcell = a.sm_lcell, hc = a.sm_hc, hcb = hc * 16;              // image cells per row; rows de-interleaved by column parity (see the file header)
    const bool patch = a.tiles_x > 0;
    const int N = a.n;
    const int GT = (a.sm_tp + 31) >> 5;                                   // groups of 32 pooled pixels per tile: the tables are sized by it (round 6: was GMAX)
    int *ldsS = reinterpret_cast<int *>(smem + PIECES * pieceb);          // [rows_cap * ncell] per-cell channel sums
    int *ldsSX = ldsS + ((a.rows_cap * ncell + 1) & ~1);                  // [G][4][32] 3x3 box sums per pre-pool pixel
    int *ldsBase = ldsSX + GT * 128;                                      // [G][4][32] image byte offset of tap (0,0)
    long *ldsCell = reinterpret_cast<long *>(ldsBase + GT * 128);         // [G][32] pooled output cell, -1: no pixel
    double *ldsMP = reinterpret_cast<double *>(smem + a.lds_param_off);   // [N] folded multiplier
    int *ldsDZ = reinterpret_cast<int *>(ldsMP + N);
    int *ldsCB = ldsDZ + N;
    int *ldsLO = ldsCB + N, *ldsHI = ldsLO + N;                           // POOL: lower end and width of the wrap-safe range (biased accumulators)
    int *ldsM0 = ldsHI + N, *ldsSH = ldsM0 + N;                           // integer requantisation: M0, s - 1 (common.h intrq_make)
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char *)smem;

    TSM(0);
    const int tid = threadIdx.x, NT = blockDim.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), nwave = NT >> 6;
    const int nq = N >> 5;                      // 32-channel quads = waves of one set
    const int wq = wave % nq, wset = wave / nq, nset = nwave / nq;  // this wave: channels [32 wq, +32), pixel groups wset, wset + nset, ..
    const int TP = a.sm_tp, G = (TP + 31) >> 5;  // pooled pixels per tile (patches: 128), groups of 32
    const int kh = lane >> 5, lj = lane & 31;
    const int W1 = a.W + 1;
    const int OH = a.H >> 1, OW = a.W >> 1, ohw = OH * OW;
    const int total_p = a.B * ohw;
    const int tpi = a.tiles_x * a.tiles_y;
    const bool pow2 = a.hdr->pow2 == 1;
    // one tile per workgroup; workgroup id w runs on XCD w % 8: every XCD gets one contiguous share of the tiles (neighbouring tiles share
    // their halo rows through that XCD's L2 -- see conv_small_pool_kernel)
    const int tile = (a.debug & 2048) ? (int)blockIdx.x
                                      : (int)(blockIdx.x & 7) * (int)(gridDim.x >> 3) + min((int)(blockIdx.x & 7), (int)(gridDim.x & 7)) + (int)(blockIdx.x >> 3);

    // ---- tile geometry (as in conv_small_pool_kernel)
    int gr_first, col0, nrows, pb = 0, pty = 0, ptx = 0;
    if (patch) {
        pb = tile / tpi;
        const int t = tile - pb * tpi;
        pty = t / a.tiles_x;
        ptx = t - pty * a.tiles_x;
        gr_first = pb * (a.H + 1) + 16 * pty + 1;
        col0 = 32 * ptx - 1;
        nrows = 18;
    } else {
        const int p0 = tile * TP;
        const int p1 = min(p0 + TP, total_p) - 1;
        const int b0 = fd_div(p0, a.fd_hw), r0 = fd_div(p0 - b0 * ohw, a.fd_w);
        const int b1 = fd_div(p1, a.fd_hw), r1 = fd_div(p1 - b1 * ohw, a.fd_w);
        gr_first = b0 * (a.H + 1) + 2 * r0 + 1;
        col0 = -1;
        nrows = b1 * (a.H + 1) + 2 * r1 + 2 - gr_first + 3;
    }
    // ---- image DMA: instruction k fills cells [64k, 64k + 64) of every piece plane (the last one shifted back)
    {
        const int ncells = nrows * ncell;
        const long org = (long)a.in_lead + (long)(gr_first - 1) * W1 + col0;
        for (int k = wave; k * 64 < ncells; k += nwave) {
            const int start = min(k * 64, a.rows_cap * ncell - 64);
            const int slot = start + lane;
            const int r = slot / ncell, cs = slot - r * ncell;
            const int c = min(cs < hc ? 2 * cs : 2 * (cs - hc) + 1, lcell - 1);  // slot -> image cell (pad slots repeat the last cell)
            long f = org + (long)r * W1 + c;
            f = f < 0 ? 0 : (f > a.in_cells - 1 ? a.in_cells - 1 : f);
            const unsigned voff = (unsigned)(f * a.in_cs);
#pragma unroll
            for (int p = 0; p < PIECES; ++p) {
                const unsigned dst = lds0 + p * pieceb + start * 16;
                const unsigned v = voff + p * 16;
                DMA_S(dst, a.x, v);
            }
        }
    }
    TSM(1);
    // ---- per-channel parameters, wrap-safe ranges, this wave's A fragments (overlap the DMA)
    // (POOL: accumulators biased by the safe range's lower end, window maxima requantised with two integer instructions where every channel
    // qualifies -- pool_requant_quad_biased, as in conv_small_pool_kernel)
    constexpr bool INTRQC = (ACT == MI355_ACT_LEAKY || ACT == MI355_ACT_RELU6) && !SAT;
    int never_l = 0, noint_l = 0;
    const bool ept_ok = POOL && !SAT && a.ept != nullptr && a.ept->key == ept_key(ACT, a.zp_act);  // the host's epilogue table (common.h)
    for (int i = tid; i < N; i += NT) {
        const double mp = a.mprime[i];
        ldsMP[i] = mp;
        ldsDZ[i] = a.dzp[i];
        int32_t m0 = 0, sh = 0;
        if (ept_ok) {
            const EptEntry e = reinterpret_cast<const EptEntry *>(a.ept + 1)[i];
            ldsCB[i] = e.cbl;
            ldsLO[i] = e.lb;
            ldsHI[i] = (int32_t)e.rg;
            m0 = e.m0; sh = e.sh;
        } else {
            int32_t lo = -2147483647 - 1, hi = 2147483647;
            if (!SAT && POOL) small_safe_range<ACT>(mp, a.zp_act, lo, hi);
            if constexpr (POOL) {
                int32_t lb = 0; uint32_t rg = 0;
                if (!biased_safe_range(lo, hi, lb, rg)) never_l = 1;
                if (INTRQC && !(pow2 && intrq_make(a.mval[i], a.shift[i], lb, (int32_t)((uint32_t)lb + rg), m0, sh, ACT == MI355_ACT_RELU6))) noint_l = 1;
                ldsCB[i] = (int32_t)((uint32_t)a.cwb[i] - (uint32_t)lb);
                ldsLO[i] = lb;
                ldsHI[i] = (int32_t)rg;
            } else {
                ldsCB[i] = a.cwb[i];
                ldsLO[i] = lo;
                ldsHI[i] = hi;
            }
        }
        ldsM0[i] = m0;
        ldsSH[i] = sh;
    }
    bool never_any, use_int;  // (`noint` apart from `never`: see pool_requant_quad_biased)
    if (ept_ok) {
        never_any = (a.ept->flags & EPT_NEVER) != 0;
        use_int = (a.ept->flags & EPT_NOINT) == 0;
        __syncthreads();
    } else {
        never_any = __syncthreads_or(never_l) != 0;
        use_int = __syncthreads_or(noint_l) == 0;
    }
    const bool never = POOL && (never_any || !pow2);
    v4i wf[KST];
#pragma unroll
    for (int s = 0; s < KST; ++s) wf[s] = *reinterpret_cast<const v4i *>(a.ws + ((size_t)(wq * KST + s) * 64 + lane) * 16);
    // K-step s: tap s / 2, channels 32 * (s % 2) + 16 * kh .. + 15 -> piece plane 2 * (s % 2) + kh.  The lane's k-half goes
    // into its pixel base, the tap row and the channel half are wave-uniform scalars, the tap column an immediate.
    const int khoff = kh * pieceb;

    // ---- pooled pixels of the tile: image offsets, output cells (one thread per (group, lane, window position))
    for (int idx = tid; idx < G * 128; idx += NT) {
        const int g = idx >> 7, j = (idx >> 5) & 3, l = idx & 31;
        int b, prow, pcol;
        bool valid;
        if (patch) {
            b = pb;
            prow = 8 * pty + 2 * g + (l >> 4);
            pcol = 16 * ptx + (l & 15);
            valid = prow < OH && pcol < OW;
        } else {
            const int pp = tile * TP + g * 32 + l;
            valid = pp < total_p && g * 32 + l < TP;
            const int ppc = valid ? pp : min(tile * TP + TP, total_p) - 1;  // idle lanes shadow the tile's last pixel
            b = fd_div(ppc, a.fd_hw);
            const int prem = ppc - b * ohw;
            prow = fd_div(prem, a.fd_w);
            pcol = prem - prow * OW;
        }
        const int rr = b * (a.H + 1) + 2 * prow + 1 - gr_first + (j >> 1);
        const int lch = (2 * pcol - 1 - col0) >> 1;  // half the (even) cell of the left window column's left tap; the window column is in the tap offsets
        ldsBase[idx] = rr * rowb + lch * 16;
        if (j == 0)  // output cell: the pooled pixel, or (no pool) the top-left of its four conv pixels
            ldsCell[g * 32 + l] = !valid ? -1L
                                  : POOL ? (long)a.pool_lead + ((long)b * (OH + 1) + (prow + 1)) * (OW + 1) + pcol
                                  : MODE == 2 ? (long)a.out_lead + ((long)b * (OH + 1) + (prow + 1)) * (OW + 1) + pcol
                                              : (long)a.out_lead + ((long)b * (a.H + 1) + (2 * prow + 1)) * W1 + 2 * pcol;
    }
    TSM(2);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();  // image, parameters, pixel tables
    TSM(3);

    // ---- per-cell channel sums, then the 3x3 box sum of every pre-pool pixel
    const char *X = smem;
    for (int id = tid; id < nrows * ncell; id += NT) {
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
    for (int idx = tid; idx < G * 128; idx += NT) {
        const int c0 = ldsBase[idx] >> 4;  // slot of the window row's even cell lcol: rr * ncell + (lcol >> 1)
        const int jx = (idx >> 5) & 1;
        int t = 0;
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                const int e = jx + dx;
                t += ldsS[c0 + dy * ncell + (e & 1) * hc + (e >> 1)];
            }
        ldsSX[idx] = t;
    }
    __syncthreads();

    TSM(4);
    // ---- this wave's 32 channels over the tile's four pixel groups
    const int chw = 32 * wq;
#pragma unroll 1
    for (int g = wset; g < G; g += nset) {
        int base[4], sx[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            base[j] = ldsBase[(g * 4 + j) * 32 + lj] + khoff;
            sx[j] = ldsSX[(g * 4 + j) * 32 + lj];
        }
        const long pcell = ldsCell[g * 32 + lj];
        v16i acc[4];
#pragma unroll
        for (int grp = 0; grp < 4; ++grp) {
            const int4 c4 = *reinterpret_cast<const int4 *>(ldsCB + chw + 16 * kh + 4 * grp);
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                acc[j][grp * 4 + 0] = c4.x; acc[j][grp * 4 + 1] = c4.y;
                acc[j][grp * 4 + 2] = c4.z; acc[j][grp * 4 + 3] = c4.w;
            }
        }
        if (!(a.debug & 131072)) __builtin_amdgcn_s_setprio(3);  // see conv_small_pool_kernel
#pragma unroll
        for (int s = 0; s < KST; ++s) {
            const int soff = ((s >> 1) / 3) * rowb + (s & 1) * 2 * pieceb;  // scalar
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const int e = (j & 1) + (s >> 1) % 3;  // window column + tap column -> parity half and slot of the de-interleaved row
                const v4i bf = *reinterpret_cast<const v4i *>(X + base[j] + soff + (e & 1) * hcb + (e >> 1) * 16);
                acc[j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(wf[s], bf, acc[j], 0, 0, 0);
            }
            if (s & 1) __builtin_amdgcn_sched_barrier(0);  // keep the B fragments of at most two K-steps live (registers)
        }
        __builtin_amdgcn_s_setprio(0);
        uint32_t pk4[4] = {0, 0, 0, 0};
        uint32_t pj[POOL ? 1 : NJ][4];  // no-pool / stride-2 modes: packed bytes of the four channel groups, stored together
        (void)pj; (void)pk4;
        // POOL: a group's constants are read while the group before it is requantised (see conv_small_pool_kernel)
        struct GroupConst { int4 dz, lo, hi, m0, sh; };
        auto group_const = [&](int grp) {
            const int ch0 = chw + 16 * kh + 4 * grp;
            GroupConst g;
            g.dz = *reinterpret_cast<const int4 *>(ldsDZ + ch0);
            g.lo = *reinterpret_cast<const int4 *>(ldsLO + ch0);
            g.hi = *reinterpret_cast<const int4 *>(ldsHI + ch0);
            g.m0 = *reinterpret_cast<const int4 *>(ldsM0 + ch0);
            g.sh = *reinterpret_cast<const int4 *>(ldsSH + ch0);
            return g;
        };
        GroupConst gnext = {};
        if constexpr (POOL) {
            gnext = group_const(0);
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int grp = 0; grp < 4; ++grp) {
            const int ch0 = chw + 16 * kh + 4 * grp;  // (ws_row_filter)
            GroupConst gc = gnext;
            if constexpr (POOL) {
                if (grp < 3) gnext = group_const(grp + 1);
                __builtin_amdgcn_sched_barrier(0);
            } else {
                gc.dz = *reinterpret_cast<const int4 *>(ldsDZ + ch0);
                gc.lo = *reinterpret_cast<const int4 *>(ldsLO + ch0);
                gc.hi = *reinterpret_cast<const int4 *>(ldsHI + ch0);
            }
            const int4 dz4 = gc.dz, lo4 = gc.lo, hi4 = gc.hi;
            const int dzv[4] = {dz4.x, dz4.y, dz4.z, dz4.w};
            const int lov[4] = {lo4.x, lo4.y, lo4.z, lo4.w}, hiv[4] = {hi4.x, hi4.y, hi4.z, hi4.w};
            int32_t accb[4][4];
            double mp[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                mp[r] = ldsMP[ch0 + r];
#pragma unroll
                for (int j = 0; j < 4; ++j) accb[r][j] = j >= NJ ? 0 : acc[j][grp * 4 + r] + __mul24(dzv[r], sx[j]);
            }
            if constexpr (POOL) {
                const int4 m04 = gc.m0, sh4 = gc.sh;
                const int m0v[4] = {m04.x, m04.y, m04.z, m04.w}, shv[4] = {sh4.x, sh4.y, sh4.z, sh4.w};
                uint32_t ub[4][4];
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int j = 0; j < 4; ++j) ub[r][j] = (uint32_t)accb[r][j];
                pk4[grp] = pool_requant_quad_biased<ACT, SAT>(ub, lov, hiv, never, m0v, shv, ldsMP + ch0, a.zp_act, pow2, a.mval + ch0, a.sval + ch0, use_int);
                if (grp == 3 && pcell >= 0)  // the lane's sixteen consecutive filters 16 kh .. + 15 (ws_row_filter): one store
                    *reinterpret_cast<uint4 *>(a.ypool + (size_t)pcell * a.pool_cs + chw + 16 * kh) = uint4{pk4[0], pk4[1], pk4[2], pk4[3]};
            } else if constexpr (MODE == 2) {  // stride 2: window position 0 is the output pixel
                int32_t a1[4][1], v1[4][1];
#pragma unroll
                for (int r = 0; r < 4; ++r) a1[r][0] = accb[r][0];
                if (pow2) {
                    requant_values<ACT, SAT, 1>(a1, mp, a.zp_act, v1);
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        v1[r][0] = (int32_t)requant_u8(a1[r][0], 0, a.mval[ch0 + r], a.sval[ch0 + r], a.zp_act, ACT,
                                                       SAT ? MI355_STORE_SATURATE : MI355_STORE_WRAP);
                }
                pj[0][grp] = pack4_biased(v1[0][0], v1[1][0], v1[2][0], v1[3][0]);
                if (grp == 3 && pcell >= 0)
                    *reinterpret_cast<uint4 *>(a.y + (size_t)pcell * a.out_cs + chw + 16 * kh) = uint4{pj[0][0], pj[0][1], pj[0][2], pj[0][3]};
            } else {  // no pool: four output pixels per lane, sixteen requantisations
                int32_t v[4][4];
                if (pow2) {
                    requant_values<ACT, SAT, 4>(accb, mp, a.zp_act, v);
                } else {
                    requant16_two_step<ACT, SAT>(accb, a.mval + ch0, a.sval + ch0, a.zp_act, v);
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) pj[j][grp] = pack4_biased(v[0][j], v[1][j], v[2][j], v[3][j]);
                if (grp == 3 && pcell >= 0) {
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        *reinterpret_cast<uint4 *>(a.y + (size_t)(pcell + (j >> 1) * W1 + (j & 1)) * a.out_cs + chw + 16 * kh) =
                            uint4{pj[j][0], pj[j][1], pj[j][2], pj[j][3]};
                }
            }
        }
    }
    TSM(5);
}

template <int ACT>
static int mid_launch_sat(ConvArgs &a, hipStream_t st, int grid, int threads, size_t lds)
{
    const bool sat = a.store_mode == MI355_STORE_SATURATE;
    if (a.ypool) return sat ? launch_big_lds<conv_mid_pool_kernel<ACT, true, 0>>(grid, threads, lds, st, a) : launch_big_lds<conv_mid_pool_kernel<ACT, false, 0>>(grid, threads, lds, st, a);
    if (a.stride == 2) return sat ? launch_big_lds<conv_mid_pool_kernel<ACT, true, 2>>(grid, threads, lds, st, a) : launch_big_lds<conv_mid_pool_kernel<ACT, false, 2>>(grid, threads, lds, st, a);
    return sat ? launch_big_lds<conv_mid_pool_kernel<ACT, true, 1>>(grid, threads, lds, st, a) : launch_big_lds<conv_mid_pool_kernel<ACT, false, 1>>(grid, threads, lds, st, a);
}

// ---------------------------------------------------------------------------------------------------------------
// Persistent 4-wave workgroups: `grid` is what the LDS allows per CU; what the registers of THIS instantiation allow is asked of the
// runtime once (numRegs -> allocation granule of 8 -> waves per SIMD = workgroups per CU) -- a workgroup beyond that does not run beside
// the others but after them, a second round of whole tile walks (measured on the 16-channel kernel: 38 -> 53 us at 170 registers with
// three launched per CU; its RELU6 instantiations need 188-200)
template <void (*kern)(const ConvArgs)>
static int small_launch_kern(ConvArgs &a, hipStream_t st, int grid, size_t lds)
{
    static std::atomic<int> per_cu_regs[64];  // per kernel instantiation (the template argument) and per device; zero-initialised.  Two host
                                              // threads that race here both compute the same value (a property of the code object): atomic, no lock
    int dev = 0;
    (void)hipGetDevice(&dev);
    int pc = per_cu_regs[dev & 63].load(std::memory_order_relaxed);
    if (pc == 0) {
        hipFuncAttributes fa;
        pc = 8;
        if (hipFuncGetAttributes(&fa, reinterpret_cast<const void *>(kern)) == hipSuccess && fa.numRegs > 0) {
            const int alloc = (fa.numRegs + 7) & ~7;
            pc = 512 / alloc < 1 ? 1 : 512 / alloc;
        } else {
            (void)hipGetLastError();  // not fatal: the launch below must not report this query's error as its own
        }
        per_cu_regs[dev & 63].store(pc, std::memory_order_relaxed);
    }
    if (grid > 256 * pc) grid = 256 * pc;
    return launch_big_lds<kern>(grid, 256, lds, st, a);
}

template <int C, int NM, int ACT, int POOL>
static int small_launch_sat2(ConvArgs &a, hipStream_t st, int grid, size_t lds)
{
#ifndef MI355_SMALL_NO_VDZ  // (A/B builds: the MFMA form of the zero-point correction under both plans)
    if (C == 16 && POOL == 0 && a.plan == MI355_PLAN_THROUGHPUT && a.store_mode != MI355_STORE_SATURATE)
        return small_launch_kern<conv_small_pool_kernel<C, NM, ACT, false, POOL, (C == 16)>>(a, st, grid, lds);
#endif
    if (a.store_mode == MI355_STORE_SATURATE) return small_launch_kern<conv_small_pool_kernel<C, NM, ACT, true, POOL>>(a, st, grid, lds);
    return small_launch_kern<conv_small_pool_kernel<C, NM, ACT, false, POOL>>(a, st, grid, lds);
}

template <int C, int NM, int ACT>
static int small_launch_sat(ConvArgs &a, hipStream_t st, int grid, size_t lds)
{
    if (a.ypool) return small_launch_sat2<C, NM, ACT, 0>(a, st, grid, lds);
    return a.stride == 2 ? small_launch_sat2<C, NM, ACT, 2>(a, st, grid, lds) : small_launch_sat2<C, NM, ACT, 1>(a, st, grid, lds);
}

template <int C, int NM>
static int small_launch_act(ConvArgs &a, hipStream_t st, int grid, size_t lds)
{
    if (a.act == MI355_ACT_LEAKY) return small_launch_sat<C, NM, MI355_ACT_LEAKY>(a, st, grid, lds);
    if (a.act == MI355_ACT_RELU6) return small_launch_sat<C, NM, MI355_ACT_RELU6>(a, st, grid, lds);
    return small_launch_sat<C, NM, MI355_ACT_LINEAR>(a, st, grid, lds);
}

bool conv_small_eligible(int n, int c, int ksize)
{
    if (ksize != 3) return false;
    if (c == 64) return n % 32 == 0 && n >= 64 && n <= 128;  // conv_mid_pool_kernel: n / 32 waves (2..4) per workgroup
    return (c == 16 || c == 32) && (n == 32 || n == 64);
}

// returns MI355_EINVAL when the shape is outside this kernel's domain (the caller falls back to conv_igemm.hip)
int conv_small_pool_launch(ConvArgs &a, hipStream_t st)
{
    const int c = a.cb * a.nchunks;
    // conv + maxpool (ypool, no y), or the same kernels without the pool (y, no ypool): four output pixels per lane
    if (!conv_small_eligible(a.n, c, a.ksize) || a.acc_out || a.y_f32 || !a.ws) return MI355_EINVAL;
    if (a.ypool ? (a.y != nullptr || a.stride != 1) : (a.y == nullptr || a.out_w < a.n || a.up != 1)) return MI355_EINVAL;
    a.debug = mi355_debug_flags_get();
    if ((a.H & 1) || (a.W & 1) || a.in_cs != c) return MI355_EINVAL;
    if (c == 64 && (a.debug & (1 << 26))) return MI355_EINVAL;  // A/B switch (tools/dbg): the 64-channel layer on the row-image kernel
    // no fused residual add here: measured (YOLOv3 @608, batch 32) 412 us fused against 150 us + an 88 us stand-alone add for
    // 32 -> 64 @304 -- these kernels' stores are already their bottleneck, the add's loads queue in front of them
    if (a.res) return MI355_EINVAL;
    if ((size_t)a.in_cells * (size_t)a.in_cs >= ((size_t)1 << 32)) return MI355_EINVAL;  // 32-bit DMA lane offsets
    const int OH = a.H / 2, OW = a.W / 2;
    const long total_p = (long)a.B * OH * OW;
    if (total_p + 256 >= (1L << 31)) return MI355_EINVAL;
    a.fd_hw = fastdiv_make((uint32_t)(OH * OW));  // the kernels' flat-tile geometry: pooled pixel -> (image, row, column)
    a.fd_w = fastdiv_make((uint32_t)OW);
    int ntiles;
    if (OW >= 64) {  // wide map: 8 x 16 pooled patches (a flat run of 128 would load two mostly unused rows)
        a.tiles_x = (OW + 15) / 16;
        a.tiles_y = (OH + 7) / 8;
        a.sm_lcell = 34;
        a.sm_ncell = 40;  // pitch: the wave's second pooled row (two image rows down) starts on the same 16-byte bank group (2 * 40 = 0 mod 16)
        a.rows_cap = 18;
        ntiles = a.B * a.tiles_x * a.tiles_y;
    } else {
        // rows of a 128-pooled-pixel run: pooled rows it can touch, two image rows each, one pad row per image boundary
        // crossed, one halo row above and below
        a.tiles_x = a.tiles_y = 0;
        a.sm_lcell = a.W + 2;
        // pitch: a wave whose pixels wrap into the next pooled row (slot + 2 * pitch - OW) continues the sequence of bank groups when
        // 2 * pitch = OW mod 16 (even OW; with an odd one the wrap costs a conflict)
        a.sm_ncell = a.sm_lcell + ((OW & 1) ? 0 : ((OW / 2 - a.sm_lcell) % 8 + 8) % 8);
        a.rows_cap = 2 * ((SM_PPB - 2 + OW) / OW + 1) + (SM_PPB - 2 + OH * OW) / (OH * OW) + 2;
        ntiles = (int)((total_p + SM_PPB - 1) / SM_PPB);
    }
    a.sm_hc = a.sm_lcell / 2;  // (W is even: so is the cell count)
    a.sm_pieceb = a.rows_cap * a.sm_ncell * 16;
    if (c == 64) {  // channels split over n / 32 waves x 2 sets, one single-buffered tile per workgroup
        int tp = SM_PPB;
        if (a.tiles_x == 0) {
            // flat tiles: as few rounds of 256 workgroups as the 256-pixel tile limit allows, tiles of equal size
            // (half-size tiles -- two workgroups per CU -- measured slower alone (r02) and with three batches in flight (r03:
            // 0.2820 -> 0.2865 ms per step): every workgroup loads its own copy of the A fragments)
            const long rounds = (total_p + 256L * SM_GMAX * 32 - 1) / (256L * SM_GMAX * 32);
            tp = (int)((total_p + 256 * rounds - 1) / (256 * rounds));
            if (tp < 32) tp = 32;
            // Round 6, throughput plan (half workgroups, see below): tiles of WHOLE groups of 32 pooled pixels, at most four -- the 169 pixels
            // that fill one round of the chip are five groups and nine lanes of a sixth, and nothing about a half workgroup that shares its CU
            // asks for one round.  Layer 6 in flight, same box: 169 pixels 14.7 us, 160 13.8, 128 13.9 (and 25.9 instead of 29.5 us alone),
            // 96 15.0, 64 15.8 (profiles/r06_mid_tile_sizes_flood.log).  MI355_MID_TP=<pixels> overrides for A/B runs.
            static const int tp_env = getenv("MI355_MID_TP") ? atoi(getenv("MI355_MID_TP")) : 0;
            static const bool mid_full_tp = getenv("MI355_MID_FULL") != nullptr;
            if (a.plan == MI355_PLAN_THROUGHPUT && !mid_full_tp) {
                if (tp_env > 0) tp = tp_env;
                else if (tp > 128) tp = 128;
                else if (tp >= 32) tp = (tp / 32) * 32;
            }
            a.rows_cap = 2 * ((tp - 2 + OW) / OW + 1) + (tp - 2 + OH * OW) / (OH * OW) + 2;
            a.sm_pieceb = a.rows_cap * a.sm_ncell * 16;
            ntiles = (int)((total_p + tp - 1) / tp);
        }
        a.sm_tp = tp;
        // Round 6: HALF workgroups -- one wave set (n / 32 waves) instead of two -- wherever two of them fit a CU and something can run beside a
        // workgroup: launches of several rounds, and every launch under the throughput plan (other batches' launches).  The kernel's 213-219
        // VGPRs allow two waves per SIMD = eight waves per CU: ONE eight-wave workgroup, whose load phase (A fragments, image DMA, tables: 5.3 of
        // 17.9 us, tools/wg_timeline.py) and tail nothing covered -- or TWO four-wave workgroups that run them under each other's group loops.
        // Flat tiles drop the bank-alignment padding of their row pitch for it (layer 6: 93.4 -> 79.6 KB of LDS).  Layer 6 with four batches in
        // flight 18.2 -> 15.2 us per launch, whole in-flight step 242.3 -> 239.3 us, bench 0.2423 -> 0.2381 ms per step; alone the layer is slower
        // (21.6 -> 27.4 us: half the waves per CU), which is the latency plan's concern and why one-round launches keep whole workgroups there
        // (profiles/r06_mid_half_*; MI355_MID_FULL=1 restores whole workgroups everywhere for A/B runs).
        static const bool mid_full = getenv("MI355_MID_FULL") != nullptr;
        const bool want_half = !mid_full && (a.plan == MI355_PLAN_THROUGHPUT || ntiles > 256);
        const int gt = (tp + 31) / 32;
        auto lds_need = [&]() {
            size_t l = 4 * (size_t)a.sm_pieceb + (size_t)((a.rows_cap * a.sm_ncell + 1) & ~1) * 4 + (size_t)gt * 128 * 8 + (size_t)gt * 32 * 8;
            l = (l + 15) & ~(size_t)15;
            a.lds_param_off = (int)l;
            return l + (size_t)a.n * 32;
        };
        const int ncell_padded = a.sm_ncell;
        bool half = false;
        if (want_half) {
            if (a.tiles_x == 0) {  // flat tiles: the unpadded pitch (a wave whose pixels wrap into the next pooled row then takes a bank conflict there)
                a.sm_ncell = a.sm_lcell;
                a.sm_pieceb = a.rows_cap * a.sm_ncell * 16;
            }
            half = 2 * lds_need() <= 160 * 1024 && a.rows_cap * a.sm_ncell >= 64;
            if (!half) {
                a.sm_ncell = ncell_padded;
                a.sm_pieceb = a.rows_cap * a.sm_ncell * 16;
            }
        }
        if (a.rows_cap * a.sm_ncell < 64) return MI355_EINVAL;
        const size_t l64 = lds_need();
        if (l64 > 160 * 1024) return MI355_EINVAL;
        const int threads = (half ? 1 : 2) * (a.n / 32) * 64;
        if (a.act == MI355_ACT_LEAKY) return mid_launch_sat<MI355_ACT_LEAKY>(a, st, ntiles, threads, l64);
        if (a.act == MI355_ACT_RELU6) return mid_launch_sat<MI355_ACT_RELU6>(a, st, ntiles, threads, l64);
        return mid_launch_sat<MI355_ACT_LINEAR>(a, st, ntiles, threads, l64);
    }
    if (a.rows_cap * a.sm_ncell < 64 || a.rows_cap * a.sm_ncell > 4 * SM_KMAX * 64) return MI355_EINVAL;
    size_t lds = 2 * (size_t)(c / 16) * a.sm_pieceb + (size_t)a.rows_cap * a.sm_ncell * 4;
    lds = (lds + 15) & ~(size_t)15;
    a.lds_param_off = (int)lds;
    lds += (size_t)a.n * 32;
    if (lds > 160 * 1024) return MI355_EINVAL;
    // persistent workgroups per CU: LDS permitting; the c = 16, n = 32 variant needs few enough registers for three
    int per_cu = (2 * lds <= 160 * 1024) ? 2 : 1;
    if (c == 16 && a.n == 32 && 3 * lds <= 160 * 1024) per_cu = 3;
    static const int per_cu_env = getenv("MI355_SMALL_PER_CU") ? atoi(getenv("MI355_SMALL_PER_CU")) : 0;  // (A/B runs)
    if (per_cu_env > 0 && per_cu_env < per_cu) per_cu = per_cu_env;
    const int grid = ntiles < 256 * per_cu ? ntiles : 256 * per_cu;
    if (c == 16 && a.n == 32) return small_launch_act<16, 1>(a, st, grid, lds);
    if (c == 16 && a.n == 64) return small_launch_act<16, 2>(a, st, grid, lds);
    if (c == 32 && a.n == 32) return small_launch_act<32, 1>(a, st, grid, lds);
    return small_launch_act<32, 2>(a, st, grid, lds);
}
