// conv_rows.hip -- the MFMA implicit-GEMM INT8 convolution for layers whose input channels come in 64-byte chunks
// (every 3x3 s1 / 1x1 layer of yolov3-tiny from the 4th conv on, and BASELINE config[1]).  Same mathematics as
// conv_igemm.hip (read its header for the signed-operand decomposition); what differs is the inner loop, which is
// built so that a K-step is almost nothing but ds_read_b128 + V_MFMA_I32_32X32X32_I8:
//
//   * LDS image of the B operand = whole image ROWS of the PHWC tensor, RS cells per row (RS = 16/32/64 >= W+2,
//     a template constant), stored per row as [16-byte piece][RS cells][16 B], rows skewed by W mod 16 cells so that a
//     wave's 32 consecutive pixels never collide in a bank across a row wrap.  A 3x3 tap (dy,dx) is a per-row table
//     entry plus the immediate dx*16: no per-step address arithmetic.
//   * the receptive-field sums  sum_k x'  (needed because V_MFMA_*_I8 is signed x signed) are reduced once per channel
//     chunk from the landed row image into an LDS int32 plane S[cell]; the epilogue forms the 3x3 box sum of S
//     (1x1: straight from the B fragments in registers).
//   * A operand: 6-stage DMA ring five K-steps ahead (4 stages in the 128-column configurations, the 1x1 loop 3-4 for
//     A and B alike); B operand: double-buffered per channel chunk, its DMA slots spread over a chunk's first four
//     K-steps.  LDS-DMA (global_load_lds_dwordx4) is written by hand with scalar bases; one s_barrier per K-step.
//   * 3x3 K loop: nine K-steps per chunk fully unrolled in a steady-state and a last-chunk variant, every vmcnt /
//     lgkmcnt wait an immediate; three fragment register sets, a set is read one K-step before its MFMAs and the
//     reads are threaded between the MFMAs of the previous sets.
//   * N tiles split the pixel range evenly (host-side tile planner in conv_igemm_launch), 32-column sub-tiles are dealt
//     round-robin to the N waves, workgroups walk the channel chunks in rotated order.
//   * epilogue: accumulators start at cw + bias, one FP64 multiply per output (folded multiplier), branch-free leaky,
//     LDS-transposed stores; optional fused nearest-neighbour upsample store and fused yolo head activations.
//
// Measured motivation and the microbenchmarks behind each of these choices: DESIGN.md section 3.1, profiles/r01_*.
#include "kargs.h"
#include <type_traits>

// timing-ablation switches (tools/conv_microbench.py --ablate) exist only in builds made with -DMI355_ABLATE;
// in the product build DBG(x) is the constant 0 and every switch folds away.
#if defined(MI355_ABLATE) && !defined(ROWS_TU_KS1)
#define DBG(bit) ((a.debug & (bit)) != 0)
// phase timestamps (100 MHz wall clock) of wave 0 of every workgroup: tools/conv_microbench.py --timeline
#define TS_PHASES 6
#define TS_BLOCKS 4096
__device__ long long g_rows_ts[TS_PHASES][TS_BLOCKS];
#define TS(k)                                                                                   \
    do {                                                                                        \
        if (threadIdx.x == 0 && blockIdx.x < TS_BLOCKS) g_rows_ts[k][blockIdx.x] = wall_clock64(); \
    } while (0)
extern "C" int mi355_debug_read_ts(long long *host)
{
    return hipMemcpyFromSymbol(host, HIP_SYMBOL(g_rows_ts), sizeof(long long) * TS_PHASES * TS_BLOCKS) == hipSuccess ? 0 : -5;
}
// in-loop stall profile (debug bit 256): per wave, shader-clock sums of the five phases of a K-step
#define WP_PHASES 5
__device__ long long g_rows_wp[TS_BLOCKS][8][WP_PHASES];
#define WP_DECL long long wp_acc[WP_PHASES] = {0, 0, 0, 0, 0}; long long wp_t = 0
#define WP_START()                                                   \
    do {                                                             \
        if (DBG(256)) { wp_t = __builtin_readcyclecounter(); }       \
    } while (0)
#define WP_MARK(k)                                                   \
    do {                                                             \
        if (DBG(256)) {                                              \
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       \
            const long long wp_n = __builtin_readcyclecounter();     \
            wp_acc[k] += wp_n - wp_t;                                \
            wp_t = wp_n;                                             \
        }                                                            \
    } while (0)
#define WP_FLUSH()                                                                                   \
    do {                                                                                             \
        if (DBG(256) && lane == 0 && blockIdx.x < TS_BLOCKS && wave < 8)                             \
            for (int k = 0; k < WP_PHASES; ++k) g_rows_wp[blockIdx.x][wave][k] = wp_acc[k];          \
    } while (0)
extern "C" int mi355_debug_read_wp(long long *host)
{
    return hipMemcpyFromSymbol(host, HIP_SYMBOL(g_rows_wp), sizeof(long long) * TS_BLOCKS * 8 * WP_PHASES) == hipSuccess ? 0 : -5;
}
#else
#define DBG(bit) (false)
#define TS(k) do { } while (0)
#define WP_DECL do { } while (0)
#define WP_START() do { } while (0)
#define WP_MARK(k) do { } while (0)
#define WP_FLUSH() do { } while (0)
#endif

#define DMA16(gsrc, ldst)                                                                               \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(gsrc),           \
                                     (__attribute__((address_space(3))) void *)(ldst), 16, 0, 0)

// A ring depth: 6 for the software-pipelined 3x3 loop (fragments of step g+1 are read while the MFMAs of step g run, so
// A(g), A(g+1) are being read while A(g+2) .. A(g+5) are in flight) -- 4 in the 128-column configurations, whose LDS
// then lets two workgroups share a CU (one's epilogue under the other's K loop); 4 (3 for the widest tiles: LDS) for the plain 1x1
// loop, which then has its DMA two or three K-steps ahead instead of one -- its K-steps took 0.7 us each, the DMA latency
template <int KS, int BN> constexpr int ra_stages() { return KS == 3 ? (BN <= 192 ? 4 : 6) : (BN <= 128 ? 4 : 3); }
// B buffers: two per-chunk row images for 3x3 (a chunk lasts nine K-steps); for 1x1 every K-step is a new chunk and the
// row image rides the same ring as the weights
template <int KS, int BN> constexpr int rb_stages() { return KS == 3 ? 2 : ra_stages<KS, BN>(); }
// B DMA slots per wave per channel-chunk load: a compile-time constant per configuration (one VGPR of source offset
// each, issued unconditionally -- slots past a tile's last LDS row repeat its last one), so that every vmcnt wait of
// the K loop is an immediate.  Sized for the rows a BN-pixel tile spans when the map is at least 3/4 as wide as the
// row image (W >= 12 / 24 / 48 for RS = 16 / 32 / 64); narrower maps are refused by the launcher (other tile or kernel).
constexpr int rows_nb_slots(int BN, int RS, int NW, int KS)
{
    const int wmin = RS * 3 / 4, halo = KS == 3 ? 1 : 0;
    const int rows = (BN - 2 + wmin) / wmin + 1 + (BN - 2 + wmin * wmin) / (wmin * wmin) + 2 * halo;
    return (rows * (RS / 16) + NW - 1) / NW;
}

// global row index (over all image blocks, pad rows included) and column of valid pixel n
__device__ __forceinline__ void row_of_pixel(int n, int H, int W, FastDiv fd_hw, FastDiv fd_w, int &grow, int &x)
{
    const int hw = H * W;
    const int b = fd_div(n, fd_hw);
    const int r = n - b * hw;
    const int y = fd_div(r, fd_w);
    x = r - y * W;
    grow = b * (H + 1) + y + 1;
}

template <int BM, int BN, int WMW, int WNW, int RS, int KS>
__global__ __launch_bounds__(64 * WMW * WNW, 2) void conv_rows_i8_kernel(const ConvArgs a)
{
    constexpr int NW = WMW * WNW, NT = 64 * NW;
    constexpr int TM = BM / WMW, TN = BN / WNW;
    constexpr int MS = TM / 32, NS = TN / 32;
    constexpr int ACH = BM / 16;
    // DMA waves: in the 8-wave 3x3 kernel only the older half of the workgroup (waves 0..3, one per SIMD) issues DMA.  The
    // younger wave of every SIMD loses the issue arbitration and reaches the barrier last anyway; freed of its DMA
    // instructions (~60-180 clocks each) the two halves arrive together.  A wave that issues nothing has vmcnt = 0: the
    // counted waits below are no-ops for it
    // (Sharing the prologue DMA out over all eight waves -- the younger ones fetching what step 0 needs -- measured the same.)
    constexpr int DW = (NW == 8 && KS == 3) ? 4 : NW;
    constexpr int APT = (ACH + DW - 1) / DW;
    constexpr int HALO = (KS == 3) ? 1 : 0;
    constexpr int PIECEB = RS * 16;     // bytes between 16-byte pieces of a row
    constexpr int CPR = RS / 16;        // 1 KiB DMA chunks per row
    constexpr int OSTR = BM + 4;
    constexpr int RA_STAGES = ra_stages<KS, BN>();
    constexpr int RB_STAGES = rb_stages<KS, BN>();
    constexpr int NBS = rows_nb_slots(BN, RS, DW, KS);  // B DMA slots per (DMA) wave per chunk
    constexpr int SPS = (NBS + 3) / 4;                  // ... issued per K-step over a chunk's first four steps (3x3 loop)
    static_assert(TM % 32 == 0 && TN % 32 == 0, "wave tile must be a multiple of the 32x32 MFMA tile");

    extern __shared__ __attribute__((aligned(16))) char smem[];
    char *ldsA = smem;                                   // [RA_STAGES][BM*64]
    char *ldsB = smem + RA_STAGES * BM * 64;             // [RB_STAGES][rows_cap*rowb]
    // bytes between LDS rows: RS*64 of data + a skew of (W mod 16) cells, so that the pixel after a row's last one
    // lands in the next 16-byte bank slot -- a wave's 32 consecutive pixels then never collide across a row wrap
    // (tools/ubench/lds_conflict.hip: 2.1x slower ds_read_b128 for W = 13 without it)
    const int rowb = a.rowb;
    const int bbytes = a.rows_cap * rowb;
    int *ldsS = reinterpret_cast<int *>(ldsB + RB_STAGES * bbytes);  // [rows_cap*RS] receptive-field partial sums per cell
    // per-channel epilogue parameters of this M tile, staged once (the epilogue would otherwise issue 5 dependent
    // global loads per output channel per lane): doubles first (8-byte aligned), then the three int planes
    double *ldsPM = reinterpret_cast<double *>(smem + a.lds_param_off);  // [BM] M_value, [BM] shift_value
    int *ldsPI = reinterpret_cast<int *>(ldsPM + 3 * BM);                  // [BM] dzp, [BM] bias
    float *ldsYL = reinterpret_cast<float *>(ldsPI + 2 * BM);              // [256] fused yolo head: logistic of every byte's dequantised value
    // ldsPM: [BM] M_value, [BM] shift_value, [BM] M_value*shift_value

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WNW, wn = wave % WNW;
    const int kh = lane >> 5, lj = lane & 31;
    const int dwave = wave & (DW - 1);  // index into the DMA tables
    const bool issuer = wave < DW;      // steady-state DMA wave
    TS(0);

    int logical;
    {
        const int nb = gridDim.x, id = blockIdx.x;
        const int q = nb >> 3, r = nb & 7, xcd = id & 7, idx = id >> 3;
        logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    // `logical` gives every XCD (workgroup id % 8) one contiguous range of tiles.  Tiles are numbered in blocks of xcd_mb M tiles:
    // inside a block N-major (the xcd_mb M tiles of one N tile are neighbours), block after block -- so an XCD's range is about
    // xcd_mb M tiles x (range / xcd_mb) N tiles whatever the tile counts are (no divisibility conditions: L12's 8 x 85 tiles of
    // the throughput plan fetched 62.7 MB per launch M-major, every XCD the whole input).  xcd_mb = 1 is the M-major order; the
    // launcher picks the block height that minimises weight slabs + input rows per L2.
    int mtile, ntile;
    {
        const int blk = fd_div(logical, a.fd_ntper);            // / (xcd_mb * ntiles_n)
        const int rem = logical - blk * (a.xcd_mb * a.ntiles_n);
        ntile = fd_div(rem, a.fd_mb);
        mtile = blk * a.xcd_mb + (rem - ntile * a.xcd_mb);
    }
    // N tiles split the flattened pixel range evenly (tile widths differ by at most one pixel and never exceed BN):
    // the host picks ntiles_n so that mtiles * ntiles_n fills whole rounds of workgroups over the 256 CUs.
    const int n0 = ntile * a.tile_q + min(ntile, a.tile_r);
    const int n_end = n0 + a.tile_q + (ntile < a.tile_r ? 1 : 0);
    const int W1 = a.W + 1, hw = a.H * a.W;

    // ---- tile rows: LDS row 0 = global row (row of first pixel) - HALO
    int gr0, x0;
    row_of_pixel(n0, a.H, a.W, a.fd_hw, a.fd_w, gr0, x0);
    int gr1, x1;
    row_of_pixel(n_end - 1, a.H, a.W, a.fd_hw, a.fd_w, gr1, x1);
    const int grow_first = gr0 - HALO;
    const int nrows = gr1 - gr0 + 1 + 2 * HALO;   // <= a.rows_cap (host guarantees)
    const int ndma = nrows * CPR;                 // B DMA instructions per chunk load for the whole workgroup

    // ---- per-lane B base: LDS byte offset of (its pixel's row - HALO, its column - HALO) for k-half kh
    int bbase[NS];
    int prow[NS], pcol[NS];
    bool nvalid[NS];
#pragma unroll
    for (int ns = 0; ns < NS; ++ns) {
        const int n = n0 + (ns * WNW + wn) * 32 + lj;  // 32-column sub-tiles are dealt round-robin to the N waves
        nvalid[ns] = n < n_end;
        int gr, x;
        row_of_pixel(nvalid[ns] ? n : n_end - 1, a.H, a.W, a.fd_hw, a.fd_w, gr, x);
        prow[ns] = gr - grow_first - HALO;  // LDS row of tap dy = 0 (top tap)
        pcol[ns] = x + 1 - HALO;            // LDS cell of tap dx = 0 (left tap); cell 0 of a row is x = -1
        bbase[ns] = prow[ns] * rowb + pcol[ns] * 16 + kh * PIECEB;
    }
    int atab[MS];
#pragma unroll
    for (int ms = 0; ms < MS; ++ms) {
        const int row = wm * TM + ms * 32 + lj;
        atab[ms] = ((row >> 4) << 10) + ((row & 15) << 4) + kh * 256;
    }

    // ---- DMA helpers.  Every wave issues exactly APT (A) / NBS (B) instructions per load.  A K-step is bounded by
    // the length of each wave's own instruction stream as much as by the matrix pipe (the in-order wave issues ~50
    // scalar instructions and a dozen branches per step if the addressing is left to the compiler), so the DMA is
    // written in its "scalar 64-bit base + 32-bit lane offset" form by hand: the lane offsets are loop invariant
    // registers and everything that moves lives on the scalar unit.  M0 carries the LDS destination; no other code in
    // this kernel uses M0 (all LDS-DMA goes through this macro), so it is not declared as clobbered.
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char *)smem;
#define DMA_S(ldsdst_u32, sbase_ptr, voff_u32)                                                                   \
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(ldsdst_u32), "v"(voff_u32), \
                 "s"(sbase_ptr)                                                                                  \
                 : "memory", "m0")
    // Channel-chunk rotation: workgroup `ntile` walks the chunks in the order rot, rot+1, .., nchunks-1, 0, .., rot-1
    // (exact int32 accumulation does not care).  All workgroups run in lockstep, and a chunk is the same 64 bytes of
    // every in_cs-byte cell: without the rotation the whole chip reads one quarter of the tensor's cache lines -- a few
    // L2 channels -- at any one time, and every workgroup of an XCD wants the same new weight slab in the same instant.
    constexpr int TAPS = KS * KS;                         // K-steps per channel chunk (= ksteps / nchunks)
    const int rot = a.debug & 512 ? 0 : ntile - fd_div(ntile, a.fd_nch) * a.nchunks;
    const int kwrap = a.ksteps;                           // taps * chunks
    int aleft = kwrap - rot * TAPS;                       // slabs until the walk wraps to slab 0
    const int8_t *aptr[APT];  // next K-step slab of this wave's A chunk(s) to fetch
    unsigned adst[APT];
#pragma unroll
    for (int i = 0; i < APT; ++i) {
        const int ch = min(dwave + i * DW, ACH - 1);
        aptr[i] = a.wp + ((size_t)(mtile * ACH + ch) * a.ksteps + (size_t)rot * TAPS) * 1024;
        adst[i] = lds0 + (ch << 10);
    }
    const unsigned lane16 = lane * 16;
    auto issueA_next = [&](unsigned stage_off) {  // fetch the next slab into the ring stage at byte offset stage_off
        if (DW < NW && !issuer) return;
        const bool wrap = --aleft == 0;
#pragma unroll
        for (int i = 0; i < APT; ++i) {
            DMA_S(adst[i] + stage_off, aptr[i], lane16);
            aptr[i] += wrap ? 1024 - (long)kwrap * 1024 : 1024;
        }
        if (wrap) aleft = kwrap;
    };
    // B: DMA instruction j covers 1 KiB of LDS row j / CPR (chunk-in-row j % CPR).  Inside a row the image is
    // [piece][RS cells][16 B]: byte o -> piece o / PIECEB, cell (o % PIECEB) / 16.  Neither the lane's source offset
    // nor the LDS destination depends on the channel chunk: both are computed once per DMA slot, the chunk only moves
    // the scalar base by 64 B and the buffer parity.
    const long cell0 = (long)a.in_lead + (long)grow_first * W1 - 1;  // global cell of LDS (row 0, cell 0)
    unsigned bvoff[NBS], bdst[NBS];
#pragma unroll
    for (int i = 0; i < NBS; ++i) {
        const int j = min(dwave + i * DW, ndma - 1);
        const int r = j / CPR, cj = j - r * CPR;
        const int o = cj * 1024 + lane * 16;
        const int p = o / PIECEB, c = (o - p * PIECEB) >> 4;
        long f = cell0 + (long)r * W1 + c;
        f = f < 0 ? 0 : (f > a.in_cells - 1 ? a.in_cells - 1 : f);
        bvoff[i] = (unsigned)(f * a.in_cs + p * 16);  // < 2^32: the host rejects tensors of 4 GiB and more
        bdst[i] = lds0 + RA_STAGES * BM * 64 + r * rowb + (cj << 10);
    }
    auto issueB_slots = [&](int chunk, auto lo_c, auto hi_c) {  // slots [LO, HI) of channel chunk `chunk`
        constexpr int LO = decltype(lo_c)::value, HI = decltype(hi_c)::value;
        if (DW < NW && !issuer) return;
        const unsigned boff = (chunk % RB_STAGES) * bbytes;
        int phys = chunk + rot;  // rotated walk, see above
        if (phys >= a.nchunks) phys -= a.nchunks;
        const int8_t *base = a.x + (size_t)phys * 64;
#pragma unroll
        for (int i = LO; i < HI; ++i) {
            const unsigned d = bdst[i] + boff, v = bvoff[i];  // locals: asm operands may not name captured arrays
            DMA_S(d, base, v);
        }
    };
    auto issueB = [&](int chunk) { issueB_slots(chunk, std::integral_constant<int, 0>{}, std::integral_constant<int, NBS>{}); };

    // K-loop prologue DMA, issued as early as its addresses exist: the first weight slabs and the first row image fly
    // while the accumulators and the epilogue parameters are fetched (3x3: B(0), A(0..4); 1x1: groups 0 .. R-2)
    TS(1);
    if constexpr (KS == 3) {
        issueB(0);
#pragma unroll
        for (int st = 0; st < RA_STAGES - 1; ++st) issueA_next(st * (BM * 64));
    } else {
#pragma unroll
        for (int st = 0; st < RA_STAGES - 1; ++st)
            if (st < a.ksteps) {
                issueB(st);
                issueA_next(st * (BM * 64));
            }
    }

    // accumulators start at the per-channel constant cw + bias (blob plane cwb), so the epilogue does not add it:
    // register grp*4+r of a 32x32 tile holds channel row 8*grp + 4*kh + r (parameter planes are padded to mpad)
    v16i acc[MS][NS];
#pragma unroll
    for (int ms = 0; ms < MS; ++ms)
#pragma unroll
        for (int grp = 0; grp < 4; ++grp) {
            const int4 c4 = *reinterpret_cast<const int4 *>(a.cwb + mtile * BM + wm * TM + ms * 32 + 8 * grp + 4 * kh);
#pragma unroll
            for (int ns = 0; ns < NS; ++ns) {
                acc[ms][ns][grp * 4 + 0] = c4.x;
                acc[ms][ns][grp * 4 + 1] = c4.y;
                acc[ms][ns][grp * 4 + 2] = c4.z;
                acc[ms][ns][grp * 4 + 3] = c4.w;
            }
        }

    // zero the S plane, stage the epilogue parameters (both visible after the first barrier of the K loop)
    for (int i = tid; i < a.rows_cap * RS; i += NT) ldsS[i] = 0;
    if (a.yolo_out)  // a head's float outputs take 256 values: one table instead of a double-precision exp per element
        for (int i = tid; i < 256; i += NT) ldsYL[i] = yolo_entry_act((float)(i - a.zp_act) * a.s_act, 0);
    for (int i = tid; i < BM; i += NT) {
        const int oc = mtile * BM + i;  // parameter arrays are padded to mpad
        ldsPM[i] = a.mval[oc];
        ldsPM[BM + i] = a.sval[oc];
        ldsPI[i] = a.dzp[oc];
        ldsPI[BM + i] = a.bias[oc];
        ldsPM[2 * BM + i] = a.mprime[oc];
    }

    int sxr[NS];  // 1x1 only: the receptive field is the pixel itself -- its channel sum accumulates from the B fragments
#pragma unroll
    for (int ns = 0; ns < NS; ++ns) sxr[ns] = 0;
    auto compute = [&](const char *A, const char *Bt, int tapoff) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            v4i af[MS];
#pragma unroll
            for (int ms = 0; ms < MS; ++ms) af[ms] = *reinterpret_cast<const v4i *>(A + atab[ms] + h * 512);
#pragma unroll
            for (int ns = 0; ns < NS; ++ns) {
                const v4i bf = *reinterpret_cast<const v4i *>(Bt + bbase[ns] + tapoff + h * 2 * PIECEB);
                int t = sxr[ns];
                t = __builtin_amdgcn_sdot4(bf[0], 0x01010101, t, false);
                t = __builtin_amdgcn_sdot4(bf[1], 0x01010101, t, false);
                t = __builtin_amdgcn_sdot4(bf[2], 0x01010101, t, false);
                t = __builtin_amdgcn_sdot4(bf[3], 0x01010101, t, false);
                sxr[ns] = t;
#pragma unroll
                for (int ms = 0; ms < MS; ++ms)
                    acc[ms][ns] = __builtin_amdgcn_mfma_i32_32x32x32_i8(af[ms], bf, acc[ms][ns], 0, 0, 0);
            }
        }
    };
    // per channel chunk: every thread reduces cells of the freshly landed B buffer into S
    auto cell_sums = [&](const char *Bt) {
        const int ncells = nrows * RS;
        for (int id = tid; id < ncells; id += NT) {
            const int r = id / RS, c = id - r * RS;
            const char *p0 = Bt + r * rowb + c * 16;
            int t = 0;
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const v4i v = *reinterpret_cast<const v4i *>(p0 + p * PIECEB);
                t = __builtin_amdgcn_sdot4(v[0], 0x01010101, t, false);
                t = __builtin_amdgcn_sdot4(v[1], 0x01010101, t, false);
                t = __builtin_amdgcn_sdot4(v[2], 0x01010101, t, false);
                t = __builtin_amdgcn_sdot4(v[3], 0x01010101, t, false);
            }
            ldsS[id] += t;  // each cell is owned by exactly one thread: plain read-modify-write
        }
    };

    if constexpr (KS == 3) {
        // ---- software-pipelined 3x3 loop.  R = 6 A ring stages, three fragment register sets (one k-half each):
        //        step g:  wait A(g+1) landed; s_barrier; issue DMA A(g+5) [B(chunk+1) on a chunk's first step]
        //                 MFMA H0(g)  ||  ds_read H0(g+1)  -> the set H1(g-1) vacated
        //                 MFMA H1(g)  ||  ds_read H1(g+1)  -> the set H0(g) vacated
        //      A fragment set is read one whole K-step before its MFMAs, and a DMA is issued four steps (~2 us) before
        //      its barrier: the in-loop stall profile (tools/conv_microbench.py --waveprof) showed the previous
        //      two-set / four-stage loop waiting 8% of its time on vmcnt and ~15% on lgkmcnt.
        constexpr int R = RA_STAGES;
        static_assert(R == 6 || R == 4, "ring depths the stage bookkeeping below is written for");
        constexpr int ROT = 9 % R;  // ring phase advance of one channel chunk (nine K-steps)
        v4i fa[3][MS], fb[3][NS];
        // Fragment reads are issued as inline asm so that hipcc does not account for them: its own bookkeeping puts an
        // s_waitcnt lgkmcnt(0) in front of the first MFMA after ANY ds_read.  We count instead: LDS returns in order,
        // every k-half issues exactly MS+NS reads, and a k-half starts with lgkmcnt(MS+NS) (the set read during the
        // previous k-half may still be in flight, the one before has landed).
        // Fragment addresses live in registers per A ring stage and per tap row, so a K-step issues no address
        // arithmetic at all: the tap column and the k-half are instruction immediates; once per channel chunk the A
        // table is rotated (9 K-steps = ring phase + 3) and the B table moves to the other buffer.
        unsigned aaddr[R][MS], baddr[3][NS];
#pragma unroll
        for (int st = 0; st < R; ++st)
#pragma unroll
            for (int ms = 0; ms < MS; ++ms) aaddr[st][ms] = lds0 + st * (BM * 64) + atab[ms];
#pragma unroll
        for (int ty = 0; ty < 3; ++ty)
#pragma unroll
            for (int ns = 0; ns < NS; ++ns) baddr[ty][ns] = lds0 + R * BM * 64 + bbase[ns] + ty * rowb;
#define LDS_READ128(dst, addr, imm) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(imm))
        // One k-half: the MFMAs of the set that has landed, with the ds_reads of a later set (and this step's DMA
        // issue, `hook`) threaded between them.  A wave issues in order, so a burst of reads in front of the MFMAs
        // leaves the matrix pipe idle while the LDS queue drains (all 8 waves leave the barrier together and the two
        // waves of a SIMD stay in phase); interleaved, every read issues under an MFMA of the same wave.
        // Every sub-tile is computed, also the ones past a narrow tile's last pixel (their results are never stored):
        // skipping them cost two branches per k-half in every wave and only ever relieved waves off the critical path.
        auto half = [&](const v4i(&ca)[MS], const v4i(&cb)[NS], v4i(&na)[MS], v4i(&nb)[NS], const unsigned(&aad)[MS],
                        const unsigned(&bad)[NS], auto tap_c, auto h_c, auto rd_c, auto prev_rd_c, auto &&hook) {
            constexpr int TAPOFF = decltype(tap_c)::value, H = decltype(h_c)::value;
            constexpr bool RD = decltype(rd_c)::value, PREV_RD = decltype(prev_rd_c)::value;
            const bool rd = RD && !DBG(8);
            if (PREV_RD) asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(MS + NS) : "memory");
            else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            if (!DBG(4)) acc[0][0] = __builtin_amdgcn_mfma_i32_32x32x32_i8(ca[0], cb[0], acc[0][0], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            hook();
            if (rd) {
#pragma unroll
                for (int ms = 0; ms < MS; ++ms) LDS_READ128(na[ms], aad[ms], H * 512);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (MS > 1 && !DBG(4)) acc[MS - 1][0] = __builtin_amdgcn_mfma_i32_32x32x32_i8(ca[MS - 1], cb[0], acc[MS - 1][0], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (rd) {
                LDS_READ128(nb[0], bad[0], TAPOFF + H * 2 * PIECEB);
                if (NS > 1) LDS_READ128(nb[NS > 1 ? 1 : 0], bad[NS > 1 ? 1 : 0], TAPOFF + H * 2 * PIECEB);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int ns = 1; ns < NS; ++ns) {
                if (!DBG(4)) {
#pragma unroll
                    for (int ms = 0; ms < MS; ++ms)
                        acc[ms][ns] = __builtin_amdgcn_mfma_i32_32x32x32_i8(ca[ms], cb[ns], acc[ms][ns], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
                if (ns + 1 < NS && rd) LDS_READ128(nb[ns + 1 < NS ? ns + 1 : 0], bad[ns + 1 < NS ? ns + 1 : 0], TAPOFF + H * 2 * PIECEB);
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        auto nohook = [] {};
        using std::integral_constant;
        using std::true_type;
        using std::false_type;
        // prologue DMA (issued above): B(0), A(0..4) (ksteps >= 9); here A(0), A(1) (and B(0), older) have to have landed;
        // then both k-halves of step 0 are put on their way
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((R - 3) * APT) : "memory");
        __builtin_amdgcn_s_barrier();
        TS(2);
#pragma unroll
        for (int ms = 0; ms < MS; ++ms) LDS_READ128(fa[0][ms], aaddr[0][ms], 0);
#pragma unroll
        for (int ns = 0; ns < NS; ++ns) LDS_READ128(fb[0][ns], baddr[0][ns], 0);
#pragma unroll
        for (int ms = 0; ms < MS; ++ms) LDS_READ128(fa[1][ms], aaddr[0][ms], 512);
#pragma unroll
        for (int ns = 0; ns < NS; ++ns) LDS_READ128(fb[1][ns], baddr[0][ns], 2 * PIECEB);
        WP_DECL;
        WP_START();
        // One channel chunk = 9 K-steps, fully unrolled.  LAST selects the variant for the final chunk; in both variants
        // every "does that slab / chunk still exist" question is answered at compile time, so the steady state carries
        // no bookkeeping branches (the previous single-variant loop spent ~50 scalar instructions and 13 branches per step).
        auto chunk_body = [&](auto last_c, int chunk) {
            constexpr bool LAST = decltype(last_c)::value;
            const char *Bt = ldsB + (chunk & 1) * bbytes;
            const bool odd = chunk & 1;  // ring stage of tap 0 is (9 * chunk) % 6 = 3 * odd
            auto step = [&](auto t_c) {
                constexpr int t = decltype(t_c)::value;
                // fragments read in this step belong to step t+1: tap row table entry TYN + the immediate TAPN, A table
                // entry SN (aaddr[s] is ring stage (3 * odd + s) % 6); on the chunk's last step they belong to tap 0 of
                // the next chunk, read through the tables rotated for it
                constexpr int TYN = ((t + 1) % 9) / 3;
                constexpr int TAPN = ((t + 1) % 3) * 16;
                constexpr int SN = (t == 8) ? 0 : (t + 1) % R;
                constexpr int C0 = (2 * t) % 3, C1 = (2 * t + 1) % 3, N0 = (2 * t + 2) % 3, N1 = (2 * t + 3) % 3;
                constexpr bool RD = (t < 8) || !LAST;            // there is a step t+1 to read fragments for
                constexpr bool ISSUE_A = !LAST || (t + R - 1 < 9);  // slab g+5 exists
                // DMA queue when this step waits for A(g+1), issued four steps ago as the last DMA of its step: younger
                // than it are the DMAs of the three steps in between -- APT A slabs per step while slab s+5 exists, and the
                // B slots of the next chunk on a chunk's first four steps (steps < 0 are the previous chunk's steps 6..8;
                // for chunk 0 the prologue issued the same DMAs in the same order)
                constexpr auto n_b = [](int st) { return (LAST || st < 0 || st >= 4) ? 0 : ((st + 1) * SPS < NBS ? (st + 1) * SPS : NBS) - (st * SPS < NBS ? st * SPS : NBS); };
                constexpr auto n_a = [](int st) { return (st < 0 || !LAST || st + R - 1 < 9) ? APT : 0; };
                constexpr int YOUNG = n_a(t - 1) + n_b(t - 1) + (R > 4 ? n_a(t - 2) + n_b(t - 2) + n_a(t - 3) + n_b(t - 3) : 0);
                constexpr int BLO = t * SPS < NBS ? t * SPS : NBS, BHI = (t + 1) * SPS < NBS ? (t + 1) * SPS : NBS;
                if (t == 8 && !LAST) {
                    const int bdelta = odd ? -bbytes : bbytes;
                    unsigned rot[R][MS];
#pragma unroll
                    for (int st = 0; st < R; ++st)
#pragma unroll
                        for (int ms = 0; ms < MS; ++ms) rot[st][ms] = aaddr[(st + ROT) % R][ms];
#pragma unroll
                    for (int st = 0; st < R; ++st)
#pragma unroll
                        for (int ms = 0; ms < MS; ++ms) aaddr[st][ms] = rot[st][ms];
#pragma unroll
                    for (int yy = 0; yy < 3; ++yy)
#pragma unroll
                        for (int ns = 0; ns < NS; ++ns) baddr[yy][ns] += bdelta;
                }
                WP_MARK(0);  // LDS reads of the previous step drained
                if (t > 0 || chunk > 0) {
                    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(YOUNG) : "memory");
                    WP_MARK(1);  // DMA wait
                    if (!DBG(2)) __builtin_amdgcn_s_barrier();
                    WP_MARK(2);  // barrier
                }
                half(fa[C0], fb[C0], fa[N0], fb[N0], aaddr[SN], baddr[TYN], integral_constant<int, TAPN>{},
                     integral_constant<int, 0>{}, integral_constant<bool, RD>{}, true_type{}, [&] {
                         if (!LAST && t < 4 && !DBG(1))
                             issueB_slots(chunk + 1, integral_constant<int, BLO>{}, integral_constant<int, BHI>{});
                         if (ISSUE_A && !DBG(1)) {  // slab g+R-1 -> ring stage (9 * chunk + t + R - 1) % R
                             if constexpr (R == 6) {  // (9 * chunk) % 6 = 3 * odd
                                 constexpr unsigned E = ((t + R - 1) % R) * (BM * 64), O = ((t + R - 1 + 3) % R) * (BM * 64);
                                 issueA_next(odd ? O : E);
                             } else {                 // (9 * chunk) % 4 = chunk % 4
                                 issueA_next((unsigned)((chunk + t + R - 1) & 3) * (BM * 64));
                             }
                         }
                     });
                WP_MARK(3);  // first k-half
                half(fa[C1], fb[C1], fa[N1], fb[N1], aaddr[SN], baddr[TYN], integral_constant<int, TAPN>{},
                     integral_constant<int, 1>{}, integral_constant<bool, RD>{}, integral_constant<bool, RD>{}, nohook);
                WP_MARK(4);  // second k-half
                if (t == 1 && !DBG(16)) {
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // our reads are invisible to hipcc's counters
                    cell_sums(Bt);
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_sched_barrier(0);
                }
            };
            step(integral_constant<int, 0>{}); step(integral_constant<int, 1>{}); step(integral_constant<int, 2>{});
            step(integral_constant<int, 3>{}); step(integral_constant<int, 4>{}); step(integral_constant<int, 5>{});
            step(integral_constant<int, 6>{}); step(integral_constant<int, 7>{}); step(integral_constant<int, 8>{});
        };
        for (int chunk = 0; chunk + 1 < a.nchunks; ++chunk) chunk_body(false_type{}, chunk);
        chunk_body(true_type{}, a.nchunks - 1);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        WP_FLUSH();
#undef LDS_READ128
    } else {
        TS(2);
        // 1x1: every K-step is one 64-channel chunk; stage g % R1 of both rings holds step g.  A step's DMA group is its
        // B slots followed by its A slabs (G instructions per wave), issued R1 - 1 steps ahead.
        constexpr int R1 = RA_STAGES;
        constexpr int G = NBS + APT;
        static_assert(RB_STAGES == R1 && (R1 == 3 || R1 == 4), "1x1 loop: common ring depth");
        for (int g0 = 0; g0 < a.ksteps; g0 += R1) {
#pragma unroll
            for (int u = 0; u < R1; ++u) {
                const int g = g0 + u;
                if (g < a.ksteps) {
                    // group g has to have landed; the groups issued after it (at most R1 - 2) may stay in flight
                    const int young = min(R1 - 2, a.ksteps - 1 - g);
                    if (young == R1 - 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((R1 - 2) * G) : "memory");
                    else if (young == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(G) : "memory");
                    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    __builtin_amdgcn_s_barrier();  // ... for every wave, and every wave is past step g - 1: its stage is free
                    if (g + R1 - 1 < a.ksteps) {
                        issueB(g + R1 - 1);
                        issueA_next(((u + R1 - 1) % R1) * (BM * 64));
                    }
                    const char *Bt = ldsB + u * bbytes;
                    compute(ldsA + u * (BM * 64), Bt, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
    }

    TS(3);
    if (DBG(32)) {  // timing ablation: no epilogue (keep the accumulators alive)
        if (acc[0][0][0] == 0x7fffffff && a.y) a.y[tid] = 1;
        return;
    }
    // ---- epilogue
    __syncthreads();  // S complete, all fragment reads done
    int sx[NS];
#pragma unroll
    for (int ns = 0; ns < NS; ++ns) {
        int t = 0;
        if (KS == 1) {
            t = sxr[ns] + __shfl_xor(sxr[ns], 32);  // the two 16-byte k-halves of every K-step
        } else {
            const int c0 = prow[ns] * RS + pcol[ns];
#pragma unroll
            for (int dy = 0; dy < KS; ++dy)
#pragma unroll
                for (int dx = 0; dx < KS; ++dx) t += ldsS[c0 + dy * RS + dx];
        }
        sx[ns] = t;
    }
    __syncthreads();  // before the LDS is reused as the [BN][BM+4] uint8 output tile
    char *otile = smem;
    int *celltab = reinterpret_cast<int *>(smem + BN * OSTR);
    const int m0 = mtile * BM;

    int pb_[NS], rem[NS], nl_[NS];
#pragma unroll
    for (int ns = 0; ns < NS; ++ns) {
        const int nl = (ns * WNW + wn) * 32 + lj;
        nl_[ns] = nl;
        const int nn = nvalid[ns] ? n0 + nl : 0;
        const int b = fd_div(nn, a.fd_hw);
        const int rem0 = nn - b * hw;
        const int y = fd_div(rem0, a.fd_w), xx = rem0 - y * a.W;
        pb_[ns] = b;
        rem[ns] = rem0;
        if (wm == 0 && kh == 0) {
            // output cell; with a fused nearest-neighbour upsample (ref: src/upsample_layer.c:96-113) the top-left cell of the
            // pixel's up x up block in the (up*H) x (up*W) tensor
            const int up = a.up;
            celltab[nl] = !nvalid[ns] ? -1
                          : up == 1   ? a.out_lead + (b * (a.H + 1) + (y + 1)) * W1 + xx
                                      : a.out_lead + (b * (up * a.H + 1) + (up * y + 1)) * (up * a.W + 1) + up * xx;
        }
    }
    // Fast path: no int32 parity dump, power-of-two shifts (always true for the reference's prep): activation and store
    // mode become compile-time constants and the per-output code is branch free.  Channels past n (the parameter planes
    // are zero-padded to mpad) produce bytes nobody reads; the float tail of a quant_stop head skips them.
    const bool fast = !a.acc_out && a.hdr->pow2 == 1;
    auto epi_fast = [&](auto act_c, auto sat_c) {
        constexpr int ACT = decltype(act_c)::value;
        constexpr bool SAT = decltype(sat_c)::value != 0;
#pragma unroll
        for (int ms = 0; ms < MS; ++ms) {
#pragma unroll
            for (int grp = 0; grp < 4; ++grp) {
                const int ocl = wm * TM + ms * 32 + 8 * grp + 4 * kh;
                const int4 dz4 = *reinterpret_cast<const int4 *>(ldsPI + ocl);
                const int dzv[4] = {dz4.x, dz4.y, dz4.z, dz4.w};
                int32_t accb[4][NS];
                double mp[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    mp[r] = ldsPM[2 * BM + ocl + r];
#pragma unroll
                    for (int ns = 0; ns < NS; ++ns)  // |dz| <= 128, |sx| < 2^23 (K <= 64K): 24-bit multiply, full rate
                        accb[r][ns] = acc[ms][ns][grp * 4 + r] + __mul24(dzv[r], sx[ns]);
                }
                int32_t v[4][NS];
                requant_values<ACT, SAT, NS>(accb, mp, a.zp_act, v);
#pragma unroll
                for (int ns = 0; ns < NS; ++ns)
                    *reinterpret_cast<uint32_t *>(otile + nl_[ns] * OSTR + ocl) = pack4_biased(v[0][ns], v[1][ns], v[2][ns], v[3][ns]);
                if (a.y_f32) {  // quant_stop tail (ref :752-760) and, fused, the yolo layer's activations
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int oc = m0 + ocl + r;
                        if (oc < a.n) {
                            const int e = a.yolo_out ? oc % a.yolo_per : 0;
#pragma unroll
                            for (int ns = 0; ns < NS; ++ns)
                                if (nvalid[ns]) {
                                    const int u8 = v[r][ns] & 0xFF;
                                    const float f = (float)(u8 - a.zp_act) * a.s_act;
                                    const size_t ridx = ((size_t)pb_[ns] * a.n + oc) * hw + rem[ns];
                                    a.y_f32[ridx] = f;
                                    if (a.yolo_out) a.yolo_out[ridx] = (e == 2 || e == 3) ? f : ldsYL[u8];
                                }
                        }
                    }
                }
            }
        }
    };
    using std::integral_constant;
    if (DBG(64)) {
    } else if (fast) {
#ifdef MI355_EPI_ONE  // A/B build: only the epilogue the benchmarked launch needs is instantiated
        epi_fast(integral_constant<int, MI355_ACT_LEAKY>{}, integral_constant<int, 0>{});
    }
    if (false) {
#endif
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
    }
#ifndef MI355_EPI_ONE
    else {
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
                    const int dzv = ldsPI[ocl + r], biv = ldsPI[BM + ocl + r];
                    const double mv = ldsPM[ocl + r], sv = ldsPM[BM + ocl + r];
    #pragma unroll
                    for (int ns = 0; ns < NS; ++ns) {
                        const int32_t accv = acc[ms][ns][grp * 4 + r] - biv + dzv * sx[ns];  // acc started at cw + bias
                        uint32_t u8 = 0;
                        if (oc < a.n) {
                            u8 = requant_u8(accv, biv, mv, sv, a.zp_act, a.act, a.store_mode);
                            if (nvalid[ns] && (a.acc_out || a.y_f32)) {
                                const size_t ridx = ((size_t)pb_[ns] * a.n + oc) * hw + rem[ns];
                                if (a.acc_out) a.acc_out[ridx] = accv;
                                if (a.y_f32) {
                                    const float f = (float)((int)u8 - a.zp_act) * a.s_act;  // ref :757
                                    a.y_f32[ridx] = f;
                                    if (a.yolo_out) {
                                        const int e = oc % a.yolo_per;
                                        a.yolo_out[ridx] = (e == 2 || e == 3) ? f : ldsYL[u8];
                                    }
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
#endif
    __syncthreads();
    TS(4);
    if (a.y && !DBG(128)) {
        const int dwords = min(BM, a.out_w - m0) >> 2;
        if (a.up > 1) {  // every pixel is stored up x up times
            const int total = BN * dwords, rowc = a.up * a.W + 1;
            for (int p = tid; p < total; p += NT) {
                const int pix = p / dwords, d = p - pix * dwords;
                const int cell = celltab[pix];
                if (cell >= 0) {
                    const uint32_t v = *reinterpret_cast<const uint32_t *>(otile + pix * OSTR + d * 4);
                    for (int uy = 0; uy < a.up; ++uy)
                        for (int ux = 0; ux < a.up; ++ux)
                            *reinterpret_cast<uint32_t *>(a.y + (size_t)(cell + uy * rowc + ux) * a.out_cs + m0 + d * 4) = v;
                }
            }
        } else if (dwords == BM / 4 && !(a.out_cs & 15) && !((size_t)a.y & 15)) {  // common case: 16-byte stores (a quarter of the store instructions)
#pragma unroll 2
            for (int p = tid; p < BN * (BM / 16); p += NT) {
                const int pix = p / (BM / 16), q = p % (BM / 16);
                const int cell = celltab[pix];
                if (cell >= 0) {
                    const uint32_t *sp = reinterpret_cast<const uint32_t *>(otile + pix * OSTR + q * 16);  // rows are 4-byte aligned only
                    const uint4 v = make_uint4(sp[0], sp[1], sp[2], sp[3]);
                    *reinterpret_cast<uint4 *>(a.y + (size_t)cell * a.out_cs + m0 + q * 16) = v;
                }
            }
        } else if (dwords == BM / 4) {  // constant divisor
#pragma unroll 4
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
    TS(5);
}

// ---------------------------------------------------------------------------------------------------------------
template <int BM, int BN, int WMW, int WNW, int RS, int KS>
static int rows_launch_cfg(ConvArgs &a, hipStream_t st)
{
    constexpr int NW = WMW * WNW, NT = 64 * NW;
    constexpr int DW = (NW == 8 && KS == 3) ? 4 : NW;  // DMA waves, as in the kernel
    constexpr int HALO = (KS == 3) ? 1 : 0;
    if (a.mpad % BM) return MI355_EINVAL;
    a.mtiles = a.mpad / BM;
    if (a.ntiles_n < (a.total_n + BN - 1) / BN) a.ntiles_n = (a.total_n + BN - 1) / BN;  // tiles must fit BN
    // rows spanned by BN consecutive pixels: pixel rows + one pad row per image boundary crossed, + halo rows
    a.rows_cap = (BN - 2 + a.W) / a.W + 1 + (BN - 2 + a.H * a.W) / (a.H * a.W) + 2 * HALO;
    const int ndma = a.rows_cap * (RS / 16);
    if ((ndma + DW - 1) / DW > rows_nb_slots(BN, RS, DW, KS)) return MI355_EINVAL;  // map too narrow for this tile
    if ((size_t)a.in_cells * (size_t)a.in_cs >= ((size_t)1 << 32)) return MI355_EINVAL;  // 32-bit DMA lane offsets
    a.rowb = RS * 64 + 16 * (a.W & 15);
    a.fd_hw = fastdiv_make((uint32_t)(a.H * a.W)); a.fd_w = fastdiv_make((uint32_t)a.W);
    a.fd_ntn = fastdiv_make((uint32_t)a.ntiles_n); a.fd_nch = fastdiv_make((uint32_t)a.nchunks);
    a.tile_q = a.total_n / a.ntiles_n; a.tile_r = a.total_n % a.ntiles_n;
    size_t lds = (size_t)ra_stages<KS, BN>() * BM * 64 + (size_t)rb_stages<KS, BN>() * a.rows_cap * a.rowb + (size_t)a.rows_cap * RS * 4;
    const size_t lds_epi = (size_t)BN * (BM + 4) + (size_t)BN * 4;
    if (lds_epi > lds) lds = lds_epi;
    lds = (lds + 15) & ~(size_t)15;
    a.lds_param_off = (int)lds;  // beyond both the K-loop buffers and the epilogue tile
    lds += (size_t)BM * 32 + (a.yolo_out ? 1024 : 0);  // the logistic table only when a yolo head is fused (three 64 x 128 workgroups then fit a CU on 13-wide maps)
    if (lds > 160 * 1024) return MI355_EINVAL;
    auto kern = conv_rows_i8_kernel<BM, BN, WMW, WNW, RS, KS>;
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
    // XCD tile order: minimise the bytes one XCD's L2 has to pull in = its M tiles' weight slabs + its N tiles' input rows
    a.xcd_mb = 1;
    if (!(a.debug & (1 << 24))) {
        const long nb = (long)a.ntiles_n * a.mtiles, per = (nb + 7) / 8;
        const double wslab = (double)BM * a.ksteps * 64, itile = (double)a.total_n / a.ntiles_n * a.cb * a.nchunks;
        double best = 0;
        for (int mb = 1; mb <= a.mtiles; ++mb) {
            if (a.mtiles % mb) continue;
            const long blk = (long)mb * a.ntiles_n;
            const long mt = std::min<long>(a.mtiles, mb * ((per + blk - 1) / blk)), nt = std::min<long>(a.ntiles_n, (per + mb - 1) / mb + (per % mb || blk % per || nb % 8 ? 1 : 0));  // ranges that start inside an N tile's column touch one more
            const double bytes = mt * wslab + nt * itile;
            if (mb == 1 || bytes < best) { best = bytes; a.xcd_mb = mb; }
        }
    }
    a.fd_ntper = fastdiv_make((uint32_t)(a.xcd_mb * a.ntiles_n));
    a.fd_mb = fastdiv_make((uint32_t)a.xcd_mb);
    dim3 grid(a.ntiles_n * a.mtiles), block(NT);
    hipLaunchKernelGGL(kern, grid, block, lds, st, a);
    return hipGetLastError() == hipSuccess ? MI355_OK : MI355_EHIP;
}

// Round 4, measured and not kept (profiles/r04_rows_12wave_ab.log): the 128 x 384 tile on TWELVE waves (2 x 6 of 64 x 64 wave tiles: 163
// registers, three waves per SIMD instead of two; rows_launch_cfg<128, 384, 2, 6, RS, 3> with launch bounds (768, 3) and four DMA
// waves) -- rocprofv3 averages over 53 launches of the L12 shape: 55.00 us against 55.06 us.  Together with conv_rows16.hip's
// variants (other MFMA shape, A operand from global memory, four waves of 128 x 96) that is four structurally different K loops
// within 3 % of each other on the same operand bytes, and all of them 20-25 % faster on constant operands: on these layers the clock
// the power management grants under matrix load sets the time, not the loop.
template <int RS, int KS>
static int rows_launch_tile(ConvArgs &a, hipStream_t st, int bm, int bn)
{
    if (bm == 128 && bn == 384) return rows_launch_cfg<128, 384, 2, 4, RS, KS>(a, st);
    if (bm == 128 && bn == 256) return rows_launch_cfg<128, 256, 2, 4, RS, KS>(a, st);
    if constexpr (KS == 3) {
        if (bm == 128 && bn == 192) return rows_launch_cfg<128, 192, 2, 2, RS, KS>(a, st);  // four waves of 64 x 96 (round 5)
    }
    if (bm == 128 && bn == 128) return rows_launch_cfg<128, 128, 2, 2, RS, KS>(a, st);
    if (bm == 64 && bn == 256) return rows_launch_cfg<64, 256, 1, 4, RS, KS>(a, st);
    if (bm == 64 && bn == 128) return rows_launch_cfg<64, 128, 1, 4, RS, KS>(a, st);
    if (bm == 32 && bn == 256) return rows_launch_cfg<32, 256, 1, 4, RS, KS>(a, st);
    if (bm == 32 && bn == 128) return rows_launch_cfg<32, 128, 1, 4, RS, KS>(a, st);
    return MI355_EINVAL;
}

// The 3x3 and the 1x1 instantiations live in two translation units (conv_rows.hip, conv_rows_k1.hip = this file
// compiled with ROWS_TU_KS1) so that they compile in parallel: 21 kernel instantiations each.
#ifdef ROWS_TU_KS1
int conv_rows_launch_k1(ConvArgs &a, hipStream_t st, int bm, int bn)
{
    const int need = a.W + 2;
    if (need <= 16) return rows_launch_tile<16, 1>(a, st, bm, bn);
    if (need <= 32) return rows_launch_tile<32, 1>(a, st, bm, bn);
    if (need <= 64) return rows_launch_tile<64, 1>(a, st, bm, bn);
    return MI355_EINVAL;
}
#else
int conv_rows_launch_k1(ConvArgs &a, hipStream_t st, int bm, int bn);

// returns MI355_EINVAL when the shape is outside this kernel's domain (caller falls back to conv_igemm.hip)
int conv_rows16_launch(ConvArgs &a, hipStream_t st, int bm, int bn);  // conv_rows16.hip: the 3x3 kernel on 16x16x64 MFMAs

int conv_rows_launch(ConvArgs &a, hipStream_t st, int bm, int bn)
{
    if (a.cb != 64) return MI355_EINVAL;
    if (a.ksize != 3) return conv_rows_launch_k1(a, st, bm, bn);
    if (!(a.debug & (1 << 20))) {  // (bit 20: the 32x32x32 kernel below, for A/B runs)
        const int rc = conv_rows16_launch(a, st, bm, bn);
        if (rc != MI355_EINVAL) return rc;
    }
    const int need = a.W + 2;
    if (need <= 16) return rows_launch_tile<16, 3>(a, st, bm, bn);
    if (need <= 32) return rows_launch_tile<32, 3>(a, st, bm, bn);
    if (need <= 64) return rows_launch_tile<64, 3>(a, st, bm, bn);
    return MI355_EINVAL;
}
#endif
