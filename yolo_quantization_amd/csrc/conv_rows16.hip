// conv_rows16.hip -- conv_rows.hip's 3x3 kernel on V_MFMA_I32_16X16X64_I8 instead of V_MFMA_I32_32X32X32_I8.
//
// Why: the 3x3 layers on this kernel are bound by the shader clock, and the clock by the power the matrix pipe draws, which
// depends on the operand bytes AND on the instruction (tools/ubench/mfma_data_power.hip): a loop of
// nothing but MFMAs on uniform random bytes sustains 3 450-3 510 TOP/s with the 32x32x32 instruction and 3 860 TOP/s with
// the 16x16x64 one (a quarter of the accumulator registers written per operation).
//
// Everything outside the K-step body is conv_rows.hip's design, unchanged: row-image LDS layout of the B operand
// ([16-byte piece][RS cells][16 B] per row: piece = the lane's k quarter), A slabs of [16-row chunk][4 pieces][16 rows][16 B]
// (a 16x16x64 A fragment is 1 KiB of consecutive bytes: lane l reads bytes [16 l, +16)), 6-stage A ring five K-steps ahead,
// double-buffered B row images, counted vmcnt waits, one s_barrier per K-step, DMA issued by the older four waves, tile plan,
// XCD grid, epilogue (folded FP64 requantise, LDS-transposed 16-byte stores, fused upsample / yolo head).
//
// What differs: a wave tile of 64 x 96 is 4 x 6 tiles of 16 x 16 (the same 96 accumulator registers); a K-step is
// 24 MFMAs on 4 A + 6 B fragments of the whole 64-channel chunk (the same 10 ds_read_b128 per K-step), and the fragments
// are prefetched IN PLACE: a B fragment is dead after its four MFMAs and is reloaded for the next K-step at once, the A
// fragments are reloaded during the last two rounds -- 40 fragment registers instead of 60, no register-set rotation.
//   step:  lgkmcnt(2) | for ni in 0..3: 4 MFMAs (A0..A3, B[ni]); read B[ni]' | lgkmcnt(4) |
//          for mi in 0..3: MFMA(A[mi], B4), MFMA(A[mi], B5); read A[mi]' | read B4', B5'
// D layout of the instruction: lane l holds rows (= filters) 4 (l / 16) .. + 3 of column (= pixel) l % 16.
#include "kargs.h"
#include <type_traits>

#define DBG(bit) (false)
#if defined(MI355_ABLATE)
// phase timestamps (100 MHz wall clock) of wave 0 of every workgroup: tools/conv_microbench.py --timeline16 (-DMI355_ABLATE builds only)
__device__ long long g_rows16_ts[6][4096];
#define TS(k) do { if (threadIdx.x == 0 && blockIdx.x < 4096) g_rows16_ts[k][blockIdx.x] = wall_clock64(); } while (0)
extern "C" int mi355_debug_read_ts16(long long *host)
{
    return hipMemcpyFromSymbol(host, HIP_SYMBOL(g_rows16_ts), sizeof(long long) * 6 * 4096) == hipSuccess ? 0 : -5;
}
#else
#define TS(k) do { } while (0)
#endif
#define WP_DECL do { } while (0)
#define WP_START() do { } while (0)
#define WP_MARK(k) do { } while (0)
#define WP_FLUSH() do { } while (0)

#define DMA16(gsrc, ldst)                                                                               \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(gsrc),           \
                                     (__attribute__((address_space(3))) void *)(ldst), 16, 0, 0)

// A ring depth: 6 for the software-pipelined 3x3 loop (fragments of step g+1 are read while the MFMAs of step g run, so
// A(g), A(g+1) are being read while A(g+2) .. A(g+5) are in flight) -- 4 in the 128-column configurations, whose LDS
// then lets two workgroups share a CU (one's epilogue under the other's K loop); 4 (3 for the widest tiles: LDS) for the plain 1x1
// loop, which then has its DMA two or three K-steps ahead instead of one -- its K-steps took 0.7 us each, the DMA latency
// (round 3: the K loop also takes FIVE stages -- a slab issued three K-steps ahead instead of two; measured on the 128-column tiles,
// 80 448 bytes of LDS on 13-wide maps, two workgroups per CU still: L12 44.6 vs 44.2-45.3 us per launch with three batches in flight,
// 360 parity tests green -- the 128 x 128 K loop at 58 % of the matrix pipe is not waiting for its DMA.  Back to four.)
template <int KS, int BN> constexpr int ra16_stages() { return KS == 3 ? (BN <= 192 ? 4 : 6) : (BN <= 128 ? 4 : 3); }
// B buffers: two per-chunk row images for 3x3 (a chunk lasts nine K-steps); for 1x1 every K-step is a new chunk and the
// row image rides the same ring as the weights
template <int KS, int BN> constexpr int rb16_stages() { return KS == 3 ? 2 : ra16_stages<KS, BN>(); }
// B DMA slots per wave per channel-chunk load: a compile-time constant per configuration (one VGPR of source offset
// each, issued unconditionally -- slots past a tile's last LDS row repeat its last one), so that every vmcnt wait of
// the K loop is an immediate.  Sized for the rows a BN-pixel tile spans when the map is at least 3/4 as wide as the
// row image (W >= 12 / 24 / 48 for RS = 16 / 32 / 64); narrower maps are refused by the launcher (other tile or kernel).
constexpr int rows16_nb_slots(int BN, int RS, int NW, int KS)
{
    const int wmin = RS * 3 / 4, halo = KS == 3 ? 1 : 0;
    const int rows = (BN - 2 + wmin) / wmin + 1 + (BN - 2 + wmin * wmin) / (wmin * wmin) + 2 * halo;
    return (rows * (RS / 16) + NW - 1) / NW;
}

// global row index (over all image blocks, pad rows included) and column of valid pixel n
__device__ __forceinline__ void row16_of_pixel(int n, int H, int W, FastDiv fd_hw, FastDiv fd_w, int &grow, int &x)
{
    const int hw = H * W;
    const int b = fd_div(n, fd_hw);
    const int r = n - b * hw;
    const int y = fd_div(r, fd_w);
    x = r - y * W;
    grow = b * (H + 1) + y + 1;
}

// NBX: extra B DMA slots per DMA wave, for maps that fill their LDS row image badly (19 + 2 cells in a 32-slot row: a 384-pixel tile
// then spans more rows than the slot count sized for W >= 3/4 RS covers).  Its own instantiation: the common one keeps its registers.
template <int BM, int BN, int WMW, int WNW, int RS, int KS, int NBX = 0>
__global__ __launch_bounds__(64 * WMW * WNW, (BM / WMW > 64 ? 1 : 2)) void conv_rows16_i8_kernel(const ConvArgs a)
{
    constexpr int NW = WMW * WNW, NT = 64 * NW;
    constexpr int TM = BM / WMW, TN = BN / WNW;
    constexpr int MI = TM / 16, NI = TN / 16;  // 16 x 16 accumulator tiles of a wave: MI x NI
    constexpr int ACH = BM / 16;
    static_assert(KS == 3 && TN % 32 == 0, "3x3 only; pixels are dealt in 32-column sub-tiles (two 16-column tiles each)");
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
    constexpr int RA_STAGES = ra16_stages<KS, BN>();
    constexpr int RB_STAGES = rb16_stages<KS, BN>();
    constexpr int NBS = rows16_nb_slots(BN, RS, DW, KS) + NBX;  // B DMA slots per (DMA) wave per chunk
    constexpr int SPS = (NBS + 3) / 4;                  // ... issued per K-step over a chunk's first four steps (3x3 loop)
    static_assert(TM % 16 == 0 && TN % 32 == 0, "wave tile: 16-row tiles, 32-column sub-tiles");

    extern __shared__ __attribute__((aligned(16))) char smem[];
    // smem: [RA_STAGES][BM*64] A ring first
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
    const int kq = lane >> 4, lj = lane & 15;  // the lane's k quarter (16 of the chunk's 64 channels) and its row / column in a 16 x 16 tile
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
    row16_of_pixel(n0, a.H, a.W, a.fd_hw, a.fd_w, gr0, x0);
    int gr1, x1;
    row16_of_pixel(n_end - 1, a.H, a.W, a.fd_hw, a.fd_w, gr1, x1);
    const int grow_first = gr0 - HALO;
    const int nrows = gr1 - gr0 + 1 + 2 * HALO;   // <= a.rows_cap (host guarantees)
    const int ndma = nrows * CPR;                 // B DMA instructions per chunk load for the whole workgroup

    // ---- per-lane B base: LDS byte offset of (its pixel's row - HALO, its column - HALO) for k quarter kq
    constexpr int NS = NI;  // (the per-pixel tables below are indexed by the 16-column tile)
    int bbase[NS];
    // (the pixel's LDS row / cell and validity are recomputed in the epilogue rather than kept in 18 registers across the K loop)
#pragma unroll
    for (int ns = 0; ns < NS; ++ns) {
        const int n = n0 + ((ns >> 1) * WNW + wn) * 32 + (ns & 1) * 16 + lj;  // 32-column sub-tiles are dealt round-robin to the N waves
        int gr, x;
        row16_of_pixel(n < n_end ? n : n_end - 1, a.H, a.W, a.fd_hw, a.fd_w, gr, x);
        const int prow = gr - grow_first - HALO;  // LDS row of tap dy = 0 (top tap)
        const int pcol = x + 1 - HALO;            // LDS cell of tap dx = 0 (left tap); cell 0 of a row is x = -1
        bbase[ns] = prow * rowb + pcol * 16 + kq * PIECEB;
    }
    // A fragment mi of the wave = 16-row chunk wm * MI + mi of the slab = 1 KiB of consecutive bytes: [piece kq][row lj][16 B]
    const int atab0 = (wm * MI) * 1024 + lane * 16;

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
    // register r of tile (mi, ni) holds filter row 16 mi + 4 kq + r (parameter planes are padded to mpad)
    v4i acc[MI][NI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
        const int4 c4 = *reinterpret_cast<const int4 *>(a.cwb + mtile * BM + wm * TM + mi * 16 + 4 * kq);
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = v4i{c4.x, c4.y, c4.z, c4.w};
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

    // per channel chunk: every thread reduces cells of the freshly landed B buffer into S
    auto cell_sums = [&](const char *Bt) {
        const int ncells = nrows * RS;
        for (int id = tid; id < ncells; id += NT) {
            const int r = id / RS, c = id - r * RS;
            const char *p0 = Bt + r * rowb + c * 16;
            int t = 0;
#pragma unroll 1  // one piece (4 registers) at a time: this runs inside the K loop, where ~215 registers hold its state
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

    {
        // ---- software-pipelined 3x3 loop.  R = 6 A ring stages; per K-step g:  wait A(g+1) landed; s_barrier; issue DMA A(g+5)
        //      [B(chunk+1) on a chunk's first steps]; the step's 24 MFMAs with the reads of step g+1's fragments threaded between
        //      them, IN PLACE: B fragment ni is reloaded right after its four MFMAs, the A fragments during the last two rounds.
        constexpr int R = RA_STAGES;
        static_assert(R >= 4 && R <= 6, "ring depths the stage bookkeeping below is written for");
        static_assert((MI == 4 || MI == 8) && NI >= 2, "written for 64-row wave tiles (128-row ones: one wave per SIMD, accumulators in the AGPRs)");
        constexpr int ROT = 9 % R;  // ring phase advance of one channel chunk (nine K-steps)
        v4i fa[MI], fb[NI];
        // Fragment reads are issued as inline asm so that hipcc does not account for them (its own bookkeeping puts an
        // s_waitcnt lgkmcnt(0) in front of the first MFMA after ANY ds_read).  We count instead: LDS returns in order and a
        // K-step issues its reads in the order B0' .. B(NI-3)', A0' .. A3', B(NI-2)', B(NI-1)'.
        // Addresses: one register per A ring stage (the fragment index is an immediate), one per B fragment for the tap ROW being
        // read (moved on by rowb every third step: NI additions, instead of 3 x NI registers).
        unsigned aaddr[R], baddr[NI];
        unsigned sst[R];  // (R == 5) byte offset of ring stage (9 * chunk + s) % R: wave-uniform, rotated with aaddr once per chunk
#pragma unroll
        for (int st = 0; st < R; ++st) sst[st] = st * (BM * 64);
#pragma unroll
        for (int st = 0; st < R; ++st) aaddr[st] = lds0 + st * (BM * 64) + atab0;
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) baddr[ni] = lds0 + R * BM * 64 + bbase[ni];
#define LDS_READ128(dst, addr, imm) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(imm))
        using std::integral_constant;
        using std::true_type;
        using std::false_type;
        // prologue DMA (issued above): B(0), A(0..4) (ksteps >= 9); here A(0), A(1) (and B(0), older) have to have landed;
        // then step 0's fragments are put on their way in the steady state's order
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((R - 3) * APT) : "memory");
        __builtin_amdgcn_s_barrier();
        TS(2);
#pragma unroll
        for (int ni = 0; ni < NI - 2; ++ni) LDS_READ128(fb[ni], baddr[ni], 0);
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) LDS_READ128(fa[mi], aaddr[0], mi * 1024);
        LDS_READ128(fb[NI - 2], baddr[NI - 2], 0);
        LDS_READ128(fb[NI - 1], baddr[NI - 1], 0);
        // One channel chunk = 9 K-steps, fully unrolled.  LAST selects the variant for the final chunk; in both variants
        // every "does that slab / chunk still exist" question is answered at compile time.
        auto chunk_body = [&](auto last_c, int chunk) {
            constexpr bool LAST = decltype(last_c)::value;
            const char *Bt = ldsB + (chunk & 1) * bbytes;
            const bool odd = chunk & 1;  // ring stage of tap 0 is (9 * chunk) % 6 = 3 * odd
            auto step = [&](auto t_c) {
                constexpr int t = decltype(t_c)::value;
                // fragments read in this step belong to step t+1: tap row table entry TYN + the immediate TAPN, A stage
                // register SN (aaddr[s] is ring stage (3 * odd + s) % 6); on the chunk's last step they belong to tap 0 of
                // the next chunk, read through the tables rotated for it
                constexpr int TAPN = ((t + 1) % 3) * 16;
                constexpr int SN = (t == 8) ? 0 : (t + 1) % R;
                constexpr bool RD = (t < 8) || !LAST;            // there is a step t+1 to read fragments for
                constexpr bool ISSUE_A = !LAST || (t + R - 1 < 9);  // slab g+5 exists
                constexpr auto n_b = [](int st) { return (LAST || st < 0 || st >= 4) ? 0 : ((st + 1) * SPS < NBS ? (st + 1) * SPS : NBS) - (st * SPS < NBS ? st * SPS : NBS); };
                constexpr auto n_a = [](int st) { return (st < 0 || !LAST || st + R - 1 < 9) ? APT : 0; };
                // A(g+1) was issued R - 2 steps ago as the last DMA of its step: younger than it are the DMAs of the R - 3 steps in between
                constexpr int YOUNG = n_a(t - 1) + n_b(t - 1) + (R > 4 ? n_a(t - 2) + n_b(t - 2) : 0) + (R > 5 ? n_a(t - 3) + n_b(t - 3) : 0);
                constexpr int BLO = t * SPS < NBS ? t * SPS : NBS, BHI = (t + 1) * SPS < NBS ? (t + 1) * SPS : NBS;
                if (t == 8 && !LAST) {
                    const int bdelta = (odd ? -bbytes : bbytes) - 2 * rowb;  // the other row image, back to its tap row 0
                    unsigned rot[R];
#pragma unroll
                    for (int st = 0; st < R; ++st) rot[st] = aaddr[(st + ROT) % R];
#pragma unroll
                    for (int st = 0; st < R; ++st) aaddr[st] = rot[st];
                    if constexpr (R == 5) {
                        unsigned srot[R];
#pragma unroll
                        for (int st = 0; st < R; ++st) srot[st] = sst[(st + ROT) % R];
#pragma unroll
                        for (int st = 0; st < R; ++st) sst[st] = srot[st];
                    }
#pragma unroll
                    for (int ni = 0; ni < NI; ++ni) baddr[ni] += bdelta;
                }
                if (t == 2 || t == 5) {  // the fragments read from this step on belong to the next tap row
#pragma unroll
                    for (int ni = 0; ni < NI; ++ni) baddr[ni] += rowb;
                }
                if (t > 0 || chunk > 0) {
                    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(YOUNG) : "memory");
                    __builtin_amdgcn_s_barrier();
                }
                auto dma = [&] {
                    __builtin_amdgcn_sched_barrier(0);
                    if (!LAST && t < 4) issueB_slots(chunk + 1, integral_constant<int, BLO>{}, integral_constant<int, BHI>{});
                    if (ISSUE_A) {  // slab g+R-1 -> ring stage (9 * chunk + t + R - 1) % R
                        if constexpr (R == 6) {  // (9 * chunk) % 6 = 3 * odd
                            constexpr unsigned E = ((t + R - 1) % R) * (BM * 64), O = ((t + R - 1 + 3) % R) * (BM * 64);
                            issueA_next(odd ? O : E);
                        } else if constexpr (R == 5) {
                            // (on a chunk's last step the table has already been rotated for the next chunk: 9 steps further on)
                            issueA_next(sst[t == 8 ? (t + R - 1 - 9) % R : (t + R - 1) % R]);
                        } else {                 // (9 * chunk) % 4 = chunk % 4
                            issueA_next((unsigned)((chunk + t + R - 1) & 3) * (BM * 64));
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                };
                // this step's A fragments and B fragments 0 .. NI-3 have landed (the two reads issued after them may be in flight)
                asm volatile("s_waitcnt lgkmcnt(2)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int ni = 0; ni < NI - 2; ++ni) {
#pragma unroll
                    for (int mi = 0; mi < MI; ++mi) {
                        acc[mi][ni] = __builtin_amdgcn_mfma_i32_16x16x64_i8(fa[mi], fb[ni], acc[mi][ni], 0, 0, 0);
                        if (ni == 0 && mi == 0) dma();  // this step's DMA issue, behind the first MFMA
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    if (RD) LDS_READ128(fb[ni], baddr[ni], TAPN);  // in place: its MFMAs have been issued
                    __builtin_amdgcn_sched_barrier(0);
                }
                // the last two B fragments of this step (read at the end of the previous one): the reads issued since may be in flight
                if (RD) asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(NI - 2) : "memory");
                else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int mi = 0; mi < MI; ++mi) {
                    acc[mi][NI - 2] = __builtin_amdgcn_mfma_i32_16x16x64_i8(fa[mi], fb[NI - 2], acc[mi][NI - 2], 0, 0, 0);
                    if (NI == 2 && mi == 0) dma();  // (32-column wave tiles: there is no round in front of these)
                    acc[mi][NI - 1] = __builtin_amdgcn_mfma_i32_16x16x64_i8(fa[mi], fb[NI - 1], acc[mi][NI - 1], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    if (RD) LDS_READ128(fa[mi], aaddr[SN], mi * 1024);
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (RD) {
                    LDS_READ128(fb[NI - 2], baddr[NI - 2], TAPN);
                    LDS_READ128(fb[NI - 1], baddr[NI - 1], TAPN);
                }
                __builtin_amdgcn_sched_barrier(0);
                if (t == 1) {
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
#undef LDS_READ128
    }

    TS(3);
    if (DBG(32)) {  // timing ablation: no epilogue (keep the accumulators alive)
        if (acc[0][0][0] == 0x7fffffff && a.y) a.y[tid] = 1;
        return;
    }
    // ---- epilogue
    __syncthreads();  // S complete, all fragment reads done
    int sx[NS];
    bool nvalid[NS];
#pragma unroll
    for (int ns = 0; ns < NS; ++ns) {
        int t = 0;
        const int n = n0 + ((ns >> 1) * WNW + wn) * 32 + (ns & 1) * 16 + lj;
        nvalid[ns] = n < n_end;
        int gr, x;
        row16_of_pixel(nvalid[ns] ? n : n_end - 1, a.H, a.W, a.fd_hw, a.fd_w, gr, x);  // idle lanes shadow the tile's last pixel, as in the K loop
        const int c0 = (gr - grow_first - HALO) * RS + x + 1 - HALO;
#pragma unroll
        for (int dy = 0; dy < KS; ++dy)
#pragma unroll
            for (int dx = 0; dx < KS; ++dx) t += ldsS[c0 + dy * RS + dx];
        sx[ns] = t;
    }
    __syncthreads();  // before the LDS is reused as the [BN][BM+4] uint8 output tile
    char *otile = smem;
    int *celltab = reinterpret_cast<int *>(smem + BN * OSTR);
    const int m0 = mtile * BM;

    int pb_[NS], rem[NS], nl_[NS];
#pragma unroll
    for (int ns = 0; ns < NS; ++ns) {
        const int nl = ((ns >> 1) * WNW + wn) * 32 + (ns & 1) * 16 + lj;
        nl_[ns] = nl;
        const int nn = nvalid[ns] ? n0 + nl : 0;
        const int b = fd_div(nn, a.fd_hw);
        const int rem0 = nn - b * hw;
        const int y = fd_div(rem0, a.fd_w), xx = rem0 - y * a.W;
        pb_[ns] = b;
        rem[ns] = rem0;
        if (wm == 0 && kq == 0) {
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
        constexpr int NH = (NS % 3 == 0) ? 3 : 2;  // pixels requantised per call: 12 or 8 values in flight (registers)
        static_assert(NS % NH == 0, "pixel tiles per lane");
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
            const int ocl = wm * TM + mi * 16 + 4 * kq;  // 4 consecutive filters held by this lane
            const int4 dz4 = *reinterpret_cast<const int4 *>(ldsPI + ocl);
            const int dzv[4] = {dz4.x, dz4.y, dz4.z, dz4.w};
            double mp[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) mp[r] = ldsPM[2 * BM + ocl + r];
#pragma unroll
            for (int n0h = 0; n0h < NS; n0h += NH) {
                int32_t accb[4][NH];
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int k = 0; k < NH; ++k)  // |dz| <= 128, |sx| < 2^23 (K <= 64K): 24-bit multiply, full rate
                        accb[r][k] = acc[mi][n0h + k][r] + __mul24(dzv[r], sx[n0h + k]);
                int32_t v[4][NH];
                requant_values<ACT, SAT, NH>(accb, mp, a.zp_act, v);
#pragma unroll
                for (int k = 0; k < NH; ++k)
                    *reinterpret_cast<uint32_t *>(otile + nl_[n0h + k] * OSTR + ocl) = pack4_biased(v[0][k], v[1][k], v[2][k], v[3][k]);
                if (a.y_f32) {  // quant_stop tail (ref :752-760) and, fused, the yolo layer's activations
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int oc = m0 + ocl + r;
                        if (oc < a.n) {
                            const int e = a.yolo_out ? oc % a.yolo_per : 0;
#pragma unroll
                            for (int k = 0; k < NH; ++k)
                                if (nvalid[n0h + k]) {
                                    const int u8 = v[r][k] & 0xFF;
                                    const float f = (float)(u8 - a.zp_act) * a.s_act;
                                    const size_t ridx = ((size_t)pb_[n0h + k] * a.n + oc) * hw + rem[n0h + k];
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
#ifdef MI355_EPI_ONE  // A/B build: only the epilogue the benchmarked launch needs is instantiated -- is the cold epilogue an instruction-fetch problem?
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
        for (int mi = 0; mi < MI; ++mi) {
            const int ocl = wm * TM + mi * 16 + 4 * kq;  // 4 consecutive filters held by this lane
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
                    const int32_t accv = acc[mi][ns][r] - biv + dzv * sx[ns];  // acc started at cw + bias
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
template <int BM, int BN, int WMW, int WNW, int RS, int KS, int NBX = 0>
static int rows16_launch_cfg(ConvArgs &a, hipStream_t st)
{
    constexpr int NW = WMW * WNW, NT = 64 * NW;
    constexpr int DW = (NW == 8 && KS == 3) ? 4 : NW;  // DMA waves, as in the kernel
    constexpr int HALO = (KS == 3) ? 1 : 0;
    if (a.mpad % BM) return MI355_EINVAL;
    a.mtiles = a.mpad / BM;
    if (a.ntiles_n < (a.total_n + BN - 1) / BN) a.ntiles_n = (a.total_n + BN - 1) / BN;  // tiles must fit BN
    // LDS rows of the row image: the most any tile of THIS plan spans (pixel rows + one pad row per image boundary crossed) + halo
    // rows -- not the bound for BN arbitrary pixels, which costs narrow maps the LDS their widest tiles need
    a.tile_q = a.total_n / a.ntiles_n; a.tile_r = a.total_n % a.ntiles_n;
    {
        const int hw = a.H * a.W;
        auto grow = [&](long n) { const long b = n / hw, r = n - b * hw; return (int)(b * (a.H + 1) + r / a.W + 1); };
        int span = 0;
        for (int t = 0; t < a.ntiles_n; ++t) {
            const long n0 = (long)t * a.tile_q + (t < a.tile_r ? t : a.tile_r), n1 = n0 + a.tile_q + (t < a.tile_r ? 1 : 0) - 1;
            if (n1 < n0) continue;
            const int sp = grow(n1) - grow(n0) + 1;
            if (sp > span) span = sp;
        }
        a.rows_cap = span + 2 * HALO;
    }
    const int ndma = a.rows_cap * (RS / 16);
    if ((ndma + DW - 1) / DW > rows16_nb_slots(BN, RS, DW, KS) + NBX) return MI355_EINVAL;  // map too narrow for this tile
    if ((size_t)a.in_cells * (size_t)a.in_cs >= ((size_t)1 << 32)) return MI355_EINVAL;  // 32-bit DMA lane offsets
    a.rowb = RS * 64 + 16 * (a.W & 15);
    a.fd_hw = fastdiv_make((uint32_t)(a.H * a.W)); a.fd_w = fastdiv_make((uint32_t)a.W);
    a.fd_ntn = fastdiv_make((uint32_t)a.ntiles_n); a.fd_nch = fastdiv_make((uint32_t)a.nchunks);
    size_t lds = (size_t)ra16_stages<KS, BN>() * BM * 64 + (size_t)rb16_stages<KS, BN>() * a.rows_cap * a.rowb + (size_t)a.rows_cap * RS * 4;
    const size_t lds_epi = (size_t)BN * (BM + 4) + (size_t)BN * 4;
    if (lds_epi > lds) lds = lds_epi;
    lds = (lds + 15) & ~(size_t)15;
    a.lds_param_off = (int)lds;  // beyond both the K-loop buffers and the epilogue tile
    lds += (size_t)BM * 32 + (a.yolo_out ? 1024 : 0);  // the logistic table only when a yolo head is fused (three 64 x 128 workgroups then fit a CU on 13-wide maps)
    if (lds > 160 * 1024) return MI355_EINVAL;
    auto kern = conv_rows16_i8_kernel<BM, BN, WMW, WNW, RS, KS, NBX>;
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

template <int RS>
static int rows16_launch_tile(ConvArgs &a, hipStream_t st, int bm, int bn)
{
    if (bm == 128 && bn == 384) return rows16_launch_cfg<128, 384, 2, 4, RS, 3>(a, st);
    if (bm == 128 && bn == 256) return rows16_launch_cfg<128, 256, 2, 4, RS, 3>(a, st);
    if (bm == 128 && bn == 192) return rows16_launch_cfg<128, 192, 2, 2, RS, 3>(a, st);  // four waves of 64 x 96: 0.42 fragment reads per MFMA instead of 0.5 (round 5)
    if (bm == 128 && bn == 128) return rows16_launch_cfg<128, 128, 2, 2, RS, 3>(a, st);
    if (bm == 64 && bn == 256) return rows16_launch_cfg<64, 256, 1, 4, RS, 3>(a, st);
    if (bm == 64 && bn == 128) return rows16_launch_cfg<64, 128, 1, 4, RS, 3>(a, st);
    return MI355_EINVAL;  // 32-filter tiles: conv_rows.hip
}

// Which 3x3 layers take this kernel is a measured choice (same-box A/B against conv_rows.hip, profiles/r02_v4_ab_and_startup.log):
// 16-slot row images (maps up to 14 wide: yolov3-tiny's 512 -> 1024 @13, -1 us) and the narrow maps on 32-slot rows that only the
// variant with exact LDS rows + extra DMA slots can give 128 x 384 tiles (YOLOv3-608's 512 -> 1024 @19: 92.7 -> 66.2 us).  On
// well-filled 32-slot rows (384 -> 256 @26) the 32x32x32 kernel is ~1 us faster (this kernel's 128 x 384 instantiation sits at the
// 256-register limit there) and keeps the layer; 64-slot rows were not measured and stay with it too.
// Round 4, measured and not kept: the 128 x 384 tile on FOUR waves of 128 x 96 (rows16_launch_cfg<128, 384, 1, 4, RS, 3>: one wave per
// SIMD, 192 accumulators in the AGPRs, 14 fragment reads per 48 MFMAs instead of 10 per 24; compiles without scratch only with
// -mllvm -pragma-unroll-threshold raised, see build.sh) -- bit-identical, L12 62-64 us against 54-57, L21 50-51 against 43-45
// (profiles/r04_rows16_4wave_ab.log): with one wave per SIMD nothing covers the issuing wave's DMA instructions and barrier waits.
// Round 4, measured and not kept (profiles/r04_rows16_adir_experiment.log): the A operand straight from global memory
// (the packed weights ARE the fragment images: one coalesced global_load_dwordx4 per fragment, three register sets, two K-steps
// ahead), no A ring, one barrier per channel chunk instead of one per K-step.  Bit-identical; the same time alone (L12 58-60 us on
// 128 x 128 tiles) and with four batches in flight (0.267-0.269 ms per step either way): two K-steps of fragments in registers do
// not cover the L2 latency when a wave is alone on its SIMD (K loop 22.9 us against 16.8 for the LDS ring, which holds five), a
// third step of fragments or 64 x 96 wave tiles spill at 256 registers.
// MI355_EINVAL -> conv_rows.hip's kernel.
int conv_rows16_launch(ConvArgs &a, hipStream_t st, int bm, int bn)
{
    if (a.cb != 64 || a.ksize != 3) return MI355_EINVAL;
    const int need = a.W + 2;
    if (need <= 16) return rows16_launch_tile<16>(a, st, bm, bn);
    if (need <= 32 && need < 24 && bm == 128 && bn == 384) return rows16_launch_cfg<128, 384, 2, 4, 32, 3, 4>(a, st);  // e.g. 19-wide maps
    return MI355_EINVAL;
}
