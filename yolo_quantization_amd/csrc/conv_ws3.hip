// conv_ws3.hip -- 3x3 stride-1 INT8 convolution with the weights stationary in registers, for the middle of the net:
// 128 or 256 input channels (K = 1152 / 2304: layers 8, 10 and 14 of yolov3-tiny at batch 64), when the whole layer is
// one round of workgroups.  Through the row-image kernel these layers ran 18-36 barrier-synchronised K-steps behind
// ~13 us of per-launch fixed cost with a third of every 128 x 384 tile idle (169-pixel images); here
//   * a wave keeps 36 K-steps (tap x 4 blocks of 32 channels = 144 VGPRs) of the A fragments of its 32 filters for the
//     whole launch.  With 128 input channels that is the whole K and a workgroup of 8 waves covers 256 filters; with
//     256 input channels two waves (K parts) share a filter quad and a workgroup covers 128 filters;
//   * the workgroup's tile is TP consecutive pixels (a whole 13x13 image, a quarter of a 26x26 one), staged once with
//     coalesced 16-byte loads into a cell-major LDS image (cell = its C channels + 16 B of bank skew); per-cell channel
//     sums come from the staging registers, 3x3 box sums and the pixel tables are built once per tile;
//   * the K loop of a group of 32 pixels is 36 MFMAs fed by one ds_read_b128 each, four K-steps ahead, whose tap and
//     channel-block offsets are immediates: no A reads from LDS, no barriers, no address arithmetic; the wave inside
//     its MFMA loop runs at raised priority so that the other wave's requantisation fills the gaps, not the reverse;
//   * K parts are chained through LDS with a single barrier: every wave first computes the partial sums of the groups
//     it does not own and parks them; after the barrier it seeds the accumulators of its own groups with its
//     partner's partial sums and finishes them (no extra additions).
// Same mathematics and the same bytes as conv_rows.hip (signed-operand decomposition: see conv_igemm.hip).
#include "kargs.h"
#include <type_traits>

#ifdef MI355_ABLATE
// per-wave phase timestamps (100 MHz wall clock): tools/conv_microbench.py --timeline3
#define W3_PHASES 8
__device__ long long g_ws3_ts[W3_PHASES][4096][8];
#define TS3(k)                                                                                                      \
    do {                                                                                                            \
        if ((threadIdx.x & 63) == 0 && blockIdx.x < 4096) g_ws3_ts[k][blockIdx.x][threadIdx.x >> 6] = wall_clock64(); \
    } while (0)
extern "C" int mi355_debug_read_ts3(long long *host)
{
    return hipMemcpyFromSymbol(host, HIP_SYMBOL(g_ws3_ts), sizeof(long long) * W3_PHASES * 4096 * 8) == hipSuccess ? 0 : -5;
}
// shader-clock sums per wave: [0] tap offsets + first B reads, [1] 36-MFMA loop, [2] epilogue / partial store, [3] groups
__device__ long long g_ws3_wp[4096][8][4];
#define WP3_DECL long long wp3[4] = {0, 0, 0, 0}; long long wp3_t = 0
#define WP3_START() do { wp3_t = __builtin_readcyclecounter(); } while (0)
#define WP3_MARK(k)                                              \
    do {                                                         \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       \
        const long long n_ = __builtin_readcyclecounter();       \
        wp3[k] += n_ - wp3_t;                                    \
        wp3_t = n_;                                              \
    } while (0)
#define WP3_STORE()                                                                                      \
    do {                                                                                                 \
        if ((threadIdx.x & 63) == 0 && blockIdx.x < 4096)                                                \
            for (int k = 0; k < 4; ++k) g_ws3_wp[blockIdx.x][threadIdx.x >> 6][k] = wp3[k];              \
    } while (0)
extern "C" int mi355_debug_read_wp3(long long *host)
{
    return hipMemcpyFromSymbol(host, HIP_SYMBOL(g_ws3_wp), sizeof(long long) * 4096 * 8 * 4) == hipSuccess ? 0 : -5;
}
// per-tile phase sums (wall clock, thread 0 of every workgroup) of the persistent form: [0] wait + barrier at the top of a tile (non-DB: the staging pass),
// [1] DMA issue of the next tile, [2] cell sums, [3] parameters + pixel tables + barrier, [4] box sums + barrier, [5] group loops, [6] tiles
__device__ long long g_ws3_acc[8][4096];
#define ACC3_DECL long long acc3[8] = {0, 0, 0, 0, 0, 0, 0, 0}; long long acc3_t = wall_clock64()
#define ACC3(k) do { const long long n_ = wall_clock64(); acc3[k] += n_ - acc3_t; acc3_t = n_; } while (0)
#define ACC3_STORE() do { if (threadIdx.x == 0 && blockIdx.x < 4096) for (int k = 0; k < 8; ++k) g_ws3_acc[k][blockIdx.x] = acc3[k]; } while (0)
extern "C" int mi355_debug_read_acc3(long long *host)
{
    return hipMemcpyFromSymbol(host, HIP_SYMBOL(g_ws3_acc), sizeof(long long) * 8 * 4096) == hipSuccess ? 0 : -5;
}
#else
#define ACC3_DECL do { } while (0)
#define ACC3(k) do { } while (0)
#define ACC3_STORE() do { } while (0)
#define TS3(k) do { } while (0)
#define WP3_DECL do { } while (0)
#define WP3_START() do { } while (0)
#define WP3_MARK(k) do { } while (0)
#define WP3_STORE() do { } while (0)
#endif

// Fused stride-2 pool: pixel order inside a block of 32 (= 8 windows = 2 image rows x 16 columns).  A ds_read_b128 is served
// in lane groups {0-3, 12-15, 20-27} and {4-11, 16-19, 28-31} (one LDS cycle each when the 16 lanes hit 16 different 16-byte
// bank groups): 16 consecutive cells of ONE image row always do (cells are an odd number of 16-byte units apart), cells of two
// rows generally do not.  So the first lane group takes the windows' upper row, the second their lower row.
__host__ __device__ constexpr int ws3_pm2_cell(int l)  // lane of the block -> row << 4 | column
{
    return l < 4 ? l : l < 12 ? 16 | (l - 4) : l < 16 ? l - 8 : l < 20 ? 16 | (l - 8) : l < 28 ? l - 12 : 16 | (l - 16);
}
__host__ __device__ constexpr int ws3_pm2_lane(int row, int col)  // inverse
{
    return row == 0 ? (col < 4 ? col : col < 8 ? col + 8 : col + 12) : (col < 8 ? col + 4 : col < 12 ? col + 8 : col + 16);
}
// LDS-DMA in its "scalar 64-bit base + 32-bit lane offset" form (see conv_rows16.hip): M0 carries the LDS destination of lane 0, lane i lands 16 i
// bytes further on; nothing else in this kernel uses M0
#define WS3_DMA(ldsdst_u32, sbase_ptr, voff_u32)                                                                  \
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(ldsdst_u32), "v"(voff_u32), \
                 "s"(sbase_ptr)                                                                                  \
                 : "memory", "m0")
constexpr int WS3_GMAX = 8;    // groups of 32 pixels per tile
constexpr int WS3_UB = 8;      // 16-byte staging loads in flight per thread

// PM: fused maxpool after the conv (ConvArgs::pool_mode).  The requantised bytes of the tile are staged in LDS (one 32 pixel x
// 32 filter slot per wave and group: for KP == 2 the slot of the partial sums the wave has just consumed), and after the last
// group the workgroup forms the window maxima from the bytes -- the reference's order (bytes first, then the maximum:
// src/maxpool_layer.c:134-146), so wrapped bytes need no special case.  PM == 2: the tile is a run of whole 2x2 windows,
// pixel 4 w + j = position j of window w, and the conv's own tensor is stored too when a.y is given (a route reads it);
// PM == 1: whole-image tiles, window = the pixel, its right, lower and lower-right neighbours inside the image.
template <int KP, int ACT, bool SAT, int PM>
__global__ __launch_bounds__(512, 2) void conv_ws3_kernel(const ConvArgs a)
{
    constexpr int KST = 36, PIECES = 8 * KP, PSH = (KP == 1) ? 3 : 4;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int CELLB = (PIECES + 1) * 16;                    // bytes of an LDS image cell: its channels + 16 B of bank skew
    const int ncell = a.sm_ncell;                               // cells per LDS image row = W + 2
    const int RS = a.rows_cap;                                  // LDS image rows (row R lives at R % RS)
    const int NQ = a.sm_nq, NF = 32 * NQ;
    const int TP = a.sm_tp, G = ((PM == 2 ? 4 * TP : TP) + 31) >> 5;  // PM == 2: TP counts 2x2 windows (8 per group of 32 pixels)
    int *ldsS = reinterpret_cast<int *>(smem + a.sm_pieceb);              // [RS * ncell] per-cell channel sums
    int *ldsSX = ldsS + ((RS * ncell + 3) & ~3);                          // [G][32] 3x3 box sum of the pixel
    int *ldsBase = ldsSX + G * 32;                                        // [G][32] image row | column << 16 of tap (0,0)
    int *ldsCell = ldsBase + G * 32;                                      // [G][32] output cell, -1: no pixel
    double *ldsMP = reinterpret_cast<double *>(smem + a.lds_param_off);   // [NF] folded multiplier
    int *ldsDZ = reinterpret_cast<int *>(ldsMP + NF);                     // [NF] 128 - zp_w
    int *ldsCB = ldsDZ + NF;                                              // [NF] cw + bias
    char *ldsRed = smem + a.sm_red_off;                                   // K-part partial sums, 4 KiB per (quad, set, group)
    char *ldsPT = smem + a.sm_pt_off;                                     // PM: staged bytes, slot (set, quad, group) = [32 px][36 B]
    int *ldsPCell = ldsCell + G * 32;                                     // PM == 2: [G * 8] pooled cell of window w, -1: none

    // Round 5 -- double-buffered DMA staging (DB): the persistent 128-channel form (several tiles per workgroup: YOLOv3-608's 128 -> 256 @76 layers)
    // staged a tile's image through registers, waited, computed, and nothing of tile t + 1 was on its way meanwhile: 35 of 83 us per launch
    // (profiles/r04_conv_ws3_608_ablation.log).  The skewed cell-major layout IS writable by the LDS-DMA after all: the image is a linear array of
    // 16-byte units, nine per cell (eight pieces + the skew unit), and an instruction writes 64 consecutive units -- the lanes that fall on a skew
    // unit fetch the cell's first piece again (1/9 of the DMA's bytes).  With two image buffers (a.sm_hc = their byte distance, 0: single buffer) a
    // tile's image is fetched while the previous tile computes; the per-cell channel sums then come from LDS instead of the staging registers.
    const bool DB = KP == 1 && PM == 0 && a.sm_hc != 0;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char *)smem;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kh = lane >> 5, lj = lane & 31;
    const int wq = wave % NQ, kp = (wave / NQ) % KP, wset = wave / (NQ * KP), nset = 8 / (NQ * KP);
    // persistent over pixel tiles: workgroup (mt, wg) keeps the A fragments of filter tile mt in its registers and walks
    // the tiles wg, wg + nwg, .. (one tile per workgroup when the whole layer is a single round: the yolov3-tiny layers)
    const int mt = blockIdx.x % a.mtiles, wg = blockIdx.x / a.mtiles, nwg = gridDim.x / a.mtiles;
    const int f0 = mt * NF;                      // first filter of the workgroup
    // a.H x a.W is the INPUT map; pixels (tiles, groups, lanes) enumerate the OUTPUT map OHd x OWd = the input map for
    // stride 1, its even positions for stride 2 (3x3, pad 1): output (y, x) reads input rows S y - 1 .. S y + 1, columns
    // S x - 1 .. S x + 1, so only the pixel -> image-cell tables know about the stride
    const int W1 = a.W + 1, S = a.stride, OHd = a.OH, OWd = a.OW, hw = OHd * OWd;
    const bool pow2 = a.hdr->pow2 == 1;
    const int OWp = OWd >> 1, ohwp = (OHd >> 1) * OWp;  // PM == 2: the pooled map
    const int total_u = PM == 2 ? a.B * ohwp : a.total_n;  // tile units: windows or pixels
    // first image row (flattened (b, y) row space) and row count of a tile's LDS image, halo rows included
    auto tile_rows = [&](int u0, int u1, int &gr_first, int &nrows) {
        int b0, r0, b1, r1;
        if (PM == 2) {  // windows [u0, u1): conv rows 2 py .. 2 py + 1
            b0 = u0 / ohwp; r0 = 2 * ((u0 - b0 * ohwp) / OWp);
            b1 = (u1 - 1) / ohwp; r1 = 2 * (((u1 - 1) - b1 * ohwp) / OWp) + 1;
        } else {
            b0 = u0 / hw; r0 = (u0 - b0 * hw) / OWd;
            b1 = (u1 - 1) / hw; r1 = ((u1 - 1) - b1 * hw) / OWd;
        }
        gr_first = b0 * (a.H + 1) + S * r0 + 1;
        nrows = b1 * (a.H + 1) + S * r1 + 1 - gr_first + 3;
    };
    v4i wf[KST];
    WP3_DECL;
    TS3(0);
    // staging of one tile's image (and, for the workgroup's first tile, its A fragments behind the first image batch)
    auto stage = [&](int tile, auto with_a_c) {
        constexpr bool WITH_A = decltype(with_a_c)::value;
        constexpr int UB = WITH_A ? WS3_UB : 4;  // later tiles stage with fewer loads in flight: the A fragments hold 144 VGPRs (6 / 8 measured: no gain)
        const int p0 = tile * TP, p1 = min(p0 + TP, total_u);  // this tile's pixels (PM == 2: windows) [p0, p1)

        // ---- tile geometry: image rows [first - 1, last + 1] of the flattened (b, y) row space, columns -1 .. W
        int gr_first, nrows;
        tile_rows(p0, p1, gr_first, nrows);
        const long org = (long)a.in_lead + (long)(gr_first - 1) * W1 - 1;
        // LDS image: cell-major, row R (0 = the row above the tile's first pixel row) at cells [(R % RS) ncell, +ncell), cell
        // c = column c - 1; a cell is its C channels followed by 16 B of skew (CELLB / 16 is odd: the 16 lanes of a
        // B-fragment read, one cell apart, hit 16 different 16-byte bank groups).  Every tap / channel-block offset of a
        // K-step is then a compile-time immediate of the ds_read.  RS < nrows only for tiles that are one whole image:
        // the pad row below it aliases the pad row above it.
        const int srows = min(nrows, RS);

        // ---- stage the image: unit u = (cell, piece); a wave instruction reads 1 KiB of consecutive bytes
        {
            const int total_u = (srows * ncell) << PSH;
            for (int u0 = 0; u0 < total_u; u0 += 512 * UB) {
                v4i v[UB];
#pragma unroll
                for (int i = 0; i < UB; ++i) {
                    const int u = min(u0 + i * 512 + tid, total_u - 1);
                    const int lin = u >> PSH, piece = u & (PIECES - 1);
                    const int r = lin / ncell, c = lin - r * ncell;
                    long f = org + (long)r * W1 + c;
                    f = f < 0 ? 0 : (f > a.in_cells - 1 ? a.in_cells - 1 : f);
                    v[i] = *reinterpret_cast<const v4i *>(a.x + (size_t)f * a.in_cs + piece * 16);
                }
                if (WITH_A && u0 == 0) {  // this wave's A fragments queue behind the first image batch of the first tile
#pragma unroll
                    for (int s = 0; s < KST; ++s)
                        wf[s] = *reinterpret_cast<const v4i *>(a.ws + ((size_t)(((f0 >> 5) + wq) * KP + kp) * KST + s) * 1024 + lane * 16);
                }
#pragma unroll
                for (int i = 0; i < UB; ++i) {
                    const int u = u0 + i * 512 + tid;
                    const int lin = u >> PSH, piece = u & (PIECES - 1);
                    int t = 0;
                    t = __builtin_amdgcn_sdot4(v[i][0], 0x01010101, t, false);
                    t = __builtin_amdgcn_sdot4(v[i][1], 0x01010101, t, false);
                    t = __builtin_amdgcn_sdot4(v[i][2], 0x01010101, t, false);
                    t = __builtin_amdgcn_sdot4(v[i][3], 0x01010101, t, false);
#pragma unroll
                    for (int m = 1; m < PIECES; m <<= 1) t += __shfl_xor(t, m);
                    if (u < total_u) {
                        *reinterpret_cast<v4i *>(smem + lin * CELLB + piece * 16) = v[i];
                        if (piece == 0) ldsS[lin] = t;
                    }
                }
            }
        }
    };
    // DB: fetch a tile's image into buffer `b` (asynchronous; counted in vmcnt like any load)
    auto dma_tile = [&](int tile, int b) {
        const int p0 = tile * TP, p1 = min(p0 + TP, total_u);
        int gr_first, nrows;
        tile_rows(p0, p1, gr_first, nrows);
        const long org = (long)a.in_lead + (long)(gr_first - 1) * W1 - 1;
        const int units = min(nrows, RS) * ncell * (PIECES + 1);  // 16-byte units of the image, skew units included
        const int ninstr = (units + 63) >> 6;                     // (the buffer is a whole number of KiB: the launcher rounds it up)
        const unsigned dst0 = lds0 + (unsigned)b * (unsigned)a.sm_hc;
        const long maxcell = (long)a.in_cells - 1;
        for (int k = wave; k < ninstr; k += 8) {
            const int q = min(k * 64 + lane, units - 1);
            const int cell = q / (PIECES + 1), piece = q - cell * (PIECES + 1);
            const int r = fd_div(cell, a.fd_w), c = cell - r * ncell;  // (fd_w: the launcher's division by ncell)
            long f = org + (long)r * W1 + c;
            f = f < 0 ? 0 : (f > maxcell ? maxcell : f);
            const unsigned voff = (unsigned)(f * a.in_cs) + (unsigned)(piece < PIECES ? piece : 0) * 16u;  // < 2^32: the launcher checks
            const unsigned d = dst0 + (unsigned)k * 1024u;
            WS3_DMA(d, a.x, voff);
        }
    };
    if (DB) {
        if (wg < a.ntiles_n) dma_tile(wg, 0);
        // this wave's A fragments queue behind the first image
#pragma unroll
        for (int s = 0; s < KST; ++s)
            wf[s] = *reinterpret_cast<const v4i *>(a.ws + ((size_t)(((f0 >> 5) + wq) * KP + kp) * KST + s) * 1024 + lane * 16);
    } else if (wg < a.ntiles_n) {
        stage(wg, std::true_type{});
    }
    int dbuf = 0;
    ACC3_DECL;
    for (int tile = wg, first = 1; tile < a.ntiles_n; tile += nwg, first = 0, dbuf ^= 1) {
    ACC3(7);
    if (DB) {
        // this tile's image has landed (every wave's share: barrier) and every wave is done with the previous tile, whose buffer the next
        // tile's image may now overwrite.  As a builtin, so that the compiler's wait-count pass knows nothing is pending (the A fragments, first time round)
        __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
        __syncthreads();
        ACC3(0);
        if (tile + nwg < a.ntiles_n) dma_tile(tile + nwg, dbuf ^ 1);
        ACC3(1);
        // per-cell channel sums of this tile's image, from LDS
        {
            int gr_first, nrows;
            tile_rows(tile * TP, min(tile * TP + TP, total_u), gr_first, nrows);
            const int cells = min(nrows, RS) * ncell;
            const char *img = smem + (size_t)dbuf * a.sm_hc;
            for (int id = tid; id < cells; id += 512) {
                int t = 0;
#pragma unroll
                for (int pc = 0; pc < PIECES; ++pc) {
                    const v4i v = *reinterpret_cast<const v4i *>(img + id * CELLB + pc * 16);
                    t = __builtin_amdgcn_sdot4(v[0], 0x01010101, t, false);
                    t = __builtin_amdgcn_sdot4(v[1], 0x01010101, t, false);
                    t = __builtin_amdgcn_sdot4(v[2], 0x01010101, t, false);
                    t = __builtin_amdgcn_sdot4(v[3], 0x01010101, t, false);
                }
                ldsS[id] = t;
            }
        }
        ACC3(2);
    } else if (!first) {
        __syncthreads();  // every wave is done with the previous tile's image, tables and parked partial sums
        stage(tile, std::false_type{});
        ACC3(0);
    }
    const int p0 = tile * TP, p1 = min(p0 + TP, total_u);  // this tile's pixels (PM == 2: windows) [p0, p1)

    // ---- tile geometry: image rows [first - 1, last + 1] of the flattened (b, y) row space, columns -1 .. W
    int gr_first, nrows;
    tile_rows(p0, p1, gr_first, nrows);
    // LDS image: cell-major, row R (0 = the row above the tile's first pixel row) at cells [(R % RS) ncell, +ncell), cell
    // c = column c - 1; a cell is its C channels followed by 16 B of skew (CELLB / 16 is odd: the 16 lanes of a
    // B-fragment read, one cell apart, hit 16 different 16-byte bank groups).  Every tap / channel-block offset of a
    // K-step is then a compile-time immediate of the ds_read.  RS < nrows only for tiles that are one whole image:
    // the pad row below it aliases the pad row above it.

    TS3(1);
    // ---- per-channel parameters of the workgroup's filters
    if (first) for (int i = tid; i < NF; i += 512) {
        ldsMP[i] = a.mprime[f0 + i];
        ldsDZ[i] = a.dzp[f0 + i];
        ldsCB[i] = a.cwb[f0 + i];
    }
    // ---- pixels of the tile: image offset of tap (0,0), output cell
    for (int idx = tid; idx < G * 32; idx += 512) {
        if (PM == 2) {  // a block of 32 pixels = 8 windows = 16 columns x 2 rows; lane -> (row, column) by ws3_pm2_cell
            const int rc = ws3_pm2_cell(idx & 31), j = ((rc >> 4) << 1) | (rc & 1);
            const int q = p0 + (idx >> 5) * 8 + ((rc & 15) >> 1);
            const bool valid = q < p1;
            const int qc = valid ? q : p1 - 1;  // idle lanes shadow the tile's last window
            const int b = qc / ohwp, rem = qc - b * ohwp;
            const int py = rem / OWp, px = rem - py * OWp;
            const int y = 2 * py + (j >> 1), x = 2 * px + (j & 1);
            ldsBase[idx] = (b * (a.H + 1) + y + 1 - gr_first) | (x << 16);
            ldsCell[idx] = valid ? a.out_lead + (b * (OHd + 1) + (y + 1)) * (OWd + 1) + x : -1;
            if (j == 0) ldsPCell[q - p0] = valid ? a.pool_lead + (b * ((OHd >> 1) + 1) + (py + 1)) * (OWp + 1) + px : -1;
        } else {
            const int p = p0 + idx;
            const bool valid = p < p1;
            const int pc = valid ? p : p1 - 1;  // idle lanes shadow the tile's last pixel
            const int b = pc / hw, rem = pc - b * hw;
            const int y = rem / OWd, x = rem - y * OWd;
            ldsBase[idx] = (b * (a.H + 1) + S * y + 1 - gr_first) | ((S * x) << 16);
            ldsCell[idx] = valid ? a.out_lead + (b * (OHd + 1) + (y + 1)) * (OWd + 1) + x : -1;
        }
    }
    __syncthreads();
    ACC3(3);
    TS3(2);
    // first cell of image row rr + dy (rows past RS alias row 0: whole-image tiles)
    auto row_cell = [&](int rr, int dy) {
        const int R = rr + dy;
        return (R >= RS ? R - RS : R) * ncell;
    };
    for (int idx = tid; idx < G * 32; idx += 512) {
        const int rx = ldsBase[idx], rr = rx & 0xFFFF, x = rx >> 16;
        int t = 0;
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) t += ldsS[row_cell(rr, dy) + x + dx];
        ldsSX[idx] = t;
    }
    __syncthreads();
    ACC3(4);
    TS3(3);
    // All A fragments have to be in before the group loops: with the waits inside the (shared) loop body every group
    // would end on an s_waitcnt vmcnt(0), and vmcnt also counts the previous group's stores -- a store round trip per
    // group.  As a builtin, so that the compiler's wait-count pass knows nothing is pending any more.
    if (!DB) __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)   (DB: waited at the top of the tile; here the NEXT tile's image is in flight, on purpose)
    TS3(4);

    // K-step s of this wave: tap s / 4, channels 128 kp + 32 (s % 4) + 16 kh .. + 15 = bytes [128 kp + 32 (s % 4) + 16 kh, +16)
    // of the tap's cell.  The lane's k-half and the wave's K part go into its three row bases; the tap column and the
    // channel block are immediates of the ds_read: no address arithmetic inside the K loop.
    const char *X = smem + (DB ? (size_t)dbuf * a.sm_hc : 0);
    const int lane_off = (8 * kp + kh) * 16;
    const int lw = 32 * wq;  // first filter of the wave within the workgroup's parameter tables
    const int Gs = (G - wset + nset - 1) / nset;      // groups of this wave set: wset, wset + nset, ..
    const int hown = (Gs + 1) >> 1;                   // K part 0 owns the first hown of them, K part 1 the rest
    const int gsmax = (G + nset - 1) / nset;
    char *red = ldsRed + (size_t)((wset * NQ + wq) * gsmax) * 4096 + lane * 16;

    // B fragments are fetched four K-steps ahead of the MFMA that consumes them (a ring of four register sets; the
    // sched_barrier after every step keeps the compiler from sinking the reads back to their use)
    auto kloop = [&](v16i &acc, int rx) {
        WP3_START();
        int rowoff[3];
        {
            const int rr = rx & 0xFFFF, x = rx >> 16;
#pragma unroll
            for (int dy = 0; dy < 3; ++dy) rowoff[dy] = (row_cell(rr, dy) + x) * CELLB + lane_off;
        }
        auto ldb = [&](int s) {
            const int tap = s >> 2;
            return *reinterpret_cast<const v4i *>(X + rowoff[tap / 3] + (tap % 3) * CELLB + (s & 3) * 32);
        };
        v4i bq[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) bq[j] = ldb(j);
        __builtin_amdgcn_sched_barrier(0);
        WP3_MARK(0);
        // the wave inside its MFMA loop outranks the one that requantises: the matrix pipe never waits for VALU traffic
        __builtin_amdgcn_s_setprio(3);
#ifdef MI355_ABLATE
        if (a.debug & (1 << 17)) {  // timing ablation: no MFMA loop
            __builtin_amdgcn_s_setprio(0);
            WP3_MARK(1);
            return;
        }
#endif
#pragma unroll
        for (int s = 0; s < KST; ++s) {
            acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(wf[s], bq[s & 3], acc, 0, 0, 0);
            if (s + 4 < KST) bq[s & 3] = ldb(s + 4);
            __builtin_amdgcn_sched_barrier(0);
        }
        __builtin_amdgcn_s_setprio(0);
#ifdef MI355_ABLATE
        asm volatile("s_nop 0" ::"v"(acc[0]));  // the last MFMA's result: the mark below waits for it
#endif
        WP3_MARK(1);
    };
    auto seed_bias = [&](v16i &acc) {
#pragma unroll
        for (int grp = 0; grp < 4; ++grp) {
            const int4 c4 = *reinterpret_cast<const int4 *>(ldsCB + lw + 16 * kh + 4 * grp);  // rows 8 grp + 4 kh + r = filters 16 kh + 4 grp + r (kargs.h ws_row_filter)
            acc[grp * 4 + 0] = c4.x; acc[grp * 4 + 1] = c4.y; acc[grp * 4 + 2] = c4.z; acc[grp * 4 + 3] = c4.w;
        }
    };
    auto finish = [&](const v16i &acc, int g, int i) {  // i: index of group g within this wave set (its staging slot)
        const int sx = ldsSX[g * 32 + lj];
        const int cell = (PM && !a.y) ? -1 : ldsCell[g * 32 + lj];
        // (+ 32 B per quad: the slots are a multiple of 128 B apart, and the pool pass reads one pixel's 32 dwords of all quads at once)
        char *pt = ldsPT + (size_t)((wset * NQ + wq) * gsmax + i) * a.sm_pt_stride + (wq & 3) * 32 + lj * 36 + 16 * kh;
        // the lane's sixteen consecutive filters 16 kh .. + 15 of its quad: one 16-byte store (and one 16-byte residual load) per group of pixels
        uint8_t *dst = a.y + (size_t)(cell < 0 ? 0 : cell) * a.out_cs + f0 + lw + 16 * kh;
        // fused residual add: the `from` tensor's bytes of the same pixel and channels, fetched before the requantisation
        uint32_t resv[4] = {0, 0, 0, 0};
        if (a.res) {
            const uint4 r4 = *reinterpret_cast<const uint4 *>(a.res + (size_t)(cell < 0 ? 0 : cell + a.res_delta) * a.res_cs + f0 + lw + 16 * kh);
            resv[0] = r4.x; resv[1] = r4.y; resv[2] = r4.z; resv[3] = r4.w;
        }
#ifdef MI355_ABLATE
        if (a.debug & (1 << 18)) {  // timing ablation: stores only
            if (cell >= 0)
                for (int grp = 0; grp < 4; ++grp) *reinterpret_cast<uint32_t *>(dst + 4 * grp) = acc[grp * 4] + sx;
            return;
        }
        if (a.debug & (1 << 19)) return;  // timing ablation: no epilogue at all
#endif
        uint32_t o4[4];
        if (pow2) {
#pragma unroll
            for (int half = 0; half < 2; ++half) {  // 8 channels per call: eight independent chains, one fallback ballot
                int32_t accb[8], v[8];
                double mp[8];
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int grp = 2 * half + h, cl = lw + 16 * kh + 4 * grp;
                    const int4 dz4 = *reinterpret_cast<const int4 *>(ldsDZ + cl);
                    const int dzv[4] = {dz4.x, dz4.y, dz4.z, dz4.w};
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        mp[4 * h + r] = ldsMP[cl + r];
                        accb[4 * h + r] = acc[grp * 4 + r] + __mul24(dzv[r], sx);
                    }
                }
                requant_values_mp<ACT, SAT, 8>(accb, mp, a.zp_act, v);
                uint32_t o0 = pack4_biased(v[0], v[1], v[2], v[3]), o1 = pack4_biased(v[4], v[5], v[6], v[7]);
                if (a.res) {
                    o0 = shortcut4_biased(o0, resv[2 * half], a.sc_ka, a.sc_kb, a.sc_k0);
                    o1 = shortcut4_biased(o1, resv[2 * half + 1], a.sc_ka, a.sc_kb, a.sc_k0);
                }
                o4[2 * half] = o0;
                o4[2 * half + 1] = o1;
            }
        } else {  // shift_value not a power of two: the reference's two-step form (never produced by its own prep)
            // (the sixteen two-step requantisations as a ROLLED loop over a private array: unrolled, this cold path took part in sizing the
            // kernel's registers -- it sits at the 256-register limit, three of its instantiations with spills; see conv_small.hip)
            int32_t tmp[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) tmp[i] = acc[i];
#pragma unroll 1
            for (int i = 0; i < 16; ++i) {
                const int cl = lw + 16 * kh + i;  // group i >> 2, channel i & 3 of it
                tmp[i] = (int32_t)requant_u8(tmp[i] + __mul24(ldsDZ[cl], sx), 0, a.mval[f0 + cl], a.sval[f0 + cl], a.zp_act, ACT,
                                             SAT ? MI355_STORE_SATURATE : MI355_STORE_WRAP);
            }
#pragma unroll
            for (int grp = 0; grp < 4; ++grp) {
                const int32_t v[4] = {tmp[4 * grp], tmp[4 * grp + 1], tmp[4 * grp + 2], tmp[4 * grp + 3]};
                uint32_t o = pack4_biased(v[0], v[1], v[2], v[3]);
                if (a.res) o = shortcut4_biased(o, resv[grp], a.sc_ka, a.sc_kb, a.sc_k0);
                o4[grp] = o;
            }
        }
        if (cell >= 0) *reinterpret_cast<uint4 *>(dst) = uint4{o4[0], o4[1], o4[2], o4[3]};
        if (PM) {
#pragma unroll
            for (int grp = 0; grp < 4; ++grp) *reinterpret_cast<uint32_t *>(pt + 4 * grp) = o4[grp];
        }
    };
    if (KP == 1) {
#pragma unroll 1
        for (int i = 0; i < Gs; ++i) {
            const int g = wset + i * nset;
            v16i acc;
            seed_bias(acc);
            kloop(acc, ldsBase[g * 32 + lj]);
            finish(acc, g, i);
            WP3_MARK(2);
        }
    } else {
        // phase 1: partial sums of the partner's groups
        const int i0 = kp == 0 ? hown : 0, i1 = kp == 0 ? Gs : hown;
#pragma unroll 1
        for (int i = i0; i < i1; ++i) {
            const int g = wset + i * nset;
            v16i acc;
            seed_bias(acc);
            kloop(acc, ldsBase[g * 32 + lj]);
            v4i *dst = reinterpret_cast<v4i *>(red + i * 4096);
#pragma unroll
            for (int j = 0; j < 4; ++j) dst[j * 64] = v4i{acc[4 * j], acc[4 * j + 1], acc[4 * j + 2], acc[4 * j + 3]};
            WP3_MARK(2);
        }
        TS3(5);
        __syncthreads();
        TS3(6);
        // phase 2: this wave's own groups, seeded with the partner's partial sums
        const int j0 = kp == 0 ? 0 : hown, j1 = kp == 0 ? hown : Gs;
#pragma unroll 1
        for (int i = j0; i < j1; ++i) {
            const int g = wset + i * nset;
            v16i acc;
            const v4i *src = reinterpret_cast<const v4i *>(red + i * 4096);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const v4i t = src[j * 64];
                acc[4 * j] = t[0]; acc[4 * j + 1] = t[1]; acc[4 * j + 2] = t[2]; acc[4 * j + 3] = t[3];
            }
            kloop(acc, ldsBase[g * 32 + lj]);
            finish(acc, g, i);
            WP3_MARK(2);
        }
    }
#ifdef MI355_ABLATE
    const bool pool_pass = PM && !(a.debug & (1 << 22));  // timing ablation: no pool pass (the pooled tensor is not written)
#else
    const bool pool_pass = PM;
#endif
    if (pool_pass) {
        __syncthreads();  // every group's bytes are staged
        // NQ and nset are powers of two (conv_ws3_eligible): no divisions in the loops below
        const int lgND = 3 + __builtin_ctz(NQ), ND = 1 << lgND;  // dwords of a pixel's filters in this workgroup
        const int lgns = __builtin_ctz(nset);
        // staged dword: pixel idx of the tile, dword d of the workgroup's filters (quad d >> 3, 4 (d & 7) bytes into its 32)
        // The pass is bound by its instruction count (and a 32-bit v_mul_lo / a 64-bit mad is a quarter-rate instruction): a thread
        // keeps ONE dword column d of the workgroup's filters, so everything that depends on d is computed once; 24-bit multiplies
        // and 32-bit offsets elsewhere.  Four independent pixels / windows per iteration, branch free (LDS round trips overlap).
        const unsigned pts = (unsigned)a.sm_pt_stride, pcs = (unsigned)a.pool_cs;
        const __attribute__((address_space(3))) char *pt3 = (const __attribute__((address_space(3))) char *)ldsPT;
        const unsigned d = (unsigned)tid & (unsigned)(ND - 1), dq = d >> 3;
        const unsigned qoff = __umul24(__umul24(dq, (unsigned)gsmax), pts) + (dq & 3) * 32 + (d & 7) * 4;  // this thread's quad and dword in a slot
        const unsigned setstride = __umul24((unsigned)(NQ * gsmax), pts);                                  // between wave sets
        auto staged = [&](int idx) {
            const unsigned g = (unsigned)idx >> 5, st = g & (unsigned)(nset - 1), i = g >> lgns;
            const unsigned off = qoff + __umul24(st, setstride) + __umul24(i, pts) + __umul24((unsigned)idx & 31, 36u);
            return *reinterpret_cast<const __attribute__((address_space(3))) uint32_t *>(pt3 + off);
        };
        uint8_t *pout = a.ypool + f0 + 4 * d;  // + cell * pool_cs: below 2^32 (the launcher checks the pooled tensor's size)
        constexpr int UN = 4;
        const int PPI = 512 >> lgND;  // pixels / windows per sweep of the workgroup
        const int nu = p1 - p0;
        for (int u0 = tid >> lgND; u0 < nu; u0 += PPI * UN) {
            uint32_t m[UN];
            int pc[UN];
#pragma unroll
            for (int u = 0; u < UN; ++u) {
                const int w = min(u0 + u * PPI, nu - 1);
                if (PM == 2) {
                    const int blk = (w >> 3) << 5, c0 = (w & 7) << 1;  // the window's two columns in its block
                    m[u] = max4_s8x4(staged(blk + ws3_pm2_lane(0, c0)), staged(blk + ws3_pm2_lane(0, c0 + 1)),
                                     staged(blk + ws3_pm2_lane(1, c0)), staged(blk + ws3_pm2_lane(1, c0 + 1)));
                    pc[u] = ldsPCell[w];
                } else {  // whole-image tiles (the launcher guarantees it): pixel w = (y, x) of image p0 / hw; ldsBase holds (y | x << 16)
                    const int rx = ldsBase[w], y = rx & 0xFFFF, x = rx >> 16;
                    // neighbours outside the image fall back on pixels of the window that are inside it
                    const int i1 = w + (x + 1 < OWd ? 1 : 0), i2 = w + (y + 1 < OHd ? OWd : 0), i3 = i2 + (i1 - w);
                    m[u] = max4_s8x4(staged(w), staged(i1), staged(i2), staged(i3));
                    pc[u] = ldsCell[w] + (a.pool_lead - a.out_lead);  // the pooled map has the conv map's geometry
                }
            }
#pragma unroll
            for (int u = 0; u < UN; ++u)
                if (u0 + u * PPI < nu) *reinterpret_cast<uint32_t *>(pout + __umul24((unsigned)pc[u], pcs)) = m[u];
        }
    }
    ACC3(5);
#ifdef MI355_ABLATE
    acc3[6] += 1;
#endif
    }  // tiles
    ACC3_STORE();
    TS3(7);
    WP3_STORE();
}

template <int KP, int ACT, int PM>
static int w3_launch_sat(ConvArgs &a, hipStream_t st, int grid, size_t lds)
{
    if (a.store_mode == MI355_STORE_SATURATE) {
        return launch_big_lds<conv_ws3_kernel<KP, ACT, true, PM>>(grid, 512, lds, st, a);
    }
    return launch_big_lds<conv_ws3_kernel<KP, ACT, false, PM>>(grid, 512, lds, st, a);
}

template <int KP, int PM>
static int w3_launch_act(ConvArgs &a, hipStream_t st, int grid, size_t lds)
{
    if (a.act == MI355_ACT_LEAKY) return w3_launch_sat<KP, MI355_ACT_LEAKY, PM>(a, st, grid, lds);
    if (a.act == MI355_ACT_RELU6) return w3_launch_sat<KP, MI355_ACT_RELU6, PM>(a, st, grid, lds);
    return w3_launch_sat<KP, MI355_ACT_LINEAR, PM>(a, st, grid, lds);
}

template <int KP>
static int w3_launch_pm(ConvArgs &a, hipStream_t st, int grid, size_t lds)
{
    if (a.pool_mode == 2) return w3_launch_act<KP, 2>(a, st, grid, lds);
    if (a.pool_mode == 1) return w3_launch_act<KP, 1>(a, st, grid, lds);
    return w3_launch_act<KP, 0>(a, st, grid, lds);
}

static int ws3_quads(int n, int c)
{
    const int kp = c / 128, q = n / 32, qmax = 8 / kp;
    return q < qmax ? q : qmax;
}

// shapes whose blob carries the weights-stationary plane (off_ws) for this kernel: [n/32 quads][K part][36][64 lanes][16 B]
bool conv_ws3_eligible(int n, int c, int ksize)
{
    if (ksize != 3 || (c != 128 && c != 256) || n % 32) return false;
    const int nq = ws3_quads(n, c);
    return (nq & (nq - 1)) == 0 && (n / 32) % nq == 0;
}

// returns MI355_EINVAL when the shape is outside this kernel's domain (the caller falls back to conv_rows / conv_igemm)
int conv_ws3_launch(ConvArgs &a, hipStream_t st)
{
    const int c = a.cb * a.nchunks;
    if (!conv_ws3_eligible(a.n, c, a.ksize) || !a.ws || a.acc_out || a.y_f32 || a.yolo_out) return MI355_EINVAL;
    // A/B switches (tools/dbg): bit 22 / 23 send the 256- / 128-channel layers to the row-image kernel
    if ((c == 256 && (mi355_debug_flags_get() & (1 << 22))) || (c == 128 && (mi355_debug_flags_get() & (1 << 23)))) return MI355_EINVAL;
    // throughput plan: the 256-channel form needs the whole LDS of a CU (image + 96 KB of parked K-part sums) -- with other
    // batches' small workgroups trickling onto every CU its launch starves (89 us average per launch against 22 alone,
    // profiles/r03_overlap_*.md); the row-image kernel's 128 x 128 tiles share a CU and take its place
    // (the 128-channel form, 122 KB: in-flight step 0.2836 -> 0.2786 ms without it, one box, profiles/r03_plan_ab.log)
    if (plan_one_round(a) && !(mi355_debug_flags_get() & (1 << 27))) return MI355_EINVAL;
    if (!a.ypool) a.pool_mode = 0;
    const int pm = a.pool_mode;
    if (!a.y && !pm) return MI355_EINVAL;
    if (pm && (a.stride != 1 || a.res || a.pool_w < a.n || (pm == 2 && ((a.OH | a.OW) & 1)))) return MI355_EINVAL;
    if (pm) {  // the pool pass addresses the pooled tensor with 24-bit cell indices and 32-bit byte offsets
        const long pcells = (long)a.pool_lead + (long)a.B * (a.OH / (pm == 2 ? 2 : 1) + 1) * (a.OW / (pm == 2 ? 2 : 1) + 1);
        if (pcells >= (1L << 24) || a.pool_cs >= (1 << 24) || pcells * a.pool_cs >= (1L << 32)) return MI355_EINVAL;
    }
    if ((a.stride != 1 && a.stride != 2) || a.up != 1 || (a.y && a.out_w < a.n)) return MI355_EINVAL;
    if (a.stride == 2 && ((a.H & 1) || (a.W & 1))) return MI355_EINVAL;  // even maps: output = the even positions
    const int kp = c / 128, nq = ws3_quads(a.n, c), pieces = 8 * kp;
    const int mtiles = a.n / (32 * nq), nset = 8 / (nq * kp);
    const int S = a.stride, OHd = a.OH, OWd = a.OW, hw = OHd * OWd;  // pixels enumerate the output map (see the kernel)
    const int OWp = OWd / 2, ohwp = (OHd / 2) * OWp;
    const long total = pm == 2 ? (long)a.B * ohwp : a.total_n;  // tile units: 2x2 windows with the fused stride-2 pool, else pixels
    const int upx = pm == 2 ? 4 : 1;                             // pixels per unit
    const int want = 256 / mtiles > 0 ? 256 / mtiles : 1;  // workgroups per filter tile: one round of the chip
    // LDS need of a plan with tiles of tp pixels (0: does not fit); fills the geometry fields of `a`
    auto plan = [&](int tp, size_t &lds_out, bool db = false) {
        const int ntiles = (int)((total + tp - 1) / tp);
        const int G = (tp * upx + 31) / 32;
        int rows_cap = 0;
        for (int t = 0; t < ntiles; ++t) {
            const long p0 = (long)t * tp, p1 = (p0 + tp < total ? p0 + tp : total) - 1;
            int b0, r0, b1, r1;
            if (pm == 2) {
                b0 = (int)(p0 / ohwp); r0 = 2 * (int)((p0 - (long)b0 * ohwp) / OWp);
                b1 = (int)(p1 / ohwp); r1 = 2 * (int)((p1 - (long)b1 * ohwp) / OWp) + 1;
            } else {
                b0 = (int)(p0 / hw); r0 = (int)((p0 - (long)b0 * hw) / OWd);
                b1 = (int)(p1 / hw); r1 = (int)((p1 - (long)b1 * hw) / OWd);
            }
            const int nrows = b1 * (a.H + 1) + S * r1 - b0 * (a.H + 1) - S * r0 + 3;
            if (nrows > rows_cap) rows_cap = nrows;
        }
        if (S == 1 && pm != 2 && tp == hw && total % hw == 0 && rows_cap == a.H + 2) rows_cap = a.H + 1;  // whole-image tiles: the two pad rows alias
        const int cells = rows_cap * (a.W + 2);
        size_t lds = (size_t)cells * (pieces + 1) * 16;
        // db: two image buffers, each a whole number of KiB (the LDS-DMA writes 1 KiB per instruction), the next tile's image lands in one
        // while the other is computed on
        const size_t imgstride = db ? ((lds + 1023) & ~(size_t)1023) : 0;
        if (db) lds = 2 * imgstride;
        const size_t imgb = lds;
        lds += (size_t)((cells + 3) & ~3) * 4 + (size_t)G * 32 * 12 + (pm == 2 ? (size_t)G * 8 * 4 : 0);
        lds = (lds + 15) & ~(size_t)15;
        const size_t poff = lds;
        lds += (size_t)32 * nq * 16;
        const size_t roff = lds;
        if (kp == 2) lds += (size_t)nq * nset * ((G + nset - 1) / nset) * 4096;
        // staged bytes of the fused pool: the consumed partial-sum slots when there are K parts, else slots of their own
        a.sm_pt_off = (int)roff;
        a.sm_pt_stride = 4096;
        if (pm && kp == 1) {
            a.sm_pt_off = (int)lds;
            a.sm_pt_stride = 32 * 36 + 128;  // + the per-quad bank offset (3 x 32 B), rounded up to a multiple of 128 B
            lds += (size_t)nq * nset * ((G + nset - 1) / nset) * (32 * 36 + 128);
        }
        if (lds > 160 * 1024) return false;
        a.sm_tp = tp;
        a.ntiles_n = ntiles;
        a.sm_ncell = a.W + 2;
        a.rows_cap = rows_cap;
        a.sm_pieceb = (int)imgb;
        a.sm_hc = (int)imgstride;  // (a conv_small field: here the distance of the two image buffers, 0 = one buffer, staged through registers)
        a.lds_param_off = (int)poff;
        a.sm_red_off = (int)roff;
        lds_out = lds;
        return true;
    };
    size_t lds = 0;
    int nwg = 0;
    const int tp1 = (int)((total + want - 1) / want);  // one tile per workgroup
    if (pm == 1) {  // the stride-1 pool needs whole-image tiles: one per workgroup, or several per (persistent) workgroup
        if (hw < 64 || hw > WS3_GMAX * 32 || a.B * mtiles < 128 || !plan(hw, lds)) return MI355_EINVAL;  // (half the chip idle: not worth it)
        nwg = a.ntiles_n < want ? a.ntiles_n : want;
    } else if (tp1 * upx < 64) {
        return MI355_EINVAL;                           // tiny batches: the row-image kernel is the better fit
    } else if (tp1 * upx <= WS3_GMAX * 32 && plan(tp1, lds)) {
        nwg = a.ntiles_n;
    } else if (pm) {
        return MI355_EINVAL;  // the fused pools exist for one tile per workgroup
    } else {
        // Several tiles per workgroup cost a staging pass and four barriers per tile: measured 94 us against the row-image
        // kernel's 68 us on 256->256 @52x52 (BASELINE config[1]), but 70 / 74 us against 104 / 103 us where that kernel
        // cannot tile the map well (128->256 @76x76: wider than its 62-pixel row image; 256->512 @38x38: 40 cells in
        // a 64-slot row).  Only those maps take this path.
        const int rs = a.W + 2 <= 16 ? 16 : (a.W + 2 <= 32 ? 32 : 64);
        if (S == 1 && a.W <= 62 && (a.W + 2) >= 0.7 * rs) return MI355_EINVAL;  // (stride 2 has no row-image kernel)
        // several tiles per (persistent) workgroup: the largest tile that fits LDS, the tile count rounded up to whole
        // rounds of the `want` workgroups and the tile size shrunk to match, so that every workgroup walks as many tiles
        bool ok = false;
        // the 128-channel form first tries tiles of which TWO images fit (double-buffered DMA staging, see the kernel): smaller tiles, more of
        // them per workgroup, each fetched under the previous one's compute
        if (kp == 1 && !pm && !(mi355_debug_flags_get() & (1 << 29)) && (size_t)a.in_cells * (size_t)a.in_cs < ((size_t)1 << 32)) {
            for (int tpm = WS3_GMAX * 32; tpm >= 96 && !ok; tpm -= 32) {
                const long per = (total + (long)want * tpm - 1) / ((long)want * tpm);
                const int tp = (int)((total + per * want - 1) / (per * want));
                ok = tp >= 64 && per >= 2 && plan(tp, lds, true);
            }
            for (int k = (WS3_GMAX * 32) / OWd; k >= 1 && !ok; --k)
                if (OHd % k == 0 && k * OWd >= 64 && (long)a.B * (OHd / k) >= 2L * want) ok = plan(k * OWd, lds, true);
        }
        for (int tpm = WS3_GMAX * 32; tpm >= 64 && !ok; tpm -= 32) {
            const long per = (total + (long)want * tpm - 1) / ((long)want * tpm);  // tiles per workgroup
            const int tp = (int)((total + per * want - 1) / (per * want));
            ok = tp >= 48 && plan(tp, lds);
        }
        // wide rows (stride 2 reads two input rows per output row): tiles of whole output rows that never straddle an
        // image need the fewest LDS rows
        for (int k = (WS3_GMAX * 32) / OWd; k >= 1 && !ok; --k)
            if (OHd % k == 0 && k * OWd >= 32) ok = plan(k * OWd, lds);
        if (!ok) return MI355_EINVAL;
        nwg = a.ntiles_n < want ? a.ntiles_n : want;
    }
    a.debug = mi355_debug_flags_get();
    a.sm_nq = nq;
    a.mtiles = mtiles;
    a.fd_w = fastdiv_make((uint32_t)(a.W + 2));  // cells per LDS image row (the DMA staging's unit -> (row, column))
    return kp == 1 ? w3_launch_pm<1>(a, st, mtiles * nwg, lds) : w3_launch_pm<2>(a, st, mtiles * nwg, lds);
}
