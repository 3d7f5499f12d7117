// conv1x1.hip -- 1x1 INT8 convolution with the weights stationary in registers: the bottleneck / head layers of
// yolov3-tiny (1024->256, 512->30, 256->128, 256->30 at 13x13 / 26x26).  Through the row-image kernel these layers
// spent 12-21 us each on 1-6 GOP: a K loop of 4-16 barrier-synchronised steps behind 11 us of per-launch fixed cost.
// Here a workgroup of 8 waves owns one tile of consecutive pixels (sized so that the tiles fill whole rounds of the
// 256 CUs); the output channels are split in quads of 32 over the waves, a wave keeps the A fragments of its quad for
// the whole K (C / 32 K-steps, up to 128 VGPRs at C = 1024), `sets` such wave sets deal the tile's groups of 32 pixels
// between them.  The tile's input cells are DMAed once (global_load_lds) into LDS in the rows kernel's chunk layout
// [64-channel chunk][16 pixels x 4 pieces x 16 B]; there is no K loop, no barrier besides the one after the load, and
// the receptive field of a 1x1 tap is the pixel itself: sum(x') is the pixel's channel sum, taken once per tile by all waves together
// (one filter quad: by the wave from its B fragments).
// Same mathematics and the same bytes as conv_rows.hip (signed-operand decomposition: see conv_igemm.hip); optional
// float tail of a quant_stop head, fused yolo activations (byte -> logistic table) and nearest-neighbour upsample store.
#include "kargs.h"

#define DMA_S(ldsdst_u32, sbase_ptr, voff_u32)                                                                   \
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(ldsdst_u32), "v"(voff_u32), \
                 "s"(sbase_ptr)                                                                                  \
                 : "memory", "m0")

constexpr int P1_GMAX = 8;  // groups of 32 pixels per tile

#ifdef MI355_ABLATE
// phase timestamps (100 MHz wall clock) of wave 0 of every workgroup: tools/conv_microbench.py --timeline1 (-DMI355_ABLATE builds only)
__device__ long long g_c1_ts[4][8192];
#define TS1(k) do { if (threadIdx.x == 0 && blockIdx.x < 8192) g_c1_ts[k][blockIdx.x] = wall_clock64(); } while (0)
extern "C" int mi355_debug_read_ts1(long long *host)
{
    return hipMemcpyFromSymbol(host, HIP_SYMBOL(g_c1_ts), sizeof(long long) * 4 * 8192) == hipSuccess ? 0 : -5;
}
#else
#define TS1(k) do { } while (0)
#endif

// (HIP's second launch-bound is WAVES PER SIMD, not workgroups per CU: with (512, 2) the LEAKY instantiations took 131-138 registers,
// three waves per SIMD, i.e. ONE 8-wave workgroup per CU -- every multi-round launch (YOLOv3's 1x1 layers) ran its tile loads with
// nothing beside them.  Four waves per SIMD = two workgroups per CU wherever the stationary weights leave room: C <= 256.)
template <int KST, int ACT, bool SAT, bool COOP>  // COOP: the per-pixel channel sums are taken once per tile by all waves (more than one filter quad)
__global__ __launch_bounds__(512, (KST <= 8 ? 4 : 2)) void conv1x1_ws_kernel(const ConvArgs a)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // filter tiles: layers with more than 256 filters (YOLOv3's 1024 -> 512 necks) split them over a.mtiles workgroups per pixel
    // tile, 256 each (workgroup = tile * mtiles + filter tile: the filter tiles of a pixel tile run next to each other and share
    // its input lines in L2).  f0 = first filter of this workgroup, N32 = its filter count rounded up to a quad
    const int mtl = a.mtiles > 1 ? a.mtiles : 1;
    const int f0 = (int)(blockIdx.x % mtl) * 256;
    const int N32 = min(256, ((a.n + 31) & ~31) - f0);
    const int chunks = a.sm_ncell;               // 16-pixel chunks of the tile image (2 per group)
    const int qb = chunks * 1024;                // bytes of one 64-channel chunk plane
    double *ldsMP = reinterpret_cast<double *>(smem + a.lds_param_off);   // [N32] folded multiplier
    int *ldsDZ = reinterpret_cast<int *>(ldsMP + N32);                    // [N32] 128 - zp_w
    int *ldsCB = ldsDZ + N32;                                             // [N32] cw + bias
    float *ldsYL = reinterpret_cast<float *>(ldsCB + N32);                // [256] fused yolo head: logistic per byte
    // per pixel of the tile: output cell (-1: no pixel), image index and offset inside the image -- computed once per pixel by
    // the DMA loop below (which divides for its input cell anyway) instead of twice per group by every wave of the tile
    int *ldsCell = reinterpret_cast<int *>(ldsYL + 256);                  // [chunks * 16]
    int *ldsImg = ldsCell + chunks * 16, *ldsRem = ldsImg + chunks * 16;  // [chunks * 16] each
    int *ldsSX = ldsRem + chunks * 16;                                     // [chunks * 16] sum of x' over the pixel's channels (the 1x1 tap's receptive field)
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char *)smem;

    const int tid = threadIdx.x, NT = blockDim.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), nwave = NT >> 6;
    const int kh = lane >> 5, lj = lane & 31;
    const int nq = N32 >> 5;
    const int wq = wave % nq, wset = wave / nq, nset = nwave / nq;
    const int W1 = a.W + 1, hw = a.H * a.W;
    const int TP = a.sm_tp, G = (TP + 31) >> 5;
    const int tile = blockIdx.x / mtl;
    const int n0 = tile * TP, n1 = min(n0 + TP, a.total_n);  // this tile's pixels [n0, n1)
    const bool pow2 = a.hdr->pow2 == 1;
    TS1(0);

    // ---- tile image DMA: instruction (Q, chunk) = 16 pixels x 4 pieces of 64-channel chunk Q; lane -> (piece lane >> 4,
    //      pixel lane & 15).  The pixel -> cell division is done once per chunk and reused for every Q.
    {
        const int nck = (n1 - n0 + 15) >> 4;
        const int nQ = KST >> 1;
        for (int ck = wave; ck < chunks; ck += nwave) {  // chunks past the tile's last pixel only fill the tables (no pixel)
            const int n = min(n0 + ck * 16 + (lane & 15), n1 - 1);
            const int b = n / hw, rem = n - b * hw;
            const int y = rem / a.W, x = rem - y * a.W;
            const long cell = (long)a.in_lead + ((long)b * (a.H + 1) + (y + 1)) * W1 + x;
            const unsigned voff = (unsigned)(cell * a.in_cs) + (lane >> 4) * 16;
            if (lane < 16) {
                const int up = a.up;
                const long oc = up == 1 ? cell - a.in_lead + a.out_lead
                                        : (long)a.out_lead + ((long)b * (up * a.H + 1) + (up * y + 1)) * (up * a.W + 1) + up * x;
                const int pi = ck * 16 + lane;
                ldsCell[pi] = n0 + pi < n1 ? (int)oc : -1;  // output cells fit an int: the launcher checks
                ldsImg[pi] = b;
                ldsRem[pi] = rem;
                ldsSX[pi] = 0;
            }
            if (ck < nck)
                for (int Q = 0; Q < nQ; ++Q) {
                    const unsigned dst = lds0 + Q * qb + ck * 1024;
                    const unsigned v = voff + Q * 64;
                    DMA_S(dst, a.x, v);
                }
        }
    }
    TS1(1);
    // ---- parameters and this wave's A fragments (overlap the DMA)
    for (int i = tid; i < N32; i += NT) {
        ldsMP[i] = a.mprime[f0 + i];
        ldsDZ[i] = a.dzp[f0 + i];
        ldsCB[i] = a.cwb[f0 + i];
    }
    if (a.yolo_out)
        for (int i = tid; i < 256; i += NT) ldsYL[i] = yolo_entry_act((float)(i - a.zp_act) * a.s_act, 0);
    v4i wf[KST];
#pragma unroll
    for (int s = 0; s < KST; ++s) wf[s] = *reinterpret_cast<const v4i *>(a.ws + ((size_t)(((f0 >> 5) + wq) * KST + s) * 64 + lane) * 16);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    TS1(2);
    // ---- sum of x' per pixel, ONCE per tile by all waves together (round 5).  Every wave used to sum its B fragments with four V_DOT4 per
    //      K-step -- the same pixels in each of the workgroup's (up to eight) filter quads: 128 of a 1024 -> 256 wave's ~480 VALU instructions per
    //      group of 32 pixels, in a kernel whose VALU clocks are twice its MFMA clocks.  The tile image is a run of 16-byte slots
    //      [chunk Q][16-pixel chunk][piece][pixel]: a wave takes 64 consecutive slots = the four pieces of 16 pixels of one chunk.
    // (one quad -- the 30-filter heads: nobody to share with, the wave sums its own fragments as before: COOP = false)
    if constexpr (COOP) {
        const int nslot64 = (KST >> 1) * chunks;  // runs of 64 slots
        for (int q = wave; q < nslot64; q += nwave) {
            const v4i b = *reinterpret_cast<const v4i *>(smem + ((size_t)q * 64 + lane) * 16);
            int t = __builtin_amdgcn_sdot4(b[0], 0x01010101, 0, false);
            t = __builtin_amdgcn_sdot4(b[1], 0x01010101, t, false);
            t = __builtin_amdgcn_sdot4(b[2], 0x01010101, t, false);
            t = __builtin_amdgcn_sdot4(b[3], 0x01010101, t, false);
            t += __shfl_xor(t, 16);
            t += __shfl_xor(t, 32);
            if (lane < 16) atomicAdd(&ldsSX[(q % chunks) * 16 + lane], t);
        }
        __syncthreads();
    }

    const char *X = smem;
    const int chw = 32 * wq;
#pragma unroll 1
    for (int g = wset; g < G; g += nset) {
        // K-step s: channels 32 s + 16 kh .. + 15 = chunk s / 2, piece 2 (s % 2) + kh
        const int base = (2 * g + (lj >> 4)) * 1024 + (lj & 15) * 16 + kh * 256;
        v16i acc;
#pragma unroll
        for (int grp = 0; grp < 4; ++grp) {
            const int4 c4 = *reinterpret_cast<const int4 *>(ldsCB + chw + 16 * kh + 4 * grp);  // accumulator rows 8 grp + 4 kh + r hold filters 16 kh + 4 grp + r (kargs.h ws_row_filter)
            acc[grp * 4 + 0] = c4.x; acc[grp * 4 + 1] = c4.y; acc[grp * 4 + 2] = c4.z; acc[grp * 4 + 3] = c4.w;
        }
        int sx;
        if constexpr (COOP) {
#pragma unroll
            for (int s = 0; s < KST; ++s) {
                const v4i bf = *reinterpret_cast<const v4i *>(X + base + (s >> 1) * qb + (s & 1) * 512);
                acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(wf[s], bf, acc, 0, 0, 0);
                if ((s & 3) == 3) __builtin_amdgcn_sched_barrier(0);  // bound the number of B fragments in flight (registers)
            }
            sx = ldsSX[g * 32 + lj];
        } else {
            int sxr = 0;
#pragma unroll
            for (int s = 0; s < KST; ++s) {
                const v4i bf = *reinterpret_cast<const v4i *>(X + base + (s >> 1) * qb + (s & 1) * 512);
                sxr = __builtin_amdgcn_sdot4(bf[0], 0x01010101, sxr, false);
                sxr = __builtin_amdgcn_sdot4(bf[1], 0x01010101, sxr, false);
                sxr = __builtin_amdgcn_sdot4(bf[2], 0x01010101, sxr, false);
                sxr = __builtin_amdgcn_sdot4(bf[3], 0x01010101, sxr, false);
                acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(wf[s], bf, acc, 0, 0, 0);
                if ((s & 3) == 3) __builtin_amdgcn_sched_barrier(0);
            }
            sx = sxr + __shfl_xor(sxr, 32);  // the two 16-byte k-halves of every K-step
        }

        // ---- this lane's pixel (tables filled by the DMA loop)
        const int ocell = ldsCell[g * 32 + lj];
        const bool valid = ocell >= 0;
        const int b = ldsImg[g * 32 + lj], rem = ldsRem[g * 32 + lj];
        const int up = a.up;
        uint32_t pk[4];  // the lane's sixteen consecutive filters 16 kh .. + 15 of its quad, stored together
        // a group's constants are read while the group before it is requantised (conv_small.hip has the measurement: read where they are used,
        // every group waits for its LDS round trips with one other wave on the SIMD to cover them)
        struct GroupConst { int4 dz; double mp[4]; };
        auto group_const = [&](int grp) {
            const int ch0 = chw + 16 * kh + 4 * grp;
            GroupConst gc;
            gc.dz = *reinterpret_cast<const int4 *>(ldsDZ + ch0);
#pragma unroll
            for (int r = 0; r < 4; ++r) gc.mp[r] = ldsMP[ch0 + r];
            return gc;
        };
        constexpr bool AHEAD = KST == 32;  // (measured: c = 1024 -0.3 us in flight; c = 512 (the 30-filter head) +1 us; c <= 256 runs four waves per SIMD at 128 registers and would spill)
        GroupConst gnext = {};
        if constexpr (AHEAD) {
            gnext = group_const(0);
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int grp = 0; grp < 4; ++grp) {
            const int ch0 = chw + 16 * kh + 4 * grp;
            GroupConst gc = gnext;
            if constexpr (AHEAD) {
                if (grp < 3) gnext = group_const(grp + 1);
                __builtin_amdgcn_sched_barrier(0);
            } else {
                gc = group_const(grp);
            }
            const int dzv[4] = {gc.dz.x, gc.dz.y, gc.dz.z, gc.dz.w};
            int32_t accb[4][1], v[4][1];
            double mp[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                mp[r] = gc.mp[r];
                accb[r][0] = acc[grp * 4 + r] + __mul24(dzv[r], sx);
            }
            if (pow2) {
                requant_values<ACT, SAT, 1>(accb, mp, a.zp_act, v);
            } else {  // shift_value not a power of two: the reference's two-step form (never produced by its own prep)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    v[r][0] = (int32_t)requant_u8(accb[r][0], 0, a.mval[f0 + ch0 + r], a.sval[f0 + ch0 + r], a.zp_act, ACT,
                                                  SAT ? MI355_STORE_SATURATE : MI355_STORE_WRAP);
            }
            pk[grp] = pack4_biased(v[0][0], v[1][0], v[2][0], v[3][0]);
            // (out_w is a multiple of 16: the lane's 16-byte run lies inside the cell or outside it as a whole)
            if (grp == 3 && valid && f0 + chw + 16 * kh < a.out_w) {
                const uint4 packed = {pk[0], pk[1], pk[2], pk[3]};
                const size_t co = (size_t)(f0 + chw + 16 * kh);
#ifdef MI355_ABLATE
                if (a.debug & 1) {  // timing ablation: no stores (keep the value alive)
                    if (packed.x == 0x12345678u) a.y[0] = 1;
                } else
#endif
                if (up == 1) {
                    *reinterpret_cast<uint4 *>(a.y + (size_t)ocell * a.out_cs + co) = packed;
                } else {
                    const int rowc = up * a.W + 1;
                    for (int uy = 0; uy < up; ++uy)
                        for (int ux = 0; ux < up; ++ux)
                            *reinterpret_cast<uint4 *>(a.y + (size_t)(ocell + uy * rowc + ux) * a.out_cs + co) = packed;
                }
            }
            if (valid && f0 + ch0 < a.out_w) {
                if (a.y_f32 || a.yolo_out) {  // quant_stop tail (ref :752-760) and, fused, the yolo layer's activations (y_f32 may be
                                              // null then: the head's own float tensor is an intermediate nobody reads)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int oc = f0 + ch0 + r;
                        if (oc < a.n) {
                            const int u8 = v[r][0] & 0xFF;
                            const float f = (float)(u8 - a.zp_act) * a.s_act;
                            const size_t ridx = ((size_t)b * a.n + oc) * hw + rem;
                            if (a.y_f32) a.y_f32[ridx] = f;
                            if (a.yolo_out) {
                                const int e = oc % a.yolo_per;
                                a.yolo_out[ridx] = (e == 2 || e == 3) ? f : ldsYL[u8];
                            }
                        }
                    }
                }
            }
        }
    }
    TS1(3);
}

template <int KST, int ACT, bool SAT>
static int c1_launch_kern(ConvArgs &a, hipStream_t st, int grid, int threads, size_t lds)
{
    const int n32 = (a.n + 31) & ~31;
    if (n32 > 32) return launch_big_lds<conv1x1_ws_kernel<KST, ACT, SAT, true>>(grid, threads, lds, st, a);
    return launch_big_lds<conv1x1_ws_kernel<KST, ACT, SAT, false>>(grid, threads, lds, st, a);
}

template <int KST, int ACT>
static int c1_launch_sat(ConvArgs &a, hipStream_t st, int grid, int threads, size_t lds)
{
    if (a.store_mode == MI355_STORE_SATURATE) return c1_launch_kern<KST, ACT, true>(a, st, grid, threads, lds);
    return c1_launch_kern<KST, ACT, false>(a, st, grid, threads, lds);
}

template <int KST>
static int c1_launch_act(ConvArgs &a, hipStream_t st, int grid, int threads, size_t lds)
{
    if (a.act == MI355_ACT_LEAKY) return c1_launch_sat<KST, MI355_ACT_LEAKY>(a, st, grid, threads, lds);
    if (a.act == MI355_ACT_RELU6) return c1_launch_sat<KST, MI355_ACT_RELU6>(a, st, grid, threads, lds);
    return c1_launch_sat<KST, MI355_ACT_LINEAR>(a, st, grid, threads, lds);
}

// shapes whose blob carries the weights-stationary plane (off_ws) for this kernel
bool conv1x1_ws_eligible(int n, int c, int ksize)
{
    return ksize == 1 && (c == 64 || c == 128 || c == 256 || c == 512 || c == 1024) && n >= 1 && (n <= 256 || (n <= 1024 && n % 256 == 0));
}

// returns MI355_EINVAL when the shape is outside this kernel's domain (the caller falls back to conv_rows / conv_igemm)
int conv1x1_ws_launch(ConvArgs &a, hipStream_t st)
{
    if (a.res) return MI355_EINVAL;  // no fused residual add in this kernel
    const int c = a.cb * a.nchunks;
    if (!conv1x1_ws_eligible(a.n, c, a.ksize) || !a.ws || !a.y || a.acc_out || a.ypool || a.stride != 1) return MI355_EINVAL;
    if ((size_t)a.in_cells * (size_t)a.in_cs >= ((size_t)1 << 32)) return MI355_EINVAL;  // 32-bit DMA lane offsets
    a.debug = mi355_debug_flags_get();
    // tiles of equal size, as few rounds of 256 workgroups as the 256-pixel tile limit allows
    const long total = a.total_n;
    // few input channels: a pixel costs little LDS, and layers with millions of pixels (64 -> 32 at 304 x 304) are bound
    // by the per-workgroup latencies unless a workgroup streams a long tile
    const int gmax = c <= 64 ? 32 : (c == 128 ? 16 : P1_GMAX);
    const int n32 = (a.n + 31) & ~31;
    const int mtiles = n32 > 256 ? (n32 + 255) / 256 : 1;  // filter tiles of 256 (conv1x1_ws_eligible: n % 256 == 0 then)
    const int want = 256 / mtiles;                         // pixel tiles per round of the chip
    const long rounds = (total + (long)want * gmax * 32 - 1) / ((long)want * gmax * 32);
    int tp = (int)((total + want * rounds - 1) / (want * rounds));
    if (tp < 16) tp = 16;
    // throughput plan (batches in flight): whole groups of 32 pixels per workgroup instead of one round of the chip -- a tile of 43
    // pixels pays for two groups (LDS, MFMAs, the workgroup's copy of the A fragments) anyway; fewer, full workgroups cost the chip
    // less and the other batches' launches fill the CUs this one leaves free (layer 13: 252 -> 169 workgroups)
    if (a.plan == MI355_PLAN_THROUGHPUT && !(mi355_debug_flags_get() & (1 << 29))) tp = (tp + 31) & ~31;
    // ... and where a workgroup's copy of the A fragments (256 KB for 1024 -> 256) outweighs its pixels, more pixels per workgroup: chip
    // time per launch is (A load + pixels) x workgroups.  Layer 13 at 128 pixels: flood 9.8 -> 8.5 us per launch, in-flight step
    // -4 us (same box, profiles/r04_conv1x1_tp_ab_flood.log); alone it is slower (15 -> 23 us: 85 workgroups), which is the latency
    // plan's concern.  Layer 18 (256 -> 128: 32 KB of A) is better off at 64.
    if (a.plan == MI355_PLAN_THROUGHPUT && rounds == 1 && !(mi355_debug_flags_get() & (1 << 29))) {
        const long abytes = (long)(n32 < 256 ? n32 : 256) * c;
        while (abytes > 2L * tp * c && (long)(c / 64) * 2 * ((tp + 32) / 32) * 1024 <= 128 * 1024) tp += 32;
    }
    const int ntiles = (int)((total + tp - 1) / tp);
    const int G = (tp + 31) / 32;
    a.sm_tp = tp;
    a.sm_ncell = 2 * G;  // 16-pixel chunks
    size_t lds = (size_t)(c / 64) * a.sm_ncell * 1024;
    a.lds_param_off = (int)lds;
    const int nfw = n32 < 256 ? n32 : 256;  // filters of a workgroup
    lds += (size_t)nfw * 16 + 1024 + (size_t)a.sm_ncell * 16 * 16;  // parameters, logistic table, the four per-pixel tables
    // two workgroups per CU when a layer needs more than one round; a single round may take the whole LDS
    if (lds > (rounds == 1 && (long)ntiles * mtiles <= 256 ? 160 : 96) * 1024) return MI355_EINVAL;
    if (mtiles > 1 && (a.y_f32 || a.yolo_out || a.up != 1)) return MI355_EINVAL;  // (heads / fused upsample: single filter tile only)
    a.mtiles = mtiles;
    const int nq = nfw / 32;
    const int sets = 8 / nq > 0 ? 8 / nq : 1;  // at most 8 waves per workgroup
    const int threads = sets * nq * 64;
    switch (c) {
    case 64: return c1_launch_act<2>(a, st, ntiles * mtiles, threads, lds);
    case 128: return c1_launch_act<4>(a, st, ntiles * mtiles, threads, lds);
    case 256: return c1_launch_act<8>(a, st, ntiles * mtiles, threads, lds);
    case 512: return c1_launch_act<16>(a, st, ntiles * mtiles, threads, lds);
    default: return c1_launch_act<32>(a, st, ntiles * mtiles, threads, lds);
    }
}
