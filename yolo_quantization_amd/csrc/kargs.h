// kargs.h -- kernel argument blocks and host-side launcher prototypes shared by the translation units.
#pragma once
#include "common.h"
#include <atomic>
#include <mutex>

struct ConvArgs {
    const int8_t *x;
    const int8_t *wp;
    const int32_t *cw, *dzp, *bias;
    const double *mval, *sval;
    uint8_t *y;
    int32_t *acc_out;
    float *y_f32;
    int in_cs, in_lead, in_cells;
    int out_cs, out_lead;
    int out_w, pool_w;       // bytes of an output / pooled cell this layer may write (16-aligned channel count)
    int B, H, W, n;          // H, W: input map
    int stride, OH, OW;      // output map = (H + 2*pad - ksize) / stride + 1 (== H, W for the stride-1 kernels)
    int ksize, cb, nchunks, upc, spc, ksteps;
    int total_n, ntiles_n, mtiles;
    int xcd_mb;              // conv_rows: tiles are numbered in blocks of xcd_mb M tiles, N-major inside a block (1 = M-major): what one XCD's contiguous range covers
    // conv_rows: launch constants of the index arithmetic (divisions by them are multiplications, common.h FastDiv)
    FastDiv fd_hw, fd_w, fd_ntn, fd_ntper, fd_nch, fd_mb;  // H * W, W, ntiles_n, xcd_mb * ntiles_n, channel chunks, xcd_mb
    int tile_q, tile_r;      // N tile t covers pixels [t q + min(t, r), + q + (t < r))
    int zp_act, act, store_mode;
    float s_act;
    int mpad;
    int bchunks, bpt;        // LDS B buffer size in KiB chunks; DMA instructions per wave per chunk load
    int tiles_x, tiles_y;    // PATCH mode tiling of one image
    int rows_cap;            // conv_rows: LDS rows per B buffer
    int rowb;                // conv_rows: bytes between LDS rows (row data + bank skew)
    const int32_t *shift;    // per-channel right shift (valid when hdr->pow2)
    const double *mprime;    // M_value * shift_value
    const int32_t *cwb;      // cw + bias
    const ConvBlobHeader *hdr;
    uint8_t *ypool;          // fused 2x2/2 maxpool output (PHWC, biased) or null
    int pool_cs, pool_lead;
    int lds_param_off;       // conv_rows: byte offset of the staged per-channel epilogue parameters in LDS
    int debug;               // timing-ablation switches (results are wrong when non-zero): see mi355_debug_flags
    const int8_t *ws;        // conv_small: weights-stationary A fragments [m-tile][k-step][lane][16 B] or null
    int sm_ncell, sm_pieceb; // conv_small: slots (16 B per piece plane) per LDS image row = its pitch; bytes of one 16-channel piece plane
    int sm_lcell, sm_hc;     // conv_small: image cells per row (<= sm_ncell) and the slot of the row's first ODD cell: a row holds its even
                             // cells 0, 2, .. in slots 0 .., its odd cells in slots sm_hc .. (see conv_small.hip: conflict-free stride-2 reads)
    int sm_tp;               // conv_mid_pool: pooled pixels per tile
    int sm_nq, sm_red_off;   // conv_ws3: filter quads (waves) per K part and workgroup; LDS offset of the K-part partial sums
    int pool_mode;           // conv_ws3: 0 = none, 2 = fused 2x2 / stride-2 maxpool (ypool = the pooled map), 1 = fused 2x2 / stride-1
                             // maxpool (ypool = a map of the conv's own size; ref: pad = 1 -> window rows y..y+1, columns x..x+1, clipped)
    int sm_pt_off, sm_pt_stride;  // conv_ws3: LDS offset / slot stride of the staged output bytes the fused pool reads
    int up;                  // conv_rows: fused nearest-neighbour upsample factor of the stored tensor (1 = none)
    float *yolo_out;         // fused yolo head: activated copy of y_f32 (same layout) or null
    int yolo_per;            // classes + 5
    // fused quantized residual add (mi355_conv_shortcut_forward): the stored byte is shortcut(conv byte, res byte); res is
    // the `from` layer's tensor, same map as y.  res_delta = (res.lead - y.lead): res cell = y cell + res_delta
    const uint8_t *res;
    int res_cs, res_delta;
    int sc_ka, sc_kb, sc_k0;
    int plan;                // MI355_PLAN_*: throughput plan = prefer kernels of which two workgroups fit a CU
    const EptHeader *ept;    // the blob's epilogue table (common.h) or null
};

struct AuxArgs {
    const uint8_t *x;  // input tensor (cs==4: plain uint8 image cells; else biased PHWC)
    int in_cs, in_lead, in_cells;
    const uint32_t *wfirst;  // first-layer packing [n][9] dwords
    const uint8_t *w_u8;     // raw reference weights [n][c*k*k] (ref-f32 kernel)
    const uint8_t *zp_w;     // [n]
    const int32_t *dzp, *bias;
    const double *mval, *sval;
    uint8_t *y;
    int out_cs, out_lead;
    int32_t *acc_out;
    float *y_f32;
    int B, H, W, c, n, ksize, pad;
    int zp_in, zp_act, act, store_mode;
    float s_act;
    int total_n;
    const double *mprime;    // folded M_value * shift_value
    uint8_t *ypool;          // fused 2x2/2 maxpool output or null
    int pool_cs, pool_lead;
    const ConvBlobHeader *hdr;  // device copy of the blob header (data-dependent pow2 flag)
    const int32_t *cwb;         // cw + bias (first-layer MFMA kernel)
    int planar;                 // x is the reference's [B][3][H][W] uint8 planes (no cells, no pads), read in place
    int debug_flags;            // mi355_debug_flags (2048: plain tile walk instead of the XCD-aware one, A/B runs)
    const EptHeader *ept;       // the blob's epilogue table (common.h) or null
    FastDiv fd_tpi, fd_tx;      // first-layer MFMA kernels: divisions by tiles per image / tiles per row
};

struct PoolArgs {
    const uint8_t *x;
    uint8_t *y;
    int B, H, W, OH, OW, cs_in, cs_out, in_lead, out_lead, groups;  // groups = C/16
    int size, stride, offset;                                       // offset = -pad/2
};

struct CopyArgs {
    const uint8_t *x;
    uint8_t *y;
    int B, H, W, OH, OW, cs_in, cs_out, in_lead, out_lead, groups, stride, coff;  // coff: channel offset in y
};

struct DequantArgs {
    const uint8_t *t;
    float *out;
    int B, H, W, cs, lead, c0, nc, out_C, out_c0, zp;
    float scale;
};

struct ShortcutArgs {
    const uint8_t *a, *b;
    uint8_t *y;
    int B, H, W, groups;
    int a_cs, b_cs, y_cs, a_lead, b_lead, y_lead;
    int ka, kb, k0;
};

struct LayoutArgs {
    uint8_t *nchw;
    uint8_t *t;
    int B, H, W, C, cs, lead;
};

int yolo_detections_launch(const float *out, int B, int n, int classes, int h, int w, const float *biases, const int *mask,
                           int netw, int neth, int imw, int imh, float thresh, int relative, float *recs, int max_recs,
                           int *counts, hipStream_t st);
int conv_igemm_launch(ConvArgs &a, hipStream_t st);
int conv_rows_launch(ConvArgs &a, hipStream_t st, int bm, int bn);
int conv_small_pool_launch(ConvArgs &a, hipStream_t st);
// conv_small.hip's weights-stationary plane: A row R of a 32-filter m-tile holds filter ws_row_filter(R), so that the accumulator rows of
// one lane (8 grp + 4 kh + r in the 32 x 32 MFMA's D layout) are sixteen consecutive filters 16 kh + 4 grp + r
__host__ __device__ constexpr int ws_row_filter(int R) { return 16 * ((R >> 2) & 1) + 4 * (R >> 3) + (R & 3); }
bool conv_small_eligible(int n, int c, int ksize);
bool conv_pool16_eligible(int n, int c, int ksize);
int conv_pool16_launch(ConvArgs &a, hipStream_t st);  // conv_pool16.hip: c 16 | 32 + maxpool on 16 x 16 x 64 tiles (needs the blob's epilogue table)
bool conv_small32_eligible(int n, int c, int ksize);
int conv_small32_launch(ConvArgs &a, hipStream_t st);  // conv_small32.hip: c 32 -> n 64 + maxpool, eight waves per workgroup at four per SIMD (round 6)
int conv1x1_ws_launch(ConvArgs &a, hipStream_t st);
bool conv1x1_ws_eligible(int n, int c, int ksize);
int conv_ws3_launch(ConvArgs &a, hipStream_t st);
bool conv_ws3_eligible(int n, int c, int ksize);
int mi355_debug_flags_get();
// Throughput plan (mi355_conv_desc.plan): does this launch fit ONE round of whole-CU workgroups (128 x 384 output tiles, one
// per CU)?  Only such launches are re-planned into half-CU workgroups that share a CU with another batch's layer; a launch of
// several rounds keeps the whole-chip kernels, which are the more efficient ones and have their own tail to overlap
// (YOLOv3-608, batch 32: 128 -> 256 @76 83 -> 116 us, 256 -> 512 @38 84 -> 113 us when re-planned; profiles/r03_v3_plan_layers.log).
static inline bool plan_one_round(const ConvArgs &a)
{
    // ... and only maps that fill the row image of the 128-column kernel well (19-wide maps in 32-slot rows do not: YOLOv3's
    // 512 -> 1024 @19 layers 82 -> 100 us alone, 4.58 -> 4.64 ms per step with three batches in flight)
    const int rs = a.W + 2 <= 16 ? 16 : (a.W + 2 <= 32 ? 32 : 64);
    return a.plan == MI355_PLAN_THROUGHPUT && (long)((a.mpad + 127) / 128) * ((a.total_n + 383) / 384) <= 256 && 10 * (a.W + 2) >= 7 * rs;
}
int conv_first_launch(AuxArgs &a, hipStream_t st);
int conv_first_pool_launch(AuxArgs &a, hipStream_t st);
int conv_first_mfma_pool_launch(AuxArgs &a, hipStream_t st);
int conv_first_mfma_launch(AuxArgs &a, hipStream_t st);
int conv_ref_f32_launch(AuxArgs &a, hipStream_t st);
int maxpool_launch(const PoolArgs &a, hipStream_t st);
int copy_cells_launch(const CopyArgs &a, hipStream_t st);
int copy_cell_bytes_launch(const CopyArgs &a, int nbytes, hipStream_t st);
int nchw_to_phwc_launch(const LayoutArgs &a, hipStream_t st);
int phwc_to_nchw_launch(const LayoutArgs &a, hipStream_t st);
int dequant_cells_launch(const DequantArgs &a, hipStream_t st);
int shortcut_launch(const ShortcutArgs &a, hipStream_t st);
int fill_u32_launch(uint32_t *p, uint32_t v, long n, hipStream_t st);
int letterbox_launch(const float *im, int imw, int imh, int c, float *out, int w, int h, hipStream_t st);
int image_minmax_launch(const float *x, long count, uint32_t *mm, hipStream_t st);
int image_quantize_launch(const float *x, long count, float scale, int zp, uint8_t *out, hipStream_t st);
int yolo_logistic_launch(const float *in, float *out, int B, int n, int classes, int hw, hipStream_t st);
int checksum_u32_launch(const uint32_t *p, long n, unsigned long long *out, hipStream_t st);

// Kernels with more than 64 KB of dynamic LDS need their limit raised: per kernel instantiation (the template argument) and per device --
// function attributes are per device, `darknet -gpus` drives several from one process -- and only when a launch needs more than the
// limit set so far, not once per launch.  (The limit counts against 160 KB TOGETHER with the kernel's static LDS: ask for what is needed.)
template <void (*kern)(const ConvArgs)>
static inline bool lds_limit_for(size_t lds)
{
    static size_t have[64] = {0};
    static std::mutex mu;  // host threads that drive replicas of one device may first-launch the same instantiation at the same time:
                           // the check, the attribute call and the record are one critical section (only taken when the limit has to grow)
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (lds <= 64 * 1024) return true;
    std::lock_guard<std::mutex> lk(mu);
    size_t &h = have[dev & 63];
    if (lds > h) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return false;
        h = lds;
    }
    return true;
}
template <void (*kern)(const ConvArgs)>
static inline int launch_big_lds(int grid, int threads, size_t lds, hipStream_t st, const ConvArgs &a)
{
    if (!lds_limit_for<kern>(lds)) return MI355_EHIP;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(threads), lds, st, a);
    return hipGetLastError() == hipSuccess ? MI355_OK : MI355_EHIP;
}
