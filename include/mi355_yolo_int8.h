/*
 * mi355_yolo_int8.h -- C-ABI of libmi355yolo.so: the MI355X (gfx950) INT8 quantized-convolution inference
 * kernels for the darknet uint8-quantization fork ArtyZe/yolo_quantization.
 *
 * This is the drop-in boundary: plain C, raw device pointers and sizes, no HIP / torch types.  A darknet host
 * (ours: yolo_quantization_amd/host, or the reference itself -- see INTEGRATION.md) binds these from the
 * `layer.forward_gpu` function pointers (reference include/darknet.h:158-163).  Every entry point returns 0 on
 * success or a negative MI355_E* code; nothing aborts, nothing falls back to the CPU.
 *
 * Paths cited as `ref:` are relative to the reference repository root.
 *
 * Device activation layout ("PHWC"): a tensor of B images x H x W pixels x C channels is an array of *cells*
 * of `cs` bytes (cs >= C, cs % 16 == 0, or cs == 4 for the 3-channel network input):
 *
 *     cell(b, y, x) = lead + (b*(H+1) + (y+1))*(W+1) + x          0 <= y < H, 0 <= x < W
 *
 * Row 0 of every image block and column W of every row are *pad cells* shared between neighbours (the right pad
 * of row y is the left pad of row y+1; the bottom pad row of image b is the top pad row of image b+1), so a 3x3
 * tap is the constant cell offset dy*(W+1)+dx everywhere.  Pad cells hold the consumer's input zero point
 * (ref: src/convolutional_layer.c:703-705,715 -- im2col pads with the input zero point).  Bytes are stored
 * *biased*: stored = uint8 ^ 0x80, i.e. the signed value (uint8 - 128) that V_MFMA_I32_*_I8 consumes directly.
 */
#ifndef MI355_YOLO_INT8_H
#define MI355_YOLO_INT8_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- error codes ---------------------------------------------------------------------------------------- */
#define MI355_OK 0
#define MI355_EINVAL (-22)   /* bad argument / unsupported shape */
#define MI355_ENOMEM (-12)   /* device allocation failed */
#define MI355_EHIP (-5)      /* HIP runtime error: mi355_last_error() has the text */
#define MI355_ENODEV (-19)   /* no gfx950 device */

/* ACTIVATION enum values of the reference (ref: include/darknet.h:87-89) */
#define MI355_ACT_RELU 1
#define MI355_ACT_LINEAR 3
#define MI355_ACT_RELU6 8
#define MI355_ACT_LEAKY 9

/* uint8 store behaviour of the requantise epilogue */
#define MI355_STORE_WRAP 0      /* ref default path: stored before clamp -> wraps mod 256 (convolutional_layer.c:737-749) */
#define MI355_STORE_SATURATE 1  /* builder-defined: the default path's formulas with clamp(0,255) BEFORE the store.  Equals the MKL
                                   flavour's epilogue (convolutional_layer.c:572-596) for LEAKY / LINEAR / RELU6 on every int32
                                   (proved exhaustively, tests/test_host_cpu.py), differs for RELU (:591 adds no zero point);
                                   that flavour cannot be built here, so this mode is NOT pinned against a reference output */

/* accumulation semantics */
#define MI355_ACC_EXACT 0    /* exact int32 on V_MFMA_I32_*_I8: the integers the reference's formula defines; equals the default build
                                wherever its fp32 accumulation is exact (pinned: tests/test_gpu_refpin.py) and the MKL flavour's
                                accumulators: that flavour cannot be built here, but the MKL entry point it calls is in the image --
                                cblas_gemm_s16s16s32 with the argument lists of ref :557-569 returns exactly these integers beyond 2^24
                                as well (tests/test_mkl_pin.py: the GEMM half pinned against the real library, the epilogue not) */
#define MI355_ACC_REF_F32 1  /* bit-faithful emulation of ref src/gemm.c:279-299 (fp32 step-wise accumulate,
                                two passes); slow verification kernel, never on the throughput path */

/* ---- runtime (replaces ref src/cuda.c: cuda_set_device :9, cuda_make_array :90-104, cuda_push/pull_array
 *      :151-167, cuda_free :144-149, check_error :27-49) ---------------------------------------------------- */
/* ABI version of this header.  Structs that cross the boundary (mi355_conv_desc, mi355_tensor) carry no size field: a field is only ever
 * APPENDED, and every append bumps this number (6: mi355_conv_desc.epilogue_packed).  A caller compares MI355_ABI_VERSION (what it was built
 * against) with mi355_abi_version() (what it loaded) once at start-up -- the darknet host, integration/mi355_glue.c and the Python binding
 * all do -- so that a caller built against an older, shorter struct fails loudly instead of having the shim read past its end. */
#define MI355_ABI_VERSION 6
int mi355_abi_version(void);
int mi355_init(int device);                       /* select device, verify gfx950 */
const char *mi355_last_error(void);
int mi355_device_count(void);
int mi355_alloc(void **dptr, size_t bytes);
int mi355_free(void *dptr);
int mi355_memset(void *dptr, int byte, size_t bytes, void *stream);
int mi355_h2d(void *dst, const void *src, size_t bytes, void *stream);
int mi355_d2h(void *dst, const void *src, size_t bytes, void *stream);
int mi355_d2d(void *dst, const void *src, size_t bytes, void *stream);
int mi355_stream_create(void **stream);
int mi355_stream_destroy(void *stream);
/* A stream that was MEASURED to run side by side with the default stream and with every other acquired stream of the device
 * (HIP deals created streams round-robin onto the device's four hardware queues, the default stream owns one; streams on one
 * queue serialise, and which created stream shares whose queue depends on every stream the process created before).  The first
 * three acquisitions per device come from the measured pool, later ones are plain streams.  Release instead of destroy. */
int mi355_stream_acquire(void **stream);
int mi355_stream_release(void *stream);
int mi355_stream_sync(void *stream);              /* stream == NULL: device synchronize */
/* HIP events on the launch stream (bench.py times kernels with these, not with torch.cuda.Event) */
int mi355_event_create(void **ev);
int mi355_event_destroy(void *ev);
int mi355_event_record(void *ev, void *stream);
int mi355_event_elapsed_ms(void *start, void *stop, float *ms); /* synchronises on `stop` */
/* hipGraph capture of a layer sequence (launch-bound tail of the net) */
int mi355_graph_begin(void *stream);
int mi355_graph_end(void *stream, void **graph_exec);
int mi355_graph_launch(void *graph_exec, void *stream);
int mi355_graph_destroy(void *graph_exec);

/* ---- the path's one collective: one-shot RCCL broadcast of the packed weights over xGMI -----------------
 * The reference has no inference-time communication (its multi-GPU code is host-staged weight averaging for training,
 * ref: src/network.c:1100-1194); images shard by rank and every device holds a replica of the packed blobs.  librccl is
 * bound with dlopen at the first call (MI355_ENODEV when absent).  One communicator rank per device: call mi355_init(dev)
 * on the thread first.  id128: the 128-byte ncclUniqueId, created on one rank and handed to the others by the launcher
 * (a shared variable between host threads, the process launcher's store between processes). */
int mi355_comm_unique_id(void *id128);
int mi355_comm_init(void **comm, int nranks, const void *id128, int rank);
int mi355_bcast_blob(void *comm, void *dev_buf, size_t bytes, int root, void *stream); /* in place, asynchronous on `stream` */
int mi355_comm_destroy(void *comm);
const char *mi355_comm_last_error(void);

/* ---- tensors ----------------------------------------------------------------------------------------------- */
typedef struct mi355_tensor {
    void *data;   /* device pointer to cell 0 */
    int B, H, W;  /* images, rows, columns */
    int C;        /* logical channels */
    int cs;       /* bytes per cell */
    int lead;     /* pad cells in front of image 0 */
    int tail;     /* pad cells behind the last image */
} mi355_tensor;

/* Fill in cs/lead/tail for (B,H,W,C) and return the byte size of the buffer (0 on bad dims). data untouched. */
size_t mi355_tensor_describe(mi355_tensor *t, int B, int H, int W, int C);
/* The reference's own input layout as a tensor descriptor: [B][C][H][W] uint8 planes, plain bytes, no pad cells (cs = 1,
 * lead = tail = 0) -- `net.input_uint8` as it is (ref: src/network.c:248).  Accepted as `x` by mi355_conv_forward /
 * mi355_conv_pool_forward for the 3-channel first layer only (exact mode, 16 or 32 filters, even map, W % 4 == 0): the kernel
 * reads the three colour planes in place and pads with desc.zp_in, so the network input needs no layout conversion pass.
 * Other shapes return MI355_EINVAL: convert with mi355_nchw_to_tensor then.  Returns the byte size. */
size_t mi355_tensor_describe_nchw(mi355_tensor *t, int B, int H, int W, int C);
/* Set every byte of the buffer (pads included) to the biased zero point (zp ^ 0x80). */
int mi355_tensor_fill(const mi355_tensor *t, uint8_t zero_point, void *stream);
/* Reference layout <-> device layout.  nchw is the reference's `net.input_uint8` / `l.output_uint8_final`
 * layout [B][C][H][W] uint8 (ref: src/network.c:248-250), resident on the device. */
int mi355_nchw_to_tensor(const uint8_t *nchw, const mi355_tensor *t, void *stream);
int mi355_tensor_to_nchw(const mi355_tensor *t, uint8_t *nchw, void *stream);

/* ---- quantized convolution ----------------------------------------------------------------------------- */
/* Host-side packing of one conv layer's weights (replaces the per-forward operand prep of the reference: the
 * `zero_point_uint8` table ref: src/blas.c:290-300 and GEMM pass 2 ref: src/convolutional_layer.c:721 are
 * folded into per-channel constants).  weights_uint8: [n][c*k*k] in the reference's (ci,ky,kx) order
 * (ref: src/parser.c:1146), zp_w: [n].  Returns the packed blob size; writes it when `blob` != NULL.
 * The blob is position independent (offsets only) so it can be broadcast to other GPUs as bytes. */
size_t mi355_conv_pack_size(int n, int c, int ksize);
int mi355_conv_pack(int n, int c, int ksize, const uint8_t *weights_uint8, const uint8_t *zp_w,
                    const int32_t *biases_int32, const double *M_value, const double *shift_value, void *blob);
/* Optional second packing step, on the same HOST blob before it is uploaded: the per-channel constants of the requantise
 * epilogue (ref: src/convolutional_layer.c:726-751) that the fused conv + maxpool kernels need for ONE (activation, zero
 * point) -- the range of accumulators whose stored byte cannot wrap (:737-749, where max-pooling commutes with the
 * requantisation) and the integer form M0 / shift of M_value (ref: src/blas.c:387-418).  Without it (or when a launch's
 * activation / zp_act differ from the packed ones, or the store saturates) every workgroup derives them itself: same bytes,
 * a quarter to a third of the first layer's run time.  RELU is stored as LINEAR (the reference's integer
 * path treats them alike, :740-742). */
int mi355_conv_pack_epilogue(int n, int c, int ksize, int activation, int zp_act, void *blob);

typedef struct mi355_conv_desc {
    int n, c, ksize, stride, pad; /* filters, input channels, 1|3, 1|2, ksize/2 (stride 2: plain exact-mode convs, c % 16 == 0) */
    int activation;               /* MI355_ACT_* */
    int store_mode;               /* MI355_STORE_* */
    int accum_mode;               /* MI355_ACC_* */
    uint8_t zp_in, zp_act;        /* input / activation zero points */
    float s_act;                  /* activation scale (only for y_f32) */
    int plan;                     /* MI355_PLAN_*: which kernel / tile the launcher should prefer (results are identical) */
    int epilogue_packed;          /* 1: the caller finished `blob` with mi355_conv_pack_epilogue(activation, zp_act) of THIS desc.  A hint for the
                                   * kernel choice only (results are identical): kernels that live off the table (conv + maxpool with 16 / 32
                                   * input channels on 16 x 16 x 64 tiles) are picked when it is set; a blob that does not carry the promised
                                   * table still yields the right bytes, slowly (every window takes the reference's order) */
} mi355_conv_desc;
/* MI355_PLAN_LATENCY (0, default): one batch at a time -- every launch is sized to fill the whole chip on its own (one big
 * workgroup per CU: 128 x 384 row-image tiles, the weights-stationary 3x3 kernel with up to 160 KB of LDS).
 * MI355_PLAN_THROUGHPUT: several independent batches are in flight on separate streams (darknet_q.h network_replica) --
 * prefer kernels of which TWO workgroups fit a CU (<= 80 KB of LDS, 4 waves), so that a workgroup of another batch's
 * layer can move in next to it: a launch that needs an empty CU waits until every small workgroup of the other streams
 * has drained from one.  Measured in DESIGN.md 4.3. */
#define MI355_PLAN_LATENCY 0
#define MI355_PLAN_THROUGHPUT 1

/* forward_convolutional_layer_quant_inputi_outputi (ref: src/convolutional_layer.c:694-761) for a whole batch,
 * "apply the batch-1 function independently per image".
 *   x        input tensor (C == desc.c).  desc.c == 3 selects the first-layer kernel and x.cs must be 4.
 *   blob     device copy of the mi355_conv_pack blob
 *   w_u8     device copy of the raw weights_uint8 / zp_w (only read when accum_mode == MI355_ACC_REF_F32)
 *   y        output tensor (C == desc.n), nullable
 *   acc_out  nullable: pre-requant int32 accumulators, reference layout [B][n][H*W] == l.output_int32
 *   y_f32    nullable: quant_stop tail (ref :752-760), reference layout [B][n][H*W] float == l.output */
int mi355_conv_forward(const mi355_conv_desc *desc, const mi355_tensor *x, const void *blob, const uint8_t *w_u8,
                       const uint8_t *zp_w, const mi355_tensor *y, int32_t *acc_out, float *y_f32, void *stream);

/* The same convolution fused with the size-2 / stride-2 maxpool that follows it (ref: forward_maxpool_layer_quant,
 * src/maxpool_layer.c:109-172, window offset 0 on even maps): writes ypool (B, H/2, W/2, n) directly and, when y is
 * non-NULL, the pre-pool tensor as well.  3x3 convs on even H, W only, exact accumulation mode; returns MI355_EINVAL
 * otherwise and the caller runs the two layers separately.  Results are identical to conv_forward + maxpool_forward. */
int mi355_conv_pool_forward(const mi355_conv_desc *desc, const mi355_tensor *x, const void *blob, const mi355_tensor *y,
                            const mi355_tensor *ypool, void *stream);

/* A quant_stop head convolution fused with the yolo layer that follows it (ref: forward_yolo_layer, src/yolo_layer.c:132-146):
 * writes y (uint8), y_f32 (== l.output of the conv, ref :752-760) and yolo_out (== l.output of the yolo layer: logistic on
 * x, y, objectness and class scores of every anchor) from one kernel.  desc.n must be a multiple of classes + 5; exact
 * accumulation mode.  Results are identical to mi355_conv_forward + mi355_yolo_forward.  y_f32 may be NULL (the conv's own float
 * tensor is an intermediate only the yolo layer reads): the 1x1 heads then write yolo_out alone -- MI355_EINVAL where the kernel
 * that serves the shape cannot (pass the buffer then). */
int mi355_conv_yolo_forward(const mi355_conv_desc *desc, const mi355_tensor *x, const void *blob, const mi355_tensor *y,
                            float *y_f32, float *yolo_out, int classes, void *stream);

/* A convolution fused with the nearest-neighbour upsample layer that follows it (ref: forward_upsample_layer_quant,
 * src/upsample_layer.c:96-113): y_up is the (stride*H) x (stride*W) tensor, every output pixel is stored stride x stride
 * times; the conv's own tensor is not stored.  Layers with c % 64 == 0, exact mode.  Identical to conv_forward + upsample_forward. */
int mi355_conv_upsample_forward(const mi355_conv_desc *desc, const mi355_tensor *x, const void *blob, const mi355_tensor *y_up,
                                int stride, void *stream);

/* A convolution fused with the quantized residual add that follows it (`[shortcut] quantized=1`, mi355_shortcut_forward below):
 * every requantised byte a of the convolution (zero point desc.zp_act) is combined with the byte b of `from` at the same
 * pixel and channel, y_sum = clamp(zp_out + ((Ka*(a - zp_act) + Kb*(b - zp_from) + 2^15) >> 16)); the convolution's own
 * tensor is not stored.  Stride-1 exact-mode 3x3 layers with 128 / 256 input channels (conv_ws3.hip: 16 of YOLOv3's 23 residual
 * blocks); MI355_EINVAL otherwise and the caller runs the two layers separately (measured faster for the other kernels).  Bytes are identical to
 * mi355_conv_forward + mi355_shortcut_forward. */
int mi355_conv_shortcut_forward(const mi355_conv_desc *desc, const mi355_tensor *x, const void *blob, const mi355_tensor *from,
                                const mi355_tensor *y_sum, int32_t Ka, int32_t Kb, uint8_t zp_from, uint8_t zp_out, void *stream);

/* Tile configuration override for benchmarking (0 = auto). */
int mi355_conv_set_tile(int bm, int bn);
/* Development switches.  Bits 0..8 are timing ablations of the K loop (no DMA / no s_barrier / no MFMA / ...), compiled
 * in only with -DMI355_ABLATE: results are WRONG when set; tools/conv_microbench.py --ablate only.  Two bits select
 * among equivalent kernels and leave results unchanged (tests use them to cross-check): 512 = conv_rows.hip walks the
 * channel chunks unrotated, 1024 = fused conv+maxpool never uses conv_small.hip / the first-layer MFMA kernel, 8192 = 1x1
 * layers never use conv1x1.hip, 4096 = 32 -> 64 conv + maxpool on conv_small32.hip (round 6: eight waves per workgroup, measured slower; conv_small.hip without the bit), 16384 = 3x3 layers never use conv_ws3.hip, 2^20 = the row-image 3x3 kernel on the 32x32x32 MFMA
 * (conv_rows.hip) instead of the 16x16x64 one (conv_rows16.hip), 2^21 = mi355_conv_pool_forward refuses the 128 / 256-channel
 * layers (the host then runs conv and maxpool separately), 2048 = plain tile walk in the first-layer kernels, 131072 = no raised
 * priority around the MFMA chains of the conv+pool kernels. */
int mi355_debug_flags(int flags);
/* Which kernel family served the calling thread's most recent mi355_conv_*forward call (tests assert that a shape reaches
 * the kernel it is meant to exercise): 0 none yet, 1 first layer (conv_aux.hip), 2 conv_small.hip (few-channel / 64-channel
 * conv + maxpool), 3 conv1x1.hip, 4 conv_ws3.hip, 5 conv_igemm.hip / conv_rows.hip / conv_rows16.hip, 6 fp32-accumulate emulation,
 * 7 conv_pool16.hip (16 -> 32 + maxpool with the packed epilogue table), 8 conv_small32.hip (32 -> 64 + maxpool). */
int mi355_last_conv_kernel(void);

/* ---- glue layers ---------------------------------------------------------------------------------------- */
/* forward_maxpool_layer_quant (ref: src/maxpool_layer.c:109-172): window offset -pad/2, OOB taps = uint8 0 */
int mi355_maxpool_forward(const mi355_tensor *x, const mi355_tensor *y, int size, int stride, int pad, void *stream);
/* forward_upsample_layer_quant (ref: src/upsample_layer.c:96-113, src/blas.c:781-803), nearest x stride */
int mi355_upsample_forward(const mi355_tensor *x, const mi355_tensor *y, int stride, void *stream);
/* forward_route_layer_quant (ref: src/route_layer.c:107-130): channel concat of n inputs, no rescale */
int mi355_route_forward(const mi355_tensor *const *xs, int n, const mi355_tensor *y, void *stream);
/* quant_stop tail of a glue layer (ref: src/maxpool_layer.c:163-171, src/upsample_layer.c:104-112, src/route_layer.c:121-129):
 * out_f32[b][out_c0 + c][pix] = (int)(x[b][pix][c0 + c] - zero_point) * scale for c < nc; out_f32 is the reference's
 * `l.output` layout [B][out_C][H*W].  A route with quant_stop calls it once per input with that input's own scale / zero
 * point (ref :125: `net.layers[index].activ_data_uint8_*`). */
int mi355_dequant_forward(const mi355_tensor *x, int c0, int nc, uint8_t zero_point, float scale, float *out_f32, int out_C,
                          int out_c0, void *stream);
/* Quantized residual add, `[shortcut] quantized=1`.  The reference has NO integer shortcut (src/shortcut_layer.c:62-75 and
 * src/blas.c:490-514 are float only), so this op is builder-specified (parity status: unpinned, the
 * oracle is oracle.c:orc_shortcut_u8):
 *     K = round((float)(s_in / s_out) * 2^16)                          mi355_shortcut_multiplier, 1 <= K < 2^21
 *     q = zp_out + ((Ka*(a - zp_a) + Kb*(b - zp_b) + 2^15) >> 16)      arithmetic shift: round half up
 *     y = clamp(q, 0, 255)                                             linear activation only
 * a = the previous layer's tensor, b = the tensor of layer `from`; same batch / map / channels (YOLOv3's residual blocks). */
int mi355_shortcut_multiplier(float s_in, float s_out, int32_t *K);
int mi355_shortcut_forward(const mi355_tensor *a, const mi355_tensor *b, const mi355_tensor *y, int32_t Ka, int32_t Kb,
                           uint8_t zp_a, uint8_t zp_b, uint8_t zp_out, void *stream);
/* letterbox_image (ref: src/image.c:812-831; bilinear resize_image :1199-1242, embed_image :428-439, fill 0.5): planar
 * float image [c][imh][imw] in device memory -> [c][h][w], aspect ratio kept, centred.  Bit-identical floats. */
int mi355_letterbox_forward(const float *im_f32, int imw, int imh, int c, float *out_f32, int w, int h, void *stream);
/* Layer-0 input quantiser (ref: quant_weights_with_min_max_channel with one channel, src/blas.c:108-168, called on the
 * float image at src/blas.c:279), both halves on device memory.  mi355_image_minmax: minmax[0] = max(x, 0.0f),
 * minmax[1] = min(x, 0.0f) (-0.0f when no element is negative) over `count` floats, the reference's seeds and
 * comparisons (NaNs are skipped).  mi355_image_quantize: out[k] = clamp((int)(float)(round((double)(x[k] / scale)) +
 * (double)zero_point), 0, 255), the reference's evaluation order (:160-165). */
int mi355_image_minmax(const float *x_f32, long count, float *minmax, void *stream);
int mi355_image_quantize(const float *x_f32, long count, float scale, int zero_point, uint8_t *out_u8, void *stream);
/* *sum_dev += an order-independent 64-bit checksum of `dwords` 32-bit words at buf (device pointers; zero *sum_dev first).  The
 * host's determinism self-check compares it between passes over the same input (network_selfcheck, darknet_q.h). */
int mi355_checksum_u32(const void *buf, long dwords, uint64_t *sum_dev, void *stream);
/* yolo head activations (ref: src/yolo_layer.c:132-146) on the float head tensor [B][n*(classes+5)][H*W] */
int mi355_yolo_forward(const float *in, float *out, int B, int n, int classes, int H, int W, void *stream);

/* get_yolo_detections + correct_yolo_boxes (ref: src/yolo_layer.c:246-277, 316-345) for a batch, on the device: only
 * detections cross PCIe.  yolo_out: the yolo layer's output [B][n*(classes+5)][H*W]; anchors [2*num] and mask [n] on the
 * device.  Every (cell, anchor) whose objectness exceeds thresh yields a record of 6 + classes floats
 *   {rank = cell * n + anchor, x, y, w, h, objectness, prob[classes] (objectness * class score if > thresh, else 0)}
 * in recs[b][slot] (slot order is arbitrary: sort by rank for the reference's order); counts[b] = number of detections of
 * image b (may exceed max_recs, the surplus is dropped).  Box centre, objectness and scores equal the reference's bits;
 * width / height agree to a few ulp (the reference is built with -Ofast, its exp() is not reproducible). */
int mi355_yolo_detections(const float *yolo_out, int B, int n, int classes, int H, int W, const float *anchors,
                          const int *mask, int netw, int neth, int imw, int imh, float thresh, int relative, float *recs,
                          int max_recs, int *counts, void *stream);

#ifdef __cplusplus
}
#endif
#endif
