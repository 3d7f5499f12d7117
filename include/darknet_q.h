/*
 * darknet_q.h -- plain-C host side of the MI355X INT8 path (libdarknet_q.so, ./darknet).
 *
 * Mirrors the reference's host interface for the quantized inference path only, with the same names, argument
 * meaning and error behaviour (die via error(), ref: src/utils.c:232-237), so a darknet user finds what they know:
 *
 *   struct layer { forward, forward_gpu, ... quant fields }   ref: include/darknet.h:154-226, 471-491
 *   struct network                                            ref: include/darknet.h:562-639
 *   parse_network_cfg / load_weights / load_network           ref: src/parser.c:682-815, 1201-1305; src/network.c:49
 *   set_batch_network                                         ref: src/network.c:383-397
 *   quantization_weights_and_activations                      ref: src/blas.c:259-346
 *   forward_network / forward_network_gpu / network_predict   ref: src/network.c:229-261, 835-861, 570-581
 *
 * What differs, on purpose:
 *   * `forward` (the CPU function pointer) is not a compute path here: it is set to a function that dies with a
 *     message.  This build has exactly one data path, `forward_gpu` -> libmi355yolo.so (HIP, gfx950).
 *   * the network carries a device uint8 activation pointer (the reference's GPU executor only threads floats,
 *     ref: src/network.c:835-861) and batch > 1 means "the batch-1 function applied per image" (the reference's
 *     epilogue handles image 0 only, ref: src/convolutional_layer.c:726-751).
 *   * quantization_weights_and_activations() is idempotent (the reference's accumulates weights_sum_int and
 *     re-folds batch-norm on every call, ref: src/blas.c:285-286,309 -- valid for the first call only, which is the
 *     behaviour reproduced).
 */
#ifndef DARKNET_Q_H
#define DARKNET_Q_H
#include <stddef.h>
#include <stdint.h>
#include "mi355_yolo_int8.h"

#ifdef __cplusplus
extern "C" {
#endif

/* same enumerator values as the reference (include/darknet.h:87-89, 99-130) */
typedef enum { LOGISTIC = 0, RELU = 1, LINEAR = 3, RELU6 = 8, LEAKY = 9 } ACTIVATION;
typedef enum { CONVOLUTIONAL = 0, MAXPOOL = 3, ROUTE = 8, SHORTCUT = 13, YOLO = 23, UPSAMPLE = 26 } LAYER_TYPE;

#define QUANT_POSITIVE_LIMIT 255
#define QUANT_NEGATIVE_LIMIT 0

struct network;
typedef struct network network;
struct layer;
typedef struct layer layer;

struct layer {
    LAYER_TYPE type;
    ACTIVATION activation;
    void (*forward)(struct layer, struct network);     /* dies: no CPU data path in this build */
    void (*forward_gpu)(struct layer, struct network); /* HIP path */

    int batch, h, w, c, out_h, out_w, out_c;
    int n, size, stride, pad, groups;
    int inputs, outputs, nweights, batch_normalize;
    int count; /* layer index */

    /* quantization fields, same names as the reference */
    int close_quantization, layer_quant_flag, quant_stop_flag, fisrt_time_train_fag;
    float *input_data_uint8_scales, *activ_data_uint8_scales, *weight_data_uint8_scales;
    uint8_t *input_data_uint8_zero_point, *activ_data_uint8_zero_point, *weight_data_uint8_zero_point;
    int32_t *weights_sum_int;
    uint32_t *mult_zero_point;
    float *M;
    int32_t *M0;
    int *M0_right_shift;
    double *M_value, *M0_right_shift_value;
    uint8_t *weights_uint8;
    int32_t *biases_int32;
    float *biases, *scales, *rolling_mean, *rolling_variance;

    /* route */
    int *input_layers, *input_sizes;
    /* shortcut (quantized residual add, builder-specified: mi355_shortcut_forward): `index` = the layer added to the
     * previous layer's output (ref field name, src/shortcut_layer.c:28), the two 16.16 multipliers of the prep */
    int index;
    int32_t shortcut_Ka, shortcut_Kb;
    /* yolo */
    int classes, total;
    int *mask;
    float *anchors; /* the reference calls this l.biases for yolo layers */
    float *anchors_gpu, *det_recs_gpu; /* on-device box decode (network_yolo_detections_gpu): anchors, records, counts */
    int *mask_gpu, *det_counts_gpu, det_cap;

    /* host mirrors of the outputs, reference layout (filled by pull_layer_output) */
    float *output;               /* [batch][outputs] float (quant_stop convs, yolo) */
    int32_t *output_int32;       /* [batch][outputs] pre-requant accumulators (conv) */
    uint8_t *output_uint8_final; /* [batch][outputs] */

    /* device side */
    mi355_tensor out_t;            /* PHWC uint8 activations */
    int out_view;                  /* out_t.data is a channel window of a later route layer's buffer (not owned) */
    int route_elided;              /* route: every input already writes into out_t (see plan_views) -- no copy at run time */
    void *blob_gpu;                /* packed weights + per-channel params (mi355_conv_pack) */
    int blob_shared;               /* 1: blob_gpu belongs to the network this one is a replica of (network_replica): never freed / re-uploaded here */
    size_t blob_bytes;
    void *blob_host;
    uint8_t *weights_uint8_gpu;    /* raw weights / zero points for the ref-f32 verification mode */
    uint8_t *weight_zero_point_gpu;
    int32_t *output_int32_gpu;     /* reference layout; allocated only when net->dump_int32 */
    float *output_gpu;             /* reference layout float (quant_stop convs, yolo) */
    uint8_t *output_uint8_nchw_gpu; /* scratch for pull_layer_output */
    int fuse_next_pool; /* this conv and the 2x2 maxpool (stride 2, or stride 1 behind a 128 / 256-channel conv) after it run as one kernel (set by the prep) */
    int fuse_pool_keep; /* ... which also stores the conv's own tensor: a route reads it */
    int fuse_next_upsample; /* this conv stores its pixels straight into the upsample layer's tensor after it */
    int conv_kernel;    /* kernel family that served this conv's last forward (mi355_last_conv_kernel) */
    int fuse_next_yolo; /* this quant_stop head conv also writes the activations of the yolo layer after it */
    int fuse_next_shortcut; /* this conv's epilogue also does the quantized residual add of the [shortcut] after it */
    int prepared;
};

struct network {
    int n, batch;
    layer *layers;
    int h, w, c, inputs, outputs;
    size_t *seen;
    float *input;         /* host float CHW image(s) (layer-0 quantiser input) */
    uint8_t *input_uint8; /* host [batch][c][h][w] */
    float *output;
    int train, index;
    int close_quantization;

    /* device side */
    int gpu_index;
    void *stream;
    uint8_t *input_uint8_gpu; /* reference layout on the device */
    float *quant_mm_gpu;      /* device scratch of the layer-0 quantiser: max, min of the float image */
    float *input_gpu;         /* batch x inputs floats on the device: the letterboxed images of the device input path */
    mi355_tensor input_t;     /* cs==4 image tensor */
    mi355_tensor input_nchw_t; /* input_uint8_gpu described as it is (planar): layer 0 reads it in place where its kernel can */
    int *input_direct_p;      /* executor -> layer 0: where to clear input_direct (the network travels by value) */
    int input_direct;         /* 1: layer 0 is fed input_nchw_t, no conversion pass; cleared by the first MI355_EINVAL */
    int input_direct_off;     /* user knob (dnq_net_set "input_direct" 0): never feed the planes directly, whatever is re-allocated */
    const mi355_tensor *cur_t; /* uint8 hand-off: the reference's `net.input_uint8 = l.output_uint8_final` */
    const float *cur_f32_gpu;  /* float hand-off: `net.input = l.output` */

    /* knobs (CLI: -accum exact|ref-f32, -parity wrap|saturate) */
    int accum_mode, store_mode;
    int dump_int32; /* keep int32 accumulators of every conv (parity runs) */
    int keep_head_float; /* 0 (default): a head conv fused with its yolo layer does not store its own float tensor (l.output: an
                            intermediate only the yolo layer reads); 1: it does (per-layer parity dumps) */
    int fuse_maxpool; /* 1 (default): conv + following 2x2/2 maxpool fused, the pre-pool tensor is not stored.
                         0: every layer writes its own tensor like the reference (per-layer parity dumps) */
    const mi355_tensor *fused_up_t;   /* run-time: upsampled tensor the conv being run has to fill, or NULL */
    int fused_up_stride;
    float *fused_yolo_out;            /* run-time: yolo output buffer the conv being run has to fill, or NULL */
    int fused_yolo_classes;
    const mi355_tensor *fused_pool_t; /* executor -> conv forward_gpu: pooled output tensor of the fused pair */
    const struct layer *fused_shortcut; /* executor -> conv forward_gpu: the [shortcut] layer whose add this conv performs, or NULL */
    int verbose;
    int prepared;
    int range_lo, range_hi; /* diagnostic (tools/layer_flood.py): forward_network_gpu runs layers [range_lo, range_hi) only, on the tensors the
                               last full pass left behind; 0, 0 = the whole network */
    int on_default_stream; /* this executor launches on the device's default (NULL) stream and owns no stream of its own */
    int replica_default_stream; /* set before network_replica(): the NEXT replica runs on the default stream.  HIP keeps one of the
                                   device's four hardware queues for that stream and maps every created stream onto the other three:
                                   a fourth batch in flight only runs beside the other three from there (bench: 0.2719 -> 0.2671 ms
                                   per step; a fourth created stream shares a queue with the first: 0.2719 -> 0.30).  Only for hosts
                                   that put nothing else on the default stream. */
    int plan;             /* MI355_PLAN_*: handed to every conv launch; network_replica switches parent and replica to the throughput plan
                             (change it with network_set_plan, which also re-derives the fusion plan and drops a captured graph) */
    int plan_user;        /* the plan the caller asked for (network_set_plan); a parent returns to it when its last replica is freed */
    int has_host_weights; /* load_weights ran: raw weights_uint8 / biases / scales of every layer are on the host */
    int has_l0_weights;   /* imported from a packed exchange: blobs only, plus layer 0's raw record (re-prep on a new input scale) */
    char *cfg_path;       /* the cfg this network was parsed from (network_replica parses it again) */
    struct network *replica_of; /* non-NULL: a replica (network_replica): packed weights on the device are the parent's */
    int n_replicas;             /* live replicas of this network: it cannot be freed, re-batched or re-prepared while > 0 */
    uint64_t *selfcheck_gpu; /* [passes] checksums of the pending self-check */
    int selfcheck_passes;
    void *graph; /* hipGraph of the layer loop, built lazily when use_graph */
    int use_graph;
    /* per-layer HIP-event timing on net->stream (replaces the commented what_time_is_it_now() probes of
     * ref: src/network.c:244-246) */
    void **prof_ev;      /* [prof_cap][n+2] events */
    int prof_cap, prof_used;
    int prof_stride, prof_calls; /* events are recorded on every prof_stride-th forward (they cost ~2.4 us per layer) ... */
    int prof_phase;              /* ... the ones with call index % prof_stride == prof_phase */
};

/* ---- construction / IO ------------------------------------------------------------------------------------ */
network *parse_network_cfg(char *filename, int close_quantization);
void load_weights(network *net, char *filename);
network *load_network(char *cfg, char *weights, int clear);
void set_batch_network(network *net, int b);
void free_network(network *net);

/* ---- host prep (M0/shift/biases_int32, packing, upload, device buffers) ------------------------------------ */
void quant_multi_smaller_than_one_to_scale_and_shift(float real_multiplier, int32_t *quantized_multiplier,
                                                     int *right_shift);
void quant_image_with_min_max(int count, const float *input, uint8_t *out, float *scale, uint8_t *zero_point);
void quantization_weights_and_activations(network *net);
/* same, but with the layer-0 input scale / zero point given instead of derived from net->input (serving mode:
 * uint8 images arrive already quantised) */
void quantization_weights_and_activations_fixed_input(network *net, float in_scale, uint8_t in_zp);
/* the layer-0 quantiser (ref: src/blas.c:279 -> :108-168) on float images already in HBM: device min / max + quantise,
 * layer 0 re-derived only when scale / zero point change */
void quantization_weights_and_activations_gpu(network *net, const float *input_gpu);
/* Input path on the device: letterbox_image (ref: src/image.c:812-831) of a planar float image in HBM
 * into batch slot `slot` of the network's float input, then the layer-0 quantiser over the whole batch. */
void network_letterbox_input_gpu(network *net, int slot, const float *im_gpu, int imw, int imh);
void network_quantize_input_gpu(network *net);
/* the host half of the prep only (per-channel integers + packed blobs, no device needed) */
void quantization_prep_host(network *net, float in_scale, uint8_t in_zp);

/* Batches in flight (serving throughput): a replica is a second executor of the SAME prepared model on the same device --
 * its own activation tensors, network input and HIP stream, the parent's packed weights in HBM (read-only in every kernel).
 * forward_network_gpu on parent and replicas in turn puts independent batches on separate streams, so that the device
 * overlaps one batch's launch gaps, pipeline fills and VALU-bound layers with another batch's MFMA-bound layers
 * (measured: DESIGN.md 4.3).  The parent must be prepared and must outlive its replicas; a replica cannot re-derive
 * layer 0 for another input scale, serve MI355_ACC_REF_F32 or be re-batched (error()). */
network *network_replica(network *parent);
/* the same with the stream choice as a parameter instead of the one-shot field: default_stream != 0 -> the replica launches on the
 * device's default (NULL) stream -- HIP's fourth hardware queue -- and owns no stream of its own */
network *network_replica_ex(network *parent, int default_stream);
/* MI355_PLAN_LATENCY / MI355_PLAN_THROUGHPUT for every conv launch of this executor from the next pass on.  Waits for the executor's
 * stream, drops a captured hipGraph (it holds the other plan's kernels) and re-derives the conv + pool / upsample / yolo / residual
 * fusion flags: a launcher that declined a fused call under one plan (flag cleared at run time) is asked again under the other. */
void network_set_plan(network *net, int plan);

/* ---- execution ----------------------------------------------------------------------------------------------- */
void forward_network(network *net);     /* == forward_network_gpu; dies if no device */
void forward_network_gpu(network *net); /* input: net->input_uint8_gpu already holds [batch][c][h][w] uint8 */
float *network_predict(network *net, float *input);
/* upload net->input_uint8 (host) -> device */
void push_network_input_uint8(network *net, const uint8_t *host_nchw);
/* device -> host mirrors of layer i in the reference layout */
void pull_layer_output(network *net, int i);
/* get_yolo_detections + correct_yolo_boxes (ref: src/yolo_layer.c:246-277,316-345, as called by get_network_boxes,
 * src/network.c:583-640) of yolo layer i for the whole batch on the device; only the detections are copied back.
 * recs: [batch][max_recs][6 + classes] floats {rank, x, y, w, h, objectness, prob[classes]} sorted by rank (the order of
 * the reference's loop), counts: [batch] detections found (records beyond max_recs are dropped). */
void network_yolo_detections_gpu(network *net, int i, int imw, int imh, float thresh, int relative, float *recs,
                                 int max_recs, int *counts);

/* ---- detections (what `detector test` does after network_predict; ref: include/darknet.h:658-669) -------------------- */
typedef struct { float x, y, w, h; } box;
typedef struct detection {
    box bbox;
    int classes;
    float *prob;
    float *mask;
    float objectness;
    int sort_class;
} detection;
/* ref: src/network.c:583-640 -- yolo layers in network order, image 0 of the batch (get_network_boxes_batch: image b) */
detection *get_network_boxes(network *net, int w, int h, float thresh, float hier, int *map, int relative, int *num);
detection *get_network_boxes_batch(network *net, int b, int w, int h, float thresh, float hier, int *map, int relative, int *num);
void free_detections(detection *dets, int n);
void do_nms_sort(detection *dets, int total, int classes, float thresh); /* ref: src/box.c:58-89 */
void do_nms_sort_arrays(const float *boxes, float *probs, const float *objectness, int n, int classes, float thresh);
float box_iou(box a, box b);
char *data_cfg_find(const char *datacfg, const char *key); /* `key = value` of a .data file, or NULL */
char **get_labels(char *filename, int *count);

/* per-layer profiling: record HIP events around every layer for the next `max_steps` forward passes (eager
 * launches only), then read the per-layer sums in ms: out[0] = input layout conversion, out[1+i] = layer i. */
void network_profile_begin(network *net, int max_steps);
void network_profile_set_stride(network *net, int stride);
/* Determinism self-check (a race detector for the hand-scheduled kernels: counted waits, in-place register reuse): `passes` forward
 * passes over the input that is on the device, a device-side checksum of every yolo layer's output after each; nothing is synchronised.
 * network_selfcheck_result() waits, reads the checksums back and returns the number of passes that differ from the first (0 = all
 * identical, -1 = no self-check pending). */
void network_selfcheck(network *net, int passes);
int network_selfcheck_result(network *net);
void network_profile_set_phase(network *net, int phase);
int network_profile_read(network *net, float *ms_sum /* [n+1] */);

/* packed-weight exchange for multi-GPU start-up: rank 0 exports, the bytes travel by RCCL broadcast, the other
 * ranks import into a network parsed from the same cfg (no weights file needed there). */
size_t network_packed_size(network *net);
void network_export_packed(network *net, void *buf);
void network_import_packed(network *net, const void *buf, size_t bytes);
void network_import_packed_host(network *net, const void *buf, size_t bytes); /* host half only (no device) */
/* same exchange with the buffer already on the device (what bench.py hands over after torch.distributed.broadcast) */
void network_import_packed_gpu(network *net, const void *dev_buf, size_t bytes);
/* the whole start-up exchange behind the C ABI: root exports + uploads, one mi355_bcast_blob (RCCL over xGMI), the other
 * ranks import from HBM.  comm: mi355_comm_init handle of the calling rank. */
void network_bcast_packed(network *net, void *comm, int rank, int root);
/* the same bytes as a file: written once after load_weights + prep, read back with one fread */
void network_save_packed(network *net, char *filename);
void network_load_packed(network *net, char *filename);

/* misc */
void error(const char *s);
void file_error(const char *s);
double what_time_is_it_now(void);
const char *get_layer_string(LAYER_TYPE t);

#ifdef __cplusplus
}
#endif
#endif
