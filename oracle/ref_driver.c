/*
 * ref_driver.c -- TEST INFRASTRUCTURE ONLY (never linked into the product).
 *
 * A thin C-ABI driver around the *unmodified* reference sources (compiled in place from
 * /root/reference by oracle/build_ref.sh into oracle/_ref/).  It exists so that tests and the
 * golden-vector generator can execute the reference's own functions:
 *
 *   load_network                                   src/network.c:49
 *   quantization_weights_and_activations           src/blas.c:259-346
 *   layer.forward (function pointer)               include/darknet.h:158
 *   forward_network's uint8 hand-off               src/network.c:229-261
 *   gemm_nn_uint8_int32_te                         src/gemm.c:279-299
 *   im2col_cpu_uint8                               src/im2col.c:26-50
 *   quant_multi_smaller_than_one_to_scale_and_shift src/blas.c:387-418
 *
 * Everything in this file is our own code; it only *calls* the reference's public API.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
#include <unistd.h>
#include <fcntl.h>
#include "darknet.h"

/* not in the public header, but exported by the reference objects */
void gemm_nn_uint8_int32_te(int M, int N, int K, float ALPHA, uint8_t *A, int lda, uint8_t *B, int ldb,
                            int BETA, int32_t *C, int ldc);
void im2col_cpu_uint8(uint8_t *data_im, int channels, int height, int width, int ksize, int stride, int pad,
                      uint8_t *data_col, uint8_t return_data);
void quant_multi_smaller_than_one_to_scale_and_shift(float real_multiplier, int32_t *quantized_multiplier,
                                                     int *right_shift);
void quant_weights_with_min_max_channel(int size_channel, float *input, uint8_t *input_int8, int16_t *input_int16,
                                        int16_t *zero_point_int16, int size_feature, float *quantzation_scale,
                                        uint8_t *quantization_zero_point, int zp_flag);

/* The reference printf()s per layer; silence it around calls. */
static int g_saved_stdout = -1;
static void hush(void)
{
    fflush(stdout);
    g_saved_stdout = dup(1);
    int fd = open("/dev/null", O_WRONLY);
    dup2(fd, 1);
    close(fd);
}
static void unhush(void)
{
    fflush(stdout);
    if (g_saved_stdout >= 0) {
        dup2(g_saved_stdout, 1);
        close(g_saved_stdout);
        g_saved_stdout = -1;
    }
}

typedef struct {
    network *net;
    int prepared;
} refnet;

void *refdrv_load(const char *cfg, const char *weights)
{
    hush();
    refnet *r = calloc(1, sizeof(refnet));
    r->net = load_network((char *)cfg, (char *)weights, 0);
    set_batch_network(r->net, 1);
    unhush();
    return r;
}

int refdrv_nlayers(void *h) { return ((refnet *)h)->net->n; }

/* info[0..11] = type, out_c, out_h, out_w, c, h, w, n(filters), size, stride, pad, activation,
 * info[12..15] = batch_normalize, layer_quant_flag, quant_stop_flag, outputs */
int refdrv_layer_info(void *h, int i, int *info)
{
    network *net = ((refnet *)h)->net;
    if (i < 0 || i >= net->n) return -1;
    layer *l = &net->layers[i];
    info[0] = l->type; info[1] = l->out_c; info[2] = l->out_h; info[3] = l->out_w;
    info[4] = l->c; info[5] = l->h; info[6] = l->w; info[7] = l->n;
    info[8] = l->size; info[9] = l->stride; info[10] = l->pad; info[11] = l->activation;
    info[12] = l->batch_normalize; info[13] = l->layer_quant_flag; info[14] = l->quant_stop_flag;
    info[15] = l->outputs;
    return 0;
}

/* Run the reference's host prep exactly once (it is not idempotent: src/blas.c:309 accumulates
 * weights_sum_int and 285-286 re-folds batch-norm on every call).  X = float CHW image. */
int refdrv_prepare(void *h, float *X)
{
    refnet *r = h;
    if (r->prepared) return -1;
    r->net->input = X;
    hush();
    quantization_weights_and_activations(r->net);
    unhush();
    r->prepared = 1;
    return 0;
}

/* Overwrite the quantised network input (after refdrv_prepare fixed the layer-0 scale / zero point). */
int refdrv_set_input_u8(void *h, const uint8_t *x)
{
    refnet *r = h;
    memcpy(r->net->input_uint8, x, (size_t)r->net->inputs);
    return 0;
}
const uint8_t *refdrv_input_u8(void *h) { return ((refnet *)h)->net->input_uint8; }

/* The layer loop of forward_network (src/network.c:238-259) with the reference's own function
 * pointers; identical hand-off rule.  Returns wall seconds spent per layer in t[] (nullable). */
int refdrv_forward(void *h, double *t)
{
    refnet *r = h;
    if (!r->prepared) return -1;
    network net = *r->net;
    net.train = 0;
    hush();
    for (int i = 0; i < net.n; ++i) {
        net.index = i;
        layer l = net.layers[i];
        double t0 = what_time_is_it_now();
        l.forward(l, net);
        if (t) t[i] = what_time_is_it_now() - t0;
        if (l.layer_quant_flag && !net.train) {
            net.input_uint8 = l.output_uint8_final;
            net.input = l.output;
        } else {
            net.input = l.output;
        }
        /* the reference leaks the im2col workspace every call (src/convolutional_layer.c:702);
           nothing to free here: net is a by-value copy and the pointer is lost, as upstream. */
    }
    unhush();
    return 0;
}

/* Whole-net predict through the reference's own entry point (for timing "Predicted in"). */
double refdrv_network_predict(void *h, float *X)
{
    refnet *r = h;
    hush();
    double t0 = what_time_is_it_now();
    network_predict(r->net, X);
    double dt = what_time_is_it_now() - t0;
    unhush();
    return dt;
}

const int32_t *refdrv_layer_int32(void *h, int i) { return ((refnet *)h)->net->layers[i].output_int32; }
const uint8_t *refdrv_layer_u8(void *h, int i) { return ((refnet *)h)->net->layers[i].output_uint8_final; }
const float *refdrv_layer_f32(void *h, int i) { return ((refnet *)h)->net->layers[i].output; }

/* Host-prep products of conv layer i (arrays of l.n entries each, caller-allocated; any may be NULL). */
int refdrv_layer_prep(void *h, int i, int32_t *biases_int32, double *M_value, double *shift_value, int32_t *M0,
                      int *shift, float *in_scale_zp_act_scale_zp /* [4]: s_in, zp_in, s_act, zp_act */)
{
    network *net = ((refnet *)h)->net;
    layer *l = &net->layers[i];
    if (in_scale_zp_act_scale_zp && l->activ_data_uint8_scales) {
        in_scale_zp_act_scale_zp[2] = l->activ_data_uint8_scales[0];
        in_scale_zp_act_scale_zp[3] = l->activ_data_uint8_zero_point[0];
        if (l->type == CONVOLUTIONAL) {
            in_scale_zp_act_scale_zp[0] = l->input_data_uint8_scales[0];
            in_scale_zp_act_scale_zp[1] = l->input_data_uint8_zero_point[0];
        }
    }
    if (l->type != CONVOLUTIONAL) return 1;
    for (int k = 0; k < l->n; ++k) {
        if (biases_int32) biases_int32[k] = l->biases_int32[k];
        if (M_value) M_value[k] = l->M_value[k];
        if (shift_value) shift_value[k] = l->M0_right_shift_value[k];
        if (M0) M0[k] = l->M0[k];
        if (shift) shift[k] = l->M0_right_shift[k];
    }
    return 0;
}

/* ---- single-function entry points (known-answer vectors for the restatement) ---- */
void refdrv_gemm_u8(int M, int N, int K, float ALPHA, uint8_t *A, int lda, uint8_t *B, int ldb, int BETA,
                    int32_t *C, int ldc)
{
    gemm_nn_uint8_int32_te(M, N, K, ALPHA, A, lda, B, ldb, BETA, C, ldc);
}
void refdrv_im2col_u8(uint8_t *im, int c, int h, int w, int k, int stride, int pad, uint8_t *col, uint8_t padv)
{
    im2col_cpu_uint8(im, c, h, w, k, stride, pad, col, padv);
}
void refdrv_quant_multiplier(float m, int32_t *M0, int *shift)
{
    quant_multi_smaller_than_one_to_scale_and_shift(m, M0, shift);
}
void refdrv_quantize_image(float *x, int n, uint8_t *out, float *scale, uint8_t *zp)
{
    int16_t *tmp16 = calloc(n, sizeof(int16_t));
    int16_t zp16 = 0;
    hush();
    quant_weights_with_min_max_channel(1, x, out, tmp16, &zp16, n, scale, zp, 0);
    unhush();
    free(tmp16);
}

/* letterbox_image of the reference itself (src/image.c:812-831) on a planar float image */
void refdrv_letterbox(float *im, int imw, int imh, int c, int w, int h, float *out)
{
    image src;
    src.w = imw; src.h = imh; src.c = c; src.data = im;
    image boxed = letterbox_image(src, w, h);
    memcpy(out, boxed.data, sizeof(float) * (size_t)w * h * c);
    free_image(boxed);
}

/* load_image_color of the reference itself (ref: src/image.c load_image_color -> stb_image, examples/detector.c:903): decoded
 * planar float image [c][h][w] in 0..1.  Returns a malloc'd buffer (refdrv_free_floats) and the dimensions; NULL on failure. */
float *refdrv_load_image_color(const char *path, int *w, int *h, int *c)
{
    hush();
    image im = load_image_color((char *)path, 0, 0);
    unhush();
    if (!im.data) return NULL;
    *w = im.w; *h = im.h; *c = im.c;
    return im.data;
}
void refdrv_free_floats(float *p) { free(p); }

/* get_yolo_detections of the reference itself (src/yolo_layer.c:316-345) on yolo layer i after a forward: records in
 * the layout of orc_yolo_detections (oracle.c).  Also hands out the layer's anchors / mask for the restatement. */
int refdrv_yolo_detections(void *h, int i, int imw, int imh, float thresh, int relative, float *recs, int max_recs)
{
    refnet *r = h;
    layer l = r->net->layers[i];
    if (l.type != YOLO) return -1;
    const int cand = l.w * l.h * l.n, rl = 6 + l.classes;
    detection *dets = calloc(cand, sizeof(detection));
    for (int k = 0; k < cand; ++k) dets[k].prob = calloc(l.classes, sizeof(float));
    const int count = get_yolo_detections(l, imw, imh, r->net->w, r->net->h, thresh, 0, relative, dets);
    /* the rank of a record in the reference's loop: recount the objectness test in the same order */
    int c = 0;
    for (int p = 0; p < l.w * l.h && c < count; ++p)
        for (int n = 0; n < l.n && c < count; ++n) {
            const float obj = l.output[n * l.w * l.h * (4 + l.classes + 1) + 4 * l.w * l.h + p];
            if (obj <= thresh) continue;
            if (c < max_recs) {
                float *o = recs + (size_t)c * rl;
                o[0] = (float)(p * l.n + n);
                o[1] = dets[c].bbox.x; o[2] = dets[c].bbox.y; o[3] = dets[c].bbox.w; o[4] = dets[c].bbox.h;
                o[5] = dets[c].objectness;
                for (int j = 0; j < l.classes; ++j) o[6 + j] = dets[c].prob[j];
            }
            ++c;
        }
    for (int k = 0; k < cand; ++k) free(dets[k].prob);
    free(dets);
    return count;
}
int refdrv_yolo_params(void *h, int i, float *biases, int *mask, int *total)
{
    layer l = ((refnet *)h)->net->layers[i];
    if (l.type != YOLO) return -1;
    for (int k = 0; k < 2 * l.total; ++k) biases[k] = l.biases[k];
    for (int k = 0; k < l.n; ++k) mask[k] = l.mask[k];
    *total = l.total;
    return l.n;
}

/* The quantised 0.1 the reference derives for LEAKY layers (src/blas.c:318-323: M0_lut0, M0_right_shift_lut0); only the
 * MKL forward reads it (src/convolutional_layer.c:583), but the default build computes it too. */
void refdrv_leaky_lut(void *h, int i, int32_t *M0_lut0, int *shift_lut0)
{
    layer *l = &((refnet *)h)->net->layers[i];
    *M0_lut0 = l->M0_lut0;
    *shift_lut0 = l->M0_right_shift_lut0;
}

/* One layer of a prepared network on a caller-supplied uint8 input (function-level known-answer vectors: the layer's own
 * forward pointer, src/convolutional_layer.c:258-273, with net.input_uint8 pointing at `input`). */
int refdrv_forward_layer(void *h, int i, uint8_t *input)
{
    refnet *r = h;
    if (!r->prepared || i < 0 || i >= r->net->n) return -1;
    network net = *r->net;
    net.train = 0;
    net.index = i;
    net.input_uint8 = input;
    layer l = net.layers[i];
    hush();
    l.forward(l, net);
    unhush();
    return 0;
}

#ifdef MI355
/* ---- the reference with its layer.forward_gpu pointers bound to libmi355yolo.so (integration/mi355_glue.c) ----------
 * The network is the REFERENCE's: its parser, weights loader, host prep and `struct layer`; only the forward_gpu function
 * pointers (include/darknet.h:161) are ours.  After refdrv_forward_mi355 the layers' output_uint8_final / output_int32 /
 * output hold what the device computed, so the same accessors serve both paths. */
#include "mi355_glue.h"
int refdrv_mi355_bind(void *h, int gpu, int accum_mode, int store_mode)
{
    refnet *r = h;
    if (!r->prepared) return -1;
    mi355_bind_network(r->net, gpu, accum_mode, store_mode);
    return 0;
}
int refdrv_forward_mi355(void *h, int pull_all)
{
    refnet *r = h;
    forward_network_mi355(r->net, pull_all);
    return 0;
}
void refdrv_mi355_unbind(void *h) { mi355_unbind_network(((refnet *)h)->net); }
#endif

/* do_nms_sort of the reference itself (src/box.c:58-89) on flat arrays: boxes [n][4], probs [n][classes] (in/out),
 * objectness [n].  qsort reorders the detections; each carries its row id (in the unused `classes` field) so the
 * result lands back in its own row. */
void refdrv_nms_sort(const float *boxes, float *probs, const float *objectness, int n, int classes, float thresh)
{
    detection *dets = calloc(n > 0 ? n : 1, sizeof(detection));
    for (int i = 0; i < n; ++i) {
        dets[i].bbox.x = boxes[4 * i]; dets[i].bbox.y = boxes[4 * i + 1]; dets[i].bbox.w = boxes[4 * i + 2]; dets[i].bbox.h = boxes[4 * i + 3];
        dets[i].objectness = objectness[i];
        dets[i].classes = i;
        dets[i].prob = calloc(classes, sizeof(float));
        memcpy(dets[i].prob, probs + (size_t)i * classes, sizeof(float) * classes);
    }
    do_nms_sort(dets, n, classes, thresh);
    for (int i = 0; i < n; ++i) {
        memcpy(probs + (size_t)dets[i].classes * classes, dets[i].prob, sizeof(float) * classes);
        free(dets[i].prob);
    }
    free(dets);
}
