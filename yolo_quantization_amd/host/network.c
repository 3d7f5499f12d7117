/*
 * network.c -- host prep, device buffer management and the layer-loop executor.
 *
 *   quantization_weights_and_activations   ref: src/blas.c:259-346
 *   quant_multi_smaller_than_one_...       ref: src/blas.c:387-418
 *   quant_image_with_min_max               ref: src/blas.c:108-168 (quant_weights_with_min_max_channel, 1 channel)
 *   forward_network_gpu                    ref: src/network.c:835-861 with the uint8 hand-off of :248-250
 *   network_predict                        ref: src/network.c:570-581
 *   set_batch_network                      ref: src/network.c:383-397
 */
#include <assert.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/time.h>
#include "host_internal.h"

/* ------------------------------------------------------------------------------------------------------ utils */
void error(const char *s) /* ref: src/utils.c:232-237 */
{
    fprintf(stderr, "darknet_q: %s\n", s);
    fflush(stderr);
    exit(-1);
}
void file_error(const char *s)
{
    fprintf(stderr, "Couldn't open file: %s\n", s);
    exit(0);
}
void check_mi355(int rc, const char *what) /* ref: check_error, src/cuda.c:27-49 */
{
    if (rc != MI355_OK) {
        fprintf(stderr, "MI355 error %d in %s: %s\n", rc, what, mi355_last_error());
        error(what);
    }
}
double what_time_is_it_now(void)
{
    struct timeval time;
    if (gettimeofday(&time, NULL)) return 0;
    return (double)time.tv_sec + (double)time.tv_usec * .000001;
}
const char *get_layer_string(LAYER_TYPE t)
{
    switch (t) {
    case CONVOLUTIONAL: return "CONV";
    case MAXPOOL: return "MAX";
    case ROUTE: return "ROUTE";
    case UPSAMPLE: return "UPSAMPLE";
    case SHORTCUT: return "SHORTCUT";
    case YOLO: return "YOLO";
    }
    return "?";
}

static void plan_fusion(network *net);

/* ------------------------------------------------------------------------------------------------- host prep */
void quant_multi_smaller_than_one_to_scale_and_shift(float real_multiplier, int32_t *quantized_multiplier,
                                                     int *right_shift)
{
    if (!(real_multiplier > 0.f) || !(real_multiplier < 1.f)) { /* ref :391-392 asserts */
        fprintf(stderr, "requantisation multiplier %g is outside (0,1)\n", real_multiplier);
        error("quant_multi_smaller_than_one_to_scale_and_shift");
    }
    int s = 0;
    while (real_multiplier < 0.5f) {
        real_multiplier *= 2.0f;
        s++;
    }
    int64_t q = (int64_t)round((double)(real_multiplier * (float)(1ll << 31)));
    if (q == (1ll << 31)) {
        q /= 2;
        s--;
    }
    if (s < 0) error("quant_multi: negative shift");
    *quantized_multiplier = (int32_t)q;
    *right_shift = s;
}

/* scale / zero point of the layer-0 quantiser from the image's min / max (both seeded with 0.0f), ref: src/blas.c:125-150 */
static void image_scale_zero_point(float min_value, float max_value, float *scale, uint8_t *zero_point)
{
    if (min_value == 0 && max_value == 0) error("input image is all zero (ref: src/blas.c:125-128 assert)");
    /* the reference binary (gcc -Ofast) multiplies by the reciprocal constant here; see oracle/oracle.c */
    float nudged_scale = (max_value - min_value) * (1.0f / 255.0f);
    if (nudged_scale == 0) error("zero input scale");
    const double initial_zero_point = (double)(0.0f - min_value / nudged_scale);
    uint8_t zp;
    if (initial_zero_point < QUANT_NEGATIVE_LIMIT) zp = QUANT_NEGATIVE_LIMIT;
    else if (initial_zero_point > QUANT_POSITIVE_LIMIT) zp = QUANT_POSITIVE_LIMIT;
    else zp = (uint8_t)round(initial_zero_point);
    *scale = nudged_scale;
    *zero_point = zp;
}

void quant_image_with_min_max(int count, const float *input, uint8_t *out, float *scale, uint8_t *zero_point)
{
    float min_value = 0.0f, max_value = 0.0f;
    for (int j = 0; j < count; ++j) {
        max_value = input[j] > max_value ? input[j] : max_value;
        min_value = input[j] < min_value ? input[j] : min_value;
    }
    image_scale_zero_point(min_value, max_value, scale, zero_point);
    const float nudged_scale = *scale;
    const uint8_t zp = *zero_point;
    for (int k = 0; k < count; ++k) {
        float t = (float)(round((double)(input[k] / nudged_scale)) + (double)zp);
        int v = (int)t;
        out[k] = (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
    }
}

static void prep_conv_layer(network *net, int i)
{
    layer *l = &net->layers[i];
    if (i > 0) { /* ref :301-305: input scale / zero point are the previous layer's activation record */
        l->input_data_uint8_scales[0] = net->layers[i - 1].activ_data_uint8_scales[0];
        l->input_data_uint8_zero_point[0] = net->layers[i - 1].activ_data_uint8_zero_point[0];
    }
    const int K = l->c * l->size * l->size; /* ref :306 */
    const float s_in = l->input_data_uint8_scales[0];
    const int zp_in = l->input_data_uint8_zero_point[0];
    if (s_in == 0 || l->activ_data_uint8_scales[0] == 0) error("zero quantisation scale (ref: src/blas.c:312,332 asserts)");
    for (int ii = 0; ii < l->n; ++ii) {
        float b = l->biases[ii];
        if (l->batch_normalize) /* ref :286 -> :594-600, evaluated in double as the C expression promotes */
            b = (float)((double)b - (double)(l->scales[ii] * l->rolling_mean[ii]) /
                                        (sqrt((double)l->rolling_variance[ii]) + (double).000001f));
        if (l->weight_data_uint8_scales[ii] == 0) error("zero weight scale (ref: src/blas.c:293 assert)");
        l->mult_zero_point[ii] = (uint32_t)(K * zp_in * (int)l->weight_data_uint8_zero_point[ii]);
        int32_t wsum = 0;
        for (int jj = 0; jj < K; ++jj) wsum += l->weights_uint8[(size_t)ii * K + jj];
        l->weights_sum_int[ii] = (int32_t)(l->mult_zero_point[ii] - (uint32_t)(wsum * zp_in));
        l->M[ii] = s_in * l->weight_data_uint8_scales[ii] / l->activ_data_uint8_scales[0];
        quant_multi_smaller_than_one_to_scale_and_shift(l->M[ii], &l->M0[ii], &l->M0_right_shift[ii]);
        l->M0_right_shift_value[ii] = pow(2, -l->M0_right_shift[ii]);
        l->M_value[ii] = pow(2, -31) * l->M0[ii];
        float t = b / (s_in * l->weight_data_uint8_scales[ii]) + (float)l->weights_sum_int[ii]; /* ref :333 */
        l->biases_int32[ii] = (int32_t)t;
    }
}

static void alloc_layer_device(network *net, int i)
{
    layer *l = &net->layers[i];
    void *shared_blob = l->blob_shared ? l->blob_gpu : NULL; /* a replica's weights are its parent's: survive the re-allocation */
    free_layer_device(l);
    l->blob_gpu = shared_blob;
    const int B = net->batch;
    l->batch = B;
    if (l->type != YOLO) {
        size_t bytes = mi355_tensor_describe(&l->out_t, B, l->out_h, l->out_w, l->out_c);
        if (!bytes) error("bad tensor dims");
        check_mi355(mi355_alloc(&l->out_t.data, bytes), "alloc activations");
        check_mi355(mi355_tensor_fill(&l->out_t, l->activ_data_uint8_zero_point[0], net->stream), "fill activations");
        check_mi355(mi355_alloc((void **)&l->output_uint8_nchw_gpu, (size_t)B * l->outputs), "alloc nchw scratch");
    }
    if (l->type == YOLO || l->quant_stop_flag)
        check_mi355(mi355_alloc((void **)&l->output_gpu, (size_t)B * l->outputs * sizeof(float)), "alloc float out");
    if (l->type == CONVOLUTIONAL && net->dump_int32)
        check_mi355(mi355_alloc((void **)&l->output_int32_gpu, (size_t)B * l->outputs * sizeof(int32_t)), "alloc int32 out");
    free(l->output); free(l->output_int32); free(l->output_uint8_final);
    l->output = calloc((size_t)B * l->outputs, sizeof(float));
    l->output_int32 = l->type == CONVOLUTIONAL ? calloc((size_t)B * l->outputs, sizeof(int32_t)) : NULL;
    l->output_uint8_final = calloc((size_t)B * l->outputs, sizeof(uint8_t));
}

static void upload_conv(network *net, int i, int with_raw)
{
    layer *l = &net->layers[i];
    if (l->blob_shared) error("upload_conv on a replica (its packed weights belong to the parent network)");
    if (l->blob_gpu) { mi355_free(l->blob_gpu); l->blob_gpu = NULL; }
    check_mi355(mi355_alloc(&l->blob_gpu, l->blob_bytes), "alloc blob");
    check_mi355(mi355_h2d(l->blob_gpu, l->blob_host, l->blob_bytes, net->stream), "upload blob");
    if (with_raw) {
        check_mi355(mi355_alloc((void **)&l->weights_uint8_gpu, (size_t)l->nweights), "alloc raw weights");
        check_mi355(mi355_h2d(l->weights_uint8_gpu, l->weights_uint8, (size_t)l->nweights, net->stream), "upload raw weights");
        check_mi355(mi355_alloc((void **)&l->weight_zero_point_gpu, (size_t)l->n), "alloc zp_w");
        check_mi355(mi355_h2d(l->weight_zero_point_gpu, l->weight_data_uint8_zero_point, (size_t)l->n, net->stream), "upload zp_w");
    }
    l->prepared = 1;
}

/* Concat elimination.  forward_route_layer_quant (ref: src/route_layer.c:107-130) copies its inputs' bytes side by side,
 * unscaled.  When every input of a route can write its channels straight into the route's buffer -- its out_t becomes a
 * window (data + channel offset, the route's cell stride) of that buffer -- the copy disappears; a one-input route
 * simply shares its input's tensor.  A producer qualifies when nothing reads ITS pad cells with a different zero point
 * (the window's pads are the route's: a 3x3 conv consuming the producer directly needs them to be its own input zero
 * point), its channel count is a multiple of 16, and it stores its tensor at all (no conv+pool fusion). */
static int view_producer_ok(network *net, int j, int r)
{
    layer *p = &net->layers[j];
    if (j >= r || p->out_view || p->out_c % 16) return 0;
    if (p->type != CONVOLUTIONAL && p->type != UPSAMPLE && p->type != MAXPOOL) return 0;
    if (p->type == CONVOLUTIONAL && (p->fuse_next_pool && !p->fuse_pool_keep && net->fuse_maxpool)) return 0;
    if (j > 0 && net->layers[j - 1].fuse_next_pool && net->fuse_maxpool && p->type == MAXPOOL) return 0; /* written by the fused conv */
    const int zp_differs = p->activ_data_uint8_zero_point[0] != net->layers[r].activ_data_uint8_zero_point[0];
    if (j + 1 < net->n) {
        layer *c = &net->layers[j + 1];
        if (c->type == CONVOLUTIONAL && c->size != 1 && zp_differs) return 0;
    }
    /* a one-input route that plan_views elides shares the producer's tensor: the layer after THAT route reads the same
     * pad cells */
    for (int r2 = 0; r2 + 1 < net->n; ++r2) {
        layer *q = &net->layers[r2];
        if (q->type != ROUTE || q->n != 1 || q->input_layers[0] != j) continue;
        layer *c = &net->layers[r2 + 1];
        if (c->type == CONVOLUTIONAL && c->size != 1 && zp_differs) return 0;
    }
    return 1;
}

static void plan_views(network *net)
{
    if (net->dump_int32) return; /* parity dumps keep every tensor in its own buffer */
    for (int r = 0; r < net->n; ++r) { /* concatenating routes first: their inputs must not be shared tensors yet */
        layer *l = &net->layers[r];
        if (l->type != ROUTE || l->n < 2) continue;
        int ok = 1;
        for (int k = 0; k < l->n; ++k) {
            ok &= view_producer_ok(net, l->input_layers[k], r);
            for (int k2 = 0; k2 < k; ++k2) ok &= l->input_layers[k2] != l->input_layers[k];
        }
        if (!ok) continue;
        int off = 0;
        for (int k = 0; k < l->n; ++k) {
            layer *p = &net->layers[l->input_layers[k]];
            mi355_free(p->out_t.data);
            p->out_t = l->out_t;
            p->out_t.data = (char *)l->out_t.data + off;
            p->out_t.C = p->out_c;
            p->out_view = 1;
            off += p->out_c;
        }
        l->route_elided = 1;
    }
    for (int r = 0; r < net->n; ++r) {
        layer *l = &net->layers[r];
        if (l->type != ROUTE || l->n != 1) continue;
        layer *p = &net->layers[l->input_layers[0]];
        if (l->input_layers[0] >= r || p->type == YOLO || !p->out_t.data) continue;
        if (p->type == CONVOLUTIONAL && p->fuse_next_pool && !p->fuse_pool_keep && net->fuse_maxpool) continue;
        mi355_free(l->out_t.data);
        l->out_t = p->out_t;
        l->out_view = 1;
        l->route_elided = 1;
    }
}

static void alloc_network_device(network *net)
{
    if (mi355_abi_version() != MI355_ABI_VERSION) {
        fprintf(stderr, "libmi355yolo.so speaks ABI %d, this host was built against %d\n", mi355_abi_version(), MI355_ABI_VERSION);
        error("mi355 ABI mismatch");
    }
    check_mi355(mi355_init(net->gpu_index), "mi355_init");
    if (!net->stream && !net->on_default_stream) check_mi355(mi355_stream_acquire(&net->stream), "stream");
    if (net->input_uint8_gpu) mi355_free(net->input_uint8_gpu);
    if (net->input_t.data) mi355_free(net->input_t.data);
    check_mi355(mi355_alloc((void **)&net->input_uint8_gpu, (size_t)net->batch * net->inputs), "alloc input");
    size_t bytes = mi355_tensor_describe(&net->input_t, net->batch, net->h, net->w, net->c);
    check_mi355(mi355_alloc(&net->input_t.data, bytes), "alloc input tensor");
    check_mi355(mi355_tensor_fill(&net->input_t, net->layers[0].input_data_uint8_zero_point[0], net->stream), "fill input");
    mi355_tensor_describe_nchw(&net->input_nchw_t, net->batch, net->h, net->w, net->c);
    net->input_nchw_t.data = net->input_uint8_gpu;
    net->input_direct = net->c == 3 && !net->dump_int32 && !net->input_direct_off;
    for (int i = 0; i < net->n; ++i) alloc_layer_device(net, i);
    plan_views(net);
    if (net->graph) { mi355_graph_destroy(net->graph); net->graph = NULL; }
}

/* 16.16 multipliers of the two addends of every quantized [shortcut] (mi355_shortcut_multiplier) */
static void prep_shortcut_layers(network *net)
{
    for (int i = 1; i < net->n; ++i) {
        layer *l = &net->layers[i];
        if (l->type != SHORTCUT) continue;
        const layer *a = &net->layers[i - 1], *b = &net->layers[l->index];
        if (a->type == YOLO || b->type == YOLO) error("shortcut: inputs must be quantized layers");
        check_mi355(mi355_shortcut_multiplier(a->activ_data_uint8_scales[0], l->activ_data_uint8_scales[0], &l->shortcut_Ka),
                    "mi355_shortcut_multiplier (previous layer)");
        check_mi355(mi355_shortcut_multiplier(b->activ_data_uint8_scales[0], l->activ_data_uint8_scales[0], &l->shortcut_Kb),
                    "mi355_shortcut_multiplier (`from` layer)");
    }
}

/* does any layer other than i + 1 read layer i's own tensor? (a route input, the `from` of a shortcut) */
static int output_read_elsewhere(const network *net, int i)
{
    for (int j = 0; j < net->n; ++j) {
        const layer *q = &net->layers[j];
        if (q->type == ROUTE)
            for (int k = 0; k < q->n; ++k)
                if (q->input_layers[k] == i) return 1;
        if (q->type == SHORTCUT && q->index == i) return 1;
    }
    return 0;
}

void quantization_prep_host(network *net, float in_scale, uint8_t in_zp)
{
    layer *l0 = &net->layers[0];
    if (l0->type != CONVOLUTIONAL) error("first layer must be convolutional");
    prep_shortcut_layers(net);
    l0->input_data_uint8_scales[0] = in_scale;
    l0->input_data_uint8_zero_point[0] = in_zp;
    for (int i = 0; i < net->n; ++i) {
        layer *l = &net->layers[i];
        if (l->type != CONVOLUTIONAL) continue;
        prep_conv_layer(net, i);
        size_t sz = mi355_conv_pack_size(l->n, l->c, l->size);
        if (!sz) {
            fprintf(stderr, "layer %d: conv %dx%d, %d->%d channels is not supported by the gfx950 kernels "
                            "(need size 1|3 and c==3 or c%%16==0)\n", i, l->size, l->size, l->c, l->n);
            error("unsupported convolution shape");
        }
        free(l->blob_host);
        l->blob_host = malloc(sz);
        l->blob_bytes = sz;
        check_mi355(mi355_conv_pack(l->n, l->c, l->size, l->weights_uint8, l->weight_data_uint8_zero_point,
                                    l->biases_int32, l->M_value, l->M0_right_shift_value, l->blob_host),
                    "mi355_conv_pack");
        /* the fused conv + maxpool kernels' per-channel epilogue constants for this layer's activation and zero point */
        check_mi355(mi355_conv_pack_epilogue(l->n, l->c, l->size, l->activation, l->activ_data_uint8_zero_point[0], l->blob_host),
                    "mi355_conv_pack_epilogue");
    }
}

/* conv i + maxpool i+1 can run as one kernel when the pool is the reference's size-2 / stride-2 / offset-0 window on an
 * even map, the conv is 3x3 and nothing else (a route) reads the conv's own output */
static void plan_fusion(network *net)
{
    for (int i = 0; i < net->n; ++i)
        net->layers[i].fuse_next_pool = net->layers[i].fuse_pool_keep = net->layers[i].fuse_next_yolo = net->layers[i].fuse_next_shortcut = 0;
    for (int i = 0; i + 1 < net->n; ++i) { /* conv + quantized residual add: the conv's own tensor has no other reader */
        layer *c = &net->layers[i], *sc = &net->layers[i + 1];
        if (c->type == CONVOLUTIONAL && sc->type == SHORTCUT && c->stride == 1 && !c->quant_stop_flag && !sc->quant_stop_flag &&
            c->c % 16 == 0 && sc->index != i && !output_read_elsewhere(net, i))
            c->fuse_next_shortcut = 1;
    }
    for (int i = 0; i < net->n; ++i) net->layers[i].fuse_next_upsample = 0;
    for (int i = 0; i + 1 < net->n; ++i) { /* conv + nearest upsample: the conv stores every pixel stride x stride times */
        layer *c = &net->layers[i], *u = &net->layers[i + 1];
        if (c->type != CONVOLUTIONAL || u->type != UPSAMPLE || c->quant_stop_flag || c->c % 64 || u->stride > 4 || c->stride != 1) continue;
        if (!output_read_elsewhere(net, i)) c->fuse_next_upsample = 1;
    }
    for (int i = 0; i + 1 < net->n; ++i) { /* quant_stop head conv + yolo: one kernel writes both float tensors */
        layer *c = &net->layers[i], *y = &net->layers[i + 1];
        if (c->type == CONVOLUTIONAL && c->quant_stop_flag && y->type == YOLO && c->c % 16 == 0 && c->stride == 1 &&
            c->n == y->n * (y->classes + 5))
            c->fuse_next_yolo = 1;
    }
    for (int i = 0; i + 1 < net->n; ++i) {
        layer *c = &net->layers[i], *p = &net->layers[i + 1];
        if (c->type != CONVOLUTIONAL || p->type != MAXPOOL) continue;
        if (c->size != 3 || c->stride != 1 || c->quant_stop_flag || p->quant_stop_flag || p->size != 2 || p->pad / 2 != 0) continue;
        if (c->c == 128 || c->c == 256) {
            /* the weights-stationary kernel of the middle layers (conv_ws3.hip) pools the bytes of its tile in LDS: the stride-2
             * window on even maps, and the reference's stride-1 window (pad = 1: same size as the conv's map; yolov3-tiny's
             * layer 11); it stores the conv's own tensor as well when a route reads it (layer 8) */
            const int s2 = p->stride == 2 && !(c->out_h & 1) && !(c->out_w & 1);
            const int s1 = p->stride == 1 && p->pad == 1 && p->out_h == c->out_h && p->out_w == c->out_w && c->out_h > 1;
            if (!s2 && !s1) continue;
            c->fuse_next_pool = 1;
            c->fuse_pool_keep = output_read_elsewhere(net, i);
            continue;
        }
        if (p->stride != 2) continue;
        if ((c->out_h & 1) || (c->out_w & 1) || (c->c != 3 && c->c % 16)) continue;
        /* 64-byte-chunk layers use the row-image kernel, which has no fused form; 64 -> 64..128 has its own fused kernel */
        if (c->c % 64 == 0 && !(c->c == 64 && c->n % 32 == 0 && c->n >= 64 && c->n <= 128)) continue;
        if (!output_read_elsewhere(net, i)) c->fuse_next_pool = 1;
    }
}

void quantization_weights_and_activations_fixed_input(network *net, float in_scale, uint8_t in_zp)
{
    if (net->n_replicas > 0 || net->replica_of) error("quantization_weights_and_activations: not while replicas share this network's packed weights");
    quantization_prep_host(net, in_scale, in_zp);
    plan_fusion(net);
    alloc_network_device(net);
    for (int i = 0; i < net->n; ++i)
        if (net->layers[i].type == CONVOLUTIONAL) upload_conv(net, i, net->has_host_weights);
    check_mi355(mi355_stream_sync(net->stream), "sync");
    net->prepared = 1;
}

void quantization_weights_and_activations(network *net)
{
    /* ref :279: dynamic layer-0 quantiser on the float image in net->input (image 0 defines scale / zero point) */
    float s; uint8_t zp;
    quant_image_with_min_max(net->inputs, net->input, net->input_uint8, &s, &zp);
    quantization_weights_and_activations_fixed_input(net, s, zp);
    for (int b = 1; b < net->batch; ++b) { /* further images: same scale (batch > 1 is our extension) */
        float *x = net->input + (size_t)b * net->inputs;
        uint8_t *o = net->input_uint8 + (size_t)b * net->inputs;
        for (int k = 0; k < net->inputs; ++k) {
            int v = (int)(float)(round((double)(x[k] / s)) + (double)zp);
            o[k] = (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
        }
    }
    push_network_input_uint8(net, net->input_uint8);
}

/* The same on the device (quantiser half): `input_gpu` holds batch x inputs floats in HBM.  Image 0
 * defines scale / zero point as in the reference (src/blas.c:279); min / max are reduced on the device, the two floats
 * come back to the host, which evaluates the reference's scale / zero-point expressions and -- only when they differ
 * from the ones layer 0 was prepared with -- re-derives and re-uploads layer 0's multipliers and blob in place (no other
 * layer depends on the input scale); the per-element quantiser then runs on the device straight into the network's uint8
 * input.  Byte-identical to quantization_weights_and_activations() on the same floats. */
void quantization_weights_and_activations_gpu(network *net, const float *input_gpu)
{
    if (!input_gpu) error("quantization_weights_and_activations_gpu: null input");
    check_mi355(mi355_init(net->gpu_index), "mi355_init");
    if (!net->stream && !net->on_default_stream) check_mi355(mi355_stream_acquire(&net->stream), "stream");
    if (!net->quant_mm_gpu) check_mi355(mi355_alloc((void **)&net->quant_mm_gpu, 2 * sizeof(float)), "alloc minmax");
    float mm[2];
    check_mi355(mi355_image_minmax(input_gpu, net->inputs, net->quant_mm_gpu, net->stream), "mi355_image_minmax");
    check_mi355(mi355_d2h(mm, net->quant_mm_gpu, sizeof(mm), net->stream), "minmax d2h");
    check_mi355(mi355_stream_sync(net->stream), "sync");
    float s; uint8_t zp;
    image_scale_zero_point(mm[1] + 0.0f, mm[0], &s, &zp);
    layer *l0 = &net->layers[0];
    if (!net->prepared) {
        quantization_weights_and_activations_fixed_input(net, s, zp);
    } else if (l0->input_data_uint8_scales[0] != s || l0->input_data_uint8_zero_point[0] != zp) {
        if (net->replica_of) error("the input scale changed: a replica shares its parent's layer-0 blob and cannot re-derive it");
        if (net->n_replicas > 0) error("the input scale changed: layer 0's blob is shared with replicas that may be running; free them first");
        if (!net->has_host_weights && !net->has_l0_weights)
            error("the input scale changed but this network holds no raw layer-0 weights to re-derive layer 0 from");
        const int zp_changed = l0->input_data_uint8_zero_point[0] != zp;
        l0->input_data_uint8_scales[0] = s;
        l0->input_data_uint8_zero_point[0] = zp;
        prep_conv_layer(net, 0);
        check_mi355(mi355_conv_pack(l0->n, l0->c, l0->size, l0->weights_uint8, l0->weight_data_uint8_zero_point,
                                    l0->biases_int32, l0->M_value, l0->M0_right_shift_value, l0->blob_host), "mi355_conv_pack");
        check_mi355(mi355_conv_pack_epilogue(l0->n, l0->c, l0->size, l0->activation, l0->activ_data_uint8_zero_point[0], l0->blob_host),
                    "mi355_conv_pack_epilogue");
        check_mi355(mi355_h2d(l0->blob_gpu, l0->blob_host, l0->blob_bytes, net->stream), "upload blob 0");
        if (zp_changed) check_mi355(mi355_tensor_fill(&net->input_t, zp, net->stream), "fill input");  /* pad cells = zero point */
        check_mi355(mi355_stream_sync(net->stream), "sync");  /* blob_host may be repacked by the next call */
        /* a captured graph holds layer 0's kernel arguments by value: the planar first-layer kernels take the pad value
         * (the input zero point) from there, not from pad cells in memory -> re-capture on the next forward */
        if (net->graph) { mi355_graph_destroy(net->graph); net->graph = NULL; }
    }
    check_mi355(mi355_image_quantize(input_gpu, (long)net->batch * net->inputs, s, zp, net->input_uint8_gpu, net->stream),
                "mi355_image_quantize");
}

void network_letterbox_input_gpu(network *net, int slot, const float *im_gpu, int imw, int imh)
{
    if (slot < 0 || slot >= net->batch || !im_gpu) error("network_letterbox_input_gpu: bad slot / null image");
    check_mi355(mi355_init(net->gpu_index), "mi355_init");
    if (!net->stream && !net->on_default_stream) check_mi355(mi355_stream_acquire(&net->stream), "stream");
    if (!net->input_gpu)
        check_mi355(mi355_alloc((void **)&net->input_gpu, (size_t)net->batch * net->inputs * sizeof(float)), "alloc float input");
    check_mi355(mi355_letterbox_forward(im_gpu, imw, imh, net->c, net->input_gpu + (size_t)slot * net->inputs, net->w, net->h,
                                        net->stream), "mi355_letterbox_forward");
}

void network_quantize_input_gpu(network *net)
{
    if (!net->input_gpu) error("network_quantize_input_gpu before network_letterbox_input_gpu");
    quantization_weights_and_activations_gpu(net, net->input_gpu);
}

void set_batch_network(network *net, int b)
{
    if (b < 1) error("set_batch_network: batch < 1");
    if (b == net->batch && net->prepared) return;
    if (net->replica_of) error("set_batch_network on a replica: re-batch the parent and make new replicas");
    if (net->n_replicas > 0) error("set_batch_network: free this network's replicas first (they borrow its packed weights on the device)");
    net->batch = b;
    free(net->input); free(net->input_uint8);
    net->input = calloc((size_t)net->inputs * b, sizeof(float));
    net->input_uint8 = calloc((size_t)net->inputs * b, sizeof(uint8_t));
    for (int i = 0; i < net->n; ++i) net->layers[i].batch = b;
    if (net->input_gpu) { /* batch x inputs floats of the device input path: re-allocated lazily at the new size */
        mi355_free(net->input_gpu);
        net->input_gpu = NULL;
    }
    if (net->prepared) { /* re-size the device buffers, keep the packed weights */
        plan_fusion(net); /* launcher acceptance depends on the batch: flags an earlier EINVAL fallback cleared are re-derived */
        alloc_network_device(net);
        for (int i = 0; i < net->n; ++i)
            if (net->layers[i].type == CONVOLUTIONAL) upload_conv(net, i, net->has_host_weights);
        check_mi355(mi355_stream_sync(net->stream), "sync");
    }
}

/* -------------------------------------------------------------------------------------------------- execution */
void push_network_input_uint8(network *net, const uint8_t *host_nchw)
{
    if (!net->prepared) error("push_network_input_uint8 before quantization_weights_and_activations");
    check_mi355(mi355_h2d(net->input_uint8_gpu, host_nchw, (size_t)net->batch * net->inputs, net->stream), "push input");
}

void network_profile_begin(network *net, int max_steps)
{
    if (net->prof_ev) {
        for (int i = 0; i < net->prof_cap * (net->n + 2); ++i) mi355_event_destroy(net->prof_ev[i]);
        free(net->prof_ev);
        net->prof_ev = NULL;
    }
    net->prof_cap = max_steps;
    net->prof_used = 0;
    net->prof_calls = 0;
    if (net->prof_stride < 1) net->prof_stride = 1;
    if (max_steps <= 0) return;
    net->prof_ev = calloc((size_t)max_steps * (net->n + 2), sizeof(void *));
    for (int i = 0; i < max_steps * (net->n + 2); ++i) check_mi355(mi355_event_create(&net->prof_ev[i]), "event create");
}

void network_profile_set_stride(network *net, int stride) { net->prof_stride = stride < 1 ? 1 : stride; }
void network_profile_set_phase(network *net, int phase) { net->prof_phase = phase < 0 ? 0 : phase; }

int network_profile_read(network *net, float *ms_sum)
{
    for (int i = 0; i < net->n + 1; ++i) ms_sum[i] = 0;
    for (int s = 0; s < net->prof_used; ++s) {
        void **e = net->prof_ev + (size_t)s * (net->n + 2);
        for (int i = 0; i < net->n + 1; ++i) {
            float ms = 0;
            check_mi355(mi355_event_elapsed_ms(e[i], e[i + 1], &ms), "event elapsed");
            ms_sum[i] += ms;
        }
    }
    return net->prof_used;
}

static void run_layers(network *netp)
{
    network net = *netp;
    void **ev = NULL;
    const int ranged = netp->range_hi > netp->range_lo;
    if (ranged) { /* diagnostic knob: validate it instead of reading tensors nobody wrote */
        if (netp->use_graph) error("forward_network_gpu: a layer range cannot be combined with use_graph (the captured graph is the whole net)");
        if (netp->prof_ev && netp->prof_used < netp->prof_cap) error("forward_network_gpu: a layer range cannot be combined with armed per-layer events");
        if (netp->range_lo > 0) {
            const layer *pl = &netp->layers[netp->range_lo - 1];
            const int stored = !(pl->type == CONVOLUTIONAL && netp->fuse_maxpool && !netp->dump_int32 && netp->accum_mode == MI355_ACC_EXACT &&
                                 ((pl->fuse_next_pool && !pl->fuse_pool_keep) || pl->fuse_next_upsample || pl->fuse_next_shortcut || pl->fuse_next_yolo));
            if (!stored) error("forward_network_gpu: layer range starts behind a conv whose own tensor is not stored (fused with the layer after it)");
        }
    }
    if (netp->prof_ev && netp->prof_used < netp->prof_cap && !netp->use_graph && netp->prof_calls++ % netp->prof_stride == netp->prof_phase % netp->prof_stride)
        ev = netp->prof_ev + (size_t)(netp->prof_used++) * (net.n + 2);
    if (ev) check_mi355(mi355_event_record(ev[0], net.stream), "event");
    /* The first layer reads the reference's [B][3][H][W] planes in place where its kernel can (no conversion pass); else the
     * input goes through the 4-byte-cell tensor.  The conv's forward_gpu falls back itself on MI355_EINVAL and clears the flag. */
    const int direct = netp->input_direct && net.accum_mode == MI355_ACC_EXACT && !net.dump_int32;
    const int lo = (netp->range_hi > netp->range_lo) ? netp->range_lo : 0, hi = (netp->range_hi > netp->range_lo) ? netp->range_hi : net.n;
    if (lo < 0 || hi > net.n) error("forward_network_gpu: bad layer range");
    if (!direct && lo == 0) check_mi355(mi355_nchw_to_tensor(netp->input_uint8_gpu, &netp->input_t, net.stream), "input layout");
    if (ev) check_mi355(mi355_event_record(ev[1], net.stream), "event");
    net.cur_t = direct ? &netp->input_nchw_t : &netp->input_t;
    net.input_direct_p = &netp->input_direct;
    net.cur_f32_gpu = NULL;
    if (lo > 0) net.cur_t = &netp->layers[lo - 1].out_t;
    for (int i = lo; i < hi; ++i) {
        net.index = i;
        layer l = net.layers[i];
        const int fuse0 = l.fuse_next_pool && net.fuse_maxpool && !net.dump_int32 && net.accum_mode == MI355_ACC_EXACT;
        const int fuse_yolo0 = l.fuse_next_yolo && net.fuse_maxpool && !net.dump_int32 && net.accum_mode == MI355_ACC_EXACT;
        net.fused_pool_t = fuse0 ? &netp->layers[i + 1].out_t : NULL;
        net.fused_yolo_out = fuse_yolo0 ? netp->layers[i + 1].output_gpu : NULL;
        net.fused_yolo_classes = fuse_yolo0 ? netp->layers[i + 1].classes : 0;
        const int fuse_sc0 = l.fuse_next_shortcut && net.fuse_maxpool && !net.dump_int32 && net.accum_mode == MI355_ACC_EXACT;
        net.fused_shortcut = fuse_sc0 ? &netp->layers[i + 1] : NULL;
        const int fuse_up0 = l.fuse_next_upsample && net.fuse_maxpool && !net.dump_int32 && net.accum_mode == MI355_ACC_EXACT;
        net.fused_up_t = fuse_up0 ? &netp->layers[i + 1].out_t : NULL;
        net.fused_up_stride = fuse_up0 ? netp->layers[i + 1].stride : 1;
        l.forward_gpu(l, net);
        if (l.type == CONVOLUTIONAL) netp->layers[i].conv_kernel = mi355_last_conv_kernel();
        /* plan_fusion marks candidates by shape; the launchers have the last word.  A fused call they refuse
         * (MI355_EINVAL) was re-run unfused by the conv's forward_gpu, which also cleared the flag for good: the layer
         * after it then runs on its own like any other. */
        const int fuse = fuse0 && netp->layers[i].fuse_next_pool;
        const int fuse_yolo = fuse_yolo0 && netp->layers[i].fuse_next_yolo;
        const int fuse_up = fuse_up0 && netp->layers[i].fuse_next_upsample;
        const int fuse_sc = fuse_sc0 && netp->layers[i].fuse_next_shortcut;
        if (fuse_sc) { /* the shortcut layer's tensor was written by the conv kernel: hand it on and skip the layer */
            if (ev) check_mi355(mi355_event_record(ev[i + 2], net.stream), "event");
            ++i;
            net.cur_t = &netp->layers[i].out_t;
            net.cur_f32_gpu = NULL;
            if (ev) check_mi355(mi355_event_record(ev[i + 2], net.stream), "event");
            continue;
        }
        if (ev) check_mi355(mi355_event_record(ev[i + 2], net.stream), "event");
        if (fuse_up) { /* the upsample layer's tensor was written by the conv kernel: hand it on and skip the layer */
            ++i;
            net.cur_t = &netp->layers[i].out_t;
            net.cur_f32_gpu = NULL;
            if (ev) check_mi355(mi355_event_record(ev[i + 2], net.stream), "event");
            continue;
        }
        if (fuse_yolo) { /* the yolo layer's activations were written by the conv kernel: skip it */
            net.cur_t = &netp->layers[i].out_t;
            ++i;
            net.cur_f32_gpu = netp->layers[i].output_gpu;
            if (ev) check_mi355(mi355_event_record(ev[i + 2], net.stream), "event");
            continue;
        }
        if (fuse) { /* the maxpool layer already ran inside the conv kernel: hand its tensor on and skip it */
            ++i;
            net.cur_t = &netp->layers[i].out_t;
            net.cur_f32_gpu = NULL;
            if (ev) check_mi355(mi355_event_record(ev[i + 2], net.stream), "event");
            continue;
        }
        if (l.layer_quant_flag && !net.train) { /* ref src/network.c:248-250 */
            net.cur_t = &netp->layers[i].out_t;
            net.cur_f32_gpu = l.output_gpu;
        } else {
            net.cur_f32_gpu = l.output_gpu;
        }
        if (net.verbose) fprintf(stderr, "layer %2d %-8s done\n", i, get_layer_string(l.type));
    }
}

void forward_network_gpu(network *netp)
{
    if (!netp->prepared) error("forward_network_gpu before quantization_weights_and_activations");
    if (netp->accum_mode == MI355_ACC_REF_F32 && !netp->has_host_weights)
        error("-accum ref-f32 reads the raw weights_uint8 of every layer; a network imported from packed blobs does not hold them");
    if (netp->use_graph && netp->on_default_stream) error("use_graph: the default stream cannot be captured; run this executor eagerly");
    if (netp->use_graph) {
        if (!netp->graph) {
            run_layers(netp); /* warm-up outside capture (module load, attribute calls) */
            check_mi355(mi355_stream_sync(netp->stream), "sync");
            check_mi355(mi355_graph_begin(netp->stream), "graph begin");
            run_layers(netp);
            check_mi355(mi355_graph_end(netp->stream, &netp->graph), "graph end");
        }
        check_mi355(mi355_graph_launch(netp->graph, netp->stream), "graph launch");
    } else {
        run_layers(netp);
    }
}

void forward_network(network *net) { forward_network_gpu(net); }

void network_selfcheck(network *net, int passes)
{
    if (passes < 2) passes = 2;
    if (net->selfcheck_gpu) mi355_free(net->selfcheck_gpu);
    check_mi355(mi355_alloc((void **)&net->selfcheck_gpu, (size_t)passes * sizeof(uint64_t)), "alloc selfcheck");
    uint64_t *zeros = calloc((size_t)passes, sizeof(uint64_t));
    check_mi355(mi355_h2d(net->selfcheck_gpu, zeros, (size_t)passes * sizeof(uint64_t), net->stream), "zero selfcheck");
    check_mi355(mi355_stream_sync(net->stream), "sync");
    free(zeros);
    net->selfcheck_passes = passes;
    for (int p = 0; p < passes; ++p) {
        forward_network_gpu(net);
        for (int i = 0; i < net->n; ++i) {
            const layer *l = &net->layers[i];
            if (l->type != YOLO || !l->output_gpu) continue;
            check_mi355(mi355_checksum_u32(l->output_gpu, (long)net->batch * l->outputs, net->selfcheck_gpu + p, net->stream), "checksum");
        }
    }
}

int network_selfcheck_result(network *net)
{
    if (!net->selfcheck_gpu || net->selfcheck_passes < 1) return -1;
    const int passes = net->selfcheck_passes;
    uint64_t *sums = calloc((size_t)passes, sizeof(uint64_t));
    check_mi355(mi355_d2h(sums, net->selfcheck_gpu, (size_t)passes * sizeof(uint64_t), net->stream), "pull selfcheck");
    check_mi355(mi355_stream_sync(net->stream), "sync");
    int bad = 0;
    for (int p = 1; p < passes; ++p) bad += sums[p] != sums[0];
    free(sums);
    mi355_free(net->selfcheck_gpu);
    net->selfcheck_gpu = NULL;
    net->selfcheck_passes = 0;
    return bad;
}

float *network_predict(network *net, float *input)
{
    /* ref src/network.c:570-581; the integer path ignores `input` after the prep quantised it (examples/detector.c
     * :914-921), so does this one: the uint8 image must already be on the device. */
    (void)input;
    net->train = 0;
    forward_network_gpu(net);
    check_mi355(mi355_stream_sync(net->stream), "sync");
    layer *last = &net->layers[net->n - 1];
    if (last->output_gpu) {
        check_mi355(mi355_d2h(last->output, last->output_gpu, (size_t)net->batch * last->outputs * sizeof(float), net->stream), "pull output");
        check_mi355(mi355_stream_sync(net->stream), "sync");
    }
    net->output = last->output;
    return net->output;
}

void pull_layer_output(network *net, int i)
{
    if (i < 0 || i >= net->n) error("pull_layer_output: index");
    layer *l = &net->layers[i];
    const size_t cnt = (size_t)net->batch * l->outputs;
    if (l->type != YOLO) {
        check_mi355(mi355_tensor_to_nchw(&l->out_t, l->output_uint8_nchw_gpu, net->stream), "layout");
        check_mi355(mi355_d2h(l->output_uint8_final, l->output_uint8_nchw_gpu, cnt, net->stream), "pull u8");
    }
    if (l->output_gpu) check_mi355(mi355_d2h(l->output, l->output_gpu, cnt * sizeof(float), net->stream), "pull f32");
    if (l->output_int32_gpu) check_mi355(mi355_d2h(l->output_int32, l->output_int32_gpu, cnt * sizeof(int32_t), net->stream), "pull i32");
    check_mi355(mi355_stream_sync(net->stream), "sync");
}

static int cmp_rec_rank(const void *a, const void *b)
{
    const float x = *(const float *)a, y = *(const float *)b;
    return (x > y) - (x < y);
}

void network_yolo_detections_gpu(network *net, int i, int imw, int imh, float thresh, int relative, float *recs,
                                 int max_recs, int *counts)
{
    if (i < 0 || i >= net->n || net->layers[i].type != YOLO) error("network_yolo_detections_gpu: not a yolo layer");
    layer *l = &net->layers[i];
    const int B = net->batch, rl = 6 + l->classes;
    if (!l->anchors_gpu) {
        check_mi355(mi355_alloc((void **)&l->anchors_gpu, sizeof(float) * 2 * (size_t)l->total), "alloc anchors");
        check_mi355(mi355_h2d(l->anchors_gpu, l->anchors, sizeof(float) * 2 * (size_t)l->total, net->stream), "anchors");
        check_mi355(mi355_alloc((void **)&l->mask_gpu, sizeof(int) * (size_t)l->n), "alloc mask");
        check_mi355(mi355_h2d(l->mask_gpu, l->mask, sizeof(int) * (size_t)l->n, net->stream), "mask");
    }
    if (l->det_cap < max_recs) {
        if (l->det_recs_gpu) { mi355_free(l->det_recs_gpu); mi355_free(l->det_counts_gpu); }
        check_mi355(mi355_alloc((void **)&l->det_recs_gpu, sizeof(float) * (size_t)B * max_recs * rl), "alloc records");
        check_mi355(mi355_alloc((void **)&l->det_counts_gpu, sizeof(int) * (size_t)B), "alloc counts");
        l->det_cap = max_recs;
    }
    check_mi355(mi355_yolo_detections(l->output_gpu, B, l->n, l->classes, l->h, l->w, l->anchors_gpu, l->mask_gpu, net->w,
                                      net->h, imw, imh, thresh, relative, l->det_recs_gpu, max_recs, l->det_counts_gpu,
                                      net->stream), "mi355_yolo_detections");
    check_mi355(mi355_d2h(counts, l->det_counts_gpu, sizeof(int) * (size_t)B, net->stream), "pull counts");
    check_mi355(mi355_stream_sync(net->stream), "sync");
    for (int b = 0; b < B; ++b) { /* only the records that exist cross PCIe; reference order = ascending rank */
        const int k = counts[b] < max_recs ? counts[b] : max_recs;
        if (!k) continue;
        float *dst = recs + (size_t)b * max_recs * rl;
        check_mi355(mi355_d2h(dst, l->det_recs_gpu + (size_t)b * max_recs * rl, sizeof(float) * (size_t)k * rl, net->stream), "pull records");
        check_mi355(mi355_stream_sync(net->stream), "sync");
        qsort(dst, (size_t)k, sizeof(float) * rl, cmp_rec_rank);
    }
}

/* ------------------------------------------------------------------------------------ packed-weight exchange */
/* Layout (little endian, position independent): pack_head | pack_rec[nlayers] | blobs (16-byte aligned, conv layers in
 * order) | layer-0 raw record.  The last part carries what prep_conv_layer needs to RE-DERIVE layer 0 when an image's
 * dynamic input scale / zero point differs from the exporter's (ref: src/blas.c:279 recomputes them per image): biases,
 * batch-norm statistics, weight scales / zero points and the 3-channel layer's few raw weights.  No other layer depends on
 * the input scale, so the other layers travel as packed blobs only (an imported network therefore cannot serve the
 * MI355_ACC_REF_F32 verification mode, which reads raw weights: has_host_weights == 0 refuses it). */
typedef struct { uint32_t magic; int32_t nlayers; float in_scale; int32_t in_zp; uint64_t total; uint64_t l0_bytes; } pack_head;
typedef struct { float s_act, s_in; int32_t zp_act, zp_in; uint64_t blob_bytes; } pack_rec;
typedef struct { int32_t n, c, size, batch_normalize; } pack_l0;
#define PACK_MAGIC 0x35444B4Eu /* "NKD5": the LEAKY byte table in every epilogue table (conv_pool16.hip); NKD4: round 5 added the epilogue table to the blobs of the conv + maxpool shapes (common.h EptHeader); NKD3: permuted A rows (kargs.h ws_row_filter) */

static size_t l0_record_bytes(const layer *l)
{
    size_t sz = sizeof(pack_l0) + (size_t)l->n * sizeof(float) * (l->batch_normalize ? 4 : 1) /* biases (+ scales, mean, var) */
                + (size_t)l->n * sizeof(float) + (size_t)l->n + (size_t)l->nweights;            /* w scales, w zp, weights */
    return (sz + 15) & ~(size_t)15;
}

size_t network_packed_size(network *net) /* a function of the cfg alone: every rank knows it without communication */
{
    size_t sz = sizeof(pack_head) + (size_t)net->n * sizeof(pack_rec);
    sz = (sz + 15) & ~(size_t)15;
    for (int i = 0; i < net->n; ++i) {
        layer *l = &net->layers[i];
        if (l->type != CONVOLUTIONAL) continue;
        const size_t b = mi355_conv_pack_size(l->n, l->c, l->size);
        if (!b) error("network_packed_size: unsupported convolution shape");
        sz += (b + 15) & ~(size_t)15;
    }
    return sz + l0_record_bytes(&net->layers[0]);
}

void network_export_packed(network *net, void *buf)
{
    if (!net->layers[0].blob_host) error("network_export_packed before the host prep");
    char *p = buf;
    const size_t total = network_packed_size(net);
    memset(buf, 0, total);
    layer *l0 = &net->layers[0];
    pack_head h = {PACK_MAGIC, net->n, l0->input_data_uint8_scales[0], l0->input_data_uint8_zero_point[0], total,
                   l0_record_bytes(l0)};
    memcpy(p, &h, sizeof(h)); p += sizeof(h);
    for (int i = 0; i < net->n; ++i) {
        layer *l = &net->layers[i];
        pack_rec r;
        memset(&r, 0, sizeof(r));
        if (l->activ_data_uint8_scales) { r.s_act = l->activ_data_uint8_scales[0]; r.zp_act = l->activ_data_uint8_zero_point[0]; }
        if (l->type == CONVOLUTIONAL) { r.s_in = l->input_data_uint8_scales[0]; r.zp_in = l->input_data_uint8_zero_point[0]; }
        r.blob_bytes = l->blob_bytes;
        memcpy(p, &r, sizeof(r)); p += sizeof(r);
    }
    p = (char *)buf + ((sizeof(pack_head) + (size_t)net->n * sizeof(pack_rec) + 15) & ~(size_t)15);
    for (int i = 0; i < net->n; ++i) {
        layer *l = &net->layers[i];
        if (!l->blob_bytes) continue;
        memcpy(p, l->blob_host, l->blob_bytes);
        p += (l->blob_bytes + 15) & ~(size_t)15;
    }
    pack_l0 r0 = {l0->n, l0->c, l0->size, l0->batch_normalize};
    memcpy(p, &r0, sizeof(r0)); p += sizeof(r0);
    memcpy(p, l0->biases, (size_t)l0->n * sizeof(float)); p += (size_t)l0->n * sizeof(float);
    if (l0->batch_normalize) {
        memcpy(p, l0->scales, (size_t)l0->n * sizeof(float)); p += (size_t)l0->n * sizeof(float);
        memcpy(p, l0->rolling_mean, (size_t)l0->n * sizeof(float)); p += (size_t)l0->n * sizeof(float);
        memcpy(p, l0->rolling_variance, (size_t)l0->n * sizeof(float)); p += (size_t)l0->n * sizeof(float);
    }
    memcpy(p, l0->weight_data_uint8_scales, (size_t)l0->n * sizeof(float)); p += (size_t)l0->n * sizeof(float);
    memcpy(p, l0->weight_data_uint8_zero_point, (size_t)l0->n); p += l0->n;
    memcpy(p, l0->weights_uint8, (size_t)l0->nweights);
}

void network_import_packed_host(network *net, const void *buf, size_t bytes)
{
    const char *p = buf;
    pack_head h;
    if (bytes < sizeof(h)) error("network_import_packed: truncated");
    memcpy(&h, p, sizeof(h)); p += sizeof(h);
    if (h.magic != PACK_MAGIC || h.nlayers != net->n || h.total != bytes) error("network_import_packed: header mismatch (different cfg or format version?)");
    /* the size is a pure function of the cfg: a buffer of any other length (truncated / corrupt file) is refused before
     * anything is read through the offsets below */
    if (bytes != network_packed_size(net)) error("network_import_packed: size does not match this cfg (truncated or corrupt packed data)");
    const pack_rec *recs = (const pack_rec *)p;
    p = (const char *)buf + ((sizeof(pack_head) + (size_t)net->n * sizeof(pack_rec) + 15) & ~(size_t)15);
    for (int i = 0; i < net->n; ++i) {
        layer *l = &net->layers[i];
        pack_rec r;
        memcpy(&r, &recs[i], sizeof(r));
        if (l->activ_data_uint8_scales) { l->activ_data_uint8_scales[0] = r.s_act; l->activ_data_uint8_zero_point[0] = (uint8_t)r.zp_act; }
        if (l->type == CONVOLUTIONAL) {
            l->input_data_uint8_scales[0] = r.s_in; l->input_data_uint8_zero_point[0] = (uint8_t)r.zp_in;
            if (r.blob_bytes != mi355_conv_pack_size(l->n, l->c, l->size)) error("network_import_packed: blob size mismatch");
            free(l->blob_host);
            l->blob_host = malloc(r.blob_bytes);
            l->blob_bytes = r.blob_bytes;
            memcpy(l->blob_host, p, r.blob_bytes);
            p += (r.blob_bytes + 15) & ~(size_t)15;
        }
    }
    layer *l0 = &net->layers[0];
    pack_l0 r0;
    memcpy(&r0, p, sizeof(r0)); p += sizeof(r0);
    if (r0.n != l0->n || r0.c != l0->c || r0.size != l0->size || r0.batch_normalize != l0->batch_normalize ||
        h.l0_bytes != l0_record_bytes(l0)) error("network_import_packed: layer-0 record mismatch");
    memcpy(l0->biases, p, (size_t)l0->n * sizeof(float)); p += (size_t)l0->n * sizeof(float);
    if (l0->batch_normalize) {
        memcpy(l0->scales, p, (size_t)l0->n * sizeof(float)); p += (size_t)l0->n * sizeof(float);
        memcpy(l0->rolling_mean, p, (size_t)l0->n * sizeof(float)); p += (size_t)l0->n * sizeof(float);
        memcpy(l0->rolling_variance, p, (size_t)l0->n * sizeof(float)); p += (size_t)l0->n * sizeof(float);
    }
    memcpy(l0->weight_data_uint8_scales, p, (size_t)l0->n * sizeof(float)); p += (size_t)l0->n * sizeof(float);
    memcpy(l0->weight_data_uint8_zero_point, p, (size_t)l0->n); p += l0->n;
    memcpy(l0->weights_uint8, p, (size_t)l0->nweights);
    prep_shortcut_layers(net);
    net->has_host_weights = 0; /* blobs only (plus layer 0's raw record) */
    net->has_l0_weights = 1;
}

/* On-disk form of the same bytes: what `load_weights` + the host prep + packing produce, written once
 * and read back with one fread -- replaces, for a deployed model, the per-layer reader ref: src/parser.c:1124-1159, the
 * per-channel prep ref: src/blas.c:285-334 and the MFMA-order packing at every start-up. */
void network_save_packed(network *net, char *filename)
{
    const size_t sz = network_packed_size(net);
    void *buf = malloc(sz);
    network_export_packed(net, buf);
    FILE *fp = fopen(filename, "wb");
    if (!fp) file_error(filename);
    if (fwrite(buf, 1, sz, fp) != sz) error("network_save_packed: short write");
    fclose(fp);
    free(buf);
}

static void *read_packed_file(const char *filename, size_t *bytes)
{
    FILE *fp = fopen(filename, "rb");
    if (!fp) file_error(filename);
    fseek(fp, 0, SEEK_END);
    const long sz = ftell(fp);
    fseek(fp, 0, SEEK_SET);
    if (sz < (long)sizeof(pack_head)) error("network_load_packed: file too small");
    void *buf = malloc((size_t)sz);
    if (fread(buf, 1, (size_t)sz, fp) != (size_t)sz) error("network_load_packed: short read");
    fclose(fp);
    *bytes = (size_t)sz;
    return buf;
}

void network_load_packed(network *net, char *filename)
{
    size_t sz = 0;
    void *buf = read_packed_file(filename, &sz);
    network_import_packed(net, buf, sz);
    free(buf);
}

void network_import_packed(network *net, const void *buf, size_t bytes)
{
    if (net->n_replicas > 0 || net->replica_of) error("network_import_packed: not while replicas share this network's packed weights");
    network_import_packed_host(net, buf, bytes);
    plan_fusion(net);
    alloc_network_device(net);
    for (int i = 0; i < net->n; ++i)
        if (net->layers[i].type == CONVOLUTIONAL) upload_conv(net, i, 0);
    check_mi355(mi355_stream_sync(net->stream), "sync");
    net->prepared = 1;
}

void network_import_packed_gpu(network *net, const void *dev_buf, size_t bytes)
{
    check_mi355(mi355_init(net->gpu_index), "mi355_init");
    void *host = malloc(bytes);
    check_mi355(mi355_d2h(host, dev_buf, bytes, NULL), "d2h packed");
    check_mi355(mi355_stream_sync(NULL), "sync");
    network_import_packed(net, host, bytes);
    free(host);
}

/* Multi-GPU start-up behind the C ABI (SURVEY 8(b) `mi355_bcast_weights`, 8(e)): the root rank holds a prepared network
 * (weights file read, per-channel integers derived, blobs packed); every rank calls this with its communicator rank and
 * ends up with the packed state on its device.  One RCCL broadcast of network_packed_size() bytes, in place in HBM. */
void network_bcast_packed(network *net, void *comm, int rank, int root)
{
    check_mi355(mi355_init(net->gpu_index), "mi355_init");
    if (!net->stream && !net->on_default_stream) check_mi355(mi355_stream_acquire(&net->stream), "stream");
    const size_t sz = network_packed_size(net);
    void *dev = NULL;
    check_mi355(mi355_alloc(&dev, sz), "alloc packed");
    void *host = NULL;
    if (rank == root) {
        host = malloc(sz);
        network_export_packed(net, host);
        check_mi355(mi355_h2d(dev, host, sz, net->stream), "upload packed");
    }
    int rc = mi355_bcast_blob(comm, dev, sz, root, net->stream);
    if (rc) { fprintf(stderr, "%s\n", mi355_comm_last_error()); error("mi355_bcast_blob"); }
    check_mi355(mi355_stream_sync(net->stream), "sync");
    free(host);
    if (rank != root) network_import_packed_gpu(net, dev, sz);
    mi355_free(dev);
}

/* A second executor of the same prepared model (darknet_q.h).  The cfg is parsed again (layer geometry, host-side arrays),
 * the per-layer quantisation records and the fusion plan's inputs are copied from the parent, every conv layer borrows the
 * parent's packed blob on the device; activations, input buffers and the stream are the replica's own. */
static void set_plan_internal(network *net, int plan)
{
    /* the executor may still have a pass (or its captured graph) in flight on its stream */
    if (net->prepared && (net->stream || net->on_default_stream)) check_mi355(mi355_stream_sync(net->on_default_stream ? NULL : net->stream), "sync");
    if (net->graph) { mi355_graph_destroy(net->graph); net->graph = NULL; } /* captured with the other plan's kernels */
    net->plan = plan;
    /* fused calls a launcher refused under the old plan (conv_ws3's fused pools under the throughput plan: flags cleared at run
     * time, layers.c) are candidates again; the tensor views were planned with the flags set, so both forms stay valid */
    if (net->prepared) plan_fusion(net);
}

void network_set_plan(network *net, int plan)
{
    net->plan_user = plan;
    set_plan_internal(net, plan);
}

network *network_replica(network *parent)
{
    if (!parent) error("network_replica: no parent");
    const int ds = parent->replica_default_stream; /* one-shot request of the parent (darknet_q.h) */
    parent->replica_default_stream = 0;
    return network_replica_ex(parent, ds);
}

network *network_replica_ex(network *parent, int default_stream)
{
    if (!parent || !parent->prepared) error("network_replica: the parent network is not prepared");
    if (!parent->cfg_path) error("network_replica: the parent network was not parsed from a cfg file");
    network *net = parse_network_cfg(parent->cfg_path, 0);
    if (net->n != parent->n) error("network_replica: the cfg changed on disk");
    net->gpu_index = parent->gpu_index;
    net->accum_mode = parent->accum_mode; net->store_mode = parent->store_mode;
    net->fuse_maxpool = parent->fuse_maxpool; net->keep_head_float = parent->keep_head_float;
    net->use_graph = parent->use_graph; net->input_direct_off = parent->input_direct_off;
    net->dump_int32 = 0;
    if (parent->accum_mode == MI355_ACC_REF_F32) error("network_replica: MI355_ACC_REF_F32 reads raw weights, which a replica does not hold");
    net->replica_of = parent;
    parent->n_replicas++;
    net->on_default_stream = default_stream != 0;
    /* more than one batch in flight from here on: both executors ask the launchers for kernels that share a CU (the parent keeps
     * a plan its caller set explicitly after the first replica: only the first replica switches it) */
    if (parent->n_replicas == 1 && parent->plan != MI355_PLAN_THROUGHPUT) set_plan_internal(parent, MI355_PLAN_THROUGHPUT);
    net->plan = net->plan_user = parent->plan;
    net->batch = parent->batch;
    free(net->input); free(net->input_uint8);
    net->input = calloc((size_t)net->inputs * net->batch, sizeof(float));
    net->input_uint8 = calloc((size_t)net->inputs * net->batch, sizeof(uint8_t));
    for (int i = 0; i < net->n; ++i) {
        layer *l = &net->layers[i];
        const layer *p = &parent->layers[i];
        if (l->type != p->type || l->outputs != p->outputs) error("network_replica: the cfg changed on disk");
        l->batch = net->batch;
        if (l->activ_data_uint8_scales && p->activ_data_uint8_scales) {
            l->activ_data_uint8_scales[0] = p->activ_data_uint8_scales[0];
            l->activ_data_uint8_zero_point[0] = p->activ_data_uint8_zero_point[0];
        }
        if (l->type == CONVOLUTIONAL) {
            l->input_data_uint8_scales[0] = p->input_data_uint8_scales[0];
            l->input_data_uint8_zero_point[0] = p->input_data_uint8_zero_point[0];
            l->blob_gpu = p->blob_gpu;
            l->blob_bytes = p->blob_bytes;
            l->blob_shared = 1;
            l->prepared = 1;
        }
    }
    prep_shortcut_layers(net);
    net->has_host_weights = 0;
    net->has_l0_weights = 0;
    plan_fusion(net);
    alloc_network_device(net);
    check_mi355(mi355_stream_sync(net->stream), "sync");
    net->prepared = 1;
    return net;
}

void free_network(network *net)
{
    if (!net) return;
    if (net->n_replicas > 0) error("free_network: free this network's replicas first (they borrow its packed weights on the device)");
    if (net->replica_of && --net->replica_of->n_replicas == 0) /* the parent is alone on the device again: back to the plan its caller chose
                                                                  (whole-chip kernels by default), fused launches re-planned */
        set_plan_internal(net->replica_of, net->replica_of->plan_user);
    for (int i = 0; i < net->n; ++i) {
        layer *l = &net->layers[i];
        free_layer_device(l);
        free(l->input_data_uint8_scales); free(l->activ_data_uint8_scales); free(l->weight_data_uint8_scales);
        free(l->input_data_uint8_zero_point); free(l->activ_data_uint8_zero_point); free(l->weight_data_uint8_zero_point);
        free(l->weights_sum_int); free(l->mult_zero_point); free(l->M); free(l->M0); free(l->M0_right_shift);
        free(l->M_value); free(l->M0_right_shift_value); free(l->weights_uint8); free(l->biases_int32);
        free(l->biases); free(l->scales); free(l->rolling_mean); free(l->rolling_variance);
        free(l->input_layers); free(l->input_sizes); free(l->mask); free(l->anchors);
        free(l->output); free(l->output_int32); free(l->output_uint8_final); free(l->blob_host);
    }
    network_profile_begin(net, 0);
    if (net->graph) mi355_graph_destroy(net->graph);
    if (net->input_uint8_gpu) mi355_free(net->input_uint8_gpu);
    if (net->input_t.data) mi355_free(net->input_t.data);
    if (net->input_gpu) mi355_free(net->input_gpu);
    if (net->quant_mm_gpu) mi355_free(net->quant_mm_gpu);
    if (net->selfcheck_gpu) mi355_free(net->selfcheck_gpu);
    if (net->stream) mi355_stream_release(net->stream);
    free(net->layers); free(net->input); free(net->input_uint8); free(net->seen); free(net->cfg_path);
    free(net);
}
