/*
 * layers.c -- layer constructors and the forward_gpu function pointers of the INT8 path.
 *
 *   make_convolutional_layer   ref: src/convolutional_layer.c:179-279 (quant fields 214-257, dispatch 258-273)
 *   make_maxpool_layer         ref: src/maxpool_layer.c:20-75
 *   make_upsample_layer        ref: src/upsample_layer.c:7-60
 *   make_route_layer           ref: src/route_layer.c:7-50
 *   make_yolo_layer            ref: src/yolo_layer.c:13-60
 * Each forward_*_gpu is the device replacement of the reference's forward_*_quant (cited at the function).
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "host_internal.h"

static void forward_cpu_not_built(layer l, network net)
{
    (void)net;
    fprintf(stderr,
            "layer %d (%s): this is the MI355X build -- there is no CPU data path; call forward_gpu / "
            "forward_network_gpu (libmi355yolo.so)\n",
            l.count, get_layer_string(l.type));
    error("forward: CPU path not built");
}

static void alloc_act_record(layer *l)
{
    l->activ_data_uint8_scales = calloc(1, sizeof(float));
    l->activ_data_uint8_zero_point = calloc(1, sizeof(uint8_t));
}

/* ref: forward_convolutional_layer_quant_inputi_outputi, src/convolutional_layer.c:694-761 */
static void forward_convolutional_layer_quant_gpu(layer l, network net)
{
    if (!l.prepared) error("forward_gpu before quantization_weights_and_activations");
    mi355_conv_desc d;
    memset(&d, 0, sizeof(d));
    d.n = l.n; d.c = l.c; d.ksize = l.size; d.stride = l.stride; d.pad = l.pad;
    d.activation = l.activation;
    d.plan = net.plan;
    d.epilogue_packed = 1;  /* quantization_weights_and_activations finishes every blob with mi355_conv_pack_epilogue(l->activation, zp_act) */
    d.store_mode = net.store_mode;
    d.accum_mode = net.accum_mode;
    d.zp_in = l.input_data_uint8_zero_point[0];
    d.zp_act = l.activ_data_uint8_zero_point[0];
    d.s_act = l.activ_data_uint8_scales[0];
    /* Fused forms: the executor's plan marks candidates by shape, the launchers decide.  MI355_EINVAL from a fused entry
     * point means "no kernel fuses this shape" (nothing was launched): the flag is cleared in the network's layer array
     * (net.layers points at it; `l` is a by-value copy) and the convolution runs unfused below, the layer after it on its
     * own. */
    layer *self = &net.layers[net.index];
    mi355_tensor converted;
    if (net.cur_t->cs == 1) { /* planar network input: try the in-place read; MI355_EINVAL -> convert, once and for all */
        int rc = MI355_EINVAL;
        if (net.fused_pool_t) rc = mi355_conv_pool_forward(&d, net.cur_t, l.blob_gpu, NULL, net.fused_pool_t, net.stream);
        else if (!net.fused_up_t && !net.fused_yolo_out && !net.fused_shortcut && !l.quant_stop_flag)
            rc = mi355_conv_forward(&d, net.cur_t, l.blob_gpu, NULL, NULL, &l.out_t, NULL, NULL, net.stream);
        if (rc != MI355_EINVAL) { check_mi355(rc, "mi355_conv_forward (planar input)"); return; }
        if (net.input_direct_p) *net.input_direct_p = 0;
        converted = net.input_t;
        check_mi355(mi355_nchw_to_tensor((const uint8_t *)net.cur_t->data, &converted, net.stream), "input layout");
        net.cur_t = &converted;
    }
    if (net.fused_pool_t) { /* this conv + the 2x2 maxpool after it as one kernel; the pre-pool tensor is stored only if a route reads it */
        const int rc = mi355_conv_pool_forward(&d, net.cur_t, l.blob_gpu, l.fuse_pool_keep ? &l.out_t : NULL, net.fused_pool_t, net.stream);
        if (rc != MI355_EINVAL) { check_mi355(rc, "mi355_conv_pool_forward"); return; }
        self->fuse_next_pool = 0;
    }
    if (net.fused_up_t) { /* this conv + the nearest-neighbour upsample after it; the conv's own tensor is not stored */
        const int rc = mi355_conv_upsample_forward(&d, net.cur_t, l.blob_gpu, net.fused_up_t, net.fused_up_stride, net.stream);
        if (rc != MI355_EINVAL) { check_mi355(rc, "mi355_conv_upsample_forward"); return; }
        self->fuse_next_upsample = 0;
    }
    if (net.fused_shortcut) { /* this conv + the quantized residual add after it; the conv's own tensor is not stored */
        const layer *sc = net.fused_shortcut, *from = &net.layers[sc->index];
        const int rc = mi355_conv_shortcut_forward(&d, net.cur_t, l.blob_gpu, &from->out_t, &sc->out_t, sc->shortcut_Ka, sc->shortcut_Kb,
                                                   from->activ_data_uint8_zero_point[0], sc->activ_data_uint8_zero_point[0], net.stream);
        if (rc != MI355_EINVAL) { check_mi355(rc, "mi355_conv_shortcut_forward"); return; }
        self->fuse_next_shortcut = 0;
    }
    if (net.fused_yolo_out) { /* quant_stop head + the yolo layer after it (ref: src/yolo_layer.c:132-146) in one kernel */
        int rc = MI355_EINVAL;
        if (!net.keep_head_float && !net.dump_int32) /* the head's own float tensor is not stored; MI355_EINVAL: this kernel needs it */
            rc = mi355_conv_yolo_forward(&d, net.cur_t, l.blob_gpu, &l.out_t, NULL, net.fused_yolo_out, net.fused_yolo_classes, net.stream);
        if (rc == MI355_EINVAL)
            rc = mi355_conv_yolo_forward(&d, net.cur_t, l.blob_gpu, &l.out_t, l.output_gpu, net.fused_yolo_out, net.fused_yolo_classes, net.stream);
        if (rc != MI355_EINVAL) { check_mi355(rc, "mi355_conv_yolo_forward"); return; }
        self->fuse_next_yolo = 0;
    }
    check_mi355(mi355_conv_forward(&d, net.cur_t, l.blob_gpu, l.weights_uint8_gpu, l.weight_zero_point_gpu, &l.out_t,
                                   net.dump_int32 ? l.output_int32_gpu : NULL,
                                   l.quant_stop_flag ? l.output_gpu : NULL, net.stream),
                "mi355_conv_forward");
}

/* quant_stop tail shared by the glue layers: l.output = (u8 - zp) * scale (ref: src/maxpool_layer.c:163-171,
 * src/upsample_layer.c:104-112) */
static void dequant_tail(layer l, network net)
{
    check_mi355(mi355_dequant_forward(&l.out_t, 0, l.out_c, l.activ_data_uint8_zero_point[0], l.activ_data_uint8_scales[0],
                                      l.output_gpu, l.out_c, 0, net.stream), "mi355_dequant_forward");
}

/* ref: forward_maxpool_layer_quant, src/maxpool_layer.c:109-172 */
static void forward_maxpool_layer_quant_gpu(layer l, network net)
{
    check_mi355(mi355_maxpool_forward(net.cur_t, &l.out_t, l.size, l.stride, l.pad, net.stream), "mi355_maxpool_forward");
    if (l.quant_stop_flag) dequant_tail(l, net);
}

/* ref: forward_upsample_layer_quant, src/upsample_layer.c:96-113 */
static void forward_upsample_layer_quant_gpu(layer l, network net)
{
    check_mi355(mi355_upsample_forward(net.cur_t, &l.out_t, l.stride, net.stream), "mi355_upsample_forward");
    if (l.quant_stop_flag) dequant_tail(l, net);
}

/* ref: forward_route_layer_quant, src/route_layer.c:107-130 */
static void forward_route_layer_quant_gpu(layer l, network net)
{
    const mi355_tensor *xs[16];
    if (l.n > 16) error("route: more than 16 inputs");
    if (!l.route_elided) { /* else the producers wrote straight into this layer's buffer */
        for (int i = 0; i < l.n; ++i) xs[i] = &net.layers[l.input_layers[i]].out_t;
        check_mi355(mi355_route_forward(xs, l.n, &l.out_t, net.stream), "mi355_route_forward");
    }
    if (l.quant_stop_flag) { /* ref :121-129: every input's channels with THAT input's scale / zero point */
        int coff = 0;
        for (int i = 0; i < l.n; ++i) {
            const layer *in = &net.layers[l.input_layers[i]];
            check_mi355(mi355_dequant_forward(&l.out_t, coff, in->out_c, in->activ_data_uint8_zero_point[0],
                                              in->activ_data_uint8_scales[0], l.output_gpu, l.out_c, coff, net.stream),
                        "mi355_dequant_forward");
            coff += in->out_c;
        }
    }
}

/* Quantized residual add.  The reference's [shortcut] is float only (forward_shortcut_layer, src/shortcut_layer.c:62-75 ->
 * shortcut_cpu, src/blas.c:490-514: out = input + layers[index].output, then the activation): no integer forward exists
 * to be bit-exact against, so the integer form is this build's own specification. */
static void forward_shortcut_layer_quant_gpu(layer l, network net)
{
    const layer *from = &net.layers[l.index];
    const layer *prev = &net.layers[net.index - 1];
    check_mi355(mi355_shortcut_forward(net.cur_t, &from->out_t, &l.out_t, l.shortcut_Ka, l.shortcut_Kb,
                                       prev->activ_data_uint8_zero_point[0], from->activ_data_uint8_zero_point[0],
                                       l.activ_data_uint8_zero_point[0], net.stream), "mi355_shortcut_forward");
    if (l.quant_stop_flag) dequant_tail(l, net);
}

/* ref: forward_yolo_layer (inference part), src/yolo_layer.c:132-146 */
static void forward_yolo_layer_gpu(layer l, network net)
{
    if (!net.cur_f32_gpu) error("yolo layer needs a float input (previous layer must have quant_stop=1)");
    check_mi355(mi355_yolo_forward(net.cur_f32_gpu, l.output_gpu, l.batch, l.n, l.classes, l.h, l.w, net.stream),
                "mi355_yolo_forward");
}

layer make_convolutional_layer(int batch, int h, int w, int c, int n, int groups, int size, int stride, int padding,
                               ACTIVATION activation, int batch_normalize, int quant_stop_flag,
                               int close_quantization, int layer_quantization, int count)
{
    layer l;
    memset(&l, 0, sizeof(l));
    l.type = CONVOLUTIONAL;
    l.groups = groups; l.h = h; l.w = w; l.c = c; l.n = n; l.batch = batch;
    l.stride = stride; l.size = size; l.pad = padding; l.batch_normalize = batch_normalize;
    l.activation = activation; l.count = count;
    l.nweights = c / groups * n * size * size;
    l.out_h = (h + 2 * padding - size) / stride + 1; /* ref: convolutional_out_height */
    l.out_w = (w + 2 * padding - size) / stride + 1;
    l.out_c = n;
    l.outputs = l.out_h * l.out_w * l.out_c;
    l.inputs = h * w * c;
    l.layer_quant_flag = layer_quantization;
    l.close_quantization = close_quantization;
    l.quant_stop_flag = quant_stop_flag;
    if (!layer_quantization) {
        fprintf(stderr, "layer %d: [convolutional] without quantized=1 would take the reference's float path "
                        "(src/convolutional_layer.c:271), which this INT8 build does not contain\n", count);
        error("unquantized convolution");
    }
    alloc_act_record(&l);
    l.weight_data_uint8_scales = calloc(n, sizeof(float));
    l.input_data_uint8_scales = calloc(1, sizeof(float));
    l.weight_data_uint8_zero_point = calloc(n, sizeof(uint8_t));
    l.input_data_uint8_zero_point = calloc(1, sizeof(uint8_t));
    l.weights_sum_int = calloc(n, sizeof(int32_t));
    l.mult_zero_point = calloc(n, sizeof(uint32_t));
    l.M = calloc(n, sizeof(float));
    l.M0 = calloc(n, sizeof(int32_t));
    l.M_value = calloc(n, sizeof(double));
    l.M0_right_shift = calloc(n, sizeof(int));
    l.M0_right_shift_value = calloc(n, sizeof(double));
    l.weights_uint8 = calloc(l.nweights, sizeof(uint8_t));
    l.biases_int32 = calloc(n, sizeof(int32_t));
    l.biases = calloc(n, sizeof(float));
    if (batch_normalize) {
        l.scales = calloc(n, sizeof(float));
        l.rolling_mean = calloc(n, sizeof(float));
        l.rolling_variance = calloc(n, sizeof(float));
        for (int i = 0; i < n; ++i) l.scales[i] = 1;
    }
    l.forward = forward_cpu_not_built;
    l.forward_gpu = forward_convolutional_layer_quant_gpu;
    return l;
}

layer make_maxpool_layer(int batch, int h, int w, int c, int size, int stride, int padding, int layer_quant_flag,
                         int quant_stop_flag, int close_quantization, int count)
{
    layer l;
    memset(&l, 0, sizeof(l));
    l.type = MAXPOOL;
    l.batch = batch; l.h = h; l.w = w; l.c = c; l.pad = padding; l.count = count;
    l.out_w = (w + padding - size) / stride + 1; /* ref :31-32 */
    l.out_h = (h + padding - size) / stride + 1;
    l.out_c = c;
    l.outputs = l.out_h * l.out_w * l.out_c;
    l.inputs = h * w * c;
    l.size = size; l.stride = stride;
    l.layer_quant_flag = layer_quant_flag; l.quant_stop_flag = quant_stop_flag;
    l.close_quantization = close_quantization;
    if (!layer_quant_flag) error("[maxpool] without quantized=1 is the reference's float path; not built");
    alloc_act_record(&l);
    l.forward = forward_cpu_not_built;
    l.forward_gpu = forward_maxpool_layer_quant_gpu;
    return l;
}

layer make_upsample_layer(int batch, int w, int h, int c, int stride, int layer_quant_flag, int quant_stop_flag,
                          int close_quantization, int count)
{
    layer l;
    memset(&l, 0, sizeof(l));
    l.type = UPSAMPLE;
    l.batch = batch; l.w = w; l.h = h; l.c = c; l.count = count;
    l.out_w = w * stride; l.out_h = h * stride; l.out_c = c;
    l.stride = stride;
    l.outputs = l.out_w * l.out_h * l.out_c;
    l.inputs = w * h * c;
    l.layer_quant_flag = layer_quant_flag; l.quant_stop_flag = quant_stop_flag;
    l.close_quantization = close_quantization;
    if (!layer_quant_flag) error("[upsample] without quantized=1 is the reference's float path; not built");
    alloc_act_record(&l);
    l.forward = forward_cpu_not_built;
    l.forward_gpu = forward_upsample_layer_quant_gpu;
    return l;
}

layer make_route_layer(int batch, int n, int *input_layers, int *input_sizes, int layer_quant_flag,
                       int quant_stop_flag, int close_quantization, int count)
{
    layer l;
    memset(&l, 0, sizeof(l));
    l.type = ROUTE;
    l.batch = batch; l.n = n; l.count = count;
    l.input_layers = input_layers; l.input_sizes = input_sizes;
    int outputs = 0;
    for (int i = 0; i < n; ++i) outputs += input_sizes[i];
    l.outputs = outputs; l.inputs = outputs;
    l.layer_quant_flag = layer_quant_flag; l.quant_stop_flag = quant_stop_flag;
    l.close_quantization = close_quantization;
    if (!layer_quant_flag) error("[route] without quantized=1 is the reference's float path; not built");
    alloc_act_record(&l);
    l.forward = forward_cpu_not_built;
    l.forward_gpu = forward_route_layer_quant_gpu;
    return l;
}

/* ref: make_shortcut_layer, src/shortcut_layer.c:9-40 (same dims only: the reference's strided / sampled variants of
 * shortcut_cpu have no use in the quantized nets) */
layer make_shortcut_layer(int batch, int index, int w, int h, int c, int w2, int h2, int c2, int layer_quant_flag,
                          int quant_stop_flag, int close_quantization, int count)
{
    layer l;
    memset(&l, 0, sizeof(l));
    l.type = SHORTCUT;
    l.batch = batch; l.w = w2; l.h = h2; l.c = c2; l.count = count;
    l.out_w = w; l.out_h = h; l.out_c = c;
    l.outputs = w * h * c;
    l.inputs = l.outputs;
    l.index = index;
    l.activation = LINEAR;
    l.layer_quant_flag = layer_quant_flag; l.quant_stop_flag = quant_stop_flag;
    l.close_quantization = close_quantization;
    if (!layer_quant_flag) error("[shortcut] without quantized=1 is the reference's float path (src/shortcut_layer.c:62-75); not built");
    if (w != w2 || h != h2 || c != c2) error("[shortcut] quantized=1 needs both inputs to share width, height and channels");
    alloc_act_record(&l);
    l.forward = forward_cpu_not_built;
    l.forward_gpu = forward_shortcut_layer_quant_gpu;
    return l;
}

layer make_yolo_layer(int batch, int w, int h, int n, int total, int *mask, int classes, int count)
{
    layer l;
    memset(&l, 0, sizeof(l));
    l.type = YOLO;
    l.n = n; l.total = total; l.batch = batch; l.h = h; l.w = w; l.count = count;
    l.c = n * (classes + 4 + 1);
    l.out_w = w; l.out_h = h; l.out_c = l.c;
    l.classes = classes;
    l.anchors = calloc(total * 2, sizeof(float));
    if (mask) l.mask = mask;
    else {
        l.mask = calloc(n, sizeof(int));
        for (int i = 0; i < n; ++i) l.mask[i] = i;
    }
    l.outputs = h * w * n * (classes + 4 + 1);
    l.inputs = l.outputs;
    for (int i = 0; i < total * 2; ++i) l.anchors[i] = .5f;
    l.forward = forward_cpu_not_built;
    l.forward_gpu = forward_yolo_layer_gpu;
    return l;
}

void free_layer_device(layer *l)
{
    if (l->out_t.data && !l->out_view) mi355_free(l->out_t.data);
    l->out_view = 0; l->route_elided = 0;
    if (l->blob_gpu && !l->blob_shared) mi355_free(l->blob_gpu);
    if (l->weights_uint8_gpu) mi355_free(l->weights_uint8_gpu);
    if (l->weight_zero_point_gpu) mi355_free(l->weight_zero_point_gpu);
    if (l->output_int32_gpu) mi355_free(l->output_int32_gpu);
    if (l->output_gpu) mi355_free(l->output_gpu);
    if (l->output_uint8_nchw_gpu) mi355_free(l->output_uint8_nchw_gpu);
    if (l->anchors_gpu) mi355_free(l->anchors_gpu);
    if (l->mask_gpu) mi355_free(l->mask_gpu);
    if (l->det_recs_gpu) mi355_free(l->det_recs_gpu);
    if (l->det_counts_gpu) mi355_free(l->det_counts_gpu);
    l->anchors_gpu = NULL; l->mask_gpu = NULL; l->det_recs_gpu = NULL; l->det_counts_gpu = NULL; l->det_cap = 0;
    l->out_t.data = NULL; l->blob_gpu = NULL; l->weights_uint8_gpu = NULL; l->weight_zero_point_gpu = NULL;
    l->output_int32_gpu = NULL; l->output_gpu = NULL; l->output_uint8_nchw_gpu = NULL;
}
