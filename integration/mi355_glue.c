/*
 * mi355_glue.c -- reference-side binding of libmi355yolo.so: the file a maintainer of ArtyZe/yolo_quantization adds as
 * src/mi355_glue.c (build: `make MI355=1`, see reference_mi355.patch).  It compiles against the reference's OWN
 * include/darknet.h: `struct layer` / `struct network` are used as they are (no new fields -- device state lives in a
 * side table indexed by layer number, the way src/cuda.c keeps its per-device handles in file-scope arrays).
 *
 * What it replaces, function pointer by function pointer (include/darknet.h:158-163):
 *   forward_convolutional_layer_quant_inputi_outputi   src/convolutional_layer.c:694-761  -> forward_conv_mi355
 *   forward_maxpool_layer_quant                        src/maxpool_layer.c:109-172        -> forward_maxpool_mi355
 *   forward_upsample_layer_quant                       src/upsample_layer.c:96-113        -> forward_upsample_mi355
 *   forward_route_layer_quant                          src/route_layer.c:107-130          -> forward_route_mi355
 *   forward_yolo_layer (inference part)                src/yolo_layer.c:132-146           -> forward_yolo_mi355
 *   forward_network's layer loop + uint8 hand-off      src/network.c:229-261              -> forward_network_mi355
 *
 * This file is exercised for real: oracle/build_ref.sh links it with the unmodified reference objects into
 * oracle/_ref/libdarknet_ref_mi355.so and tests/test_gpu_refbind.py runs the reference's own structs, parser, weights
 * loader and prep with these forward_gpu pointers on an MI355X, comparing every layer with the reference's CPU forward.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "darknet.h"
#include "mi355_yolo_int8.h"
#include "mi355_glue.h"

typedef struct {
    void *blob;                /* mi355_conv_pack blob on the device */
    uint8_t *w_raw, *zp_raw;   /* raw weights_uint8 / zero points (MI355_ACC_REF_F32 only) */
    mi355_tensor out;          /* PHWC uint8 activations */
    float *out_f32;            /* l.output on the device (quant_stop layers, yolo) */
    int32_t *out_i32;          /* l.output_int32 on the device (pull_all runs) */
    uint8_t *nchw;             /* scratch: reference layout of `out` for the copy back */
} mi355_layer_state;

static struct {
    network *net;
    mi355_layer_state *ls;
    void *stream;
    uint8_t *input_nchw;
    mi355_tensor input;
    const mi355_tensor *cur;   /* device twin of `net.input_uint8 = l.output_uint8_final` (src/network.c:248-250) */
    const float *cur_f32;      /* device twin of `net.input = l.output` */
    int accum_mode, store_mode, want_i32;
} G;

static void chk(int rc, const char *what) /* die like check_error (src/cuda.c:27-49) */
{
    if (rc) {
        fprintf(stderr, "MI355 error %d in %s: %s\n", rc, what, mi355_last_error());
        error(what);
    }
}

/* ---- layer.forward_gpu implementations ------------------------------------------------------------------------- */
static void forward_conv_mi355(layer l, network net)
{
    mi355_layer_state *s = &G.ls[net.index];
    mi355_conv_desc d;
    memset(&d, 0, sizeof(d));
    d.n = l.n; d.c = l.c; d.ksize = l.size; d.stride = l.stride; d.pad = l.pad;
    d.activation = l.activation; d.store_mode = G.store_mode; d.accum_mode = G.accum_mode;
    d.epilogue_packed = 1;  /* mi355_bind_network finished the blob with mi355_conv_pack_epilogue */
    d.zp_in = l.input_data_uint8_zero_point[0];
    d.zp_act = l.activ_data_uint8_zero_point[0];
    d.s_act = l.activ_data_uint8_scales[0];
    chk(mi355_conv_forward(&d, G.cur, s->blob, s->w_raw, s->zp_raw, &s->out, G.want_i32 ? s->out_i32 : NULL,
                           l.quant_stop_flag ? s->out_f32 : NULL, G.stream), "mi355_conv_forward");
}

static void dequant_tail(layer l, mi355_layer_state *s)
{
    chk(mi355_dequant_forward(&s->out, 0, l.out_c, l.activ_data_uint8_zero_point[0], l.activ_data_uint8_scales[0], s->out_f32,
                              l.out_c, 0, G.stream), "mi355_dequant_forward");
}

static void forward_maxpool_mi355(layer l, network net)
{
    mi355_layer_state *s = &G.ls[net.index];
    chk(mi355_maxpool_forward(G.cur, &s->out, l.size, l.stride, l.pad, G.stream), "mi355_maxpool_forward");
    if (l.quant_stop_flag) dequant_tail(l, s);
}

static void forward_upsample_mi355(layer l, network net)
{
    mi355_layer_state *s = &G.ls[net.index];
    chk(mi355_upsample_forward(G.cur, &s->out, l.stride, G.stream), "mi355_upsample_forward");
    if (l.quant_stop_flag) dequant_tail(l, s);
}

static void forward_route_mi355(layer l, network net)
{
    mi355_layer_state *s = &G.ls[net.index];
    const mi355_tensor *xs[16];
    if (l.n > 16) error("route: more than 16 inputs");
    for (int i = 0; i < l.n; ++i) xs[i] = &G.ls[l.input_layers[i]].out;
    chk(mi355_route_forward(xs, l.n, &s->out, G.stream), "mi355_route_forward");
    if (l.quant_stop_flag) {
        int coff = 0;
        for (int i = 0; i < l.n; ++i) {
            layer in = net.layers[l.input_layers[i]];
            chk(mi355_dequant_forward(&s->out, coff, in.out_c, in.activ_data_uint8_zero_point[0], in.activ_data_uint8_scales[0],
                                      s->out_f32, l.out_c, coff, G.stream), "mi355_dequant_forward");
            coff += in.out_c;
        }
    }
}

static void forward_yolo_mi355(layer l, network net)
{
    mi355_layer_state *s = &G.ls[net.index];
    if (!G.cur_f32) error("yolo layer needs a float input (previous layer must have quant_stop=1)");
    chk(mi355_yolo_forward(G.cur_f32, s->out_f32, l.batch, l.n, l.classes, l.h, l.w, G.stream), "mi355_yolo_forward");
}

/* ---- bind / run / unbind ------------------------------------------------------------------------------------------ */
void mi355_bind_network(network *net, int gpu, int accum_mode, int store_mode)
{
    if (mi355_abi_version() != MI355_ABI_VERSION) error("libmi355yolo.so and mi355_yolo_int8.h disagree on the ABI version");
    chk(mi355_init(gpu), "mi355_init");
    memset(&G, 0, sizeof(G));
    G.net = net;
    G.accum_mode = accum_mode; G.store_mode = store_mode;
    G.ls = calloc(net->n, sizeof(mi355_layer_state));
    chk(mi355_stream_acquire(&G.stream), "stream"); /* a stream measured to run beside the default stream and other acquired ones */
    chk(mi355_alloc((void **)&G.input_nchw, (size_t)net->batch * net->inputs), "alloc input");
    size_t bytes = mi355_tensor_describe(&G.input, net->batch, net->h, net->w, net->c);
    chk(mi355_alloc(&G.input.data, bytes), "alloc input tensor");
    chk(mi355_tensor_fill(&G.input, net->layers[0].input_data_uint8_zero_point[0], G.stream), "fill input"); /* pad = zp */
    for (int i = 0; i < net->n; ++i) {
        layer *l = &net->layers[i];
        mi355_layer_state *s = &G.ls[i];
        const size_t cnt = (size_t)net->batch * l->outputs;
        if (l->type == CONVOLUTIONAL) {
            if (!l->layer_quant_flag) error("mi355: unquantized convolution");
            const size_t sz = mi355_conv_pack_size(l->n, l->c, l->size);
            if (!sz) error("mi355: unsupported convolution shape");
            void *host = malloc(sz);
            chk(mi355_conv_pack(l->n, l->c, l->size, l->weights_uint8, l->weight_data_uint8_zero_point, l->biases_int32,
                                l->M_value, l->M0_right_shift_value, host), "mi355_conv_pack");
            chk(mi355_conv_pack_epilogue(l->n, l->c, l->size, l->activation, l->activ_data_uint8_zero_point[0], host),
                "mi355_conv_pack_epilogue");
            chk(mi355_alloc(&s->blob, sz), "alloc blob");
            chk(mi355_h2d(s->blob, host, sz, G.stream), "upload blob");
            chk(mi355_stream_sync(G.stream), "sync");
            free(host);
            if (accum_mode == MI355_ACC_REF_F32) {
                chk(mi355_alloc((void **)&s->w_raw, (size_t)l->nweights), "alloc raw weights");
                chk(mi355_h2d(s->w_raw, l->weights_uint8, (size_t)l->nweights, G.stream), "upload raw weights");
                chk(mi355_alloc((void **)&s->zp_raw, (size_t)l->n), "alloc zp");
                chk(mi355_h2d(s->zp_raw, l->weight_data_uint8_zero_point, (size_t)l->n, G.stream), "upload zp");
            }
            chk(mi355_alloc((void **)&s->out_i32, cnt * sizeof(int32_t)), "alloc int32");
            l->forward_gpu = forward_conv_mi355;
        } else if (l->type == MAXPOOL) l->forward_gpu = forward_maxpool_mi355;
        else if (l->type == UPSAMPLE) l->forward_gpu = forward_upsample_mi355;
        else if (l->type == ROUTE) l->forward_gpu = forward_route_mi355;
        else if (l->type == YOLO) l->forward_gpu = forward_yolo_mi355;
        else error("mi355: layer type outside the quantized inference path");
        if (l->type != YOLO) {
            bytes = mi355_tensor_describe(&s->out, net->batch, l->out_h, l->out_w, l->out_c);
            chk(mi355_alloc(&s->out.data, bytes), "alloc activations");
            chk(mi355_tensor_fill(&s->out, l->activ_data_uint8_zero_point[0], G.stream), "fill"); /* pad cells = zero point */
            chk(mi355_alloc((void **)&s->nchw, cnt), "alloc scratch");
        }
        if (l->type == YOLO || l->quant_stop_flag) chk(mi355_alloc((void **)&s->out_f32, cnt * sizeof(float)), "alloc float");
    }
    chk(mi355_stream_sync(G.stream), "sync");
}

void forward_network_mi355(network *netp, int pull_all)
{
    if (G.net != netp) error("forward_network_mi355: network is not bound");
    network net = *netp;
    G.want_i32 = pull_all;
    chk(mi355_h2d(G.input_nchw, net.input_uint8, (size_t)net.batch * net.inputs, G.stream), "push input");
    chk(mi355_nchw_to_tensor(G.input_nchw, &G.input, G.stream), "input layout");
    G.cur = &G.input;
    G.cur_f32 = NULL;
    for (int i = 0; i < net.n; ++i) { /* src/network.c:238-259 */
        net.index = i;
        layer l = net.layers[i];
        l.forward_gpu(l, net);
        if (l.layer_quant_flag && !net.train) G.cur = &G.ls[i].out;
        G.cur_f32 = G.ls[i].out_f32;
    }
    for (int i = 0; i < net.n; ++i) { /* copy back what the host side reads (the reference's pull_network_output, :863-871) */
        layer l = net.layers[i];
        mi355_layer_state *s = &G.ls[i];
        const size_t cnt = (size_t)net.batch * l.outputs;
        if (s->out_f32) chk(mi355_d2h(l.output, s->out_f32, cnt * sizeof(float), G.stream), "pull float");
        if (!pull_all) continue;
        if (l.type != YOLO) {
            chk(mi355_tensor_to_nchw(&s->out, s->nchw, G.stream), "layout");
            chk(mi355_d2h(l.output_uint8_final, s->nchw, cnt, G.stream), "pull uint8");
        }
        if (l.type == CONVOLUTIONAL) chk(mi355_d2h(l.output_int32, s->out_i32, cnt * sizeof(int32_t), G.stream), "pull int32");
    }
    chk(mi355_stream_sync(G.stream), "sync");
    netp->output = net.layers[net.n - 1].output;
}

void mi355_unbind_network(network *net)
{
    if (G.net != net) return;
    for (int i = 0; i < net->n; ++i) {
        mi355_layer_state *s = &G.ls[i];
        if (s->blob) mi355_free(s->blob);
        if (s->w_raw) mi355_free(s->w_raw);
        if (s->zp_raw) mi355_free(s->zp_raw);
        if (s->out.data) mi355_free(s->out.data);
        if (s->out_f32) mi355_free(s->out_f32);
        if (s->out_i32) mi355_free(s->out_i32);
        if (s->nchw) mi355_free(s->nchw);
    }
    mi355_free(G.input_nchw);
    mi355_free(G.input.data);
    mi355_stream_release(G.stream);
    free(G.ls);
    memset(&G, 0, sizeof(G));
}
