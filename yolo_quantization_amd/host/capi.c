/* capi.c -- flat accessors over struct network / struct layer for FFI callers (ctypes in tests and bench.py) that
 * should not replicate the struct layouts. */
#include <string.h>
#include "host_internal.h"

int dnq_net_n(network *net) { return net->n; }
int dnq_net_batch(network *net) { return net->batch; }
int dnq_net_inputs(network *net) { return net->inputs; }
void *dnq_net_stream(network *net) { return net->stream; }
void *dnq_net_input_gpu(network *net) { return net->input_uint8_gpu; }
uint8_t *dnq_net_input_host(network *net) { return net->input_uint8; }
float *dnq_net_input_float(network *net) { return net->input; }

int dnq_net_set(network *net, const char *key, int val)
{
    if (!strcmp(key, "accum_mode")) net->accum_mode = val;
    else if (!strcmp(key, "store_mode")) net->store_mode = val;
    else if (!strcmp(key, "dump_int32")) net->dump_int32 = val;
    else if (!strcmp(key, "fuse_maxpool")) net->fuse_maxpool = val;
    else if (!strcmp(key, "use_graph")) net->use_graph = val;
    else if (!strcmp(key, "gpu_index")) net->gpu_index = val;
    else if (!strcmp(key, "verbose")) net->verbose = val;
    else if (!strcmp(key, "keep_head_float")) net->keep_head_float = val;
    else if (!strcmp(key, "range_lo")) net->range_lo = val;
    else if (!strcmp(key, "range_hi")) net->range_hi = val;
    else if (!strcmp(key, "replica_default_stream")) net->replica_default_stream = val;
    else if (!strcmp(key, "plan")) network_set_plan(net, val); /* MI355_PLAN_LATENCY / MI355_PLAN_THROUGHPUT: syncs, drops the graph, re-plans fusion */
    else if (!strcmp(key, "input_direct")) { /* 0: always convert the input to 4-byte cells (A/B runs); survives re-allocation */
        net->input_direct_off = !val;
        net->input_direct = val && net->c == 3 && !net->dump_int32;
    }
    else return -1;
    return 0;
}

/* info[0..15] = type, out_c, out_h, out_w, c, h, w, n, size, stride, pad, activation, batch_normalize,
 * layer_quant_flag, quant_stop_flag, outputs  (same order as oracle/ref_driver.c) */
int dnq_layer_info(network *net, int i, int *info)
{
    if (i < 0 || i >= net->n) return -1;
    layer *l = &net->layers[i];
    info[0] = l->type; info[1] = l->out_c; info[2] = l->out_h; info[3] = l->out_w;
    info[4] = l->c; info[5] = l->h; info[6] = l->w; info[7] = l->n;
    info[8] = l->size; info[9] = l->stride; info[10] = l->pad; info[11] = l->activation;
    info[12] = l->batch_normalize; info[13] = l->layer_quant_flag; info[14] = l->quant_stop_flag;
    info[15] = l->outputs;
    return 0;
}
const uint8_t *dnq_layer_u8(network *net, int i) { return net->layers[i].output_uint8_final; }
const int32_t *dnq_layer_int32(network *net, int i) { return net->layers[i].output_int32; }
const float *dnq_layer_f32(network *net, int i) { return net->layers[i].output; }
void *dnq_layer_f32_gpu(network *net, int i) { return net->layers[i].output_gpu; }

int dnq_layer_prep(network *net, int i, int32_t *biases_int32, double *M_value, double *shift_value, int32_t *M0,
                   int *shift, float *q)
{
    layer *l = &net->layers[i];
    if (q && l->activ_data_uint8_scales) {
        q[2] = l->activ_data_uint8_scales[0];
        q[3] = l->activ_data_uint8_zero_point[0];
        if (l->type == CONVOLUTIONAL) { q[0] = l->input_data_uint8_scales[0]; q[1] = l->input_data_uint8_zero_point[0]; }
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

/* 1 when layer i (a conv) runs fused with the maxpool / upsample after it in the current configuration: its own uint8
 * tensor is then not stored (only the pooled / upsampled one is) */
/* kernel family (mi355_last_conv_kernel codes) that served conv layer i in the last forward pass, 0 for other layers */
int dnq_layer_conv_kernel(network *net, int i)
{
    if (i < 0 || i >= net->n) return 0;
    return net->layers[i].conv_kernel;
}

/* layer i + 1 runs inside layer i's kernel and layer i's own tensor is NOT stored */
int dnq_layer_is_fused(network *net, int i)
{
    if (i < 0 || i >= net->n) return 0;
    const layer *l = &net->layers[i];
    return ((l->fuse_next_pool && !l->fuse_pool_keep) || l->fuse_next_upsample || l->fuse_next_shortcut) && net->fuse_maxpool && !net->dump_int32 &&
           net->accum_mode == MI355_ACC_EXACT;
}

/* layer i + 1 runs inside layer i's kernel (whether or not layer i's own tensor is stored as well) */
int dnq_layer_fuses_next(network *net, int i)
{
    if (i < 0 || i >= net->n) return 0;
    const layer *l = &net->layers[i];
    return (l->fuse_next_pool || l->fuse_next_upsample || l->fuse_next_shortcut || l->fuse_next_yolo) && net->fuse_maxpool && !net->dump_int32 &&
           net->accum_mode == MI355_ACC_EXACT;
}

/* k[0..2] = Ka, Kb, `from` index of shortcut layer i */
int dnq_layer_shortcut(network *net, int i, int32_t *k)
{
    if (i < 0 || i >= net->n || net->layers[i].type != SHORTCUT) return -1;
    k[0] = net->layers[i].shortcut_Ka; k[1] = net->layers[i].shortcut_Kb; k[2] = net->layers[i].index;
    return 0;
}
