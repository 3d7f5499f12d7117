/* host_internal.h -- internal declarations of libdarknet_q */
#ifndef HOST_INTERNAL_H
#define HOST_INTERNAL_H
#include "darknet_q.h"

void check_mi355(int rc, const char *what); /* maps shim error codes to error(), like the reference's check_error */

layer make_convolutional_layer(int batch, int h, int w, int c, int n, int groups, int size, int stride, int padding,
                               ACTIVATION activation, int batch_normalize, int quant_stop_flag,
                               int close_quantization, int layer_quantization, int count);
layer make_maxpool_layer(int batch, int h, int w, int c, int size, int stride, int padding, int layer_quant_flag,
                         int quant_stop_flag, int close_quantization, int count);
layer make_upsample_layer(int batch, int w, int h, int c, int stride, int layer_quant_flag, int quant_stop_flag,
                          int close_quantization, int count);
layer make_route_layer(int batch, int n, int *input_layers, int *input_sizes, int layer_quant_flag,
                       int quant_stop_flag, int close_quantization, int count);
layer make_shortcut_layer(int batch, int index, int w, int h, int c, int w2, int h2, int c2, int layer_quant_flag,
                          int quant_stop_flag, int close_quantization, int count);
layer make_yolo_layer(int batch, int w, int h, int n, int total, int *mask, int classes, int count);

void free_layer_device(layer *l);
#endif
