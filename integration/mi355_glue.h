/* mi355_glue.h -- what the reference (ArtyZe/yolo_quantization) includes to run its quantized inference path on an MI355X
 * through libmi355yolo.so.  See mi355_glue.c and INTEGRATION.md section B. */
#ifndef MI355_GLUE_H
#define MI355_GLUE_H
#include "darknet.h"

/* Bind a loaded network (after quantization_weights_and_activations(net), examples/detector.c:918): packs and uploads the
 * quantized weights, allocates the device activations and points every quantized layer's `forward_gpu` (include/darknet.h:161)
 * at the MI355X implementation.  gpu = HIP device index.  accum_mode / store_mode: MI355_ACC_* / MI355_STORE_*. */
void mi355_bind_network(network *net, int gpu, int accum_mode, int store_mode);
/* forward_network (src/network.c:229-261) over the bound forward_gpu pointers, with the uint8 hand-off on the device.
 * pull_all != 0: every layer's output_uint8_final / output / output_int32 is copied back (per-layer parity runs);
 * pull_all == 0: only the float outputs the host-side post-processing reads (yolo layers, quant_stop convs). */
void forward_network_mi355(network *net, int pull_all);
void mi355_unbind_network(network *net);
#endif
