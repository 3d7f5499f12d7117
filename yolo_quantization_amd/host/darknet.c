/*
 * darknet.c -- `./darknet detector test <data> <cfg> <weights> <image> [flags]`, the drop-in entry point
 * (ref: examples/darknet.c:220 main -> examples/detector.c:952 run_detector -> :878 test_detector), INT8 path only.
 *
 * Flags of the reference kept: -i <gpu>, -thresh <t>, -close_quantization (rejected: float path not built).
 * Added: -batch <B> (replicates the image), -accum exact|ref-f32, -parity wrap|saturate, -dump <dir> (per-layer
 * tensors in the reference layout), -graph (hipGraph replay), -n <iters> (timing loop).
 *
 * Image input: binary PPM (P6) of the network size, a raw `.u8` file holding [c][h][w] bytes, or
 * `synthetic:<seed>`.  JPEG/PNG decoding (stb_image in the reference, third-party) and letterboxing are outside
 * the INT8 hot path and not built; detections are printed as raw (objectness*class) maxima,
 * box decode / NMS / drawing are SURVEY.md 8(f) rank 1 "next" work.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "darknet_q.h"

static int find_arg(int argc, char **argv, const char *arg)
{
    for (int i = 0; i < argc; ++i)
        if (argv[i] && 0 == strcmp(argv[i], arg)) { argv[i] = 0; return 1; }
    return 0;
}
static const char *find_char_arg(int argc, char **argv, const char *arg, const char *def)
{
    for (int i = 0; i < argc - 1; ++i)
        if (argv[i] && 0 == strcmp(argv[i], arg)) { def = argv[i + 1]; argv[i] = 0; argv[i + 1] = 0; break; }
    return def;
}

static void load_image_u8(const char *path, int c, int h, int w, uint8_t *out)
{
    const size_t n = (size_t)c * h * w;
    if (0 == strncmp(path, "synthetic", 9)) {
        unsigned long long s = 88172645463325252ULL;
        const char *colon = strchr(path, ':');
        if (colon) s ^= strtoull(colon + 1, NULL, 10) * 0x9E3779B97F4A7C15ULL;
        for (size_t i = 0; i < n; ++i) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; out[i] = (uint8_t)(s >> 32); }
        out[0] = 0; out[1] = 255; /* full range: the reference's dynamic quantiser is then the identity */
        return;
    }
    FILE *f = fopen(path, "rb");
    if (!f) { fprintf(stderr, "Cannot load image \"%s\"\n", path); exit(0); }
    size_t len = strlen(path);
    if (len > 3 && 0 == strcmp(path + len - 3, ".u8")) {
        if (fread(out, 1, n, f) != n) error("raw .u8 image has the wrong size");
        fclose(f);
        return;
    }
    char magic[3] = {0};
    int iw = 0, ih = 0, maxv = 0;
    if (fscanf(f, "%2s %d %d %d", magic, &iw, &ih, &maxv) != 4 || strcmp(magic, "P6") || maxv != 255)
        error("image must be a binary PPM (P6, maxval 255), a raw .u8 file or synthetic:<seed>; JPEG/PNG decoding is not built");
    fgetc(f);
    if (iw != w || ih != h || c != 3) error("PPM size must equal the network input (letterbox resize is not built)");
    uint8_t *rgb = malloc((size_t)3 * w * h);
    if (fread(rgb, 1, (size_t)3 * w * h, f) != (size_t)3 * w * h) error("PPM truncated");
    for (int k = 0; k < 3; ++k)
        for (int i = 0; i < w * h; ++i) out[(size_t)k * w * h + i] = rgb[3 * i + k];
    free(rgb);
    fclose(f);
}

static void dump_layer(const char *dir, network *net, int i)
{
    layer *l = &net->layers[i];
    pull_layer_output(net, i);
    char path[512];
    const size_t cnt = (size_t)net->batch * l->outputs;
    if (l->type != YOLO) {
        snprintf(path, sizeof(path), "%s/L%02d_u8.bin", dir, i);
        FILE *f = fopen(path, "wb"); if (!f) file_error(path);
        fwrite(l->output_uint8_final, 1, cnt, f); fclose(f);
    }
    if (l->output_int32_gpu) {
        snprintf(path, sizeof(path), "%s/L%02d_int32.bin", dir, i);
        FILE *f = fopen(path, "wb"); if (!f) file_error(path);
        fwrite(l->output_int32, sizeof(int32_t), cnt, f); fclose(f);
    }
    if (l->output_gpu) {
        snprintf(path, sizeof(path), "%s/L%02d_f32.bin", dir, i);
        FILE *f = fopen(path, "wb"); if (!f) file_error(path);
        fwrite(l->output, sizeof(float), cnt, f); fclose(f);
    }
}

static void test_detector(const char *cfgfile, const char *weightfile, const char *filename, float thresh, int batch,
                          int accum, int store, const char *dumpdir, int use_graph, int iters, int gpu)
{
    network *net = load_network((char *)cfgfile, (char *)weightfile, 0);
    net->gpu_index = gpu;
    net->accum_mode = accum;
    net->store_mode = store;
    net->dump_int32 = dumpdir != NULL;
    net->use_graph = use_graph;
    set_batch_network(net, batch);
    uint8_t *img = malloc((size_t)net->inputs);
    load_image_u8(filename, net->c, net->h, net->w, img);
    /* float image as the reference sees it (X = u8/255), so that its dynamic layer-0 quantiser runs (src/blas.c:279) */
    for (int b = 0; b < batch; ++b)
        for (int k = 0; k < net->inputs; ++k) net->input[(size_t)b * net->inputs + k] = (float)img[k] / 255.0f;
    quantization_weights_and_activations(net);
    double t0 = what_time_is_it_now();
    for (int it = 0; it < iters; ++it) network_predict(net, net->input);
    double dt = (what_time_is_it_now() - t0) / iters;
    printf("%s: Predicted in %f seconds. (batch %d, %.1f images/s, accum=%s, parity=%s)\n", filename, dt, batch,
           batch / dt, accum == MI355_ACC_EXACT ? "exact" : "ref-f32", store == MI355_STORE_WRAP ? "wrap" : "saturate");
    /* ref: get_network_boxes (src/network.c:583-640) -> get_yolo_detections, here decoded on the device; NMS / drawing are
     * host-side post-processing outside the INT8 path: the boxes above thresh are printed instead */
    for (int i = 0; i < net->n; ++i) {
        layer *l = &net->layers[i];
        if (l->type != YOLO) continue;
        const int cap = l->n * l->h * l->w, rl = 6 + l->classes;
        float *recs = calloc((size_t)batch * cap * rl, sizeof(float));
        int *counts = calloc((size_t)batch, sizeof(int));
        /* the image is handed over at network size (no letterbox resize built), so the box correction is the identity */
        network_yolo_detections_gpu(net, i, net->w, net->h, thresh, 1, recs, cap, counts);
        printf("yolo layer %d (%dx%d): %d boxes with objectness above %.2f in image 0\n", i, l->w, l->h, counts[0], thresh);
        for (int k = 0; k < counts[0] && k < 5; ++k) {
            const float *r = recs + (size_t)k * rl;
            int best = 0;
            for (int c = 1; c < l->classes; ++c)
                if (r[6 + c] > r[6 + best]) best = c;
            printf("  cell %d anchor %d: class %d %.0f%%  box x %.4f y %.4f w %.4f h %.4f\n", (int)r[0] / l->n, (int)r[0] % l->n,
                   best, 100.f * r[6 + best], r[1], r[2], r[3], r[4]);
        }
        free(recs);
        free(counts);
    }
    if (dumpdir) for (int i = 0; i < net->n; ++i) dump_layer(dumpdir, net, i);
    free(img);
    free_network(net);
}

int main(int argc, char **argv)
{
    if (argc < 2) {
        fprintf(stderr, "usage: %s detector test <data> <cfg> <weights> <image> [-thresh t] [-i gpu] [-batch B] "
                        "[-accum exact|ref-f32] [-parity wrap|saturate] [-dump dir] [-graph] [-n iters]\n", argv[0]);
        return 0;
    }
    int gpu = atoi(find_char_arg(argc, argv, "-i", "0"));
    if (find_arg(argc, argv, "-nogpu")) error("-nogpu: this build has no CPU data path");
    if (find_arg(argc, argv, "-close_quantization")) error("-close_quantization: the float path is not built");
    float thresh = (float)atof(find_char_arg(argc, argv, "-thresh", ".5"));
    int batch = atoi(find_char_arg(argc, argv, "-batch", "1"));
    const char *accum_s = find_char_arg(argc, argv, "-accum", "exact");
    const char *parity_s = find_char_arg(argc, argv, "-parity", "wrap");
    const char *dumpdir = find_char_arg(argc, argv, "-dump", NULL);
    int use_graph = find_arg(argc, argv, "-graph");
    int iters = atoi(find_char_arg(argc, argv, "-n", "1"));
    int accum = 0 == strcmp(accum_s, "ref-f32") ? MI355_ACC_REF_F32 : MI355_ACC_EXACT;
    int store = 0 == strcmp(parity_s, "saturate") ? MI355_STORE_SATURATE : MI355_STORE_WRAP;
    if (0 == strcmp(argv[1], "detector")) {
        if (argc < 7 || !argv[2] || strcmp(argv[2], "test")) {
            fprintf(stderr, "only `detector test` is part of the INT8 inference path (train/valid/recall: SURVEY.md 2 rows 12,22)\n");
            return 0;
        }
        test_detector(argv[4], argv[5], argv[6], thresh, batch, accum, store, dumpdir, use_graph, iters < 1 ? 1 : iters, gpu);
    } else {
        fprintf(stderr, "Not an option: %s\n", argv[1]);
    }
    return 0;
}
