/*
 * darknet.c -- `./darknet detector test <data> <cfg> <weights> <image> [flags]`, the drop-in entry point
 * (ref: examples/darknet.c:220 main -> examples/detector.c:952 run_detector -> :878 test_detector), INT8 path only.
 *
 * The flow is test_detector's (examples/detector.c:878-950):
 *   read_data_cfg(<data>) `names=`, get_labels            :880-882
 *   load_network, set_batch_network                        :885-886
 *   load image -> letterbox_image(net->w, net->h)          :903-904   (here: on the device, mi355_letterbox_forward)
 *   quantization_weights_and_activations (layer-0 quantiser on the float image) :918   (here: on the device)
 *   network_predict, "Predicted in"                        :922-924
 *   get_network_boxes(im.w, im.h, thresh, hier, 0, 1)      :926       (here: box decode on the device)
 *   do_nms_sort(nms = .45)                                 :930
 *   draw_detections -> prints "<name>: <p>%"               :931 (src/image.c:255); drawing / saving the image is not built
 *
 * Flags of the reference kept: -i <gpu>, -thresh <t>, -hier <t>, -gpus a,b,c (ref :954-986), -close_quantization
 * (rejected: float path not built).  Added: -batch <B> (replicates the image), -accum exact|ref-f32,
 * -parity wrap|saturate, -dump <dir> (per-layer tensors in the reference layout), -graph (hipGraph replay), -n <iters>,
 * -save_packed <file> / -packed <file>, -boxes (one machine-readable line per kept box), -bcast (with
 * -gpus: replica 0 reads the weights file, the others receive the packed blobs by one RCCL broadcast over xGMI),
 * -inflight <N> (with -n <iters>: after the timed passes, the same passes dealt round-robin to N executors of the prepared
 * model -- network_replica: own activations and stream, shared packed weights; the 4th on the default stream -- and their rate).
 *
 * Image input: binary PPM (P6) of ANY size (letterboxed like the reference does), a raw `.u8` file holding [c][h][w] bytes at
 * network size, or `synthetic:<seed>`.  JPEG/PNG decoding is third-party code in the reference (stb_image, SURVEY.md 2
 * row 21) and not built.
 */
#include <math.h>
#include <pthread.h>
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

typedef struct { int w, h, c; float *data; } image; /* planar float [c][h][w] in 0..1, like the reference's `image` */

/* load_image_color's contract (ref: src/image.c:1361-1395: bytes / 255., planar) for PPM / raw / synthetic sources */
static image load_image_any(const char *path, int netc, int neth, int netw)
{
    image im = {0, 0, 0, NULL};
    if (0 == strncmp(path, "synthetic", 9) || (strlen(path) > 3 && 0 == strcmp(path + strlen(path) - 3, ".u8"))) {
        const size_t n = (size_t)netc * neth * netw;
        uint8_t *raw = malloc(n);
        if (0 == strncmp(path, "synthetic", 9)) {
            unsigned long long s = 88172645463325252ULL;
            const char *colon = strchr(path, ':');
            if (colon) s ^= strtoull(colon + 1, NULL, 10) * 0x9E3779B97F4A7C15ULL;
            for (size_t i = 0; i < n; ++i) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; raw[i] = (uint8_t)(s >> 32); }
            raw[0] = 0; raw[1] = 255; /* full range: the reference's dynamic quantiser is then the identity */
        } else {
            FILE *f = fopen(path, "rb");
            if (!f) { fprintf(stderr, "Cannot load image \"%s\"\n", path); exit(0); }
            if (fread(raw, 1, n, f) != n) error("raw .u8 image has the wrong size");
            fclose(f);
        }
        im.w = netw; im.h = neth; im.c = netc;
        im.data = malloc(n * sizeof(float));
        for (size_t i = 0; i < n; ++i) im.data[i] = (float)raw[i] / 255.0f;
        free(raw);
        return im;
    }
    FILE *f = fopen(path, "rb");
    if (!f) { fprintf(stderr, "Cannot load image \"%s\"\n", path); exit(0); }
    char magic[3] = {0};
    int iw = 0, ih = 0, maxv = 0;
    if (fscanf(f, "%2s %d %d %d", magic, &iw, &ih, &maxv) != 4 || strcmp(magic, "P6") || maxv != 255 || iw < 1 || ih < 1)
        error("image must be a binary PPM (P6, maxval 255), a raw .u8 file or synthetic:<seed>; JPEG/PNG decoding is not built");
    fgetc(f);
    if (netc != 3) error("PPM input needs a 3-channel network");
    uint8_t *rgb = malloc((size_t)3 * iw * ih);
    if (fread(rgb, 1, (size_t)3 * iw * ih, f) != (size_t)3 * iw * ih) error("PPM truncated");
    fclose(f);
    im.w = iw; im.h = ih; im.c = 3;
    im.data = malloc((size_t)3 * iw * ih * sizeof(float));
    for (int k = 0; k < 3; ++k)
        for (int i = 0; i < iw * ih; ++i) im.data[(size_t)k * iw * ih + i] = (float)rgb[3 * i + k] / 255.f; /* ref :1386 */
    free(rgb);
    return im;
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

typedef struct {
    const char *datacfg, *cfgfile, *weightfile, *filename, *dumpdir, *packed_in, *packed_out;
    float thresh, hier_thresh;
    int batch, accum, store, use_graph, iters, gpu, boxes, quiet, inflight;
    int rank, nranks;      /* -gpus with -bcast: this replica's rank; rank 0 reads the weights file, the others receive blobs */
    const void *comm_id;   /* shared 128-byte RCCL unique id (NULL: every replica reads the file) */
    double seconds; /* out: per forward pass */
} detect_job;

/* one network on one device: load, input path, predict, boxes, NMS, print */
static void test_detector(detect_job *job)
{
    char **names = NULL;
    int nnames = 0;
    if (job->datacfg) {
        char *name_list = data_cfg_find(job->datacfg, "names");
        if (name_list) { names = get_labels(name_list, &nnames); free(name_list); }
        else fprintf(stderr, "%s has no names= entry: classes are printed by index\n", job->datacfg);
    }
    network *net;
    if (job->packed_in || (job->comm_id && job->rank != 0)) { /* deployed model / broadcast receiver: cfg only, blobs arrive packed */
        net = parse_network_cfg((char *)job->cfgfile, 0);
    } else {
        net = load_network((char *)job->cfgfile, (char *)job->weightfile, 0);
    }
    net->gpu_index = job->gpu;
    net->accum_mode = job->accum;
    net->store_mode = job->store;
    net->dump_int32 = job->dumpdir != NULL;
    net->use_graph = job->use_graph;
    set_batch_network(net, job->batch);
    if (job->packed_in) network_load_packed(net, (char *)job->packed_in);
    if (job->comm_id) { /* one-shot RCCL broadcast of the packed weights from replica 0 (xGMI), ~9 MB for yolov3-tiny */
        void *comm = NULL;
        if (mi355_init(job->gpu) || mi355_comm_init(&comm, job->nranks, job->comm_id, job->rank)) {
            fprintf(stderr, "RCCL: %s %s\n", mi355_last_error(), mi355_comm_last_error());
            error("mi355_comm_init");
        }
        if (job->rank == 0 && !job->packed_in) quantization_prep_host(net, 1.0f / 255.0f, 0);
        network_bcast_packed(net, comm, job->rank, 0);
        mi355_comm_destroy(comm);
    }

    image im = load_image_any(job->filename, net->c, net->h, net->w);
    /* input path on the device: the float image goes up once, letterbox_image + the layer-0 quantiser run in HBM */
    float *im_gpu = NULL;
    const size_t imbytes = (size_t)im.c * im.h * im.w * sizeof(float);
    if (mi355_init(job->gpu) || mi355_alloc((void **)&im_gpu, imbytes) || mi355_h2d(im_gpu, im.data, imbytes, NULL) || mi355_stream_sync(NULL)) {
        fprintf(stderr, "MI355: %s\n", mi355_last_error());
        error("cannot stage the image on the device (this build has no CPU data path)");
    }
    for (int b = 0; b < job->batch; ++b) network_letterbox_input_gpu(net, b, im_gpu, im.w, im.h);
    network_quantize_input_gpu(net); /* == quantization_weights_and_activations on the letterboxed floats (ref :918) */
    if (job->packed_out) network_save_packed(net, (char *)job->packed_out);

    double t0 = what_time_is_it_now();
    for (int it = 0; it < job->iters; ++it) network_predict(net, net->input);
    job->seconds = (what_time_is_it_now() - t0) / job->iters;
    if (!job->quiet)
        printf("%s: Predicted in %f seconds. (batch %d, %.1f images/s, gpu %d, accum=%s, parity=%s)\n", job->filename, job->seconds,
               job->batch, job->batch / job->seconds, job->gpu, job->accum == MI355_ACC_EXACT ? "exact" : "ref-f32",
               job->store == MI355_STORE_WRAP ? "wrap" : "saturate");

    if (job->inflight > 1 && job->accum == MI355_ACC_EXACT && !job->dumpdir) { /* several batches in flight (darknet_q.h network_replica) */
        const int n = job->inflight > 8 ? 8 : job->inflight;
        network *ex[8] = {net};
        for (int k = 1; k < n; ++k) {
            net->replica_default_stream = (k == 3 && !job->use_graph); /* the fourth hardware queue belongs to the default stream */
            ex[k] = network_replica(net);
            if (mi355_d2d(ex[k]->input_uint8_gpu, net->input_uint8_gpu, (size_t)net->batch * net->inputs, ex[k]->stream) ||
                mi355_stream_sync(ex[k]->stream)) error("replica input");
        }
        const int passes = job->iters * n;
        for (int k = 0; k < n; ++k) forward_network_gpu(ex[k]); /* the throughput plan's kernels: first launch outside the timing */
        for (int k = 0; k < n; ++k) if (mi355_stream_sync(ex[k]->stream)) error("sync");
        const double t1 = what_time_is_it_now();
        for (int it = 0; it < passes; ++it) forward_network_gpu(ex[it % n]);
        for (int k = 0; k < n; ++k) if (mi355_stream_sync(ex[k]->stream)) error("sync");
        const double dt = (what_time_is_it_now() - t1) / passes;
        if (!job->quiet)
            printf("%d batches in flight: %f seconds per pass (batch %d, %.1f images/s, gpu %d)\n", n, dt, job->batch, job->batch / dt, job->gpu);
        job->seconds = dt;
        for (int k = n - 1; k >= 1; --k) free_network(ex[k]);
    }

    int classes = 0;
    for (int i = 0; i < net->n; ++i)
        if (net->layers[i].type == YOLO) classes = net->layers[i].classes; /* ref: `l = net->layers[net->n-1]` (:910) */
    if (classes && !job->quiet) {
        const float nms = .45f; /* ref :892 */
        int nboxes = 0;
        detection *dets = get_network_boxes(net, im.w, im.h, job->thresh, job->hier_thresh, 0, 1, &nboxes);
        printf("%d\n", nboxes);
        printf("-----------------------\n");
        if (nms) do_nms_sort(dets, nboxes, classes, nms);
        for (int i = 0; i < nboxes; ++i) /* draw_detections' console output (src/image.c:246-257) */
            for (int j = 0; j < classes; ++j)
                if (dets[i].prob[j] > job->thresh) {
                    if (names && j < nnames) printf("%s: %.0f%%\n", names[j], dets[i].prob[j] * 100);
                    else printf("class %d: %.0f%%\n", j, dets[i].prob[j] * 100);
                    if (job->boxes)
                        printf("box: class %d prob %.9g x %.9g y %.9g w %.9g h %.9g\n", j, dets[i].prob[j], dets[i].bbox.x,
                               dets[i].bbox.y, dets[i].bbox.w, dets[i].bbox.h);
                }
        free_detections(dets, nboxes);
    }
    if (job->dumpdir) for (int i = 0; i < net->n; ++i) dump_layer(job->dumpdir, net, i);
    mi355_free(im_gpu);
    free(im.data);
    for (int i = 0; i < nnames; ++i) free(names[i]);
    free(names);
    free_network(net);
}

static void *detector_thread(void *p)
{
    test_detector((detect_job *)p);
    return NULL;
}

int main(int argc, char **argv)
{
    if (argc < 2) {
        fprintf(stderr, "usage: %s detector test <data> <cfg> <weights> <image> [-thresh t] [-i gpu | -gpus a,b,..] [-batch B] "
                        "[-accum exact|ref-f32] [-parity wrap|saturate] [-dump dir] [-graph] [-n iters] [-boxes] "
                        "[-save_packed file] [-packed file] [-bcast] [-inflight N]\n", argv[0]);
        return 0;
    }
    detect_job job;
    memset(&job, 0, sizeof(job));
    job.gpu = atoi(find_char_arg(argc, argv, "-i", "0"));
    const char *gpu_list = find_char_arg(argc, argv, "-gpus", NULL);
    if (find_arg(argc, argv, "-nogpu")) error("-nogpu: this build has no CPU data path");
    if (find_arg(argc, argv, "-close_quantization")) error("-close_quantization: the float path is not built");
    job.thresh = (float)atof(find_char_arg(argc, argv, "-thresh", ".5"));
    job.hier_thresh = (float)atof(find_char_arg(argc, argv, "-hier", ".5"));
    job.batch = atoi(find_char_arg(argc, argv, "-batch", "1"));
    const char *accum_s = find_char_arg(argc, argv, "-accum", "exact");
    const char *parity_s = find_char_arg(argc, argv, "-parity", "wrap");
    job.dumpdir = find_char_arg(argc, argv, "-dump", NULL);
    job.packed_in = find_char_arg(argc, argv, "-packed", NULL);
    job.packed_out = find_char_arg(argc, argv, "-save_packed", NULL);
    job.use_graph = find_arg(argc, argv, "-graph");
    job.boxes = find_arg(argc, argv, "-boxes");
    const int bcast = find_arg(argc, argv, "-bcast");
    job.iters = atoi(find_char_arg(argc, argv, "-n", "1"));
    job.inflight = atoi(find_char_arg(argc, argv, "-inflight", "1"));
    if (job.iters < 1) job.iters = 1;
    if (job.batch < 1) job.batch = 1;
    job.accum = 0 == strcmp(accum_s, "ref-f32") ? MI355_ACC_REF_F32 : MI355_ACC_EXACT;
    job.store = 0 == strcmp(parity_s, "saturate") ? MI355_STORE_SATURATE : MI355_STORE_WRAP;
    if (strcmp(argv[1], "detector")) {
        fprintf(stderr, "Not an option: %s\n", argv[1]);
        return 0;
    }
    if (argc < 7 || !argv[2] || strcmp(argv[2], "test") || !argv[3] || !argv[4] || !argv[5] || !argv[6]) {
        fprintf(stderr, "only `detector test <data> <cfg> <weights> <image>` is part of the INT8 inference path (train/valid/recall: SURVEY.md 2 rows 12,22)\n");
        return 0;
    }
    job.datacfg = argv[3]; job.cfgfile = argv[4]; job.weightfile = argv[5]; job.filename = argv[6];
    if (!gpu_list) {
        test_detector(&job);
        return 0;
    }
    /* -gpus a,b,c (ref: examples/detector.c:954-977 parses the same list for training): one host thread per device, each
     * with its own network replica and its own `-batch` images -- images shard embarrassingly, no steady-state traffic.
     * Weights reach every device from the file, or with -bcast from replica 0 by one RCCL broadcast (mi355_bcast_blob). */
    int gpus[64], ngpus = 0;
    char *list = malloc(strlen(gpu_list) + 1);
    strcpy(list, gpu_list);
    for (char *tok = strtok(list, ","); tok && ngpus < 64; tok = strtok(NULL, ",")) gpus[ngpus++] = atoi(tok);
    if (ngpus < 1) error("-gpus: empty list");
    if (mi355_device_count() < 1) error("no HIP device visible (this build has no CPU data path)");
    static char comm_id[128];
    if (bcast && mi355_comm_unique_id(comm_id)) {
        fprintf(stderr, "RCCL: %s\n", mi355_comm_last_error());
        error("-bcast needs librccl");
    }
    detect_job *jobs = calloc((size_t)ngpus, sizeof(detect_job));
    pthread_t *th = calloc((size_t)ngpus, sizeof(pthread_t));
    for (int g = 0; g < ngpus; ++g) {
        jobs[g] = job;
        jobs[g].gpu = gpus[g];
        jobs[g].rank = g; jobs[g].nranks = ngpus;
        jobs[g].comm_id = bcast ? comm_id : NULL;
        jobs[g].quiet = g != 0; /* device 0 of the list prints the detections */
        jobs[g].dumpdir = g == 0 ? job.dumpdir : NULL;
        jobs[g].packed_out = g == 0 ? job.packed_out : NULL;
        if (pthread_create(&th[g], NULL, detector_thread, &jobs[g])) error("pthread_create");
    }
    double slowest = 0;
    for (int g = 0; g < ngpus; ++g) {
        pthread_join(th[g], NULL);
        if (jobs[g].seconds > slowest) slowest = jobs[g].seconds;
    }
    printf("%d GPUs x batch %d: %.1f images/s (slowest replica %f s per pass)\n", ngpus, job.batch, ngpus * job.batch / slowest, slowest);
    free(jobs); free(th); free(list);
    return 0;
}
