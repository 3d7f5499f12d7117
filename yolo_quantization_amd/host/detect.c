/*
 * detect.c -- the post-processing `detector test` runs after network_predict, host side:
 *
 *   get_network_boxes / free_detections   ref: src/network.c:583-640, 642-651 (make_network_boxes, fill_network_boxes)
 *   do_nms_sort, box_iou                  ref: src/box.c:6-19, 58-89, 151-182
 *   read_data_cfg `names=` / get_labels   ref: src/option_list.c:7-33, src/data.c:657-663
 *
 * The box decode itself (get_yolo_detections + correct_yolo_boxes, src/yolo_layer.c:246-277,316-345) runs on the device
 * (mi355_yolo_detections): only the detections cross PCIe; this file turns the records into the reference's `detection`
 * array, in the reference's order (yolo layers in network order, cells row-major, anchors innermost), and runs its NMS.
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "host_internal.h"

detection *get_network_boxes_batch(network *net, int b, int w, int h, float thresh, float hier, int *map, int relative, int *num)
{
    (void)hier; (void)map; /* yolo layers ignore both (ref: src/network.c:626-629) */
    if (b < 0 || b >= net->batch) error("get_network_boxes: image index outside the batch");
    detection *dets = NULL;
    int total = 0;
    for (int i = 0; i < net->n; ++i) {
        layer *l = &net->layers[i];
        if (l->type != YOLO) continue;
        const int cap = l->n * l->h * l->w, rl = 6 + l->classes;
        float *recs = calloc((size_t)net->batch * cap * rl, sizeof(float));
        int *counts = calloc((size_t)net->batch, sizeof(int));
        network_yolo_detections_gpu(net, i, w, h, thresh, relative, recs, cap, counts);
        const int cnt = counts[b] < cap ? counts[b] : cap;
        dets = realloc(dets, (size_t)(total + cnt + 1) * sizeof(detection));
        for (int k = 0; k < cnt; ++k) {
            const float *r = recs + ((size_t)b * cap + k) * rl;
            detection *d = &dets[total + k];
            memset(d, 0, sizeof(*d));
            d->bbox.x = r[1]; d->bbox.y = r[2]; d->bbox.w = r[3]; d->bbox.h = r[4];
            d->objectness = r[5];
            d->classes = l->classes;
            d->prob = calloc(l->classes, sizeof(float));
            memcpy(d->prob, r + 6, sizeof(float) * (size_t)l->classes);
        }
        total += cnt;
        free(recs);
        free(counts);
    }
    if (!dets) dets = calloc(1, sizeof(detection));
    *num = total;
    return dets;
}

detection *get_network_boxes(network *net, int w, int h, float thresh, float hier, int *map, int relative, int *num)
{
    return get_network_boxes_batch(net, 0, w, h, thresh, hier, map, relative, num); /* the reference decodes image 0 */
}

void free_detections(detection *dets, int n)
{
    for (int i = 0; i < n; ++i) free(dets[i].prob);
    free(dets);
}

/* ref: src/box.c:151-182 */
static float overlap(float x1, float w1, float x2, float w2)
{
    float l1 = x1 - w1 / 2, l2 = x2 - w2 / 2;
    float left = l1 > l2 ? l1 : l2;
    float r1 = x1 + w1 / 2, r2 = x2 + w2 / 2;
    float right = r1 < r2 ? r1 : r2;
    return right - left;
}
static float box_intersection(box a, box b)
{
    float w = overlap(a.x, a.w, b.x, b.w), h = overlap(a.y, a.h, b.y, b.h);
    if (w < 0 || h < 0) return 0;
    return w * h;
}
static float box_union(box a, box b)
{
    float i = box_intersection(a, b);
    return a.w * a.h + b.w * b.h - i;
}
float box_iou(box a, box b) { return box_intersection(a, b) / box_union(a, b); }

static int nms_comparator(const void *pa, const void *pb) /* ref: src/box.c:6-19 */
{
    const detection *a = pa, *b = pb;
    float diff = b->sort_class >= 0 ? a->prob[b->sort_class] - b->prob[b->sort_class] : a->objectness - b->objectness;
    if (diff < 0) return 1;
    if (diff > 0) return -1;
    return 0;
}

void do_nms_sort(detection *dets, int total, int classes, float thresh) /* ref: src/box.c:58-89 */
{
    int k = total - 1;
    for (int i = 0; i <= k; ++i)
        if (dets[i].objectness == 0) {
            detection swap = dets[i];
            dets[i] = dets[k];
            dets[k] = swap;
            --k;
            --i;
        }
    total = k + 1;
    for (k = 0; k < classes; ++k) {
        for (int i = 0; i < total; ++i) dets[i].sort_class = k;
        qsort(dets, (size_t)total, sizeof(detection), nms_comparator);
        for (int i = 0; i < total; ++i) {
            if (dets[i].prob[k] == 0) continue;
            box a = dets[i].bbox;
            for (int j = i + 1; j < total; ++j)
                if (box_iou(a, dets[j].bbox) > thresh) dets[j].prob[k] = 0;
        }
    }
}

/* flat-array form for FFI callers and tests: boxes [n][4], probs [n][classes], objectness [n]; rows keep their place (an
 * id travels through the sort), suppressed scores become 0 */
void do_nms_sort_arrays(const float *boxes, float *probs, const float *objectness, int n, int classes, float thresh)
{
    detection *dets = calloc((size_t)(n > 0 ? n : 1), sizeof(detection));
    for (int i = 0; i < n; ++i) {
        dets[i].bbox.x = boxes[4 * i]; dets[i].bbox.y = boxes[4 * i + 1]; dets[i].bbox.w = boxes[4 * i + 2]; dets[i].bbox.h = boxes[4 * i + 3];
        dets[i].objectness = objectness[i];
        dets[i].classes = i; /* the id */
        dets[i].prob = calloc((size_t)classes, sizeof(float));
        memcpy(dets[i].prob, probs + (size_t)i * classes, sizeof(float) * (size_t)classes);
    }
    do_nms_sort(dets, n, classes, thresh);
    for (int i = 0; i < n; ++i) memcpy(probs + (size_t)dets[i].classes * classes, dets[i].prob, sizeof(float) * (size_t)classes);
    free_detections(dets, n);
}

/* `names = <file>` of a .data file (ref: read_data_cfg + option_find_str(options, "names", "data/voc.names"),
 * examples/detector.c:880-881); NULL when the key is missing */
char *data_cfg_find(const char *datacfg, const char *key)
{
    FILE *f = fopen(datacfg, "r");
    if (!f) file_error(datacfg);
    char line[4096], *found = NULL;
    while (fgets(line, sizeof(line), f)) {
        size_t len = strlen(line), off = 0; /* ref strip(): drop blanks, tabs, newlines anywhere */
        for (size_t i = 0; i < len; ++i) {
            char c = line[i];
            if (c == ' ' || c == '\t' || c == '\n' || c == '\r') ++off;
            else line[i - off] = c;
        }
        line[len - off] = 0;
        if (!line[0] || line[0] == '#' || line[0] == ';') continue;
        char *eq = strchr(line, '=');
        if (!eq) continue;
        *eq = 0;
        if (strcmp(line, key) == 0) {
            free(found);
            found = malloc(strlen(eq + 1) + 1);
            strcpy(found, eq + 1);
        }
    }
    fclose(f);
    return found;
}

char **get_labels(char *filename, int *count) /* ref: src/data.c:657-663 (one label per line) */
{
    FILE *f = fopen(filename, "r");
    if (!f) file_error(filename);
    char **labels = NULL;
    int n = 0, cap = 0;
    char line[1024];
    while (fgets(line, sizeof(line), f)) {
        size_t len = strlen(line);
        while (len && (line[len - 1] == '\n' || line[len - 1] == '\r')) line[--len] = 0;
        if (n == cap) { cap = cap ? cap * 2 : 32; labels = realloc(labels, (size_t)cap * sizeof(char *)); }
        labels[n] = malloc(len + 1);
        strcpy(labels[n++], line);
    }
    fclose(f);
    if (count) *count = n;
    return labels;
}
