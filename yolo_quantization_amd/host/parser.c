/*
 * parser.c -- cfg and .weights readers of the host.
 *
 * Accepts the cfg keys and the QUANTIZATION-build .weights layout of the reference:
 *   cfg sections/keys        ref: src/parser.c:170-204 (convolutional), 411-431 (maxpool), 506-518 (upsample),
 *                                 520-564 (route), 254-291 (yolo), 579-674 (net), read_cfg 817-860
 *   .weights records         ref: src/parser.c:1124-1199 (readers), 1201-1300 (load_weights_upto)
 * Layer types outside the INT8 path are rejected with error().
 */
#include <ctype.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "host_internal.h"

/* ----------------------------------------------------------------------------------------------- option store */
typedef struct { char *key, *val; int used; } kv;
typedef struct { char *type; kv *opts; int nopts, cap; } section;

static char *xstrdup(const char *s)
{
    char *r = malloc(strlen(s) + 1);
    strcpy(r, s);
    return r;
}

static void strip(char *s) /* remove blanks, tabs, newlines anywhere in the line (ref: src/utils.c strip) */
{
    size_t len = strlen(s), off = 0;
    for (size_t i = 0; i < len; ++i) {
        char c = s[i];
        if (c == ' ' || c == '\t' || c == '\n' || c == '\r') ++off;
        else s[i - off] = c;
    }
    s[len - off] = '\0';
}

static section *read_cfg(const char *filename, int *nsec)
{
    FILE *f = fopen(filename, "r");
    if (!f) file_error(filename);
    section *secs = NULL;
    int n = 0, cap = 0;
    char line[4096];
    int nu = 0;
    while (fgets(line, sizeof(line), f)) {
        ++nu;
        strip(line);
        switch (line[0]) {
        case '[':
            if (n == cap) { cap = cap ? cap * 2 : 32; secs = realloc(secs, cap * sizeof(section)); }
            secs[n].type = xstrdup(line);
            secs[n].opts = NULL; secs[n].nopts = 0; secs[n].cap = 0;
            ++n;
            break;
        case '\0': case '#': case ';':
            break;
        default: {
            char *eq = strchr(line, '=');
            if (!eq || n == 0) {
                fprintf(stderr, "Config file error line %d, could parse: %s\n", nu, line);
                break;
            }
            *eq = '\0';
            section *s = &secs[n - 1];
            if (s->nopts == s->cap) { s->cap = s->cap ? s->cap * 2 : 16; s->opts = realloc(s->opts, s->cap * sizeof(kv)); }
            s->opts[s->nopts].key = xstrdup(line);
            s->opts[s->nopts].val = xstrdup(eq + 1);
            s->opts[s->nopts].used = 0;
            s->nopts++;
        }
        }
    }
    fclose(f);
    *nsec = n;
    return secs;
}

static char *option_find(section *s, const char *key)
{
    for (int i = 0; i < s->nopts; ++i)
        if (strcmp(s->opts[i].key, key) == 0) { s->opts[i].used = 1; return s->opts[i].val; }
    return NULL;
}
static int option_find_int(section *s, const char *key, int def)
{
    char *v = option_find(s, key);
    return v ? atoi(v) : def;
}
static float option_find_float(section *s, const char *key, float def)
{
    char *v = option_find(s, key);
    return v ? (float)atof(v) : def;
}
static const char *option_find_str(section *s, const char *key, const char *def)
{
    char *v = option_find(s, key);
    return v ? v : def;
}

static ACTIVATION get_activation(const char *s)
{
    if (strcmp(s, "leaky") == 0) return LEAKY;
    if (strcmp(s, "relu6") == 0) return RELU6;
    if (strcmp(s, "relu") == 0) return RELU;
    if (strcmp(s, "linear") == 0) return LINEAR;
    if (strcmp(s, "logistic") == 0) return LOGISTIC;
    fprintf(stderr, "activation %s has no integer semantics in the reference (src/convolutional_layer.c:734-748)\n", s);
    error("unsupported activation");
    return LINEAR;
}

/* ---------------------------------------------------------------------------------------------------- parsing */
typedef struct { int batch, inputs, h, w, c, index; int close_quantization; } size_params;

static layer parse_convolutional(section *o, size_params p, int count)
{
    int n = option_find_int(o, "filters", 1);
    int size = option_find_int(o, "size", 1);
    int stride = option_find_int(o, "stride", 1);
    int pad = option_find_int(o, "pad", 0);
    int padding = option_find_int(o, "padding", 0);
    int groups = option_find_int(o, "groups", 1);
    if (pad) padding = size / 2; /* ref :178 */
    ACTIVATION act = get_activation(option_find_str(o, "activation", "logistic"));
    if (act == LOGISTIC) /* also the default when the key is missing (ref: src/parser.c:180) */
        error("[convolutional] activation=logistic (or no activation key): the reference's quantized forward stores nothing for it "
              "(`default: break`, src/convolutional_layer.c:746); use leaky, relu6, relu or linear");
    if (!(p.h && p.w && p.c)) error("Layer before convolutional layer must output image.");
    int bn = option_find_int(o, "batch_normalize", 0);
    int q = option_find_int(o, "quantized", 0);
    int qs = option_find_int(o, "quant_stop", 0);
    if (groups != 1) error("grouped convolution is outside the INT8 path of the reference (prep ignores groups, src/blas.c:306)");
    layer l = make_convolutional_layer(p.batch, p.h, p.w, p.c, n, groups, size, stride, padding, act, bn, qs,
                                       p.close_quantization, q, count);
    l.fisrt_time_train_fag = option_find_int(o, "first_time", 0);
    return l;
}

static layer parse_maxpool(section *o, size_params p, int count)
{
    int stride = option_find_int(o, "stride", 1);
    int size = option_find_int(o, "size", stride);
    int padding = option_find_int(o, "padding", size - 1); /* ref :415 */
    if (!(p.h && p.w && p.c)) error("Layer before maxpool layer must output image.");
    int q = option_find_int(o, "quantized", 0), qs = option_find_int(o, "quant_stop", 0);
    layer l = make_maxpool_layer(p.batch, p.h, p.w, p.c, size, stride, padding, q, qs, p.close_quantization, count);
    l.fisrt_time_train_fag = option_find_int(o, "first_time", 0);
    return l;
}

static layer parse_upsample(section *o, size_params p, int count)
{
    int stride = option_find_int(o, "stride", 2);
    int q = option_find_int(o, "quantized", 0), qs = option_find_int(o, "quant_stop", 0);
    float scale = option_find_float(o, "scale", 1);
    if (scale != 1) error("upsample scale must be 1 on the integer path (ref: src/blas.c:785 assert)");
    if (stride < 0) error("reverse upsample has no quantized path");
    layer l = make_upsample_layer(p.batch, p.w, p.h, p.c, stride, q, qs, p.close_quantization, count);
    l.fisrt_time_train_fag = option_find_int(o, "first_time", 0);
    return l;
}

static layer parse_route(section *o, size_params p, network *net, int count)
{
    char *l = option_find(o, "layers");
    if (!l) error("Route Layer must specify input layers");
    int n = 1;
    for (char *c = l; *c; ++c) if (*c == ',') ++n;
    int *layers = calloc(n, sizeof(int));
    int *sizes = calloc(n, sizeof(int));
    for (int i = 0; i < n; ++i) {
        int index = atoi(l);
        char *comma = strchr(l, ',');
        l = comma ? comma + 1 : l;
        if (index < 0) index = p.index + index;
        if (index < 0 || index >= p.index) error("route: bad layer index");
        layers[i] = index;
        sizes[i] = net->layers[index].outputs;
    }
    int q = option_find_int(o, "quantized", 0), qs = option_find_int(o, "quant_stop", 0);
    layer r = make_route_layer(p.batch, n, layers, sizes, q, qs, p.close_quantization, count);
    layer first = net->layers[layers[0]];
    r.out_w = first.out_w; r.out_h = first.out_h; r.out_c = first.out_c;
    for (int i = 1; i < n; ++i) {
        layer next = net->layers[layers[i]];
        if (next.out_w == first.out_w && next.out_h == first.out_h) r.out_c += next.out_c;
        else error("route inputs must share spatial dims on the integer path");
    }
    r.fisrt_time_train_fag = option_find_int(o, "first_time", 0);
    return r;
}

/* ref: parse_shortcut, src/parser.c:566-590 (`from`, `activation`; the fork adds no quant keys there -- `quantized` /
 * `quant_stop` / `first_time` follow its other glue layers) */
static layer parse_shortcut(section *o, size_params p, network *net, int count)
{
    char *l = option_find(o, "from");
    if (!l) error("Shortcut layer must specify `from`");
    int index = atoi(l);
    if (index < 0) index = p.index + index;
    if (index < 0 || index >= p.index || p.index == 0) error("shortcut: bad layer index");
    layer from = net->layers[index];
    ACTIVATION act = get_activation(option_find_str(o, "activation", "linear"));
    if (act != LINEAR) error("[shortcut] quantized=1 supports activation=linear only (YOLOv3's residual blocks)");
    int q = option_find_int(o, "quantized", 0), qs = option_find_int(o, "quant_stop", 0);
    layer s = make_shortcut_layer(p.batch, index, p.w, p.h, p.c, from.out_w, from.out_h, from.out_c, q, qs,
                                  p.close_quantization, count);
    s.fisrt_time_train_fag = option_find_int(o, "first_time", 0);
    /* first_time = 1 means "no activation record in the weights file yet" (the calibration pass of the reference's glue layers, which
     * inherit their input's record: ref src/parser.c:1174-1199).  The sum of two tensors has no input to inherit from, and a record
     * derived here would be a second spec: refuse it at parse time instead of failing later in mi355_shortcut_multiplier. */
    if (q && s.fisrt_time_train_fag)
        error("[shortcut] quantized=1 first_time=1: the residual add needs its own (scale, zero point) record in the weights file");
    return s;
}

static layer parse_yolo(section *o, size_params p, int count)
{
    int classes = option_find_int(o, "classes", 20);
    int total = option_find_int(o, "num", 1);
    char *a = option_find(o, "mask");
    int num = total, *mask = NULL;
    if (a) {
        num = 1;
        for (char *c = a; *c; ++c) if (*c == ',') ++num;
        mask = calloc(num, sizeof(int));
        for (int i = 0; i < num; ++i) {
            mask[i] = atoi(a);
            char *comma = strchr(a, ',');
            a = comma ? comma + 1 : a;
        }
    }
    layer l = make_yolo_layer(p.batch, p.w, p.h, num, total, mask, classes, count);
    if (l.outputs != p.inputs) error("yolo: previous layer must have n*(classes+5) filters");
    a = option_find(o, "anchors");
    if (a) {
        int n = 1;
        for (char *c = a; *c; ++c) if (*c == ',') ++n;
        for (int i = 0; i < n && i < total * 2; ++i) {
            l.anchors[i] = (float)atof(a);
            char *comma = strchr(a, ',');
            a = comma ? comma + 1 : a;
        }
    }
    return l;
}

network *parse_network_cfg(char *filename, int close_quantization)
{
    int nsec = 0;
    section *secs = read_cfg(filename, &nsec);
    if (nsec == 0) error("Config file has no sections");
    if (strcmp(secs[0].type, "[net]") && strcmp(secs[0].type, "[network]")) error("First section must be [net] or [network]");
    if (close_quantization)
        error("-close_quantization selects the reference's float path (src/convolutional_layer.c:264-270), which this INT8 build does not contain");
    network *net = calloc(1, sizeof(network));
    net->n = nsec - 1;
    net->layers = calloc(net->n, sizeof(layer));
    net->seen = calloc(1, sizeof(size_t));
    net->gpu_index = 0;
    section *o = &secs[0];
    net->batch = option_find_int(o, "batch", 1);
    int subdivs = option_find_int(o, "subdivisions", 1);
    net->batch /= subdivs;
    if (net->batch < 1) net->batch = 1;
    net->h = option_find_int(o, "height", 0);
    net->w = option_find_int(o, "width", 0);
    net->c = option_find_int(o, "channels", 0);
    net->inputs = option_find_int(o, "inputs", net->h * net->w * net->c);
    if (!net->inputs && !(net->h && net->w && net->c)) error("No input parameters supplied");
    net->close_quantization = close_quantization;
    net->cfg_path = strdup(filename);

    size_params p = {net->batch, net->inputs, net->h, net->w, net->c, 0, close_quantization};
    for (int i = 0; i < net->n; ++i) {
        section *s = &secs[i + 1];
        p.index = i;
        layer l;
        if (!strcmp(s->type, "[convolutional]") || !strcmp(s->type, "[conv]")) l = parse_convolutional(s, p, i);
        else if (!strcmp(s->type, "[maxpool]") || !strcmp(s->type, "[max]")) l = parse_maxpool(s, p, i);
        else if (!strcmp(s->type, "[upsample]")) l = parse_upsample(s, p, i);
        else if (!strcmp(s->type, "[route]")) l = parse_route(s, p, net, i);
        else if (!strcmp(s->type, "[shortcut]")) l = parse_shortcut(s, p, net, i);
        else if (!strcmp(s->type, "[yolo]")) l = parse_yolo(s, p, i);
        else {
            fprintf(stderr, "Layer type %s has no quantized forward in the reference (SURVEY.md 2 row 20); not built.\n", s->type);
            error("unsupported layer type");
            return NULL;
        }
        net->layers[i] = l;
        p.h = l.out_h; p.w = l.out_w; p.c = l.out_c; p.inputs = l.outputs;
    }
    for (int i = 0; i < nsec; ++i) {
        for (int j = 0; j < secs[i].nopts; ++j) { free(secs[i].opts[j].key); free(secs[i].opts[j].val); }
        free(secs[i].opts); free(secs[i].type);
    }
    free(secs);
    net->outputs = net->layers[net->n - 1].outputs;
    net->input = calloc((size_t)net->inputs * net->batch, sizeof(float));
    net->input_uint8 = calloc((size_t)net->inputs * net->batch, sizeof(uint8_t));
    net->fuse_maxpool = 1;
    net->accum_mode = MI355_ACC_EXACT;
    net->store_mode = MI355_STORE_WRAP;
    return net;
}

/* ---------------------------------------------------------------------------------------------------- weights */
static void xread(void *dst, size_t sz, size_t cnt, FILE *fp, const char *what)
{
    if (fread(dst, sz, cnt, fp) != cnt) {
        fprintf(stderr, "weights file truncated while reading %s\n", what);
        error("load_weights");
    }
}

static void load_convolutional_weights(layer l, FILE *fp) /* ref :1124-1159 */
{
    size_t num = (size_t)l.c / l.groups * l.n * l.size * l.size;
    xread(l.biases, sizeof(float), l.n, fp, "biases");
    if (l.batch_normalize) {
        xread(l.scales, sizeof(float), l.n, fp, "scales");
        xread(l.rolling_mean, sizeof(float), l.n, fp, "rolling_mean");
        xread(l.rolling_variance, sizeof(float), l.n, fp, "rolling_variance");
    }
    xread(l.input_data_uint8_scales, sizeof(float), 1, fp, "in scale");
    xread(l.input_data_uint8_zero_point, 1, 1, fp, "in zp");
    xread(l.activ_data_uint8_scales, sizeof(float), 1, fp, "act scale");
    xread(l.activ_data_uint8_zero_point, 1, 1, fp, "act zp");
    xread(l.weight_data_uint8_scales, sizeof(float), l.n, fp, "w scales");
    xread(l.weight_data_uint8_zero_point, 1, l.n, fp, "w zp");
    xread(l.weights_uint8, 1, (size_t)l.c * l.n * l.size * l.size, fp, "weights_uint8");
    /* float weights follow; the integer path never reads them (only BN-folds them in place upstream) */
    if (fseek(fp, (long)(num * sizeof(float)), SEEK_CUR)) error("load_weights: seek");
}

static void load_act_record(layer l, FILE *fp)
{
    xread(l.activ_data_uint8_scales, sizeof(float), 1, fp, "act scale");
    xread(l.activ_data_uint8_zero_point, 1, 1, fp, "act zp");
}

void load_weights(network *net, char *filename)
{
    FILE *fp = fopen(filename, "rb");
    if (!fp) file_error(filename);
    int major, minor, revision;
    xread(&major, sizeof(int), 1, fp, "major");
    xread(&minor, sizeof(int), 1, fp, "minor");
    xread(&revision, sizeof(int), 1, fp, "revision");
    if ((major * 10 + minor) >= 2 && major < 1000 && minor < 1000) {
        xread(net->seen, sizeof(size_t), 1, fp, "seen");
    } else {
        int iseen = 0;
        xread(&iseen, sizeof(int), 1, fp, "seen");
        *net->seen = iseen;
    }
    for (int i = 0; i < net->n; ++i) {
        layer l = net->layers[i];
        if (l.type == CONVOLUTIONAL) load_convolutional_weights(l, fp);
        if (l.type == MAXPOOL) load_act_record(l, fp); /* always, ref :1244-1246 */
        if (l.type == ROUTE && l.layer_quant_flag) {   /* ref :1174-1183 */
            if (l.n > 1 && !l.fisrt_time_train_fag) load_act_record(l, fp);
            else {
                l.activ_data_uint8_scales[0] = net->layers[l.input_layers[0]].activ_data_uint8_scales[0];
                l.activ_data_uint8_zero_point[0] = net->layers[l.input_layers[0]].activ_data_uint8_zero_point[0];
            }
        }
        if (l.type == UPSAMPLE && l.layer_quant_flag) { /* ref :1185-1199 */
            if (!l.fisrt_time_train_fag) load_act_record(l, fp);
        }
        if (l.type == SHORTCUT && l.layer_quant_flag)
            load_act_record(l, fp); /* builder-specified: the sum's own (scale, zero point); first_time = 1 is refused by parse_shortcut */
    }
    long here = ftell(fp);
    fseek(fp, 0, SEEK_END);
    if (ftell(fp) != here) fprintf(stderr, "warning: %ld trailing bytes in %s\n", ftell(fp) - here, filename);
    fclose(fp);
    net->prepared = 0;
    net->has_host_weights = 1;
}

network *load_network(char *cfg, char *weights, int clear)
{
    network *net = parse_network_cfg(cfg, clear); /* ref src/network.c:51: `clear` doubles as close_quantization */
    if (weights && weights[0] != 0) load_weights(net, weights);
    if (clear) (*net->seen) = 0;
    return net;
}
