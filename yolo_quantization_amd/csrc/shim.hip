// shim.hip -- the C-ABI of libmi355yolo.so (include/mi355_yolo_int8.h): runtime helpers, tensor descriptors,
// weight packing and kernel launch wrappers.  HIP only; no CUDA-compat headers, no CPU fallback: if the device or
// a kernel is unavailable the call fails with a negative code.
#include "kargs.h"
#include <chrono>
#include <mutex>
#include <stdio.h>
#include <string.h>
#include <stdlib.h>
#include <math.h>


static thread_local char g_err[512] = "";
static int hip_fail(hipError_t e, const char *what)
{
    snprintf(g_err, sizeof(g_err), "%s: %s", what, hipGetErrorString(e));
    return MI355_EHIP;
}
#define HIPCHK(call)                                      \
    do {                                                  \
        hipError_t e__ = (call);                          \
        if (e__ != hipSuccess) return hip_fail(e__, #call); \
    } while (0)
static int einval(const char *msg)
{
    snprintf(g_err, sizeof(g_err), "invalid argument: %s", msg);
    return MI355_EINVAL;
}

extern "C" {

const char *mi355_last_error(void) { return g_err; }

int mi355_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int mi355_abi_version(void) { return MI355_ABI_VERSION; }

int mi355_init(int device)
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n == 0) {
        snprintf(g_err, sizeof(g_err), "no HIP device visible (%s)", e == hipSuccess ? "count 0" : hipGetErrorString(e));
        return MI355_ENODEV;
    }
    if (device < 0 || device >= n) return einval("device index");
    HIPCHK(hipSetDevice(device));
    hipDeviceProp_t p;
    HIPCHK(hipGetDeviceProperties(&p, device));
    if (strncmp(p.gcnArchName, "gfx950", 6) != 0) {
        snprintf(g_err, sizeof(g_err), "device %d is %s; this library is built for gfx950 (MI355X) only", device,
                 p.gcnArchName);
        return MI355_ENODEV;
    }
    return MI355_OK;
}

int mi355_alloc(void **dptr, size_t bytes)
{
    if (!dptr) return einval("dptr");
    hipError_t e = hipMalloc(dptr, bytes ? bytes : 16);
    if (e == hipErrorOutOfMemory) {
        snprintf(g_err, sizeof(g_err), "hipMalloc(%zu) out of memory", bytes);
        return MI355_ENOMEM;
    }
    if (e != hipSuccess) return hip_fail(e, "hipMalloc");
    return MI355_OK;
}
int mi355_free(void *dptr)
{
    HIPCHK(hipFree(dptr));
    return MI355_OK;
}
int mi355_memset(void *dptr, int byte, size_t bytes, void *stream)
{
    HIPCHK(hipMemsetAsync(dptr, byte, bytes, (hipStream_t)stream));
    return MI355_OK;
}
int mi355_h2d(void *dst, const void *src, size_t bytes, void *stream)
{
    HIPCHK(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, (hipStream_t)stream));
    return MI355_OK;
}
int mi355_d2h(void *dst, const void *src, size_t bytes, void *stream)
{
    HIPCHK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, (hipStream_t)stream));
    return MI355_OK;
}
int mi355_d2d(void *dst, const void *src, size_t bytes, void *stream)
{
    HIPCHK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return MI355_OK;
}
int mi355_stream_create(void **stream)
{
    hipStream_t s;
    HIPCHK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    *stream = s;
    return MI355_OK;
}
int mi355_stream_destroy(void *stream)
{
    HIPCHK(hipStreamDestroy((hipStream_t)stream));
    return MI355_OK;
}
// ---- streams that really run side by side ---------------------------------------------------------------------------------
// HIP maps the streams of a process onto the device's hardware queues round-robin in creation order (four queues; the default
// stream owns one): two streams on the same queue serialise.  Which created stream lands beside which depends on every stream
// any library of the process created before (measured: one unused stream created ahead of a network's three turns 0.277 ms per
// step into 0.311).  mi355_stream_acquire therefore MEASURES: per device it creates candidate streams and keeps those whose
// 200 us spin kernel OVERLAPS, by the device's own wall clock, the spin kernels of the default stream and of every stream kept so
// far.  The test reads device-side time stamps (start / end of each spin in wall_clock64 ticks, tick rate from
// hipDeviceAttributeWallClockRate), so a slow host, a profiler or other users of the device cannot turn "side by side" into
// "rejected".  Not to be called while a stream capture is active in the process: the measurement synchronises the device and
// launches on the NULL stream (once per device; later calls only hand out pool entries).
__global__ void mi355_spin_kernel(long long ticks, long long *stamp)  // bounded by construction
{
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
    if (stamp && threadIdx.x == 0) { stamp[0] = t0; stamp[1] = wall_clock64(); }
}
// ticks the two spins overlapped (device wall clock), or -1 on a HIP error
static long long spin_pair_overlap(hipStream_t a, hipStream_t b, long long ticks, long long *stamps_dev)
{
    long long h[4] = {0, 0, 0, 0};
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    hipLaunchKernelGGL(mi355_spin_kernel, dim3(1), dim3(64), 0, a, ticks, stamps_dev);
    hipLaunchKernelGGL(mi355_spin_kernel, dim3(1), dim3(64), 0, b, ticks, stamps_dev + 2);
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    if (hipMemcpy(h, stamps_dev, sizeof(h), hipMemcpyDeviceToHost) != hipSuccess) return -1;
    const long long lo = h[0] > h[2] ? h[0] : h[2], hi = h[1] < h[3] ? h[1] : h[3];
    return hi > lo ? hi - lo : 0;
}
static bool streams_concurrent(hipStream_t a, hipStream_t b, long long ticks, long long *stamps_dev)
{
    long long best = 0;
    for (int i = 0; i < 3; ++i) { const long long o = spin_pair_overlap(a, b, ticks, stamps_dev); if (o > best) best = o; }
    return best * 2 > ticks;  // side by side: nearly the whole spin; one behind the other: none of it
}
struct StreamPool { hipStream_t s[8]; bool used[8]; int n; bool built; };
static StreamPool g_pool[64];
static std::mutex g_pool_mu;
int mi355_stream_acquire(void **stream)
{
    if (!stream) return einval("stream_acquire: null");
    int dev = 0;
    HIPCHK(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lk(g_pool_mu);
    StreamPool &P = g_pool[dev & 63];
    if (!P.built) {
        int khz = 0;
        if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, dev) != hipSuccess || khz <= 0) khz = 100000;  // gfx950: 100 MHz
        const long long ticks = (long long)khz / 5;  // 200 us
        long long *stamps = nullptr;
        HIPCHK(hipMalloc(&stamps, 4 * sizeof(long long)));
        hipLaunchKernelGGL(mi355_spin_kernel, dim3(1), dim3(64), 0, 0, 1, stamps);  // module load + warm launch path outside the measurement
        (void)hipDeviceSynchronize();
        hipStream_t kept[8], rejected[8];
        int nk = 0, nr = 0;
        bool hip_ok = true;
        for (int c = 0; c < 8 && nk < 3 && hip_ok; ++c) {  // three created streams + the default one = the device's four queues
            hipStream_t cand;
            if (hipStreamCreateWithFlags(&cand, hipStreamNonBlocking) != hipSuccess) { hip_ok = false; break; }
            bool ok = streams_concurrent(nullptr, cand, ticks, stamps);
            for (int k = 0; ok && k < nk; ++k) ok = streams_concurrent(kept[k], cand, ticks, stamps);
            if (ok) kept[nk++] = cand;
            else rejected[nr++] = cand;  // stays alive until the pool is complete: destroying it now would hand its queue slot to the next creation
        }
        for (int k = 0; k < nr; ++k) (void)hipStreamDestroy(rejected[k]);
        (void)hipFree(stamps);
        if (!hip_ok) {  // nothing half-built is kept
            for (int k = 0; k < nk; ++k) (void)hipStreamDestroy(kept[k]);
            snprintf(g_err, sizeof(g_err), "stream_acquire: stream creation failed while measuring the pool");
            return MI355_EHIP;
        }
        for (int k = 0; k < nk; ++k) { P.s[k] = kept[k]; P.used[k] = false; }
        P.n = nk;
        P.built = true;
        if (nk < 3)
            fprintf(stderr, "mi355_stream_acquire: device %d: only %d created stream(s) ran side by side with the default stream "
                            "(expected 3: four hardware queues); further executors will share a queue\n", dev, nk);
    }
    for (int k = 0; k < P.n; ++k)
        if (!P.used[k]) { P.used[k] = true; *stream = P.s[k]; return MI355_OK; }
    hipStream_t s;  // pool exhausted: a further stream shares a queue with one of the others
    HIPCHK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    *stream = s;
    return MI355_OK;
}
int mi355_stream_release(void *stream)
{
    if (!stream) return MI355_OK;
    {
        std::lock_guard<std::mutex> lk(g_pool_mu);
        for (auto &P : g_pool)
            for (int k = 0; k < P.n; ++k)
                if (P.s[k] == (hipStream_t)stream) { P.used[k] = false; return MI355_OK; }
    }
    HIPCHK(hipStreamDestroy((hipStream_t)stream));
    return MI355_OK;
}
int mi355_stream_sync(void *stream)
{
    if (stream) HIPCHK(hipStreamSynchronize((hipStream_t)stream));
    else HIPCHK(hipDeviceSynchronize());
    return MI355_OK;
}
int mi355_event_create(void **ev)
{
    hipEvent_t e;
    // timing events between launches of one stream: no system-scope fence with every record (the host only reads the time stamps)
    // (2.4 instead of 3.9 us of stream time per event)
    HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableSystemFence));
    *ev = e;
    return MI355_OK;
}
int mi355_event_destroy(void *ev)
{
    HIPCHK(hipEventDestroy((hipEvent_t)ev));
    return MI355_OK;
}
int mi355_event_record(void *ev, void *stream)
{
    HIPCHK(hipEventRecord((hipEvent_t)ev, (hipStream_t)stream));
    return MI355_OK;
}
int mi355_event_elapsed_ms(void *start, void *stop, float *ms)
{
    HIPCHK(hipEventSynchronize((hipEvent_t)stop));
    HIPCHK(hipEventElapsedTime(ms, (hipEvent_t)start, (hipEvent_t)stop));
    return MI355_OK;
}
int mi355_graph_begin(void *stream)
{
    HIPCHK(hipStreamBeginCapture((hipStream_t)stream, hipStreamCaptureModeThreadLocal));
    return MI355_OK;
}
int mi355_graph_end(void *stream, void **graph_exec)
{
    hipGraph_t g;
    HIPCHK(hipStreamEndCapture((hipStream_t)stream, &g));
    hipGraphExec_t ge;
    hipError_t e = hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    (void)hipGraphDestroy(g);
    if (e != hipSuccess) return hip_fail(e, "hipGraphInstantiate");
    *graph_exec = ge;
    return MI355_OK;
}
int mi355_graph_launch(void *graph_exec, void *stream)
{
    HIPCHK(hipGraphLaunch((hipGraphExec_t)graph_exec, (hipStream_t)stream));
    return MI355_OK;
}
int mi355_graph_destroy(void *graph_exec)
{
    HIPCHK(hipGraphExecDestroy((hipGraphExec_t)graph_exec));
    return MI355_OK;
}

// ---------------------------------------------------------------------------------------------------- tensors
static size_t tensor_cells(const mi355_tensor *t)
{
    return (size_t)t->lead + (size_t)t->B * (t->H + 1) * (t->W + 1) + t->tail;
}

size_t mi355_tensor_describe(mi355_tensor *t, int B, int H, int W, int C)
{
    if (!t || B <= 0 || H <= 0 || W <= 0 || C <= 0) return 0;
    t->B = B; t->H = H; t->W = W; t->C = C;
    t->cs = (C == 3) ? 4 : ((C + 15) / 16) * 16;
    t->lead = 2;
    t->tail = W + 3;
    return tensor_cells(t) * (size_t)t->cs;
}

size_t mi355_tensor_describe_nchw(mi355_tensor *t, int B, int H, int W, int C)
{
    if (!t || B <= 0 || H <= 0 || W <= 0 || C <= 0) return 0;
    t->B = B; t->H = H; t->W = W; t->C = C;
    t->cs = 1; t->lead = 0; t->tail = 0;
    return (size_t)B * C * H * W;
}

int mi355_tensor_fill(const mi355_tensor *t, uint8_t zero_point, void *stream)
{
    if (!t || !t->data) return einval("tensor");
    if (t->cs == 1) return einval("tensor_fill: a planar tensor has no pad cells");
    const size_t cells = tensor_cells(t);
    if (t->cs == 4) {
        const uint32_t v = (uint32_t)zero_point | ((uint32_t)zero_point << 8) | ((uint32_t)zero_point << 16);
        return fill_u32_launch((uint32_t *)t->data, v, (long)cells, (hipStream_t)stream);
    }
    HIPCHK(hipMemsetAsync(t->data, zero_point ^ 0x80, cells * t->cs, (hipStream_t)stream));
    return MI355_OK;
}

int mi355_nchw_to_tensor(const uint8_t *nchw, const mi355_tensor *t, void *stream)
{
    if (!nchw || !t || !t->data) return einval("nchw_to_tensor");
    LayoutArgs a{const_cast<uint8_t *>(nchw), (uint8_t *)t->data, t->B, t->H, t->W, t->C, t->cs, t->lead};
    return nchw_to_phwc_launch(a, (hipStream_t)stream);
}
int mi355_tensor_to_nchw(const mi355_tensor *t, uint8_t *nchw, void *stream)
{
    if (!nchw || !t || !t->data) return einval("tensor_to_nchw");
    LayoutArgs a{nchw, (uint8_t *)t->data, t->B, t->H, t->W, t->C, t->cs, t->lead};
    return phwc_to_nchw_launch(a, (hipStream_t)stream);
}

// ---------------------------------------------------------------------------------------------- weight packing
static int default_bm(int n) { return n >= 128 ? 128 : (n > 32 ? 64 : 32); }
static size_t align16(size_t v) { return (v + 15) & ~(size_t)15; }

static thread_local int g_last_kernel = 0;
extern "C" int mi355_last_conv_kernel(void) { return g_last_kernel; }

static int blob_layout(int n, int c, int ksize, ConvBlobHeader *h)
{
    if (n <= 0 || c <= 0 || (ksize != 1 && ksize != 3)) return MI355_EINVAL;
    memset(h, 0, sizeof(*h));
    h->magic = MI355_BLOB_MAGIC;
    h->n = n; h->c = c; h->ksize = ksize;
    h->ktrue = c * ksize * ksize;
    size_t off = align16(sizeof(ConvBlobHeader));
    if (c == 3 && ksize == 3) {
        h->first = 1;
        h->mpad = ((n + 3) / 4) * 4;
        h->cb = 4; h->nchunks = 1; h->upc = 9; h->spc = 9; h->ksteps = 9;
        h->off_wp = off; off = align16(off + (size_t)h->mpad * 9 * 4);
    } else {
        if (c % 16) return MI355_EINVAL;
        const int bm = default_bm(n);
        h->mpad = ((n + bm - 1) / bm) * bm;
        h->cb = (c % 64 == 0) ? 64 : (c % 32 == 0 ? 32 : 16);
        h->nchunks = c / h->cb;
        h->upc = ksize * ksize * (h->cb / 16);
        h->spc = (h->upc + 3) / 4;
        h->ksteps = h->nchunks * h->spc;
        h->off_wp = off; off = align16(off + (size_t)h->mpad * h->ksteps * 64);
    }
    h->off_cw = off;   off = align16(off + (size_t)h->mpad * 4);
    h->off_dzp = off;  off = align16(off + (size_t)h->mpad * 4);
    h->off_bias = off; off = align16(off + (size_t)h->mpad * 4);
    h->off_mval = off; off = align16(off + (size_t)h->mpad * 8);
    h->off_sval = off; off = align16(off + (size_t)h->mpad * 8);
    h->off_shift = off; off = align16(off + (size_t)h->mpad * 4);
    h->off_mprime = off; off = align16(off + (size_t)h->mpad * 8);
    h->off_cwb = off; off = align16(off + (size_t)h->mpad * 4);
    if (conv1x1_ws_eligible(n, c, ksize)) {  // conv1x1.hip: [n/32 quads][c/32 K-steps][64 lanes][16 B]
        h->off_ws = off;
        off = align16(off + (size_t)((n + 31) / 32) * (c / 32) * 1024);
    }
    if (conv_ws3_eligible(n, c, ksize)) {  // conv_ws3.hip: [n/32 quads][c/128 K parts][36 K-steps][64 lanes][16 B]
        h->off_ws = off;
        off = align16(off + (size_t)n * c * 9);
    }
    if (conv_small_eligible(n, c, ksize)) {
        h->off_ws = off;
        off = align16(off + (size_t)(n / 32) * (c == 16 ? 5 : (c == 32 ? 9 : 18)) * 1024);
    }
    if (h->first || conv_small_eligible(n, c, ksize)) {  // the conv + maxpool kernels' epilogue table (common.h EptHeader)
        h->off_ept = off;
        // header, entries, the 4 KiB LEAKY byte table (conv_first_mfma_pool, conv_pool16), first layer: the kernel's per-lane state
        off = align16(off + sizeof(EptHeader) + (size_t)h->mpad * sizeof(EptEntry) + (size_t)LUTQ_N +
                      (h->first ? (size_t)((n + 15) / 16) * 64 * sizeof(L0Lane) : 0));
    }
    h->total = off;
    return MI355_OK;
}

size_t mi355_conv_pack_size(int n, int c, int ksize)
{
    ConvBlobHeader h;
    if (blob_layout(n, c, ksize, &h) != MI355_OK) return 0;
    return (size_t)h.total;
}

int mi355_conv_pack(int n, int c, int ksize, const uint8_t *wq, const uint8_t *zp_w, const int32_t *biases_int32,
                    const double *M_value, const double *shift_value, void *blob)
{
    ConvBlobHeader h;
    if (blob_layout(n, c, ksize, &h) != MI355_OK) return einval("conv_pack: need ksize 1|3 and c==3 or c%16==0");
    if (!wq || !zp_w || !biases_int32 || !M_value || !shift_value || !blob) return einval("conv_pack: null");
    char *base = (char *)blob;
    memset(base, 0, (size_t)h.total);
    h.pow2 = 1;
    for (int oc = 0; oc < n; ++oc) {
        int e = 0;
        const double m = frexp(shift_value[oc], &e);  // shift_value = m * 2^e, m in [0.5,1)
        if (m != 0.5 || e > 1 || e < -30) h.pow2 = 0;   // 2^-s == 0.5 * 2^(1-s)  ->  s = 1 - e in [0,31]
    }
    memcpy(base, &h, sizeof(h));
    int32_t *cw = (int32_t *)(base + h.off_cw), *dzp = (int32_t *)(base + h.off_dzp);
    int32_t *bias = (int32_t *)(base + h.off_bias);
    double *mval = (double *)(base + h.off_mval), *sval = (double *)(base + h.off_sval);
    const int K = h.ktrue;
    for (int oc = 0; oc < n; ++oc)
        if (!(M_value[oc] > 0.0 && M_value[oc] < 1.0) || !(shift_value[oc] > 0.0 && shift_value[oc] <= 1.0))
            return einval("conv_pack: need 0 < M_value < 1 and 0 < shift_value <= 1 (ref asserts 0<M<1, src/blas.c:391-392)");
    for (int oc = 0; oc < n; ++oc) {
        bias[oc] = biases_int32[oc];
        mval[oc] = M_value[oc];
        sval[oc] = shift_value[oc];
        if (h.pow2) {
            int e = 0;
            frexp(shift_value[oc], &e);
            ((int32_t *)(base + h.off_shift))[oc] = 1 - e;
        }
        const int d = 128 - (int)zp_w[oc];
        dzp[oc] = d;
        long sw = 0;
        for (int k = 0; k < K; ++k) sw += (int)wq[(size_t)oc * K + k] - 128;
        cw[oc] = (int32_t)(128 * sw + 128L * K * d);  // 128*sum(w') + 128*K*d   (|.| < 2^28 for K <= 9216)
        ((int32_t *)(base + h.off_cwb))[oc] = (int32_t)((uint32_t)cw[oc] + (uint32_t)biases_int32[oc]);
        ((double *)(base + h.off_mprime))[oc] = M_value[oc] * shift_value[oc];
    }
    if (h.first) {
        uint32_t *wp = (uint32_t *)(base + h.off_wp);
        for (int oc = 0; oc < n; ++oc)
            for (int t = 0; t < 9; ++t) {
                uint32_t v = 0;
                for (int ci = 0; ci < 3; ++ci) v |= (uint32_t)wq[(size_t)oc * K + ci * 9 + t] << (8 * ci);
                wp[oc * 9 + t] = v;
            }
        return MI355_OK;
    }
    if (h.off_ws && ksize == 1) {  // weights-stationary plane of conv1x1.hip: K-step s = channels 32 s + 16 kh .. + 15
        int8_t *ws = (int8_t *)(base + h.off_ws);
        const int kst = c / 32;
        for (int q = 0; q < (n + 31) / 32; ++q)
            for (int s = 0; s < kst; ++s)
                for (int lane = 0; lane < 64; ++lane) {
                    const int oc = 32 * q + ws_row_filter(lane & 31), khalf = lane >> 5;  // (rows permuted: kargs.h)
                    int8_t *dst = ws + ((size_t)(q * kst + s) * 64 + lane) * 16;
                    for (int e = 0; e < 16; ++e)
                        dst[e] = oc < n ? (int8_t)(wq[(size_t)oc * K + 32 * s + 16 * khalf + e] ^ 0x80) : 0;
                }
    }
    if (h.off_ws && conv_ws3_eligible(n, c, ksize)) {  // conv_ws3.hip: K-step s of part kp = tap s / 4, channels 128 kp + 32 (s % 4) + 16 kh ..
        int8_t *ws = (int8_t *)(base + h.off_ws);
        const int kparts = c / 128;
        for (int q = 0; q < n / 32; ++q)
            for (int kp = 0; kp < kparts; ++kp)
                for (int s = 0; s < 36; ++s)
                    for (int lane = 0; lane < 64; ++lane) {
                        const int oc = 32 * q + ws_row_filter(lane & 31), khalf = lane >> 5, tap = s >> 2;  // (rows permuted: kargs.h)
                        int8_t *dst = ws + ((size_t)((q * kparts + kp) * 36 + s) * 64 + lane) * 16;
                        for (int e = 0; e < 16; ++e) {
                            const int ci = 128 * kp + 32 * (s & 3) + 16 * khalf + e;
                            dst[e] = (int8_t)(wq[(size_t)oc * K + (ci * 3 + tap / 3) * 3 + tap % 3] ^ 0x80);
                        }
                    }
    }
    if (h.off_ws && conv_small_eligible(n, c, ksize)) {  // weights-stationary plane of conv_small.hip: lane (row lj, k-half kh) of K-step s
        int8_t *ws = (int8_t *)(base + h.off_ws);
        const int kst = (c == 16) ? 5 : (c == 32 ? 9 : 18);
        for (int mt = 0; mt < n / 32; ++mt)
            for (int s = 0; s < kst; ++s)
                for (int lane = 0; lane < 64; ++lane) {
                    const int oc = 32 * mt + ws_row_filter(lane & 31), khalf = lane >> 5;  // (rows permuted: kargs.h)
                    // c 16: the k-half is the tap parity (tap 9: zeros); c 32: one tap per step; c 64: two steps per tap
                    const int tap = (c == 16) ? 2 * s + khalf : (c == 32 ? s : s / 2);
                    int8_t *dst = ws + ((size_t)(mt * kst + s) * 64 + lane) * 16;
                    for (int e = 0; e < 16; ++e) {
                        const int ci = (c == 16) ? e : (c == 32 ? 16 * khalf + e : 32 * (s & 1) + 16 * khalf + e);
                        dst[e] = tap > 8 ? 0 : (int8_t)(wq[(size_t)oc * K + (ci * 3 + tap / 3) * 3 + tap % 3] ^ 0x80);
                    }
                }
    }
    int8_t *wp = (int8_t *)(base + h.off_wp);
    const int bpc = h.cb / 16;
    for (int oc = 0; oc < n; ++oc) {
        const int mt = oc / 16, row = oc % 16;
        for (int chunk = 0; chunk < h.nchunks; ++chunk)
            for (int s = 0; s < h.spc; ++s) {
                const int g = chunk * h.spc + s;
                int8_t *dst = wp + ((size_t)mt * h.ksteps + g) * 1024 + row * 16;  // [piece][row][16]
                for (int kg = 0; kg < 4; ++kg) {
                    const int u = 4 * s + kg;
                    if (u >= h.upc) continue;
                    const int tap = u / bpc, blk = u % bpc;
                    const int ky = tap / ksize, kx = tap % ksize;
                    for (int e = 0; e < 16; ++e) {
                        const int ci = chunk * h.cb + blk * 16 + e;
                        const uint8_t w = wq[(size_t)oc * K + (ci * ksize + ky) * ksize + kx];  // (ci,ky,kx) order
                        dst[kg * 256 + e] = (int8_t)(w ^ 0x80);
                    }
                }
            }
    }
    return MI355_OK;
}

extern "C++" {
template <int ACT>
static void ept_fill(const ConvBlobHeader &h, char *base, int zp_act)
{
    EptHeader *eh = (EptHeader *)(base + h.off_ept);
    EptEntry *e = (EptEntry *)(eh + 1);
    const double *mprime = (const double *)(base + h.off_mprime), *mval = (const double *)(base + h.off_mval);
    const int32_t *shift = (const int32_t *)(base + h.off_shift), *cwb = (const int32_t *)(base + h.off_cwb);
    uint32_t flags = h.pow2 ? EPT_POW2 : EPT_NOINT;
    for (int oc = 0; oc < h.n; ++oc) {
        int32_t lo, hi, lb = 0, m0 = 0, sh = 0;
        uint32_t rg = 0;
        small_safe_range<ACT>(mprime[oc], zp_act, lo, hi);
        if (!biased_safe_range(lo, hi, lb, rg)) flags |= EPT_NEVER;
        if (!h.pow2 || !intrq_make(mval[oc], shift[oc], lb, (int32_t)((uint32_t)lb + rg), m0, sh, ACT == MI355_ACT_RELU6)) {
            flags |= EPT_NOINT;
            m0 = sh = 0;
        }
        e[oc].lb = lb; e[oc].rg = rg; e[oc].m0 = m0; e[oc].sh = sh;
        e[oc].qc = (int64_t)lb * (int64_t)m0;
        e[oc].cbl = (int32_t)((uint32_t)cwb[oc] - (uint32_t)lb);
        e[oc].pad_ = 0;
    }
    for (int oc = h.n; oc < h.mpad; ++oc) memset(&e[oc], 0, sizeof(EptEntry));
    // the LEAKY byte table (common.h leaky_lut_build / leaky_lutf_build): indexed by f (integer form) or by q
    uint8_t *lut = (uint8_t *)(e + h.mpad);
    {
        const bool by_floor = !(flags & EPT_NOINT);
        for (int i = 0; i < LUTQ_N; ++i) {
            const int f = i - LUTQ_OFF;
            lut[i] = ACT == MI355_ACT_LEAKY ? (uint8_t)leaky_byte_biased<false>(by_floor && f < 0 ? f + 1 : f, zp_act) : 0;
        }
    }
    if (h.first) {
        // the first-layer MFMA kernel's per-lane state (common.h L0Lane)
        L0Lane *ll = (L0Lane *)(lut + LUTQ_N);
        const uint32_t *wfirst = (const uint32_t *)(base + h.off_wp);
        const int32_t *dzp = (const int32_t *)(base + h.off_dzp);
        for (int mt = 0; mt < (h.n + 15) / 16; ++mt)
            for (int lane = 0; lane < 64; ++lane) {
                L0Lane &L = ll[mt * 64 + lane];
                memset(&L, 0, sizeof(L));
                const int pc = lane & 15, g = lane >> 4, ch = 16 * mt + pc;
                if (ch < h.n) {
                    const int dz = dzp[ch];
                    const int d1 = dz > 127 ? 127 : dz, d2 = dz - d1;  // dz in [-127, 128]
                    if (d2) flags |= EPT_D2;
                    const uint32_t m1 = (uint32_t)(d1 & 0xFF) * 0x00010101u, m2 = (uint32_t)(d2 & 0xFF) * 0x00010101u;
                    for (int jx = 0; jx < 2; ++jx)
                        for (int dx = 0; dx < 4; ++dx) {
                            const int t = dx - jx;  // tap column of cell dx for window column jx
                            if (g < 3 && t >= 0 && t < 3) {
                                L.wa[jx][dx] = (int32_t)(wfirst[ch * 9 + 3 * g + t] ^ 0x00808080u);
                                L.wd1[jx][dx] = (int32_t)m1;
                                L.wd2[jx][dx] = (int32_t)m2;
                            }
                        }
                }
                for (int r = 0; r < 4; ++r) {
                    const int c2 = 16 * mt + 4 * g + r;
                    if (c2 >= h.n) continue;
                    L.cb[r] = e[c2].cbl; L.lo[r] = e[c2].lb; L.hi[r] = (int32_t)e[c2].rg;
                    L.qm0[r] = e[c2].m0; L.qsh[r] = e[c2].sh; L.qc[r] = e[c2].qc;
                    L.mp[r] = mprime[c2];
                }
            }
    }
    eh->flags = flags;
    eh->pad_[0] = eh->pad_[1] = 0;
    eh->key = ept_key(ACT, zp_act);
}
}  // extern "C++"

int mi355_conv_pack_epilogue(int n, int c, int ksize, int activation, int zp_act, void *blob)
{
    ConvBlobHeader h;
    if (blob_layout(n, c, ksize, &h) != MI355_OK || !blob) return einval("conv_pack_epilogue: shape / null");
    ConvBlobHeader have;
    memcpy(&have, blob, sizeof(have));
    if (have.magic != MI355_BLOB_MAGIC || have.n != n || have.c != c || have.ksize != ksize || have.total != h.total)
        return einval("conv_pack_epilogue: not a mi355_conv_pack blob of this shape");
    if (zp_act < 0 || zp_act > 255) return einval("conv_pack_epilogue: zp_act");
    if (!h.off_ept) return MI355_OK;  // no kernel of this shape reads the table
    h.pow2 = have.pow2;
    char *base = (char *)blob;
    switch (activation) {
    case MI355_ACT_LEAKY: ept_fill<MI355_ACT_LEAKY>(h, base, zp_act); break;
    case MI355_ACT_RELU6: ept_fill<MI355_ACT_RELU6>(h, base, zp_act); break;
    case MI355_ACT_RELU:  // the integer path stores q + zp for RELU as for LINEAR (ref src/convolutional_layer.c:740-742); the kernels' LINEAR instantiation serves both
    case MI355_ACT_LINEAR: ept_fill<MI355_ACT_LINEAR>(h, base, zp_act); break;
    default: return einval("conv_pack_epilogue: activation");
    }
    return MI355_OK;
}

// ------------------------------------------------------------------------------------------------ convolution
static int conv_forward_impl(const mi355_conv_desc *d, const mi355_tensor *x, const void *blob, const uint8_t *w_u8,
                             const uint8_t *zp_w, const mi355_tensor *y, const mi355_tensor *ypool, int32_t *acc_out,
                             float *y_f32, void *stream, float *yolo_out = nullptr, int yolo_classes = 0, int up = 1,
                             const mi355_tensor *res = nullptr, int sc_ka = 0, int sc_kb = 0, int sc_k0 = 0)
{
    if (!d || !x || !x->data || !blob) return einval("conv_forward: null");
    if (d->stride != 1 && d->stride != 2) return einval("conv_forward: stride must be 1 or 2");
    if (d->stride == 2 && (d->ksize != 3 || ypool || up != 1 || yolo_out || d->accum_mode != MI355_ACC_EXACT || d->c % 16))
        return einval("conv_forward: stride 2 exists for plain exact-mode 3x3 convs with c % 16 == 0 only (the reference's 1x1 path "
                      "feeds the input to the GEMM unsampled, src/convolutional_layer.c:711-716)");
    const int OH = (x->H + 2 * d->pad - d->ksize) / d->stride + 1, OW = (x->W + 2 * d->pad - d->ksize) / d->stride + 1;
    if (!((d->ksize == 3 && d->pad == 1) || (d->ksize == 1 && d->pad == 0))) return einval("conv_forward: ksize/pad");
    if (x->C != d->c) return einval("conv_forward: x.C != desc.c");
    if (y && (y->C != d->n || y->B != x->B || y->H != OH * up || y->W != OW * up || !y->data))
        return einval("conv_forward: y shape");
    // ypool of the pooled size: the 2x2 / stride-2 maxpool; ypool of the conv's own size: the 2x2 / stride-1 maxpool (the
    // reference's pad = 1: windows y..y+1, x..x+1 clipped at the border, src/maxpool_layer.c:109-146)
    const bool pool_s1 = ypool && d->stride == 1 && ypool->H == OH && ypool->W == OW && OH > 1;
    if (ypool && !pool_s1) {
        if ((x->H & 1) || (x->W & 1) || ypool->C != d->n || ypool->B != x->B || ypool->H != x->H / 2 ||
            ypool->W != x->W / 2 || !ypool->data || d->ksize != 3 || d->accum_mode != MI355_ACC_EXACT || acc_out || y_f32)
            return einval("conv_pool_forward: fused 2x2/2 maxpool needs a 3x3 conv on an even map, exact mode, no dumps");
    }
    if (pool_s1 && (ypool->C != d->n || ypool->B != x->B || !ypool->data || d->ksize != 3 || d->accum_mode != MI355_ACC_EXACT || acc_out || y_f32))
        return einval("conv_pool_forward: fused 2x2/1 maxpool needs a 3x3 stride-1 conv, exact mode, no dumps");
    ConvBlobHeader h;
    if (blob_layout(d->n, d->c, d->ksize, &h) != MI355_OK) return einval("conv_forward: shape");
    const char *base = (const char *)blob;
    hipStream_t st = (hipStream_t)stream;
    const int total_n = x->B * OH * OW;
    const int in_cells = (int)tensor_cells(x);

    if (x->cs == 1 && (!h.first || d->accum_mode == MI355_ACC_REF_F32))
        return einval("conv_forward: the planar (cs==1) layout is accepted for the exact-mode 3-channel first layer only");
    if (res && (d->accum_mode == MI355_ACC_REF_F32 || h.first || !y || ypool || acc_out || y_f32 || yolo_out || up != 1 || d->stride != 1))
        return einval("conv_shortcut_forward: plain exact-mode stride-1 convolutions with c % 16 == 0 only");
    if (d->accum_mode == MI355_ACC_REF_F32 || h.first) {
        AuxArgs a;
        memset(&a, 0, sizeof(a));
        a.x = (const uint8_t *)x->data; a.in_cs = x->cs; a.in_lead = x->lead; a.in_cells = in_cells;
        a.wfirst = (const uint32_t *)(base + h.off_wp);
        a.w_u8 = w_u8; a.zp_w = zp_w;
        a.dzp = (const int32_t *)(base + h.off_dzp); a.bias = (const int32_t *)(base + h.off_bias);
        a.mval = (const double *)(base + h.off_mval); a.sval = (const double *)(base + h.off_sval);
        a.y = y ? (uint8_t *)y->data : nullptr; a.out_cs = y ? y->cs : 0; a.out_lead = y ? y->lead : 0;
        a.acc_out = acc_out; a.y_f32 = y_f32;
        a.B = x->B; a.H = x->H; a.W = x->W; a.c = d->c; a.n = d->n; a.ksize = d->ksize; a.pad = d->pad;
        a.zp_in = d->zp_in; a.zp_act = d->zp_act; a.act = d->activation; a.store_mode = d->store_mode;
        a.s_act = d->s_act; a.total_n = total_n;
        a.mprime = (const double *)(base + h.off_mprime); a.hdr = (const ConvBlobHeader *)base;
        a.cwb = (const int32_t *)(base + h.off_cwb);
        a.ept = h.off_ept ? (const EptHeader *)(base + h.off_ept) : nullptr;
        a.ypool = ypool ? (uint8_t *)ypool->data : nullptr; a.pool_cs = ypool ? ypool->cs : 0;
        a.pool_lead = ypool ? ypool->lead : 0;
        if (d->accum_mode == MI355_ACC_REF_F32) {
            if (ypool) return einval("conv_pool_forward: not in ref-f32 mode");
            if (!w_u8 || !zp_w) return einval("conv_forward: ref-f32 mode needs the raw weights_uint8 / zp_w");
            g_last_kernel = 6;
            return conv_ref_f32_launch(a, st);
        }
        g_last_kernel = 1;
        if (pool_s1) return einval("conv_pool_forward: the stride-1 pool is fused for 128 / 256-channel 3x3 layers only");
        if (x->cs != 4 && x->cs != 1) return einval("conv_forward: first layer expects a cs==4 image tensor or the planar (cs==1) reference layout");
        a.planar = x->cs == 1;
        a.debug_flags = mi355_debug_flags_get();
        if (a.planar) a.in_cells = 0;
        if (ypool) {
            int rc = (mi355_debug_flags_get() & 1024) ? MI355_EINVAL : conv_first_mfma_pool_launch(a, st);
            if (rc == MI355_EINVAL && !a.planar) rc = conv_first_pool_launch(a, st);
            return rc == MI355_OK ? MI355_OK : einval("conv_pool_forward: shape not fusable");
        }
        {  // no pool: the MFMA kernel where it applies (even maps, 16 / 32 filters, no dumps), else the VALU kernel
            const int rc = (mi355_debug_flags_get() & 1024) ? MI355_EINVAL : conv_first_mfma_launch(a, st);
            if (rc != MI355_EINVAL) return rc;
        }
        if (a.planar) return einval("conv_forward: the planar input layout is served by the first-layer MFMA kernels only (16 / 32 filters, "
                                    "even map, W % 4 == 0, no dumps); convert with mi355_nchw_to_tensor otherwise");
        return conv_first_launch(a, st);
    }
    if (x->cs % 16) return einval("conv_forward: x.cs must be a multiple of 16");
    ConvArgs a;
    memset(&a, 0, sizeof(a));
    a.x = (const int8_t *)x->data; a.wp = (const int8_t *)(base + h.off_wp);
    a.cw = (const int32_t *)(base + h.off_cw); a.dzp = (const int32_t *)(base + h.off_dzp);
    a.bias = (const int32_t *)(base + h.off_bias);
    a.mval = (const double *)(base + h.off_mval); a.sval = (const double *)(base + h.off_sval);
    a.y = y ? (uint8_t *)y->data : nullptr; a.acc_out = acc_out; a.y_f32 = y_f32;
    a.in_cs = x->cs; a.in_lead = x->lead; a.in_cells = in_cells;
    a.out_cs = y ? y->cs : 0; a.out_lead = y ? y->lead : 0;
    a.out_w = y ? ((y->C + 15) & ~15) : 0;  // bytes of a cell this layer owns (y may be a channel window of a wider tensor)
    a.B = x->B; a.H = x->H; a.W = x->W; a.n = d->n;
    a.stride = d->stride; a.OH = OH; a.OW = OW;
    a.ksize = d->ksize; a.cb = h.cb; a.nchunks = h.nchunks; a.upc = h.upc; a.spc = h.spc; a.ksteps = h.ksteps;
    a.total_n = total_n;
    a.zp_act = d->zp_act; a.act = d->activation; a.store_mode = d->store_mode; a.s_act = d->s_act;
    a.mpad = h.mpad;
    a.ypool = ypool ? (uint8_t *)ypool->data : nullptr; a.pool_cs = ypool ? ypool->cs : 0;
    a.pool_lead = ypool ? ypool->lead : 0;
    a.pool_w = ypool ? ((ypool->C + 15) & ~15) : 0;
    a.shift = (const int32_t *)(base + h.off_shift);
    a.mprime = (const double *)(base + h.off_mprime);
    a.cwb = (const int32_t *)(base + h.off_cwb);
    a.hdr = (const ConvBlobHeader *)base;  // device copy: the kernel reads the data-dependent pow2 flag from it
    a.ws = h.off_ws ? (const int8_t *)(base + h.off_ws) : nullptr;
    a.ept = h.off_ept ? (const EptHeader *)(base + h.off_ept) : nullptr;
    a.yolo_out = yolo_out; a.yolo_per = yolo_classes + 5;
    a.up = up;
    a.plan = d->plan;
    if (res) {
        a.res = (const uint8_t *)res->data; a.res_cs = res->cs; a.res_delta = res->lead - y->lead;
        a.sc_ka = sc_ka; a.sc_kb = sc_kb; a.sc_k0 = sc_k0;
    }
    if (up != 1 && (h.cb != 64 || ypool)) return einval("conv_upsample_forward: 64-channel-chunk layers only");
    int rc = MI355_EINVAL;
    a.pool_mode = ypool ? (pool_s1 ? 1 : 2) : 0;
    if (ypool && (pool_s1 || d->c == 128 || d->c == 256)) {
        // the stride-1 pool and the 128 / 256-channel layers: conv_ws3.hip or nothing
        // (the generic fused kernel would be slower than the two layers run separately)
        if (a.ws && !res && up == 1 && !yolo_out && !(mi355_debug_flags_get() & (16384 | (1 << 21)))) { rc = conv_ws3_launch(a, st); g_last_kernel = 4; }  // (bit 21: A/B runs without these fusions)
        if (rc == MI355_EINVAL) return einval("conv_pool_forward: shape not fusable");
        if (rc != MI355_OK) return hip_fail(hipGetLastError(), "conv_ws3 launch");
        return rc;
    }
    // few-channel layers with the epilogue table: the 16 x 16 x 64 form (conv_pool16.hip; debug bit 2^30 sends them back to conv_small.hip for A/B runs)
    if (ypool && !y && d->epilogue_packed && a.ept && !(mi355_debug_flags_get() & (1024 | (1 << 30)))) { rc = conv_pool16_launch(a, st); if (rc != MI355_EINVAL) g_last_kernel = 7; }
    // 32 -> 64 + maxpool: the occupancy cut of conv_small.hip (conv_small32.hip, round 6).  Measured slower than conv_small.hip alone and in flight
    // (profiles/r06_small32_*): NOT in the default path; debug bit 4096 selects it (A/B runs, parity tests)
    if (rc == MI355_EINVAL && ypool && !y && a.ws && (mi355_debug_flags_get() & (1024 | 4096)) == 4096) { rc = conv_small32_launch(a, st); if (rc != MI355_EINVAL) g_last_kernel = 8; }
    if (rc == MI355_EINVAL && ypool && !y && a.ws && !(mi355_debug_flags_get() & 1024)) { rc = conv_small_pool_launch(a, st); g_last_kernel = 2; }  // few-channel layers
    // the same kernel without the pool: few-channel 3x3 layers of the non-tiny nets (even maps, no dumps)
    if (rc == MI355_EINVAL && !ypool && y && a.ws && d->ksize == 3 && up == 1 && !acc_out && !y_f32 && !yolo_out &&
        (d->c == 16 || d->c == 32 || d->c == 64) && !(mi355_debug_flags_get() & 1024)) { rc = conv_small_pool_launch(a, st); g_last_kernel = 2; }
    if (rc == MI355_EINVAL && a.ws && d->ksize == 1 && !(mi355_debug_flags_get() & 8192)) { rc = conv1x1_ws_launch(a, st); g_last_kernel = 3; }  // 1x1 layers
    // a fused yolo head WITHOUT the conv's own float tensor (y_f32 == NULL) exists in conv1x1.hip only: the caller passes the buffer otherwise
    if (rc == MI355_EINVAL && yolo_out && !y_f32) return einval("conv_yolo_forward: this shape needs y_f32");
    if (rc == MI355_EINVAL && a.ws && d->ksize == 3 && !ypool && !(mi355_debug_flags_get() & 16384)) { rc = conv_ws3_launch(a, st); g_last_kernel = 4; }  // mid layers
    if (rc == MI355_EINVAL) { rc = conv_igemm_launch(a, st); g_last_kernel = 5; }
    if (rc == MI355_EINVAL && res) return einval("conv_shortcut_forward: no kernel fuses the residual add for this shape");
    if (rc == MI355_EINVAL) return einval("conv_forward: no tile configuration fits this shape");
    if (rc != MI355_OK) return hip_fail(hipGetLastError(), "conv_igemm launch");
    return rc;
}

int mi355_conv_forward(const mi355_conv_desc *d, const mi355_tensor *x, const void *blob, const uint8_t *w_u8,
                       const uint8_t *zp_w, const mi355_tensor *y, int32_t *acc_out, float *y_f32, void *stream)
{
    return conv_forward_impl(d, x, blob, w_u8, zp_w, y, nullptr, acc_out, y_f32, stream);
}

int mi355_conv_yolo_forward(const mi355_conv_desc *d, const mi355_tensor *x, const void *blob, const mi355_tensor *y,
                            float *y_f32, float *yolo_out, int classes, void *stream)
{
    if (!d || !yolo_out || classes < 0 || d->n % (classes + 5)) return einval("conv_yolo_forward: need yolo_out and n % (classes + 5) == 0");
    if (d->accum_mode != MI355_ACC_EXACT || d->c % 16) return einval("conv_yolo_forward: exact mode, c % 16 == 0 only");
    return conv_forward_impl(d, x, blob, nullptr, nullptr, y, nullptr, nullptr, y_f32, stream, yolo_out, classes);
}

int mi355_conv_upsample_forward(const mi355_conv_desc *d, const mi355_tensor *x, const void *blob, const mi355_tensor *y_up,
                                int stride, void *stream)
{
    if (!d || !y_up || stride < 1 || stride > 4) return einval("conv_upsample_forward: y_up, 1 <= stride <= 4");
    if (d->accum_mode != MI355_ACC_EXACT || d->c % 64) return einval("conv_upsample_forward: exact mode, c % 64 == 0 only");
    return conv_forward_impl(d, x, blob, nullptr, nullptr, y_up, nullptr, nullptr, nullptr, stream, nullptr, 0, stride);
}

int mi355_conv_shortcut_forward(const mi355_conv_desc *d, const mi355_tensor *x, const void *blob, const mi355_tensor *from,
                                const mi355_tensor *y_sum, int32_t Ka, int32_t Kb, uint8_t zp_from, uint8_t zp_out, void *stream)
{
    if (!d || !from || !from->data || !y_sum) return einval("conv_shortcut_forward: null");
    if (from->cs % 16 || from->B != y_sum->B || from->H != y_sum->H || from->W != y_sum->W || from->C != y_sum->C)
        return einval("conv_shortcut_forward: `from` and the sum must share batch, map and channels");
    if (Ka < 1 || Kb < 1 || Ka >= (1 << 21) || Kb >= (1 << 21)) return einval("conv_shortcut_forward: multipliers must be in [1, 2^21)");
    const int k0 = 32768 + ((int)zp_out << 16) - Ka * (int)d->zp_act - Kb * (int)zp_from;
    return conv_forward_impl(d, x, blob, nullptr, nullptr, y_sum, nullptr, nullptr, nullptr, stream, nullptr, 0, 1, from, Ka, Kb, k0);
}

int mi355_conv_pool_forward(const mi355_conv_desc *d, const mi355_tensor *x, const void *blob, const mi355_tensor *y,
                            const mi355_tensor *ypool, void *stream)
{
    if (!ypool) return einval("conv_pool_forward: ypool is null");
    return conv_forward_impl(d, x, blob, nullptr, nullptr, y, ypool, nullptr, nullptr, stream);
}

// -------------------------------------------------------------------------------------------------------- glue
int mi355_maxpool_forward(const mi355_tensor *x, const mi355_tensor *y, int size, int stride, int pad, void *stream)
{
    if (!x || !y || !x->data || !y->data) return einval("maxpool: null");
    if (x->cs % 16 || y->cs % 16 || x->C != y->C || x->B != y->B) return einval("maxpool: layout");
    const int oh = (x->H + pad - size) / stride + 1, ow = (x->W + pad - size) / stride + 1;  // ref :31-32
    if (oh != y->H || ow != y->W) return einval("maxpool: output dims");
    PoolArgs a{(const uint8_t *)x->data, (uint8_t *)y->data, x->B, x->H, x->W, oh, ow, x->cs, y->cs, x->lead, y->lead,
               (x->C + 15) / 16, size, stride, -pad / 2};
    return maxpool_launch(a, (hipStream_t)stream);
}

int mi355_upsample_forward(const mi355_tensor *x, const mi355_tensor *y, int stride, void *stream)
{
    if (!x || !y || !x->data || !y->data) return einval("upsample: null");
    if (x->cs % 16 || y->cs % 16 || x->C != y->C || x->B != y->B || y->H != x->H * stride || y->W != x->W * stride)
        return einval("upsample: shape");
    CopyArgs a{(const uint8_t *)x->data, (uint8_t *)y->data, x->B, x->H, x->W, y->H, y->W, x->cs, y->cs, x->lead,
               y->lead, (x->C + 15) / 16, stride, 0};
    return copy_cells_launch(a, (hipStream_t)stream);
}

int mi355_route_forward(const mi355_tensor *const *xs, int n, const mi355_tensor *y, void *stream)
{
    if (!xs || n <= 0 || !y || !y->data) return einval("route: null");
    int coff = 0;
    bool aligned = true;  // every input starts on a 16-byte group of the output cell
    for (int i = 0; i < n; ++i) {
        const mi355_tensor *x = xs[i];
        if (!x || !x->data || x->cs % 16 || x->B != y->B || x->H != y->H || x->W != y->W) return einval("route: input layout");
        if (i + 1 < n && x->C % 16) aligned = false;
        coff += x->C;
    }
    if (coff != y->C) return einval("route: channel sum != y.C");
    coff = 0;
    for (int i = 0; i < n; ++i) {
        const mi355_tensor *x = xs[i];
        CopyArgs a{(const uint8_t *)x->data, (uint8_t *)y->data, x->B, x->H, x->W, y->H, y->W, x->cs, y->cs, x->lead,
                   y->lead, (x->C + 15) / 16, 1, coff};
        // 16-byte groups when every offset is aligned (the last input may end inside a group: its pad bytes land in the output
        // cell's own padding); else byte by byte
        const int rc = aligned ? copy_cells_launch(a, (hipStream_t)stream) : copy_cell_bytes_launch(a, x->C, (hipStream_t)stream);
        if (rc) return rc;
        coff += x->C;
    }
    return MI355_OK;
}

int mi355_dequant_forward(const mi355_tensor *x, int c0, int nc, uint8_t zero_point, float scale, float *out_f32, int out_C,
                          int out_c0, void *stream)
{
    if (!x || !x->data || !out_f32 || x->cs % 16) return einval("dequant: null / layout");
    if (c0 < 0 || nc <= 0 || c0 + nc > x->C || out_c0 < 0 || out_c0 + nc > out_C) return einval("dequant: channel range");
    DequantArgs a{(const uint8_t *)x->data, out_f32, x->B, x->H, x->W, x->cs, x->lead, c0, nc, out_C, out_c0, (int)zero_point, scale};
    return dequant_cells_launch(a, (hipStream_t)stream);
}

int mi355_shortcut_multiplier(float s_in, float s_out, int32_t *K)
{
    if (!K || !(s_in > 0.0f) || !(s_out > 0.0f)) return einval("shortcut_multiplier: scales must be positive");
    const float ratio = s_in / s_out;  // float, like M in the reference's prep (src/blas.c:313)
    const double k = round((double)ratio * 65536.0);
    if (!(k >= 1.0) || !(k < 2097152.0)) return einval("shortcut_multiplier: need 2^-16 <= s_in / s_out < 32");
    *K = (int32_t)k;
    return MI355_OK;
}

int mi355_shortcut_forward(const mi355_tensor *a, const mi355_tensor *b, const mi355_tensor *y, int32_t Ka, int32_t Kb,
                           uint8_t zp_a, uint8_t zp_b, uint8_t zp_out, void *stream)
{
    if (!a || !b || !y || !a->data || !b->data || !y->data) return einval("shortcut: null");
    if (a->cs % 16 || b->cs % 16 || y->cs % 16) return einval("shortcut: layout");
    if (a->B != y->B || b->B != y->B || a->H != y->H || b->H != y->H || a->W != y->W || b->W != y->W || a->C != y->C || b->C != y->C)
        return einval("shortcut: both inputs and the output must share batch, map size and channels");
    if (Ka < 1 || Kb < 1 || Ka >= (1 << 21) || Kb >= (1 << 21)) return einval("shortcut: multipliers must be in [1, 2^21)");
    ShortcutArgs s{(const uint8_t *)a->data, (const uint8_t *)b->data, (uint8_t *)y->data, y->B, y->H, y->W, (y->C + 15) / 16,
                   a->cs, b->cs, y->cs, a->lead, b->lead, y->lead, Ka, Kb,
                   32768 + ((int)zp_out << 16) - Ka * (int)zp_a - Kb * (int)zp_b};
    return shortcut_launch(s, (hipStream_t)stream);
}

int mi355_letterbox_forward(const float *im_f32, int imw, int imh, int c, float *out_f32, int w, int h, void *stream)
{
    if (!im_f32 || !out_f32 || imw < 1 || imh < 1 || c < 1 || w < 2 || h < 2) return einval("letterbox: null / bad size");
    const int rc = letterbox_launch(im_f32, imw, imh, c, out_f32, w, h, (hipStream_t)stream);
    return rc == MI355_EINVAL ? einval("letterbox: degenerate aspect (resized side < 2)") : rc;
}

int mi355_image_minmax(const float *x_f32, long count, float *minmax, void *stream)
{
    if (!x_f32 || !minmax || count <= 0) return einval("image_minmax: null / empty");
    return image_minmax_launch(x_f32, count, reinterpret_cast<uint32_t *>(minmax), (hipStream_t)stream);
}

int mi355_image_quantize(const float *x_f32, long count, float scale, int zero_point, uint8_t *out_u8, void *stream)
{
    if (!x_f32 || !out_u8 || count <= 0) return einval("image_quantize: null / empty");
    if (!(scale > 0.0f) || zero_point < 0 || zero_point > 255) return einval("image_quantize: need scale > 0 and 0 <= zero_point <= 255");
    return image_quantize_launch(x_f32, count, scale, zero_point, out_u8, (hipStream_t)stream);
}

int mi355_checksum_u32(const void *buf, long dwords, uint64_t *sum_dev, void *stream)
{
    if (!buf || !sum_dev || dwords < 0) return einval("checksum: null");
    return checksum_u32_launch((const uint32_t *)buf, dwords, (unsigned long long *)sum_dev, (hipStream_t)stream);
}

int mi355_yolo_forward(const float *in, float *out, int B, int n, int classes, int H, int W, void *stream)
{
    if (!in || !out) return einval("yolo: null");
    return yolo_logistic_launch(in, out, B, n, classes, H * W, (hipStream_t)stream);
}

int mi355_yolo_detections(const float *yolo_out, int B, int n, int classes, int H, int W, const float *anchors,
                          const int *mask, int netw, int neth, int imw, int imh, float thresh, int relative, float *recs,
                          int max_recs, int *counts, void *stream)
{
    if (!yolo_out || !anchors || !mask || !recs || !counts || B <= 0 || n <= 0 || classes < 0 || max_recs <= 0 || imw <= 0 ||
        imh <= 0)
        return einval("yolo_detections: bad argument");
    return yolo_detections_launch(yolo_out, B, n, classes, H, W, anchors, mask, netw, neth, imw, imh, thresh, relative, recs,
                                  max_recs, counts, (hipStream_t)stream);
}

}  // extern "C"
