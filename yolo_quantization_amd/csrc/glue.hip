// glue.hip -- uint8 glue layers on the PHWC device layout (all HBM-bound: 16-byte accesses, one thread per
// (cell, 16-channel group)), layout converters and the float yolo head activations.
//
//   maxpool_u8_kernel   ref: src/maxpool_layer.c:109-172 (window offset -pad/2, out-of-image taps never win)
//   upsample_u8_kernel  ref: src/upsample_layer.c:96-113 -> src/blas.c:781-803 (nearest, forward, scale == 1)
//   route_u8_kernel     ref: src/route_layer.c:107-130 (byte concat along channels, no rescale)
//   yolo_logistic       ref: src/yolo_layer.c:132-146
#include "kargs.h"

static inline unsigned nblk(long total) { return (unsigned)((total + 255) / 256); }

__global__ __launch_bounds__(256) void maxpool_u8_kernel(const PoolArgs a)
{
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long total = (long)a.B * a.OH * a.OW * a.groups;
    if (idx >= total) return;
    const int g = (int)(idx % a.groups);
    const long p = idx / a.groups;
    const int ox = (int)(p % a.OW);
    const int oy = (int)((p / a.OW) % a.OH);
    const int b = (int)(p / ((long)a.OW * a.OH));
    // `max` starts at uint8 0 (ref :134) == biased 0x80; out-of-image taps are uint8 0 (ref :143)
    uint4 m = make_uint4(0x80808080u, 0x80808080u, 0x80808080u, 0x80808080u);
    for (int n = 0; n < a.size; ++n)
        for (int mm = 0; mm < a.size; ++mm) {
            const int iy = a.offset + oy * a.stride + n, ix = a.offset + ox * a.stride + mm;
            if (iy < 0 || iy >= a.H || ix < 0 || ix >= a.W) continue;
            const long cell = a.in_lead + ((long)b * (a.H + 1) + (iy + 1)) * (a.W + 1) + ix;
            const uint4 v = *reinterpret_cast<const uint4 *>(a.x + cell * a.cs_in + g * 16);
            m.x = max_s8x4(m.x, v.x); m.y = max_s8x4(m.y, v.y); m.z = max_s8x4(m.z, v.z); m.w = max_s8x4(m.w, v.w);
        }
    const long ocell = a.out_lead + ((long)b * (a.OH + 1) + (oy + 1)) * (a.OW + 1) + ox;
    *reinterpret_cast<uint4 *>(a.y + ocell * a.cs_out + g * 16) = m;
}


// y[b, oy, ox, coff + c] = x[b, oy/stride, ox/stride, c]   (stride 1 == route copy)
__global__ __launch_bounds__(256) void copy_cells_kernel(const CopyArgs a)
{
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long total = (long)a.B * a.OH * a.OW * a.groups;
    if (idx >= total) return;
    const int g = (int)(idx % a.groups);
    const long p = idx / a.groups;
    const int ox = (int)(p % a.OW);
    const int oy = (int)((p / a.OW) % a.OH);
    const int b = (int)(p / ((long)a.OW * a.OH));
    const int iy = oy / a.stride, ix = ox / a.stride;
    const long cell = a.in_lead + ((long)b * (a.H + 1) + (iy + 1)) * (a.W + 1) + ix;
    const long ocell = a.out_lead + ((long)b * (a.OH + 1) + (oy + 1)) * (a.OW + 1) + ox;
    *reinterpret_cast<uint4 *>(a.y + ocell * a.cs_out + a.coff + g * 16) =
        *reinterpret_cast<const uint4 *>(a.x + cell * a.cs_in + g * 16);
}


// reference layout [B][C][H][W] uint8 -> PHWC.  cs == 4: plain bytes (c0,c1,c2,0); else biased (^0x80).
__global__ __launch_bounds__(256) void nchw_to_phwc_kernel(const LayoutArgs a)
{
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int cg = (a.C + 3) / 4;
    const long hw = (long)a.H * a.W;
    const long total = (long)a.B * cg * hw;
    if (idx >= total) return;
    const long pix = idx % hw;  // pixel fastest: coalesced plane reads
    const int g = (int)((idx / hw) % cg);
    const int b = (int)(idx / (hw * cg));
    const int y = (int)(pix / a.W), x = (int)(pix % a.W);
    const uint32_t bias = a.cs == 4 ? 0u : 0x80u;
    uint32_t v = 0;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int c = g * 4 + r;
        uint32_t u = (c < a.C) ? (uint32_t)a.nchw[((long)b * a.C + c) * hw + pix] ^ bias : (a.cs == 4 ? 0u : 0x80u);
        v |= u << (8 * r);
    }
    const long cell = a.lead + ((long)b * (a.H + 1) + (y + 1)) * (a.W + 1) + x;
    *reinterpret_cast<uint32_t *>(a.t + cell * a.cs + g * 4) = v;
}

// 3-channel network input, W % 4 == 0: one thread converts 4 consecutive pixels -- three coalesced dword loads (one per
// colour plane) and one 16-byte store of four (c0,c1,c2,0) cells.  4x fewer, 4x wider memory instructions than the
// generic converter (this kernel is on the timed path: the reference hands over [B][3][H][W] uint8).
__global__ __launch_bounds__(256) void nchw3_to_phwc4_kernel(const LayoutArgs a)
{
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int wq = a.W >> 2;
    const long total = (long)a.B * a.H * wq;
    if (idx >= total) return;
    const int xq = (int)(idx % wq);
    const int y = (int)((idx / wq) % a.H);
    const int b = (int)(idx / ((long)wq * a.H));
    const long hw = (long)a.H * a.W;
    const long pix = (long)y * a.W + xq * 4;
    const uint32_t p0 = *reinterpret_cast<const uint32_t *>(a.nchw + ((long)b * 3 + 0) * hw + pix);
    const uint32_t p1 = *reinterpret_cast<const uint32_t *>(a.nchw + ((long)b * 3 + 1) * hw + pix);
    const uint32_t p2 = *reinterpret_cast<const uint32_t *>(a.nchw + ((long)b * 3 + 2) * hw + pix);
    uint32_t o[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
        o[i] = ((p0 >> (8 * i)) & 0xFFu) | (((p1 >> (8 * i)) & 0xFFu) << 8) | (((p2 >> (8 * i)) & 0xFFu) << 16);
    const long cell = a.lead + ((long)b * (a.H + 1) + (y + 1)) * (a.W + 1) + xq * 4;
    uint32_t *dst = reinterpret_cast<uint32_t *>(a.t + cell * 4);
    // cell addresses are only 4-byte aligned in general ((W+1) is odd): four dword stores, still coalesced across lanes
    dst[0] = o[0]; dst[1] = o[1]; dst[2] = o[2]; dst[3] = o[3];
}

__global__ __launch_bounds__(256) void phwc_to_nchw_kernel(const LayoutArgs a)
{
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long hw = (long)a.H * a.W;
    const long total = (long)a.B * a.C * hw;
    if (idx >= total) return;
    const long pix = idx % hw;
    const int c = (int)((idx / hw) % a.C);
    const int b = (int)(idx / (hw * a.C));
    const int y = (int)(pix / a.W), x = (int)(pix % a.W);
    const long cell = a.lead + ((long)b * (a.H + 1) + (y + 1)) * (a.W + 1) + x;
    const uint8_t bias = a.cs == 4 ? 0 : 0x80;
    a.nchw[idx] = a.t[cell * a.cs + c] ^ bias;
}

// fill a cs==4 image tensor with (zp,zp,zp,0) cells
__global__ __launch_bounds__(256) void fill_u32_kernel(uint32_t *p, uint32_t v, long n)
{
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx < n) p[idx] = v;
}

// ref src/yolo_layer.c:132-146 + src/activations.h:39: logistic on x,y and on objectness+classes
__global__ __launch_bounds__(256) void yolo_logistic_kernel(const float *in, float *out, int B, int n, int classes, int hw)
{
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int per = classes + 5;
    const long total = (long)B * n * per * hw;
    if (idx >= total) return;
    const int e = (int)((idx / hw) % per);
    const float v = in[idx];
    out[idx] = yolo_entry_act(v, e);
}


// get_yolo_detections + correct_yolo_boxes (ref: src/yolo_layer.c:83-91, 246-277, 316-345) for a whole batch: one thread per
// (image, cell, anchor); an objectness above thresh claims the next record of its image (atomic counter -- records
// carry their rank cell * n + anchor in the reference's loop, the host sorts by it).  Float / double promotion follows the
// reference's C expressions.  rec = {rank, x, y, w, h, objectness, prob[classes]}.
__global__ __launch_bounds__(256) void yolo_detections_kernel(const float *out, int B, int n, int classes, int h, int w,
                                                              const float *biases, const int *mask, int netw, int neth,
                                                              int imw, int imh, float thresh, int relative, float *recs,
                                                              int max_recs, int *counts)
{
    const int hw = h * w, per = classes + 5, rl = 6 + classes;
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long)B * hw * n) return;
    const int a = (int)(idx % n);
    const int i = (int)((idx / n) % hw);
    const int b = (int)(idx / ((long)n * hw));
    const float *p = out + ((size_t)b * n + a) * per * hw + i;
    const float objectness = p[4 * hw];
    if (objectness <= thresh) return;
    const int slot = atomicAdd(counts + b, 1);
    if (slot >= max_recs) return;
    int new_w, new_h;
    if (((float)netw / imw) < ((float)neth / imh)) { new_w = netw; new_h = (imh * netw) / imw; }
    else { new_h = neth; new_w = (imw * neth) / imh; }
    const int row = i / w, col = i % w;
    float bx = (col + p[0 * hw]) / w;
    float by = (row + p[1 * hw]) / h;
    float bw = (float)(exp((double)p[2 * hw]) * biases[2 * mask[a]] / netw);
    float bh = (float)(exp((double)p[3 * hw]) * biases[2 * mask[a] + 1] / neth);
    bx = (float)((bx - (netw - new_w) / 2. / netw) / ((float)new_w / netw));
    by = (float)((by - (neth - new_h) / 2. / neth) / ((float)new_h / neth));
    bw *= (float)netw / new_w;
    bh *= (float)neth / new_h;
    if (!relative) { bx *= imw; bw *= imw; by *= imh; bh *= imh; }
    float *r = recs + ((size_t)b * max_recs + slot) * rl;
    r[0] = (float)(i * n + a);
    r[1] = bx; r[2] = by; r[3] = bw; r[4] = bh;
    r[5] = objectness;
    for (int j = 0; j < classes; ++j) {
        const float prob = objectness * p[(5 + j) * hw];
        r[6 + j] = (prob > thresh) ? prob : 0.f;
    }
}

int yolo_detections_launch(const float *out, int B, int n, int classes, int h, int w, const float *biases, const int *mask,
                           int netw, int neth, int imw, int imh, float thresh, int relative, float *recs, int max_recs,
                           int *counts, hipStream_t st)
{
    if (hipMemsetAsync(counts, 0, sizeof(int) * (size_t)B, st) != hipSuccess) return MI355_EHIP;
    const long total = (long)B * h * w * n;
    hipLaunchKernelGGL(yolo_detections_kernel, dim3(nblk(total)), dim3(256), 0, st, out, B, n, classes, h, w, biases, mask, netw,
                       neth, imw, imh, thresh, relative, recs, max_recs, counts);
    return hipGetLastError() == hipSuccess ? MI355_OK : MI355_EHIP;
}

int maxpool_launch(const PoolArgs &a, hipStream_t st)
{
    const long total = (long)a.B * a.OH * a.OW * a.groups;
    hipLaunchKernelGGL(maxpool_u8_kernel, dim3(nblk(total)), dim3(256), 0, st, a);
    return hipGetLastError() == hipSuccess ? MI355_OK : MI355_EHIP;
}
int copy_cells_launch(const CopyArgs &a, hipStream_t st)
{
    const long total = (long)a.B * a.OH * a.OW * a.groups;
    hipLaunchKernelGGL(copy_cells_kernel, dim3(nblk(total)), dim3(256), 0, st, a);
    return hipGetLastError() == hipSuccess ? MI355_OK : MI355_EHIP;
}
int nchw_to_phwc_launch(const LayoutArgs &a, hipStream_t st)
{
    if (a.C == 3 && a.cs == 4 && (a.W & 3) == 0 && (((size_t)a.nchw) & 3) == 0) {
        const long total = (long)a.B * a.H * (a.W >> 2);
        hipLaunchKernelGGL(nchw3_to_phwc4_kernel, dim3(nblk(total)), dim3(256), 0, st, a);
        return hipGetLastError() == hipSuccess ? MI355_OK : MI355_EHIP;
    }
    const long total = (long)a.B * ((a.C + 3) / 4) * a.H * a.W;
    hipLaunchKernelGGL(nchw_to_phwc_kernel, dim3(nblk(total)), dim3(256), 0, st, a);
    return hipGetLastError() == hipSuccess ? MI355_OK : MI355_EHIP;
}
int phwc_to_nchw_launch(const LayoutArgs &a, hipStream_t st)
{
    const long total = (long)a.B * a.C * a.H * a.W;
    hipLaunchKernelGGL(phwc_to_nchw_kernel, dim3(nblk(total)), dim3(256), 0, st, a);
    return hipGetLastError() == hipSuccess ? MI355_OK : MI355_EHIP;
}
int fill_u32_launch(uint32_t *p, uint32_t v, long n, hipStream_t st)
{
    hipLaunchKernelGGL(fill_u32_kernel, dim3(nblk(n)), dim3(256), 0, st, p, v, n);
    return hipGetLastError() == hipSuccess ? MI355_OK : MI355_EHIP;
}
int yolo_logistic_launch(const float *in, float *out, int B, int n, int classes, int hw, hipStream_t st)
{
    const long total = (long)B * n * (classes + 5) * hw;
    hipLaunchKernelGGL(yolo_logistic_kernel, dim3(nblk(total)), dim3(256), 0, st, in, out, B, n, classes, hw);
    return hipGetLastError() == hipSuccess ? MI355_OK : MI355_EHIP;
}
