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


// route inputs whose channel count is not a multiple of 16 (a concat offset then falls inside a 16-byte group): one thread per
// (cell, byte).  Rare (the reference's nets concatenate 128- / 256-channel maps); correctness path, not a tuned one.
__global__ __launch_bounds__(256) void copy_cell_bytes_kernel(const CopyArgs a, int nbytes)
{
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long total = (long)a.B * a.OH * a.OW * nbytes;
    if (idx >= total) return;
    const int c = (int)(idx % nbytes);
    const long p = idx / nbytes;
    const int ox = (int)(p % a.OW);
    const int oy = (int)((p / a.OW) % a.OH);
    const int b = (int)(p / ((long)a.OW * a.OH));
    const long cell = a.in_lead + ((long)b * (a.H + 1) + (oy + 1)) * (a.W + 1) + ox;
    const long ocell = a.out_lead + ((long)b * (a.OH + 1) + (oy + 1)) * (a.OW + 1) + ox;
    a.y[ocell * a.cs_out + a.coff + c] = a.x[cell * a.cs_in + c];
}

int copy_cell_bytes_launch(const CopyArgs &a, int nbytes, hipStream_t st)
{
    const long total = (long)a.B * a.OH * a.OW * nbytes;
    hipLaunchKernelGGL(copy_cell_bytes_kernel, dim3(nblk(total)), dim3(256), 0, st, a, nbytes);
    return hipGetLastError() == hipSuccess ? MI355_OK : MI355_EHIP;
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

// quant_stop tail of the glue layers (ref: src/maxpool_layer.c:163-171, src/upsample_layer.c:104-112, src/route_layer.c:121-129):
// out[b][c0_out + c][pix] = (int)(u8[b][pix][c0 + c] - zp) * scale for c < nc, reference layout [B][C_out][H*W] float.
__global__ __launch_bounds__(256) void dequant_cells_kernel(const DequantArgs a)
{
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long hw = (long)a.H * a.W;
    const long total = (long)a.B * a.nc * hw;
    if (idx >= total) return;
    const long pix = idx % hw;
    const int c = (int)((idx / hw) % a.nc);
    const int b = (int)(idx / (hw * a.nc));
    const int y = (int)(pix / a.W), x = (int)(pix % a.W);
    const long cell = a.lead + ((long)b * (a.H + 1) + (y + 1)) * (a.W + 1) + x;
    const int u = a.t[cell * a.cs + a.c0 + c] ^ 0x80;
    a.out[((long)b * a.out_C + a.out_c0 + c) * hw + pix] = (float)(u - a.zp) * a.scale;
}

// Quantized residual add ([shortcut] quantized=1; builder-specified, DESIGN.md section 7 -- the reference's shortcut is
// float only, src/shortcut_layer.c:62-75): q = zo + ((Ka*(a - za) + Kb*(b - zb) + 2^15) >> 16), saturated to 0..255.
// k0 = 2^15 + (zo << 16) - Ka*za - Kb*zb is folded on the host.  One thread per (cell, 16-channel group).
__global__ __launch_bounds__(256) void shortcut_u8_kernel(const ShortcutArgs s)
{
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long total = (long)s.B * s.H * s.W * s.groups;
    if (idx >= total) return;
    const int g = (int)(idx % s.groups);
    const long p = idx / s.groups;
    const int x = (int)(p % s.W);
    const int y = (int)((p / s.W) % s.H);
    const int b = (int)(p / ((long)s.W * s.H));
    const long cell = ((long)b * (s.H + 1) + (y + 1)) * (s.W + 1) + x;
    const uint4 va = *reinterpret_cast<const uint4 *>(s.a + (cell + s.a_lead) * s.a_cs + g * 16);
    const uint4 vb = *reinterpret_cast<const uint4 *>(s.b + (cell + s.b_lead) * s.b_cs + g * 16);
    uint32_t o[4];
    o[0] = shortcut4_biased(va.x, vb.x, s.ka, s.kb, s.k0);
    o[1] = shortcut4_biased(va.y, vb.y, s.ka, s.kb, s.k0);
    o[2] = shortcut4_biased(va.z, vb.z, s.ka, s.kb, s.k0);
    o[3] = shortcut4_biased(va.w, vb.w, s.ka, s.kb, s.k0);
    *reinterpret_cast<uint4 *>(s.y + (cell + s.y_lead) * s.y_cs + g * 16) = make_uint4(o[0], o[1], o[2], o[3]);
}

int dequant_cells_launch(const DequantArgs &a, hipStream_t st)
{
    const long total = (long)a.B * a.nc * a.H * a.W;
    hipLaunchKernelGGL(dequant_cells_kernel, dim3(nblk(total)), dim3(256), 0, st, a);
    return hipGetLastError() == hipSuccess ? MI355_OK : MI355_EHIP;
}
int shortcut_launch(const ShortcutArgs &a, hipStream_t st)
{
    const long total = (long)a.B * a.H * a.W * a.groups;
    hipLaunchKernelGGL(shortcut_u8_kernel, dim3(nblk(total)), dim3(256), 0, st, a);
    return hipGetLastError() == hipSuccess ? MI355_OK : MI355_EHIP;
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
// ---------------------------------------------------------------------------------------------------------------
// Layer-0 input quantiser on the device (ref: quant_weights_with_min_max_channel with one channel, src/blas.c:108-168,
// called on the float image at src/blas.c:279).  Pass 1: min / max of the image against the reference's 0.0f seeds;
// the host turns them into (scale, zero point) with the reference's own expressions; pass 2: the per-element
// round(x / scale) + zp, clamped, with the reference's float / double evaluation order.
// mm[0] holds max(x, +0) as float bits (monotone as signed int for x >= 0), mm[1] the bits of min(x, -0) (monotone as
// unsigned for x <= 0): both seeds are the reference's 0.0f, NaNs compare false and are skipped exactly as there.
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void image_minmax_kernel(const float *x, long count, uint32_t *mm)
{
    float mx = 0.0f, mn = 0.0f;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (long)gridDim.x * blockDim.x) {
        const float v = x[i];
        mx = v > mx ? v : mx;
        mn = v < mn ? v : mn;
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        const float omx = __shfl_xor(mx, m), omn = __shfl_xor(mn, m);
        mx = omx > mx ? omx : mx;
        mn = omn < mn ? omn : mn;
    }
    if ((threadIdx.x & 63) == 0) {
        if (mx > 0.0f) atomicMax(reinterpret_cast<int *>(mm), __float_as_int(mx));
        if (mn < 0.0f) atomicMax(mm + 1, (uint32_t)__float_as_int(mn));
    }
}

__global__ __launch_bounds__(256) void image_quantize_kernel(const float *x, long count, float scale, int zp, uint8_t *out)
{
    // four elements per thread: one 16-byte load, one 4-byte store
    const long i4 = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i4 >= count) return;
    float v[4];
    if (i4 + 3 < count && (reinterpret_cast<size_t>(x) & 15) == 0) {
        const float4 t = *reinterpret_cast<const float4 *>(x + i4);
        v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
    } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = i4 + k < count ? x[i4 + k] : 0.0f;
    }
    uint32_t packed = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float t = (float)(round((double)(v[k] / scale)) + (double)zp);  // ref :160-165
        const int q = (int)t;
        packed |= (uint32_t)(q < 0 ? 0 : (q > 255 ? 255 : q)) << (8 * k);
    }
    if (i4 + 3 < count && (reinterpret_cast<size_t>(out) & 3) == 0) {
        *reinterpret_cast<uint32_t *>(out + i4) = packed;
    } else {
        for (int k = 0; k < 4 && i4 + k < count; ++k) out[i4 + k] = (uint8_t)(packed >> (8 * k));
    }
}

// letterbox_image (ref: src/image.c:812-831) with its separable bilinear resize_image (:1199-1242) in one pass: an output
// pixel inside the embedded rectangle interpolates its two source rows horizontally (the reference's intermediate
// `part` image, same expressions, each product and sum rounded to float on its own: built with -ffp-contract=off) and
// combines them vertically; outside it is the 0.5 fill.  Last column = the source's last column, last row = first term
// only, as in the reference.
__global__ __launch_bounds__(256) void letterbox_kernel(const float *im, int imw, int imh, int c, float *out, int w, int h,
                                                        int new_w, int new_h, int ox, int oy, float w_scale, float h_scale)
{
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long)c * h * w) return;
    const int x = (int)(idx % w), y = (int)((idx / w) % h), k = (int)(idx / ((long)w * h));
    const int xx = x - ox, r = y - oy;
    if (xx < 0 || xx >= new_w || r < 0 || r >= new_h) {
        out[idx] = .5f;
        return;
    }
    const float *plane = im + (size_t)k * imh * imw;
    auto hval = [&](int row) {
        const float *p = plane + (size_t)row * imw;
        if (xx == new_w - 1 || imw == 1) return p[imw - 1];
        const float sx = xx * w_scale;
        const int ix = (int)sx;
        const float dx = sx - ix;
        return (1 - dx) * p[ix] + dx * p[ix + 1];
    };
    const float sy = r * h_scale;
    const int iy = (int)sy;
    const float dy = sy - iy;
    float val = (1 - dy) * hval(iy);
    if (!(r == new_h - 1 || imh == 1)) val += dy * hval(iy + 1);
    out[idx] = val;
}

int letterbox_launch(const float *im, int imw, int imh, int c, float *out, int w, int h, hipStream_t st)
{
    int new_w, new_h;
    if (((float)w / imw) < ((float)h / imh)) { new_w = w; new_h = (imh * w) / imw; }
    else { new_h = h; new_w = (imw * h) / imh; }
    if (new_w < 2 || new_h < 2) return MI355_EINVAL;  // the reference divides by (w - 1), (h - 1)
    const float w_scale = (float)(imw - 1) / (new_w - 1), h_scale = (float)(imh - 1) / (new_h - 1);
    // (int)(r * h_scale) must stay a row of the source image (the reference asserts it)
    if ((int)((new_h - 1) * h_scale) > imh - 1 || (new_h > 1 && (int)((new_h - 2) * h_scale) + 1 > imh - 1) ||
        (new_w > 1 && (int)((new_w - 2) * w_scale) + 1 > imw - 1))
        return MI355_EINVAL;
    const long total = (long)c * h * w;
    hipLaunchKernelGGL(letterbox_kernel, dim3(nblk(total)), dim3(256), 0, st, im, imw, imh, c, out, w, h, new_w, new_h,
                       (w - new_w) / 2, (h - new_h) / 2, w_scale, h_scale);
    return hipGetLastError() == hipSuccess ? MI355_OK : MI355_EHIP;
}

int image_minmax_launch(const float *x, long count, uint32_t *mm, hipStream_t st)
{
    if (hipMemsetAsync(mm, 0, 4, st) != hipSuccess) return MI355_EHIP;                 // +0.0f
    if (fill_u32_launch(mm + 1, 0x80000000u, 1, st) != MI355_OK) return MI355_EHIP;   // -0.0f
    const long want = (count + 255) / 256;
    const int grid = (int)(want < 2048 ? (want > 0 ? want : 1) : 2048);
    hipLaunchKernelGGL(image_minmax_kernel, dim3(grid), dim3(256), 0, st, x, count, mm);
    return hipGetLastError() == hipSuccess ? MI355_OK : MI355_EHIP;
}

int image_quantize_launch(const float *x, long count, float scale, int zp, uint8_t *out, hipStream_t st)
{
    hipLaunchKernelGGL(image_quantize_kernel, dim3(nblk((count + 3) / 4)), dim3(256), 0, st, x, count, scale, zp, out);
    return hipGetLastError() == hipSuccess ? MI355_OK : MI355_EHIP;
}

// Order-independent 64-bit checksum of a device buffer (dwords; sum of value * (odd multiplier of the index)): two runs of the
// same kernels on the same input must give the same number -- the determinism self-check of the host (network_selfcheck).
__global__ __launch_bounds__(256) void checksum_u32_kernel(const uint32_t *p, long n, unsigned long long *out)
{
    unsigned long long s = 0;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256)
        s += (unsigned long long)p[i] * (unsigned long long)(2 * i + 1);
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) s += __shfl_xor(s, m);
    if ((threadIdx.x & 63) == 0) atomicAdd(out, s);
}

int checksum_u32_launch(const uint32_t *p, long n, unsigned long long *out, hipStream_t st)
{
    const int blocks = (int)((n + 255) / 256 < 1024 ? (n + 255) / 256 : 1024);
    hipLaunchKernelGGL(checksum_u32_kernel, dim3(blocks > 0 ? blocks : 1), dim3(256), 0, st, p, n, out);
    return hipGetLastError() == hipSuccess ? MI355_OK : MI355_EHIP;
}

int yolo_logistic_launch(const float *in, float *out, int B, int n, int classes, int hw, hipStream_t st)
{
    const long total = (long)B * n * (classes + 5) * hw;
    hipLaunchKernelGGL(yolo_logistic_kernel, dim3(nblk(total)), dim3(256), 0, st, in, out, B, n, classes, hw);
    return hipGetLastError() == hipSuccess ? MI355_OK : MI355_EHIP;
}
