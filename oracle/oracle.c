/*
 * oracle.c -- CPU restatement of the reference's INT8 inference path.  TEST INFRASTRUCTURE ONLY
 * (see oracle.h for the rules and the pinning status).  Build: oracle/Makefile, -O2, strict IEEE
 * (no -ffast-math, -ffp-contract=off) so that every float/double step is the one written here.
 */
#include "oracle.h"
#ifdef _OPENMP
#include <omp.h>
#endif
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ im2col: src/im2col.c:4-13, 26-50 */
static uint8_t im2col_get_pixel(const uint8_t *im, int height, int width, int row, int col, int channel, int pad,
                                uint8_t pad_value)
{
    row -= pad;
    col -= pad;
    if (row < 0 || col < 0 || row >= height || col >= width) return pad_value; /* :10-11 pad = input zero point */
    return im[col + width * (row + height * channel)];
}

void orc_im2col_u8(const uint8_t *im, int channels, int height, int width, int ksize, int stride, int pad,
                   uint8_t *col, uint8_t pad_value)
{
    int height_col = (height + 2 * pad - ksize) / stride + 1; /* :30 */
    int width_col = (width + 2 * pad - ksize) / stride + 1;   /* :31 */
    int channels_col = channels * ksize * ksize;              /* :33 row index = (ci*k + ky)*k + kx */
    for (int c = 0; c < channels_col; ++c) {
        int w_offset = c % ksize;
        int h_offset = (c / ksize) % ksize;
        int c_im = c / ksize / ksize;
        for (int h = 0; h < height_col; ++h) {
            for (int w = 0; w < width_col; ++w) {
                int im_row = h_offset + h * stride;
                int im_col = w_offset + w * stride;
                col[(c * height_col + h) * width_col + w] =
                    im2col_get_pixel(im, height, width, im_row, im_col, c_im, pad, pad_value);
            }
        }
    }
}

/* ------------------------------------------------------------------ GEMM: src/gemm.c:279-299
 * `C[i*ldc+j] += ALPHA*A[i*lda+k]*B[k*ldb+j]` with float ALPHA and int32_t C: each step is
 * C = (int32)((float)C + (ALPHA*(float)A)*(float)B), loops i,k,j.  The product is exact in fp32
 * (<= 65025); the int32->fp32 conversion and the add round to nearest-even once |C| > 2^24. */
void orc_gemm_nn_u8_i32_te(int M, int N, int K, float ALPHA, const uint8_t *A, int lda, const uint8_t *B, int ldb,
                           int BETA, int32_t *C, int ldc)
{
    for (int i = 0; i < M; ++i)
        for (int j = 0; j < N; ++j) C[i * ldc + j] *= BETA; /* :286-290 */
    for (int i = 0; i < M; ++i) {
        for (int k = 0; k < K; ++k) {
            float a = ALPHA * (float)A[i * lda + k];
            for (int j = 0; j < N; ++j) {
                float p = a * (float)B[k * ldb + j];
                float s = (float)C[i * ldc + j] + p;
                C[i * ldc + j] = (int32_t)s; /* fp32 -> int32 truncation (value is already integral) */
            }
        }
    }
}

/* threads the OpenMP-parallel parts (exact-mode accumulators) use; bench.py states it next to a `port` CPU baseline */
int orc_omp_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* ------------------------------------------------------------------ conv accumulators
 * src/convolutional_layer.c:699-723 for one image, groups == 1. */
void orc_conv_acc(const uint8_t *x, int c, int h, int w, const uint8_t *wq, const uint8_t *zp_w, int n, int ksize,
                  int stride, int pad, uint8_t zp_in, int accum_mode, int32_t *acc, int64_t *s1)
{
    int oh = (h + 2 * pad - ksize) / stride + 1;
    int ow = (w + 2 * pad - ksize) / stride + 1;
    int K = ksize * ksize * c; /* :700 */
    int N = oh * ow;           /* :701 */
    const uint8_t *b;
    uint8_t *col = NULL;
    if (ksize == 1) {
        b = x; /* :712-713 (the reference ignores stride/pad for 1x1; so do we) */
    } else {
        col = (uint8_t *)malloc((size_t)K * N); /* :702-705: workspace pre-filled with zp_in, then im2col */
        memset(col, zp_in, (size_t)K * N);
        orc_im2col_u8(x, c, h, w, ksize, stride, pad, col, zp_in); /* :715 */
        b = col;
    }
    if (accum_mode == ORC_ACC_REF_F32) {
        /* :718 / :721 two GEMM passes; the second uses the [n][K] table whose row oc is all zp_w[oc]
         * (src/blas.c:290-300) */
        uint8_t *zpt = (uint8_t *)malloc((size_t)n * K);
        for (int oc = 0; oc < n; ++oc) memset(zpt + (size_t)oc * K, zp_w[oc], (size_t)K);
        orc_gemm_nn_u8_i32_te(n, N, K, 1.0f, wq, K, b, N, 0, acc, N);
        orc_gemm_nn_u8_i32_te(n, N, K, -1.0f, zpt, K, b, N, 1, acc, N);
        free(zpt);
        if (s1) {
            for (int oc = 0; oc < n; ++oc)
                for (int j = 0; j < N; ++j) {
                    int64_t s = 0;
                    for (int k = 0; k < K; ++k) s += (int64_t)wq[(size_t)oc * K + k] * b[(size_t)k * N + j];
                    s1[(size_t)oc * N + j] = s;
                }
        }
    } else {
        /* exact integers: output channels are independent -> OpenMP over oc (test speed only; the bit-faithful
         * ref-f32 branch above stays sequential like the Makefile-default reference) */
#pragma omp parallel
        {
            int64_t *row = (int64_t *)malloc(sizeof(int64_t) * 2 * (size_t)N);
            int64_t *rs1 = row + N;
#pragma omp for schedule(dynamic, 1)
            for (int oc = 0; oc < n; ++oc) {
                memset(row, 0, sizeof(int64_t) * 2 * (size_t)N);
                int zw = zp_w[oc];
                for (int k = 0; k < K; ++k) {
                    int wv = wq[(size_t)oc * K + k];
                    const uint8_t *bk = b + (size_t)k * N;
                    for (int j = 0; j < N; ++j) {
                        row[j] += (int64_t)(wv - zw) * bk[j];
                        rs1[j] += (int64_t)wv * bk[j];
                    }
                }
                for (int j = 0; j < N; ++j) {
                    acc[(size_t)oc * N + j] = (int32_t)row[j];
                    if (s1) s1[(size_t)oc * N + j] = rs1[j];
                }
            }
            free(row);
        }
    }
    free(col);
}

/* ------------------------------------------------------------------ requant epilogue
 * src/convolutional_layer.c:726-751.  Double precision with two truncations:
 *   int64_t t = (acc + biases_int32[oc]) * M_value[oc];       :732
 *   int32_t q = t * M0_right_shift_value[oc];                 :733
 * then the activation switch :734-748 whose result is stored to a uint8_t *before* clamp() :749, i.e. it wraps
 * mod 256 (the reference relies on x86 double->uint8 conversion = truncate to int32, keep the low byte). */
void orc_requant(const int32_t *acc, int n, int spatial, const int32_t *biases_int32, const double *M_value,
                 const double *shift_value, uint8_t zp_act, int activation, int store_mode, uint8_t *out_u8)
{
    for (int i = 0; i < n; ++i) {
        for (int j = 0; j < spatial; ++j) {
            size_t idx = (size_t)i * spatial + j;
            int32_t q = acc[idx];
            int64_t t = (int64_t)((double)(q + biases_int32[i]) * M_value[i]);
            q = (int32_t)((double)t * shift_value[i]);
            int32_t v;
            switch (activation) {
            case ORC_LEAKY: {
                /* :737 the conditional expression has type double on both arms */
                double d = q < 0 ? (round((double)q * 0.1) + (double)zp_act) : (double)(q + (int)zp_act);
                v = (int32_t)d;
                break;
            }
            case ORC_LINEAR:
            case ORC_RELU: /* :740-742: RELU falls through to the LINEAR arm in the default path */
                v = q + (int)zp_act;
                break;
            case ORC_RELU6: /* :743-745 */
                v = q <= 0 ? (int)zp_act : q + (int)zp_act;
                break;
            default: /* :746-747: no store; output keeps its previous content. Not reachable from our cfgs. */
                continue;
            }
            if (store_mode == ORC_STORE_SATURATE) {
                v = v < 0 ? 0 : (v > 255 ? 255 : v); /* src/blas.c:443-452 applied before the store (:594) */
            }
            out_u8[idx] = (uint8_t)v; /* modular */
        }
    }
}

/* ------------------------------------------------------ MKL-path epilogue: src/convolutional_layer.c:572-596
 * PARITY STATUS OF THIS FUNCTION: UNPINNED.  forward_convolutional_layer_quant_inputi_outputi_mkl cannot be built in this
 * image (needs mkl.h / cblas_gemm_s16s16s32; stand-in headers are not allowed), so no output of it exists to check against.
 * What IS pinned: its inputs M0_lut0 / M0_right_shift_lut0 are computed by the default build too (src/blas.c:318-323) and
 * are committed in tests/golden/yolov3_tiny_leaky.json (1717986944, 3 == the float 0.1f).  The restatement follows the
 * source line by line:
 *   :578-579  same two double multiplies / truncations as the default path
 *   :583      LEAKY: q <= 0 ? round(q * 2^-31 * M0_lut0 * 2^-shift_lut0) + zp : q + zp      (evaluated left to right in double)
 *   :586      LINEAR: q + zp      :589  RELU6: q <= 0 ? zp : q + zp      :591  default (RELU included): q unchanged, NO zero point
 *   :594      clamp(q, 0, 255) and only then the uint8 store. */
void orc_requant_mkl(const int32_t *acc, int n, int spatial, const int32_t *biases_int32, const double *M_value,
                     const double *shift_value, uint8_t zp_act, int activation, int32_t M0_lut0, int shift_lut0,
                     uint8_t *out_u8)
{
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < spatial; ++j) {
            size_t idx = (size_t)i * spatial + j;
            int32_t q = acc[idx];
            int64_t t = (int64_t)((double)(q + biases_int32[i]) * M_value[i]);
            q = (int32_t)((double)t * shift_value[i]);
            switch (activation) {
            case ORC_LEAKY: {
                double d = q <= 0 ? (round((double)q * pow(2, -31) * (double)M0_lut0 * pow(2, -shift_lut0)) + (double)zp_act)
                                  : (double)(q + (int)zp_act);
                q = (int32_t)d;
                break;
            }
            case ORC_LINEAR: q = q + (int)zp_act; break;
            case ORC_RELU6: q = q <= 0 ? (int)zp_act : q + (int)zp_act; break;
            default: break;
            }
            out_u8[idx] = (uint8_t)(q < 0 ? 0 : (q > 255 ? 255 : q));
        }
}

/* Exhaustive comparison of the two LEAKY formulas after the clamp, over every requantised value q in [lo, hi]:
 *   default path (:737) + clamp  = what MI355_STORE_SATURATE / ORC_STORE_SATURATE computes
 *   MKL path (:583) + clamp (:594)
 * Returns the number of q for which the stored bytes differ for ANY zero point 0..255 (the byte is
 * clamp(r(q) + zp) with r the rounded negative branch, so the two agree for every zp iff r agrees wherever
 * r + 255 >= 0 and both are < -255 otherwise). */
long orc_mkl_leaky_mismatches(int64_t lo, int64_t hi, int32_t M0_lut0, int shift_lut0)
{
    long bad = 0;
#pragma omp parallel for reduction(+ : bad) schedule(static)
    for (int64_t qq = lo; qq <= hi; ++qq) {
        const int32_t q = (int32_t)qq;
        const double a = q < 0 ? round((double)q * 0.1) : (double)q;                                                  /* default */
        const double b = q <= 0 ? round((double)q * pow(2, -31) * (double)M0_lut0 * pow(2, -shift_lut0)) : (double)q; /* MKL */
        if (a == b) continue;
        if (a < -255.0 && b < -255.0) continue; /* both clamp to 0 for every zero point */
        ++bad;
    }
    return bad;
}

/* -------------------------------------------------------------- quantized residual add ([shortcut] quantized=1)
 * BUILDER-SPECIFIED, parity unpinned: the reference's shortcut is float only (forward_shortcut_layer,
 * src/shortcut_layer.c:62-75: copy_cpu(input) then shortcut_cpu, src/blas.c:490-514: out += add; activation).  The
 * integer form quantises that sum of two differently-scaled uint8 tensors:
 *     K  = round((float)(s_in / s_out) * 2^16), 1 <= K < 2^21       (float division as in src/blas.c:313)
 *     q  = zp_out + ((Ka*(a - zp_a) + Kb*(b - zp_b) + 2^15) >> 16)   (arithmetic shift == floor: round half up)
 *     y  = clamp(q, 0, 255)
 * This function is the normative statement of the op; the HIP kernel (glue.hip: shortcut_u8_kernel) is tested against it. */
int orc_shortcut_multiplier(float s_in, float s_out, int32_t *K)
{
    if (!(s_in > 0.0f) || !(s_out > 0.0f)) return -1;
    const float ratio = s_in / s_out;
    const double k = round((double)ratio * 65536.0);
    if (!(k >= 1.0) || !(k < 2097152.0)) return -1;
    *K = (int32_t)k;
    return 0;
}

void orc_shortcut_u8(const uint8_t *a, const uint8_t *b, long count, int32_t Ka, int32_t Kb, uint8_t zp_a, uint8_t zp_b,
                     uint8_t zp_out, uint8_t *out)
{
    for (long i = 0; i < count; ++i) {
        const int64_t t = (int64_t)Ka * ((int)a[i] - (int)zp_a) + (int64_t)Kb * ((int)b[i] - (int)zp_b) + 32768;
        int64_t q = (int64_t)zp_out + (t >> 16); /* arithmetic shift (gcc: sign-propagating) */
        out[i] = (uint8_t)(q < 0 ? 0 : (q > 255 ? 255 : q));
    }
}

void orc_dequant(const uint8_t *u8, int count, uint8_t zp_act, float s_act, float *out)
{
    /* :757  l.output = (u8 - zp) * s_act : int -> float multiply */
    for (int i = 0; i < count; ++i) out[i] = (float)((int)u8[i] - (int)zp_act) * s_act;
}

/* ------------------------------------------------------------------ maxpool: src/maxpool_layer.c:109-172
 * window offset -pad/2 (:112-113), out dims (w + pad - size)/stride + 1 (:31-32), `max` starts at 0 (:134) and
 * out-of-image taps yield (uint8_t)(-FLT_MAX) which is 0 on x86 (:143) -> OOB taps never win. */
void orc_maxpool_u8(const uint8_t *x, int c, int h, int w, int size, int stride, int pad, uint8_t *out)
{
    int w_offset = -pad / 2, h_offset = -pad / 2;
    int oh = (h + pad - size) / stride + 1;
    int ow = (w + pad - size) / stride + 1;
    for (int k = 0; k < c; ++k)
        for (int i = 0; i < oh; ++i)
            for (int j = 0; j < ow; ++j) {
                uint8_t mx = 0;
                for (int n = 0; n < size; ++n)
                    for (int m = 0; m < size; ++m) {
                        int cur_h = h_offset + i * stride + n;
                        int cur_w = w_offset + j * stride + m;
                        int valid = (cur_h >= 0 && cur_h < h && cur_w >= 0 && cur_w < w);
                        uint8_t val = valid ? x[cur_w + w * (cur_h + h * k)] : 0;
                        if (val > mx) mx = val;
                    }
                out[j + ow * (i + oh * k)] = mx;
            }
}

/* ------------------------------------------------------------------ upsample: src/blas.c:781-803 forward */
void orc_upsample_u8(const uint8_t *x, int c, int h, int w, int stride, uint8_t *out)
{
    for (int k = 0; k < c; ++k)
        for (int j = 0; j < h * stride; ++j)
            for (int i = 0; i < w * stride; ++i)
                out[k * w * h * stride * stride + j * w * stride + i] = x[k * w * h + (j / stride) * w + i / stride];
}

/* ------------------------------------------------------------------ src/blas.c:387-418 */
int orc_quant_multiplier(float real_multiplier, int32_t *M0, int *right_shift)
{
    if (!(real_multiplier > 0.f) || !(real_multiplier < 1.f)) return -1; /* :391-392 asserts */
    int s = 0;
    while (real_multiplier < 0.5f) { /* :398-401 float loop */
        real_multiplier *= 2.0f;
        s++;
    }
    /* :404  round(real_multiplier * (1ll << 31)): float * (float)2^31, then round() in double */
    int64_t q = (int64_t)round((double)(real_multiplier * (float)(1ll << 31)));
    if (q == (1ll << 31)) { /* :410-413 */
        q /= 2;
        s--;
    }
    if (s < 0) return -1;
    *M0 = (int32_t)q;
    *right_shift = s;
    return 0;
}

/* ------------------------------------------------------------------ host prep, src/blas.c:285-334 */
int orc_prep_conv(int n, int c, int ksize, const uint8_t *wq, const uint8_t *zp_w, const float *s_w, float s_in,
                  uint8_t zp_in, float s_act, const float *biases, const float *scales, const float *mean,
                  const float *var, int32_t *biases_int32, double *M_value, double *shift_value, int32_t *M0,
                  int *shift)
{
    int K = c * ksize * ksize; /* :306 note: ignores groups, as upstream */
    for (int ii = 0; ii < n; ++ii) {
        /* :286 -> :594-600 batch_normalize_bias: b - scale*mean/(sqrt(var) + 1e-6f).
         * sqrt() is the double function applied to a float; the sum with the float literal is done in double,
         * the quotient in double, then the subtraction rounds to float on the store. */
        float b = biases[ii];
        if (scales) b = (float)((double)b - (double)(scales[ii] * mean[ii]) / (sqrt((double)var[ii]) + (double).000001f));
        /* :306-311 */
        uint32_t mult_zero_point = (uint32_t)(K * (int)zp_in * (int)zp_w[ii]);
        int32_t wsum = 0;
        for (int jj = 0; jj < K; ++jj) wsum += wq[(size_t)ii * K + jj];
        int32_t weights_sum_int = (int32_t)(mult_zero_point - (uint32_t)(wsum * (int)zp_in));
        /* :313-316 */
        float M = s_in * s_w[ii] / s_act;
        if (orc_quant_multiplier(M, &M0[ii], &shift[ii])) return -1;
        shift_value[ii] = pow(2, -shift[ii]);
        M_value[ii] = pow(2, -31) * M0[ii];
        /* :333  biases_int32 = biases / (s_in * s_w) + weights_sum_int   (float expression -> int32 trunc) */
        float t = b / (s_in * s_w[ii]) + (float)weights_sum_int;
        biases_int32[ii] = (int32_t)t;
    }
    return 0;
}

/* ------------------------------------------------------------------ src/blas.c:108-168 with size_channel == 1 */
int orc_quantize_image(const float *x, int count, uint8_t *out, float *scale, uint8_t *zp)
{
    float min_value = 0.0f, max_value = 0.0f; /* :115-116: the range always includes 0 */
    for (int j = 0; j < count; ++j) {
        max_value = x[j] > max_value ? x[j] : max_value;
        min_value = x[j] < min_value ? x[j] : min_value;
    }
    if (min_value == 0 && max_value == 0) return -1; /* :125-128 assert */
    /* :136  (max - min) / (quant_max_float - quant_min_float).  The reference is built with -Ofast
     * (Makefile:36), under which gcc folds the division by the constant 255.0f into a multiplication by
     * (1.0f/255.0f) (`mulss .LC` in the object code); that is 1 ulp away from the IEEE quotient for some inputs,
     * and the golden vector 'qimg_unit' pins the multiply form. */
    float nudged_scale = (max_value - min_value) * (1.0f / 255.0f);
    if (nudged_scale == 0) return -1;
    const double initial_zero_point = (double)(0.0f - min_value / nudged_scale); /* :138 float expr widened */
    uint8_t nudged_zero_point;
    if (initial_zero_point < 0) nudged_zero_point = 0;
    else if (initial_zero_point > 255) nudged_zero_point = 255;
    else nudged_zero_point = (uint8_t)round(initial_zero_point);
    *scale = nudged_scale;
    *zp = nudged_zero_point;
    for (int k = 0; k < count; ++k) {
        /* :154 float temp = round(x / scale) + zp  (round() in double, sum in double, store to float) */
        float t = (float)(round((double)(x[k] / nudged_scale)) + (double)nudged_zero_point);
        int v = (int)t; /* :158 clamp(int input, ...) takes the float by value conversion */
        out[k] = (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
    }
    return 0;
}

/* ------------------------------------------------------------------ src/image.c:1199-1242 (resize_image) and
 * :812-831 (letterbox_image), planar float images [c][h][w].  Two separable passes, each product rounded to float on its
 * own and the two terms added afterwards (set_pixel then add_pixel); the last column copies the source's last column;
 * the last row takes only its first term -- with whatever (int)(r * h_scale) turns out to be.  The reference's -Ofast
 * build targets baseline x86-64 (no FMA) and these expressions leave nothing to reassociate, so plain IEEE float
 * arithmetic reproduces it (pinned by the 'lbx_*' golden vectors). */
static void orc_resize_image(const float *im, int imw, int imh, int c, int w, int h, float *out)
{
    float *part = malloc(sizeof(float) * (size_t)w * imh * c);
    const float w_scale = (float)(imw - 1) / (w - 1);
    const float h_scale = (float)(imh - 1) / (h - 1);
    for (int k = 0; k < c; ++k)
        for (int r = 0; r < imh; ++r)
            for (int x = 0; x < w; ++x) {
                float val;
                if (x == w - 1 || imw == 1) {
                    val = im[((size_t)k * imh + r) * imw + imw - 1];
                } else {
                    const float sx = x * w_scale;
                    const int ix = (int)sx;
                    const float dx = sx - ix;
                    val = (1 - dx) * im[((size_t)k * imh + r) * imw + ix] + dx * im[((size_t)k * imh + r) * imw + ix + 1];
                }
                part[((size_t)k * imh + r) * w + x] = val;
            }
    for (int k = 0; k < c; ++k)
        for (int r = 0; r < h; ++r) {
            const float sy = r * h_scale;
            const int iy = (int)sy;
            const float dy = sy - iy;
            for (int x = 0; x < w; ++x) {
                float val = (1 - dy) * part[((size_t)k * imh + iy) * w + x];
                if (!(r == h - 1 || imh == 1)) val += dy * part[((size_t)k * imh + iy + 1) * w + x];
                out[((size_t)k * h + r) * w + x] = val;
            }
        }
    free(part);
}

int orc_letterbox_image(const float *im, int imw, int imh, int c, int w, int h, float *out)
{
    int new_w, new_h;
    if (((float)w / imw) < ((float)h / imh)) { new_w = w; new_h = (imh * w) / imw; }
    else { new_h = h; new_w = (imw * h) / imh; }
    if (new_w < 2 || new_h < 2 || imw < 1 || imh < 1) return -1; /* the reference divides by (w - 1), (h - 1) */
    float *res = malloc(sizeof(float) * (size_t)new_w * new_h * c);
    orc_resize_image(im, imw, imh, c, new_w, new_h, res);
    for (size_t i = 0; i < (size_t)w * h * c; ++i) out[i] = .5f; /* fill_image(boxed, .5) */
    const int dx = (w - new_w) / 2, dy = (h - new_h) / 2;
    for (int k = 0; k < c; ++k)
        for (int y = 0; y < new_h; ++y)
            for (int x = 0; x < new_w; ++x)
                out[((size_t)k * h + dy + y) * w + dx + x] = res[((size_t)k * new_h + y) * new_w + x];
    free(res);
    return 0;
}

/* ------------------------------------------------------------------ src/yolo_layer.c:132-146 */
static float logistic(float x) { return (float)(1. / (1. + exp(-(double)x))); } /* src/activations.h:39 */
void orc_yolo_forward(const float *in, int n, int classes, int h, int w, float *out)
{
    int hw = h * w, per = classes + 5;
    memcpy(out, in, sizeof(float) * (size_t)n * per * hw);
    for (int a = 0; a < n; ++a) {
        float *o = out + (size_t)a * per * hw;
        for (int i = 0; i < 2 * hw; ++i) o[i] = logistic(o[i]);                       /* x, y */
        for (int i = 4 * hw; i < per * hw; ++i) o[i] = logistic(o[i]);                /* obj + classes */
    }
}

/* ------------------------------------------------------------------ src/yolo_layer.c:83-91, 246-277, 316-345
 * get_yolo_detections of one image: for every cell (row-major) and every anchor of the layer's mask, in the reference's
 * loop order, an objectness above `thresh` yields a record
 *     rec[0] = cell * n + anchor (the record's rank in the reference's loop, as a float)
 *     rec[1..4] = box x, y, w, h after correct_yolo_boxes (letterbox of an imw x imh image into netw x neth)
 *     rec[5] = objectness, rec[6 + j] = objectness * class_j if that product exceeds thresh, else 0.
 * `out` is the yolo layer's l.output for the image: [n][classes + 5][h * w].  Returns the number of detections; at most
 * `max_recs` are written.  Float / double promotion follows the reference's C expressions term by term. */
int orc_yolo_detections(const float *out, int n, int classes, int h, int w, const float *biases, const int *mask, int netw,
                        int neth, int imw, int imh, float thresh, int relative, float *recs, int max_recs)
{
    const int hw = h * w, per = classes + 5, rl = 6 + classes;
    int new_w, new_h; /* correct_yolo_boxes :250-257 */
    if (((float)netw / imw) < ((float)neth / imh)) { new_w = netw; new_h = (imh * netw) / imw; }
    else { new_h = neth; new_w = (imw * neth) / imh; }
    int count = 0;
    for (int i = 0; i < hw; ++i) {
        const int row = i / w, col = i % w;
        for (int a = 0; a < n; ++a) {
            const float *p = out + (size_t)a * per * hw + i; /* entry e of this (anchor, cell): p[e * hw] (:125-130) */
            const float objectness = p[4 * hw];
            if (objectness <= thresh) continue; /* :326 */
            if (count < max_recs) {
                float *r = recs + (size_t)count * rl;
                float bx = (col + p[0 * hw]) / w;  /* get_yolo_box :86-89 */
                float by = (row + p[1 * hw]) / h;
                float bw = exp(p[2 * hw]) * biases[2 * mask[a]] / netw;
                float bh = exp(p[3 * hw]) * biases[2 * mask[a] + 1] / neth;
                bx = (bx - (netw - new_w) / 2. / netw) / ((float)new_w / netw); /* :260-263 */
                by = (by - (neth - new_h) / 2. / neth) / ((float)new_h / neth);
                bw *= (float)netw / new_w;
                bh *= (float)neth / new_h;
                if (!relative) { bx *= imw; bw *= imw; by *= imh; bh *= imh; } /* :264-269 */
                r[0] = (float)(i * n + a);
                r[1] = bx; r[2] = by; r[3] = bw; r[4] = bh;
                r[5] = objectness;
                for (int j = 0; j < classes; ++j) { /* :333-337 */
                    const float prob = objectness * p[(5 + j) * hw];
                    r[6 + j] = (prob > thresh) ? prob : 0;
                }
            }
            ++count;
        }
    }
    return count;
}
