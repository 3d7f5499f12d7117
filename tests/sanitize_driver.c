/* TEST INFRASTRUCTURE: drives every entry point of oracle/oracle.c under -fsanitize=address,undefined
 * (tests/test_host_cpu.py::test_oracle_restatement_is_clean_under_asan_ubsan).  Data is chosen so that the paths the
 * REFERENCE leaves to undefined behaviour (out-of-range requantised values stored to uint8_t, out-of-image maxpool taps)
 * are exercised in the restatement, which must handle them with defined arithmetic. */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "oracle.h"

static uint64_t rs = 88172645463325252ULL;
static uint32_t rnd(void) { rs ^= rs << 13; rs ^= rs >> 7; rs ^= rs << 17; return (uint32_t)(rs >> 32); }

int main(void)
{
    enum { C = 16, H = 9, W = 11, N = 24, K = C * 9 };
    uint8_t *x = malloc(C * H * W), *wq = malloc(N * K), zp_w[N];
    for (int i = 0; i < C * H * W; ++i) x[i] = (uint8_t)rnd();
    for (int i = 0; i < N * K; ++i) wq[i] = (uint8_t)rnd();
    for (int i = 0; i < N; ++i) zp_w[i] = (uint8_t)(rnd() % 256);
    int32_t *acc = malloc(sizeof(int32_t) * N * H * W), *acc2 = malloc(sizeof(int32_t) * N * H * W);
    int64_t *s1 = malloc(sizeof(int64_t) * N * H * W);
    for (int stride = 1; stride <= 2; ++stride) {
        orc_conv_acc(x, C, H, W, wq, zp_w, N, 3, stride, 1, 200, ORC_ACC_EXACT, acc, s1);
        orc_conv_acc(x, C, H, W, wq, zp_w, N, 3, stride, 1, 200, ORC_ACC_REF_F32, acc2, NULL);
    }
    orc_conv_acc(x, C, H, W, wq, zp_w, N, 1, 1, 0, 7, ORC_ACC_EXACT, acc2, NULL);
    orc_conv_acc(x, C, H, W, wq, zp_w, N, 3, 1, 1, 200, ORC_ACC_EXACT, acc, s1);
    /* multipliers near 1 and extreme accumulators: requantised values far outside 0..255 and near INT32 limits */
    int32_t bias[N]; double mv[N], sv[N];
    for (int i = 0; i < N; ++i) { bias[i] = (int32_t)(rnd() % 2000000) - 1000000; mv[i] = 0.5 + (rnd() % 1000) / 2001.0; sv[i] = ldexp(1.0, -(int)(rnd() % 4)); }
    acc[0] = INT32_MAX - 1000001; acc[1] = INT32_MIN + 1000001; acc[2] = 0; acc[3] = -1;
    uint8_t *u8 = malloc(N * H * W), *u8b = malloc(N * H * W);
    const int acts[4] = {ORC_LEAKY, ORC_LINEAR, ORC_RELU6, ORC_RELU};
    for (int a = 0; a < 4; ++a)
        for (int st = 0; st < 2; ++st) orc_requant(acc, N, H * W, bias, mv, sv, 23, acts[a], st, u8);
    for (int a = 0; a < 4; ++a) orc_requant_mkl(acc, N, H * W, bias, mv, sv, 23, acts[a], 1717986944, 3, u8b);
    if (orc_mkl_leaky_mismatches(-100000, 100000, 1717986944, 3) != 0) { printf("mkl mismatch\n"); return 1; }
    float *f = malloc(sizeof(float) * N * H * W);
    orc_dequant(u8, N * H * W, 23, 0.05f, f);
    uint8_t *p = malloc(C * H * W * 4);
    orc_maxpool_u8(x, C, H, W, 2, 2, 1, p);
    orc_maxpool_u8(x, C, H, W, 2, 1, 1, p);
    orc_maxpool_u8(x, C, H, W, 3, 2, 2, p);
    orc_upsample_u8(x, C, H, W, 2, p);
    uint8_t *col = malloc(K * H * W);
    orc_im2col_u8(x, C, H, W, 3, 1, 1, col, 23);
    int32_t Ka = 0, Kb = 0;
    if (orc_shortcut_multiplier(0.02588f, 0.0372f, &Ka) || orc_shortcut_multiplier(0.0235f, 0.0372f, &Kb)) return 2;
    orc_shortcut_u8(x, x + 7, C * H * W - 7, Ka, Kb, 23, 0, 40, p);
    orc_shortcut_u8(x, x + 7, C * H * W - 7, (1 << 21) - 1, (1 << 21) - 1, 0, 0, 255, p);
    orc_shortcut_u8(x, x + 7, C * H * W - 7, (1 << 21) - 1, (1 << 21) - 1, 255, 255, 0, p);
    int32_t m0; int sh;
    const float ms[] = {0.5f, 0.25f, 0.99999994f, 1e-7f, 0.75f, 3.1e-5f};
    for (unsigned i = 0; i < sizeof(ms) / sizeof(ms[0]); ++i) if (orc_quant_multiplier(ms[i], &m0, &sh)) return 3;
    if (!orc_quant_multiplier(1.0f, &m0, &sh) || !orc_quant_multiplier(0.0f, &m0, &sh)) return 4;
    float sw[N], bf[N], sc[N], mean[N], var[N]; int32_t b32[N], M0[N]; int shv[N];
    for (int i = 0; i < N; ++i) { sw[i] = 0.003f + (rnd() % 100) * 1e-5f; bf[i] = (rnd() % 200) / 1000.f - .1f; sc[i] = 1.f; mean[i] = .01f; var[i] = 1.f; }
    if (orc_prep_conv(N, C, 3, wq, zp_w, sw, 0.0259f, 23, 0.0259f, bf, sc, mean, var, b32, mv, sv, M0, shv)) return 5;
    if (orc_prep_conv(N, C, 3, wq, zp_w, sw, 0.0259f, 23, 0.0259f, bf, NULL, NULL, NULL, b32, mv, sv, M0, shv)) return 5;
    float img[3 * 21 * 17], qs; uint8_t qz, q8[3 * 21 * 17];
    for (int i = 0; i < 3 * 21 * 17; ++i) img[i] = (rnd() % 2001) / 1000.f - 0.7f;
    if (orc_quantize_image(img, 3 * 21 * 17, q8, &qs, &qz)) return 6;
    float lb[3 * 32 * 32];
    if (orc_letterbox_image(img, 17, 21, 3, 32, 32, lb)) return 7;
    if (orc_letterbox_image(img, 21, 17, 3, 32, 24, lb)) return 7;
    enum { YN = 3, YC = 5, YH = 4, YW = 4 };
    float yin[YN * (YC + 5) * YH * YW], yout[YN * (YC + 5) * YH * YW], recs[YN * YH * YW * (6 + YC)];
    for (int i = 0; i < YN * (YC + 5) * YH * YW; ++i) yin[i] = (rnd() % 8001) / 1000.f - 4.f;
    orc_yolo_forward(yin, YN, YC, YH, YW, yout);
    const float anchors[12] = {10, 14, 23, 27, 37, 58, 81, 82, 135, 169, 344, 319};
    const int mask[3] = {3, 4, 5};
    int cnt = orc_yolo_detections(yout, YN, YC, YH, YW, anchors, mask, 64, 64, 100, 80, 0.3f, 1, recs, YN * YH * YW);
    cnt += orc_yolo_detections(yout, YN, YC, YH, YW, anchors, mask, 64, 64, 80, 100, 0.3f, 0, recs, 2);
    uint8_t A[6 * 40], B[40 * 5]; int32_t Cm[6 * 5] = {0};
    for (int i = 0; i < 6 * 40; ++i) A[i] = (uint8_t)rnd();
    for (int i = 0; i < 40 * 5; ++i) B[i] = (uint8_t)rnd();
    orc_gemm_nn_u8_i32_te(6, 5, 40, 1.0f, A, 40, B, 5, 0, Cm, 5);
    orc_gemm_nn_u8_i32_te(6, 5, 40, -1.0f, A, 40, B, 5, 1, Cm, 5);
    free(x); free(wq); free(acc); free(acc2); free(s1); free(u8); free(u8b); free(f); free(p); free(col);
    printf("sanitize_driver: OK (%d detections)\n", cnt);
    return 0;
}
