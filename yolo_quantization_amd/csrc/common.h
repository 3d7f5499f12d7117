// common.h -- device-side helpers shared by the gfx950 kernels (CDNA4 only; no CUDA compatibility layer).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/mi355_yolo_int8.h"

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

// ---------------------------------------------------------------------------------------------------------
// Packed conv blob (position independent; see mi355_conv_pack in shim.hip).
// Weights are stored signed-biased (w_u8 - 128) in "unit" order: a unit is 16 consecutive input channels of one
// tap; the K axis is  for chunk (cb bytes of channels)  for tap (ky,kx)  for 16-channel block.  One MFMA K-step
// (64 bytes) = 4 units.  Rows are grouped by 16 output channels: [mpad/16][ksteps][16 rows][64 bytes].
// ---------------------------------------------------------------------------------------------------------
#define MI355_BLOB_MAGIC 0x4D493335u /* "MI35" */
struct ConvBlobHeader {
    uint32_t magic;
    int32_t n, c, ksize;
    int32_t mpad;       // n rounded up to 16
    int32_t cb;         // channel-chunk bytes per cell: 64, 32 or 16 (4 for the 3-channel first layer)
    int32_t nchunks;    // c / cb
    int32_t upc;        // valid units per chunk = ksize*ksize*(cb/16)
    int32_t spc;        // K-steps per chunk = ceil(upc/4)
    int32_t ksteps;     // nchunks*spc
    int32_t ktrue;      // c*ksize*ksize
    int32_t first;      // 1: first-layer (c==3) packing: wp = [n][9] dwords (ky,kx)(c0,c1,c2,0) plain uint8
    uint64_t off_wp, off_cw, off_dzp, off_bias, off_mval, off_sval;  // byte offsets from blob start
    uint64_t total;
    uint64_t off_shift;  // int32[mpad]: right shift s when shift_value == 2^-s exactly for every channel (pow2 == 1)
    int32_t pow2;        // 1: every shift_value is an exact power of two 2^-s with 0 <= s <= 31 (the reference's case)
    int32_t pad_;
    uint64_t off_mprime; // f64[mpad]: M_value * shift_value (exact product when pow2)
    uint64_t off_cwb;    // int32[mpad]: cw + biases_int32 (the two per-channel additive constants folded)
};

// ---------------------------------------------------------------------------------------------------------
// Requantise epilogue: ref src/convolutional_layer.c:726-751.  FP64 with two truncations, activation, zero point,
// uint8 store that wraps (default path) or saturates (MKL path).  `acc` is the true pre-requant accumulator.
// Compile with -ffp-contract=off: every double op below must stay a separate IEEE operation.
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t requant_u8(int32_t acc, int32_t bias, double M, double S, int zp_act, int act,
                                               int store_mode)
{
    // :732  int64_t temp = (acc + bias) * M_value.  mi355_conv_pack guarantees 0 < M_value < 1 (the reference asserts
    // 0 < M < 1, src/blas.c:391-392), so |product| < 2^31 and the truncation to int64 equals the single-instruction
    // truncation to int32 (v_cvt_i32_f64) -- bit-identical, ~4x fewer instructions than a generic f64->i64.
    const int32_t t = (int32_t)((double)(acc + bias) * M);
    const int32_t q = (int32_t)((double)t * S);  // :733  q = temp * 2^-shift  (0 < S <= 1)
    int32_t v;
    if (act == MI355_ACT_LEAKY) {
        // :737  q < 0 ? round(q*0.1) + zp : q + zp.  round((double)q * 0.1) == -((|q| + 5) / 10) for every negative
        // int32 q (exhaustively verified, tests/test_host_cpu.py + DESIGN.md), so no FP64 here.
        const uint32_t uq = 0u - (uint32_t)q;
        v = q < 0 ? zp_act - (int32_t)((uq + 5u) / 10u) : q + zp_act;
    } else if (act == MI355_ACT_RELU6) {
        v = q <= 0 ? zp_act : q + zp_act;  // :744
    } else {
        v = q + zp_act;  // :740-742 LINEAR, RELU
    }
    if (store_mode == MI355_STORE_SATURATE) v = v < 0 ? 0 : (v > 255 ? 255 : v);
    return (uint32_t)v & 0xFFu;  // uint8_t store: modular
}

// Fast form (valid when every shift_value is an exact power of two 2^-s, which is how the reference builds it:
// M0_right_shift_value = pow(2, -shift), src/blas.c:315).  With d = fl((acc+bias) * M_value):
//     q = trunc(trunc(d) * 2^-s) = trunc(d * 2^-s)          (nested truncation toward zero by an integer divisor)
//       = trunc(fl((acc+bias) * (M_value * 2^-s)))          (scaling by a power of two commutes with rounding)
// so ONE double multiply by the pre-folded Mp = M_value * shift_value and ONE truncating convert give the
// reference's q bit for bit.  `accb` already contains acc + cw + dz*sx + biases_int32.  Activation / store mode
// are compile-time constants.  LEAKY: round((double)q * 0.1) == -((|q| + 5) / 10) for every negative int32 q
// (exhaustively verified); the division uses a 24-bit multiply when |q| + 5 < 2^16 and v_mul_hi otherwise.
// (An FP32 estimate of q with an FP64 fallback for lanes near an integer boundary was tried and measured slower than
// this single FP64 multiply: the epilogue is bound by instruction count, not by the FP64 rate -- profiles/r01 notes.)
__device__ __forceinline__ int32_t requant_q_exact(int32_t accb, double Mp) { return (int32_t)((double)accb * Mp); }
// Stage 2: activation, zero point, store mode (compile-time), uint8 wrap.
template <int ACT, bool SAT>
__device__ __forceinline__ uint32_t requant_finish(int32_t q, int zp_act)
{
    int32_t v;
    if (ACT == MI355_ACT_LEAKY) {
        // round((double)q * 0.1) == -((|q| + 5) / 10) for every negative int32 q (exhaustively verified); the division is
        // a 24-bit multiply when |q| + 5 < 2^16 and v_mul_hi otherwise
        const uint32_t x = (0u - (uint32_t)q) + 5u;
        const uint32_t d10 = x < 65536u ? (__umul24(x, 0xCCCDu) >> 19) : x / 10u;
        v = q < 0 ? zp_act - (int32_t)d10 : q + zp_act;
    } else if (ACT == MI355_ACT_RELU6) {
        v = q <= 0 ? zp_act : q + zp_act;
    } else {
        v = q + zp_act;
    }
    if (SAT) v = v < 0 ? 0 : (v > 255 ? 255 : v);
    return (uint32_t)v & 0xFFu;
}

// cell index of pixel n (n enumerates b,y,x) in a PHWC tensor
__device__ __forceinline__ int cell_of_pixel(int n, int H, int W, int lead)
{
    int hw = H * W;
    int b = n / hw;
    int r = n - b * hw;
    int y = r / W;
    int x = r - y * W;
    return lead + (b * (H + 1) + (y + 1)) * (W + 1) + x;
}
