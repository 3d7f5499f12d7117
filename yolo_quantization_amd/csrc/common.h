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
    uint64_t off_ws;     // conv_small.hip shapes only (3x3, c 16|32 with n 32|64, c 64 with n 64..128), else 0: A fragments of
                         // every K-step in MFMA lane order, [n/32][k-step][64 lanes][16 B] (c 16: 5 steps of two taps;
                         // c 32: 9 taps; c 64: 18 steps, two per tap)
    uint64_t off_ept;    // EptHeader + EptEntry[mpad] (+ the first layer's 4 KiB LEAKY byte table): the pooled kernels' per-channel epilogue
                         // constants for ONE (activation, zero point), written by mi355_conv_pack_epilogue; key 0 = not prepared
};

// ---------------------------------------------------------------------------------------------------------
// Epilogue table (round 5).  The conv + maxpool kernels (conv_first_mfma_pool, conv_small_pool, conv_mid_pool) requantise a window's
// MAXIMUM when no byte of the window can wrap; the per-channel constants of that test and of the integer requantisation -- the wrap-safe
// accumulator range (small_safe_range, biased_safe_range), M0 and shift (intrq_make) -- depend on the layer's multipliers, activation and
// zero point only.  Rounds 2-4 derived them in every workgroup's prologue (FP64 divisions, verification loops, 64-bit products: measured
// at 26-36 % of the first layer's run time, tools/l0_phases.py); mi355_conv_pack_epilogue derives them ONCE on the host with the same
// functions (below: __host__ __device__) and the kernels load 32 bytes per channel.  A blob whose key does not match the launch's
// (activation, zero point) -- mi355_conv_pack alone, or a saturating store -- takes the in-kernel derivation: same bytes either way,
// because ANY sub-range of the true safe range is safe (it only decides which windows take the exact path).
// ---------------------------------------------------------------------------------------------------------
struct EptHeader {
    uint32_t key;        // ept_key(activation, zp_act), 0 = not prepared
    uint32_t flags;      // EPT_NEVER: some channel has no wrap-safe range;  EPT_NOINT: some channel fails the integer form's conditions (or !pow2)
    uint32_t pad_[2];
};
struct EptEntry {        // 32 bytes per channel, WRAP store
    int32_t lb;          // lower end of the (clamped) wrap-safe range: accumulators are kept biased by it
    uint32_t rg;         // its width
    int32_t m0, sh;      // integer requantisation: f = mulhi(a, m0) >> sh  (0, 0 when the channel does not qualify)
    int64_t qc;          // lb * m0 (first layer: u * m0 + qc = a * m0 in one v_mad_u64_u32)
    int32_t cbl;         // cw + bias - lb: the biased accumulator's seed
    int32_t pad_;
};
#define EPT_NEVER 1u
#define EPT_NOINT 2u
#define EPT_POW2 4u      // the header's pow2 flag, repeated here so that the kernels' prologue reads one place
#define EPT_D2 8u        // first layer: some channel's zero-point correction is 128 (a third MFMA round)
// First-layer blobs carry, behind the entries and the 4 KiB LEAKY byte table, the kernel's whole per-lane state in MFMA lane order -- one
// 256-byte record per (m-tile, lane): the prologue of conv_first_mfma_pool_kernel is then sixteen independent 16-byte loads from ONE
// address instead of dependent loads of zero points, weights, multipliers and entries with their selects (the kernel starts 1 024
// workgroups at once and nothing runs beside their prologues).  Lane (pc = lane & 15, g = lane >> 4): A row = channel 16 mt + pc,
// k-group g; accumulator rows = channels 16 mt + 4 g + r.
struct L0Lane {
    int32_t wa[2][4];    // weights w' = w - 128 of window column jx = 0 / 1 (shifted by one cell), zero outside the 3 x 3 x 3 taps
    int32_t wd1[2][4];   // the zero-point correction's constant min(dz, 127) in every real k slot
    int32_t wd2[2][4];   // ... and dz - 127 (non-zero only for dz = 128)
    int32_t cb[4], lo[4], hi[4];   // EptEntry cbl, lb, rg of the lane's four channels
    int32_t qm0[4], qsh[4];
    int64_t qc[4];
    double mp[4];        // folded multiplier M_value * shift_value
    int32_t pad_[4];
};
static_assert(sizeof(L0Lane) == 256, "L0Lane is one 256-byte record");
// (the store mode is not part of the key: the table's ranges are derived for the WRAPPING store, under which they are the tightest; a saturating
// store is monotone everywhere, so a user may only ever NARROW a table range for it, never widen it -- today every saturating launch ignores the
// table altogether: `have_ept = !SAT && ...` in the kernels)
__host__ __device__ inline uint32_t ept_key(int act, int zp_act) { return 0x45500000u | ((uint32_t)(act & 0xFF) << 8) | (uint32_t)(zp_act & 0xFF); }

// ---------------------------------------------------------------------------------------------------------
// Requantise epilogue: ref src/convolutional_layer.c:726-751.  FP64 with two truncations, activation, zero point,
// uint8 store that wraps (default path) or saturates (builder-defined mode, see mi355_yolo_int8.h).  `acc` is the true pre-requant accumulator.
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
__host__ __device__ __forceinline__ int32_t requant_q_exact(int32_t accb, double Mp) { return (int32_t)((double)accb * Mp); }
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

// ---------------------------------------------------------------------------------------------------------
// Group form used by the conv epilogues: 4 consecutive channels (one packed dword) x NS pixels held by a lane.
// accb[r][ns] = accumulator + cw + bias + dz*sx;  mp[r] = folded multiplier.  Returns the BIASED packed bytes
// (uint8 ^ 0x80).  ~11 VALU instructions per output, all full rate on gfx950 (tools/ubench/valu_rate.hip): accb is
// formed by the caller (1-2), cvt/mul/cvt (3), activation (LEAKY: 5 + a running minimum, see requant_values),
// v_perm packing (0.75) + one xor.
// ---------------------------------------------------------------------------------------------------------
template <int ACT, bool SAT, int NS>
__device__ __forceinline__ void requant_values(const int32_t (&accb)[4][NS], const double (&mp)[4], int zp_act,
                                               int32_t (&v)[4][NS])
{
    // LEAKY, branch free in 5 VALU instructions per output.  With p = max(q,0), nq = p - q (= max(-q,0)), C = 0xCCCD:
    //     zp + p - floor((nq+5)*C / 2^19)  ==  ((zp + p) * 2^19 + (2^19 - 1 - 5C) - nq*C) >> 19      (arithmetic shift)
    // because -floor(t/N) == floor((N - 1 - t)/N).  (nq+5)*C >> 19 == (nq+5)/10 while nq + 5 < 2^16; the int32 form
    // holds while nq <= 40000; p << 19 may wrap in WRAP mode, which only drops bits above the stored byte.  Lanes
    // outside that range (never on sane data) send the whole wave through the division form below.
    const int kleaky = (zp_act << 19) + ((1 << 19) - 1 - 5 * 0xCCCD);
    int32_t qmin = 0;
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int ns = 0; ns < NS; ++ns) {
            int32_t q = requant_q_exact(accb[r][ns], mp[r]);
            if (ACT == MI355_ACT_LEAKY) {
                qmin = min(qmin, q);
                if (SAT) q = min(q, 2047);  // zp + 2047 saturates anyway; keeps p << 19 inside int32
                const int32_t p = max(q, 0);
                const int32_t nq = p - q;
                const int32_t m = __mul24(nq, -0xCCCD) + kleaky;
                v[r][ns] = (int32_t)(((uint32_t)p << 19) + (uint32_t)m) >> 19;
            } else if (ACT == MI355_ACT_RELU6) {
                v[r][ns] = zp_act + max(q, 0);
            } else {
                v[r][ns] = zp_act + q;
            }
        }
    if (ACT == MI355_ACT_LEAKY && __builtin_amdgcn_ballot_w64(qmin < -40000) != 0) {  // keeps exactness on any data
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int ns = 0; ns < NS; ++ns) {
                // (selects, not branches: as exec-masked regions these sixteen cold values were most of the s_*_saveexec of the pooled kernels' loops)
                const int32_t q = requant_q_exact(accb[r][ns], mp[r]);
                const uint32_t x = (0u - (uint32_t)q) + 5u;
                const int32_t neg = zp_act - (int32_t)(__umulhi(x, 0xCCCCCCCDu) >> 3), pos = q + zp_act, m = q >> 31;  // x / 10 for every 32-bit x
                v[r][ns] = (neg & m) | (pos & ~m);
            }
    }
    if (SAT) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int ns = 0; ns < NS; ++ns) v[r][ns] = min(max(v[r][ns], 0), 255);
    }
}

// The same for NV independent values with a multiplier each (conv_ws3.hip requantises 8 channels of one pixel per call:
// one fallback ballot, eight independent dependency chains).
template <int ACT, bool SAT, int NV>
__device__ __forceinline__ void requant_values_mp(const int32_t (&accb)[NV], const double (&mp)[NV], int zp_act, int32_t (&v)[NV])
{
    const int kleaky = (zp_act << 19) + ((1 << 19) - 1 - 5 * 0xCCCD);
    // convert / multiply / convert in three passes over the NV values: left alone the compiler threads all of them
    // through one register pair, and every FP64 instruction then waits for the full latency of the one before it
    double d[NV];
    int32_t q[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) d[i] = (double)accb[i];
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < NV; ++i) d[i] = d[i] * mp[i];
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < NV; ++i) q[i] = (int32_t)d[i];
    __builtin_amdgcn_sched_barrier(0);
    int32_t qmin = 0;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        int32_t qq = q[i];
        if (ACT == MI355_ACT_LEAKY) {
            qmin = min(qmin, qq);
            if (SAT) qq = min(qq, 2047);
            const int32_t p = max(qq, 0);
            const int32_t nq = p - qq;
            const int32_t m = __mul24(nq, -0xCCCD) + kleaky;
            v[i] = (int32_t)(((uint32_t)p << 19) + (uint32_t)m) >> 19;
        } else if (ACT == MI355_ACT_RELU6) {
            v[i] = zp_act + max(qq, 0);
        } else {
            v[i] = zp_act + qq;
        }
    }
    if (ACT == MI355_ACT_LEAKY && __builtin_amdgcn_ballot_w64(qmin < -40000) != 0) {  // keeps exactness on any data
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const uint32_t x = (0u - (uint32_t)q[i]) + 5u;
            const int32_t neg = zp_act - (int32_t)(__umulhi(x, 0xCCCCCCCDu) >> 3), pos = q[i] + zp_act, m = q[i] >> 31;
            v[i] = (neg & m) | (pos & ~m);
        }
    }
    if (SAT) {
#pragma unroll
        for (int i = 0; i < NV; ++i) v[i] = min(max(v[i], 0), 255);
    }
}

// ---------------------------------------------------------------------------------------------------------
// LEAKY through a byte table.  After q = trunc(acc * M') the rest of the epilogue -- activation, zero point, uint8 wrap,
// the ^0x80 bias of the stored byte -- is a function of q and of two per-LAYER constants only (zp_act and the store mode):
// ref src/convolutional_layer.c:737  q < 0 ? round(q * 0.1) + zp : q + zp.  A 4 KiB table over q in [-LUTQ_OFF, LUTQ_N -
// LUTQ_OFF) in LDS, filled with the arithmetic below once per workgroup, turns the 6 VALU instructions of the branch-free
// form (+ the running minimum of its fallback test) into one 2-clock add and one ds_read_u8.  Values of q outside the table
// (only possible where bytes wrap: |q| in the thousands) must take the arithmetic path: the pooled kernels know from their
// safe-range test that no byte of the window wraps, i.e. zp + q <= 255 and 10 zp + 4 >= -q: always inside the table.
// ---------------------------------------------------------------------------------------------------------
constexpr int LUTQ_OFF = 3072, LUTQ_N = 4096;
// Index of floor-form value f in the table.  The pooled kernels look the byte up BEFORE they know whether the window lay inside the wrap-safe
// range (a window outside it is redone in the reference's order and the byte discarded), so f may be any int32: the index is clamped into the
// table with one v_med3_i32 -- an unclamped index is an out-of-bounds array access in C++ even though the hardware read is harmless.
// -DMI355_LUT_NOCLAMP restores round 5's unclamped read for A/B timing only.
__device__ __forceinline__ int lutq_index(int f) {
#ifdef MI355_LUT_NOCLAMP
    return f + LUTQ_OFF;
#else
    return min(max(f, -LUTQ_OFF), LUTQ_N - LUTQ_OFF - 1) + LUTQ_OFF;
#endif
}
template <bool SAT>
__host__ __device__ __forceinline__ uint32_t leaky_byte_biased(int32_t q, int zp_act)
{
    const uint32_t x = (0u - (uint32_t)q) + 5u;
    int32_t v = q < 0 ? zp_act - (int32_t)(x / 10u) : q + zp_act;
    if (SAT) v = v < 0 ? 0 : (v > 255 ? 255 : v);
    return ((uint32_t)v & 0xFFu) ^ 0x80u;
}
// all `nthreads` threads of the workgroup; the caller puts a barrier between this and the first lookup
template <bool SAT>
__device__ __forceinline__ void leaky_lut_build(uint8_t *lut, int zp_act, int tid, int nthreads)
{
    for (int i = tid; i < LUTQ_N / 4; i += nthreads) {
        uint32_t w = 0;
#pragma unroll
        for (int e = 0; e < 4; ++e) w |= leaky_byte_biased<SAT>(4 * i + e - LUTQ_OFF, zp_act) << (8 * e);
        reinterpret_cast<uint32_t *>(lut)[i] = w;
    }
}
// ---------------------------------------------------------------------------------------------------------
// Integer requantisation (round 4).  With the reference's decomposition M = M0 * 2^-31 * 2^-s (src/blas.c:387-418: M0 an int32 in
// [2^30, 2^31), shift_value = 2^-s exactly), the epilogue's  q = trunc(fl((double)a * M_value) * 2^-s)  is, for every accumulator a with
// |a| * M0 < 2^53 (the FP64 product is then EXACT, so nothing is rounded before the truncation),
//     q = trunc(a * M0 / 2^(31+s)),       f := floor(a * M0 / 2^(31+s)) = mulhi_i32(a, M0) >> (s - 1)      (s >= 1)
// and f == q for a >= 0, f == q - 1 for a < 0 unless a * M0 is an exact multiple of 2^(31+s) -- impossible for -2^(31+s-tz) < a < 0,
// tz = trailing zero bits of M0.  Two full-rate integer instructions (v_mul_hi_i32, v_ashrrev_i32) replace convert / FP64 multiply /
// convert and the 64-bit register pairs that go with them.  The pooled kernels only requantise window maxima INSIDE their wrap-safe
// accumulator range [lo, hi], so the three conditions are checked per channel against that range when the kernel starts
// (intrq_make); a channel that fails keeps the FP64 form for the whole launch.  What consumes f: the LEAKY byte table indexed by f
// (leaky_lutf_build: entry f holds the byte of q = f + (f < 0)), or RELU6's zp + max(f, 0) == zp + max(q, 0).
// ---------------------------------------------------------------------------------------------------------
// neg_any (RELU / RELU6): every negative accumulator is stored as the zero point whatever its q -- zp + max(f, 0) only needs f < 0 there, which
// floor(a * M0 / 2^(31+s)) is for every a < 0 -- so the conditions apply to the non-negative end of the range only.  (Without this a RELU6
// layer never qualified: its wrap-safe range reaches down to the clamp at -2^30, and 2^30 * M0 is far beyond 2^53.)
__host__ __device__ __forceinline__ bool intrq_make(double mval, int s, int32_t lo, int32_t hi, int32_t &m0, int32_t &sh, bool neg_any = false)
{
    m0 = 0; sh = 0;
    if (!(mval > 0.0 && mval < 1.0) || s < 1 || s > 31) return false;
    const double t = mval * 2147483648.0;   // exact: a power-of-two scaling
    const int32_t m = (int32_t)t;
    if ((double)m != t || m <= 0) return false;   // M_value is not an int32 * 2^-31
    if (neg_any && lo < 0) lo = 0;
    if (hi < lo) hi = lo;
    const long amax = (-(long)lo > (long)hi) ? -(long)lo : (long)hi;
    // the FP64 product a * M_value = a * (m >> tz) * 2^(tz - 31) is exact when a * (m >> tz) has at most 53 significant bits (m = round(float * 2^31)
    // carries >= 7 trailing zeros: the plain a * m < 2^53 rejected channels that are exact and sent their launches down the slow path)
    const int tz = __builtin_ctz((unsigned)m);
    if (amax < 0 || amax >= (1l << 31) || (unsigned long)amax * (unsigned long)((unsigned)m >> tz) >= (1ul << 53)) return false;  // FP64 product not provably exact
    const int e = 31 + s - tz;   // a * M0 % 2^(31+s) == 0  <=>  a % 2^e == 0
    if (e < 40 && lo < 0 && -(long)lo >= (1l << e)) return false;  // a negative exact multiple inside the range
    m0 = m; sh = s - 1;
    return true;
}
__device__ __forceinline__ int32_t intrq_floor(int32_t a, int32_t m0, int32_t sh) { return __mulhi(a, m0) >> sh; }
// the LEAKY byte table for the floor form: entry f -> byte of q = f + (f < 0)
template <bool SAT>
__device__ __forceinline__ void leaky_lutf_build(uint8_t *lut, int zp_act, int tid, int nthreads)
{
    for (int i = tid; i < LUTQ_N / 4; i += nthreads) {
        uint32_t w = 0;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int f = 4 * i + e - LUTQ_OFF;
            w |= leaky_byte_biased<SAT>(f < 0 ? f + 1 : f, zp_act) << (8 * e);
        }
        reinterpret_cast<uint32_t *>(lut)[i] = w;
    }
}
// LEAKY on the floor form, branch free in four VALU instructions, for -40 900 <= f <= 40 000 (the int32 product f * 52428 + k; always true for window maxima inside the
// wrap-safe range: -10 zp - 6 <= f <= 255):  with q = f + (f < 0),
//     q < 0:  round(q * 0.1) = -((|q| + 5) / 10) = floor((q + 4) / 10) = floor((f + 5) / 10) = ((f + 5) * 52428) >> 19      (arithmetic shift)
//     q >= 0: q = f
// and the first expression never exceeds f for f >= 0 nor falls below it for f < 0:  v = zp + max(floor form, f).
__device__ __forceinline__ int32_t leaky_of_floor(int32_t f, int zp_act)
{
    const int32_t k = 5 * 52428 + (zp_act << 19);
    return max((__mul24(f, 52428) + k) >> 19, f + zp_act);
}
// four table bytes (each in the low byte of its dword) -> one packed dword
__device__ __forceinline__ uint32_t pack4_bytes(uint32_t b0, uint32_t b1, uint32_t b2, uint32_t b3)
{
    const uint32_t p01 = __builtin_amdgcn_perm(b1, b0, 0x0c0c0400u);
    const uint32_t p23 = __builtin_amdgcn_perm(b3, b2, 0x0c0c0400u);
    return __builtin_amdgcn_perm(p23, p01, 0x05040100u);
}

// ---------------------------------------------------------------------------------------------------------
// Wrap-safe accumulator ranges: max-pooling commutes with the requantisation where no stored byte wraps.
// ---------------------------------------------------------------------------------------------------------
// activation + zero point of a requantised value, unwrapped (the byte is this & 0xFF or its clamp)
template <int ACT>
__host__ __device__ __forceinline__ long small_v_of(int32_t accb, double mp, int zp)
{
    const int32_t q = requant_q_exact(accb, mp);
    if (ACT == MI355_ACT_LEAKY) {
        if (q >= 0) return (long)zp + q;
        const uint32_t x = (0u - (uint32_t)q) + 5u;
        return (long)zp - (long)(x / 10u);
    }
    if (ACT == MI355_ACT_RELU6) return (long)zp + (q > 0 ? q : 0);
    return (long)zp + q;
}

// [lo, hi]: accumulators (incl. bias and zero-point terms) whose stored byte does not wrap.  Any sub-range of the true
// one is safe (it only sends more waves down the exact path), so the analytic guess is moved inwards until it verifies.
template <int ACT>
__host__ __device__ inline void small_safe_range(double mp, int zp, int32_t &lo, int32_t &hi)
{
    // upper end: zp + q <= 255  <=>  q <= 255 - zp, q = trunc(a * mp)
    double gh = ((double)(256 - zp)) / mp;
    long h = gh >= 2147483000.0 ? 2147483647L : (long)gh;
    for (int it = 0; it < 64 && h > -2147483647L && small_v_of<ACT>((int32_t)h, mp, zp) > 255; ++it) h -= (it < 8 ? 1 : 4096);
    if (small_v_of<ACT>((int32_t)h, mp, zp) > 255) h = -2147483647L - 1;  // give up: nothing is safe
    hi = (int32_t)h;
    long l;
    if (ACT == MI355_ACT_RELU6) {
        l = -2147483647L - 1;  // zp + max(q, 0) >= zp >= 0
    } else {
        const double qlo = (ACT == MI355_ACT_LEAKY) ? -(10.0 * zp + 5.0) : -(double)(zp + 1);
        const double gl = qlo / mp;
        l = gl <= -2147483000.0 ? -2147483647L - 1 : (long)gl;
        for (int it = 0; it < 64 && l < 2147483647L && small_v_of<ACT>((int32_t)l, mp, zp) < 0; ++it) l += (it < 8 ? 1 : 4096);
        if (small_v_of<ACT>((int32_t)l, mp, zp) < 0) l = 2147483647L;
    }
    lo = (int32_t)l;
}

// The pooled kernels keep their accumulators BIASED by the lower end of the safe range: seeded with cw + bias - lo instead of
// cw + bias (the seed is the MFMA's C operand: free).  Then  u = acc - lo  as an unsigned number is <= hi - lo exactly when
// lo <= acc <= hi, so one unsigned maximum over the 2x2 window answers both "is every accumulator of the window inside the
// safe range" (max_u <= hi - lo) and "what is the window's largest accumulator" (max_u + lo, valid when the first holds):
// 4 VALU instructions per pooled output instead of ~8 (signed max, signed min, two compares).  Exact for every int32
// accumulator as long as -2^30 <= lo and hi < 2^30 (no modular alias of an out-of-range value lands in [0, hi - lo]); any
// sub-range of the true safe range is safe, so the ends are simply clamped.  Returns false when no accumulator is safe for
// this channel (the caller then requantises every value of every window: the reference's order).
__host__ __device__ __forceinline__ bool biased_safe_range(int32_t lo, int32_t hi, int32_t &lo_b, uint32_t &range)
{
    const int32_t L = lo < -(1 << 30) ? -(1 << 30) : lo, H = hi > (1 << 30) - 1 ? (1 << 30) - 1 : hi;
    lo_b = L;
    range = H >= L ? (uint32_t)(H - L) : 0u;
    return H >= L;
}

// yolo head activation of entry e = channel % (classes + 5): logistic on x, y, objectness and the class scores, identity on
// w, h (ref: src/yolo_layer.c:132-146, src/activations.h:39; double precision exp as the reference's logistic_activate)
__device__ __forceinline__ float yolo_entry_act(float v, int e)
{
    return (e == 2 || e == 3) ? v : (float)(1. / (1. + exp(-(double)v)));
}

// low bytes of four ints -> one dword, biased (^0x80): 3 v_perm + 1 v_xor
__device__ __forceinline__ uint32_t pack4_biased(int32_t v0, int32_t v1, int32_t v2, int32_t v3)
{
    const uint32_t p01 = __builtin_amdgcn_perm((uint32_t)v1, (uint32_t)v0, 0x0c0c0400u);
    const uint32_t p23 = __builtin_amdgcn_perm((uint32_t)v3, (uint32_t)v2, 0x0c0c0400u);
    return __builtin_amdgcn_perm(p23, p01, 0x05040100u) ^ 0x80808080u;
}

// The 16 x 16 x 64 pooled kernels' (conv_aux.hip first layer, conv_pool16.hip) exact path: every value of the 2x2 window requantised, then the maximum of the BYTES (the reference's
// order: src/convolutional_layer.c:737-749 then src/maxpool_layer.c:134-146).  Only taken by waves that see an accumulator outside the
// wrap-safe range.  acc[j][r]: window position j, channel r, biased by lo[r].
template <int ACT, bool SAT>
__device__ __forceinline__ uint32_t first_pool_exact_path(const v4i (&acc)[4], const v4i &lo, const double (&mp)[4], const double *mval4,
                                                       const double *sval4, int zp_act, bool pow2)
{
    int32_t accb[4][4], m[4];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int j = 0; j < 4; ++j) accb[r][j] = (int32_t)((uint32_t)acc[j][r] + (uint32_t)lo[r]);  // true accumulators
    if (pow2) {
        int32_t v[4][4];
        requant_values<ACT, SAT, 4>(accb, mp, zp_act, v);
#pragma unroll
        for (int r = 0; r < 4; ++r) m[r] = max(max(v[r][0] & 0xFF, v[r][1] & 0xFF), max(v[r][2] & 0xFF, v[r][3] & 0xFF));
    } else {  // (a rolled loop: unrolled, these sixteen two-step requantisations size the whole kernel's registers -- conv_small.hip)
        int32_t tmp[16];
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int j = 0; j < 4; ++j) tmp[4 * r + j] = accb[r][j];
#pragma unroll 1
        for (int idx = 0; idx < 16; ++idx)
            tmp[idx] = (int32_t)requant_u8(tmp[idx], 0, mval4[idx >> 2], sval4[idx >> 2], zp_act, ACT, SAT ? MI355_STORE_SATURATE : MI355_STORE_WRAP);
#pragma unroll
        for (int r = 0; r < 4; ++r) m[r] = max(max(tmp[4 * r], tmp[4 * r + 1]), max(tmp[4 * r + 2], tmp[4 * r + 3]));
    }
    return pack4_biased(m[0], m[1], m[2], m[3]);
}

// Quantized residual add on four packed (biased) bytes: a = this conv's requantised bytes, b = the `from` tensor's
//     q = (Ka*a + Kb*b + k0) >> 16, clamp(0, 255)          k0 = 2^15 + (zp_out << 16) - Ka*zp_a - Kb*zp_b
// (oracle.c:orc_shortcut_u8).  The empty asm keeps shift and clamp apart: fused, hipcc (ROCm 7.2) pairs
// them into V_ASHR_PK_U8_I32 and ORs further bytes into its result assuming bits 31:16 are zero, which gfx950 does not do.
__device__ __forceinline__ uint32_t shortcut4_biased(uint32_t a_biased, uint32_t b_biased, int ka, int kb, int k0)
{
    const uint32_t wa = a_biased ^ 0x80808080u, wb = b_biased ^ 0x80808080u;
    uint32_t w = 0;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int av = (int)((wa >> (8 * e)) & 0xFFu), bv = (int)((wb >> (8 * e)) & 0xFFu);
        int q = (ka * av + kb * bv + k0) >> 16;
        asm volatile("" : "+v"(q));
        q = q < 0 ? 0 : (q > 255 ? 255 : q);
        w |= (uint32_t)q << (8 * e);
    }
    return w ^ 0x80808080u;
}

template <int ACT, bool SAT, int NS>
__device__ __forceinline__ void requant_group(const int32_t (&accb)[4][NS], const double (&mp)[4], int zp_act,
                                              uint32_t (&packed)[NS])
{
    int32_t v[4][NS];
    requant_values<ACT, SAT, NS>(accb, mp, zp_act, v);
#pragma unroll
    for (int ns = 0; ns < NS; ++ns) packed[ns] = pack4_biased(v[0][ns], v[1][ns], v[2][ns], v[3][ns]);
}

// bytewise signed max of two dwords holding 4 biased (x ^ 0x80) activations each: max_u8 on the raw values ==
// max_s8 on the biased ones (maxpool, ref src/maxpool_layer.c:134-146).
// Packed 16-bit maxima: a byte compares like itself << 8 as a signed halfword, so the even bytes are shifted into the high byte of
// their halfword, the odd bytes are masked in place, and two v_pk_max_i16 do the four comparisons (8 instructions instead of
// ~20 extract / compare / insert ones: the fused pools of conv_ws3.hip are bound by this count).
typedef short v2s __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t max_s8x4(uint32_t p, uint32_t q)
{
    const v2s pe = __builtin_bit_cast(v2s, p) << 8, qe = __builtin_bit_cast(v2s, q) << 8;
    const v2s po = __builtin_bit_cast(v2s, p & 0xFF00FF00u), qo = __builtin_bit_cast(v2s, q & 0xFF00FF00u);
    const v2s me = __builtin_elementwise_max(pe, qe), mo = __builtin_elementwise_max(po, qo);
    return (__builtin_bit_cast(uint32_t, me) >> 8) | __builtin_bit_cast(uint32_t, mo);
}
// ... of four dwords (a 2x2 window)
__device__ __forceinline__ uint32_t max4_s8x4(uint32_t a, uint32_t b, uint32_t c, uint32_t d)
{
    const v2s ae = __builtin_bit_cast(v2s, a) << 8, be = __builtin_bit_cast(v2s, b) << 8, ce = __builtin_bit_cast(v2s, c) << 8,
              de = __builtin_bit_cast(v2s, d) << 8;
    const v2s ao = __builtin_bit_cast(v2s, a & 0xFF00FF00u), bo = __builtin_bit_cast(v2s, b & 0xFF00FF00u),
              co = __builtin_bit_cast(v2s, c & 0xFF00FF00u), dO = __builtin_bit_cast(v2s, d & 0xFF00FF00u);
    const v2s me = __builtin_elementwise_max(__builtin_elementwise_max(ae, be), __builtin_elementwise_max(ce, de));
    const v2s mo = __builtin_elementwise_max(__builtin_elementwise_max(ao, bo), __builtin_elementwise_max(co, dO));
    return (__builtin_bit_cast(uint32_t, me) >> 8) | __builtin_bit_cast(uint32_t, mo);
}

// Division of 0 <= n < 2^31 by a launch constant d >= 1 without the ~35-instruction sequence the compiler emits for a runtime
// divisor:  q = (mulhi(n, m) + n) >> s  with  s = ceil(log2 d),  m = floor(2^32 (2^s - d) / d) + 1  (exact for n < 2^31).
struct FastDiv {
    uint32_t m;
    int32_t s;
};
static inline FastDiv fastdiv_make(uint32_t d)
{
    FastDiv f;
    int s = 0;
    while ((1ull << s) < d) ++s;
    f.s = s;
    f.m = (uint32_t)(((1ull << 32) * ((1ull << s) - d)) / d + 1);
    return f;
}
__device__ __forceinline__ int fd_div(int n, FastDiv f) { return (int)((__umulhi((uint32_t)n, f.m) + (uint32_t)n) >> f.s); }

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
