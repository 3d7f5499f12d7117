"""Row a16: the reference's MKL flavour (`OPENBLAS=1`, ref src/convolutional_layer.c:528-605) cannot be BUILT here
(`mkl.h` is not in the image; stand-in headers are not allowed) -- but the library its GEMM calls is: `/opt/conda/lib/libmkl_rt.so` exports
`cblas_gemm_s16s16s32`, the third-party entry point of ref :557-569.  This test calls that entry point through its published C interface with
exactly the reference's two argument lists (weights as int16 with alpha = 1, the zero-point matrix with alpha = -1, beta = 1, fixed offset 0)
on operands whose pass-1 sums exceed 2^24 -- the regime in which the DEFAULT flavour's fp32 accumulation rounds -- and
checks that the result is the exact integer accumulator the oracle's MI355_ACC_EXACT mode produces (what the HIP kernels compute).

What this pins: the GEMM half of the MKL flavour = exact int32, against the real library.  What stays a restatement: its plain-C epilogue
(:572-596, oracle.c:orc_requant_mkl), covered by the exhaustive saturate / MKL-LEAKY equivalence in test_host_cpu.py.
Container-only (skips where libmkl_rt is absent, e.g. on the GPU box); not a `gpu` test."""
import ctypes as C
import glob
import os

import numpy as np
import pytest

import oracle

CblasRowMajor, CblasNoTrans, CblasFixOffset = 101, 111, 173  # mkl_cblas.h: CBLAS_LAYOUT / CBLAS_TRANSPOSE / CBLAS_OFFSET {Row 171, Col 172, Fix 173}


def _mkl():
    os.environ.setdefault("MKL_THREADING_LAYER", "SEQUENTIAL")  # no OpenMP runtime games inside pytest
    for p in sorted(glob.glob("/opt/conda/lib/libmkl_rt.so*")):
        try:
            lib = C.CDLL(p, mode=C.RTLD_GLOBAL)
        except OSError:
            continue
        if hasattr(lib, "cblas_gemm_s16s16s32"):
            f = lib.cblas_gemm_s16s16s32
            f.restype = None
            # (Layout, TransA, TransB, OffsetC, M, N, K, float alpha, const int16 *A, lda, int16 ao, const int16 *B, ldb, int16 bo,
            #  float beta, int32 *C, ldc, const int32 *cb) -- LP64 interface: MKL_INT = int
            f.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_void_p, C.c_int, C.c_int16,
                          C.c_void_p, C.c_int, C.c_int16, C.c_float, C.c_void_p, C.c_int, C.c_void_p]
            return f
    return None


@pytest.mark.parametrize("c,n,hw,lo,hi", [(512, 64, 13, 0, 256),     # L12's K = 4608 on dense bytes: pass-1 sums ~ 1e8 >> 2^24
                                          (384, 32, 26, 100, 256),   # L21's K = 3456, bright input
                                          (16, 32, 20, 0, 256)])     # small K: inside the fp32-exact regime as well
def test_mkl_gemm_s16s16s32_is_the_exact_integer_accumulator(c, n, hw, lo, hi):
    gemm = _mkl()
    if gemm is None:
        pytest.skip("libmkl_rt.so with cblas_gemm_s16s16s32 not in this image")
    rng = np.random.default_rng(c + n + hw)
    ksize, pad, zp_in = 3, 1, 17
    x = rng.integers(lo, hi, (c, hw, hw), dtype=np.uint8)
    k = c * ksize * ksize
    wq = rng.integers(0, 256, (n, k), dtype=np.uint8)
    zp_w = rng.integers(90, 170, n, dtype=np.uint8)
    want, s1 = oracle.conv_acc(x, wq, zp_w, ksize, 1, pad, zp_in, oracle.ACC_EXACT, want_s1=True)
    # the reference's operands (ref :542-555, src/blas.c:191-192): everything widened to int16, im2col padded with the input zero point
    a16 = wq.astype(np.int16)
    z16 = np.repeat(zp_w.astype(np.int16)[:, None], k, axis=1).copy()
    b16 = oracle.im2col_u8(x, ksize, 1, pad, zp_in).astype(np.int16)
    m, nn = n, hw * hw
    got = np.zeros((m, nn), np.int32)
    co = np.zeros(1, np.int32)
    for alpha, A in ((1.0, a16), (-1.0, z16)):  # ref :557-562 and :564-569
        gemm(CblasRowMajor, CblasNoTrans, CblasNoTrans, CblasFixOffset, m, nn, k, alpha, A.ctypes.data, k, 0,
             b16.ctypes.data, nn, 0, 1.0, got.ctypes.data, nn, co.ctypes.data)
    if c >= 384:
        assert int(np.abs(s1).max()) > (1 << 24), "the case must leave the regime in which fp32 accumulation is exact"
    assert np.array_equal(got, want)
    # and the default flavour's fp32 accumulation does NOT agree there: the two flavours of the reference differ, which is why the
    # product has both MI355_ACC_EXACT (this one) and MI355_ACC_REF_F32
    if c >= 384:
        f32 = oracle.conv_acc(x, wq, zp_w, ksize, 1, pad, zp_in, oracle.ACC_REF_F32)
        assert not np.array_equal(f32, want)
