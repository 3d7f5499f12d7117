"""GPU parity suite (`-m gpu`, runs on the MI355X box): every kernel is called through the C-ABI
(libmi355yolo.so / libdarknet_q.so) and compared bit-for-bit with the oracle and the committed golden fixtures.
Nothing here reads /root/reference."""
import hashlib
import json
import os

import numpy as np
import pytest

import oracle
from yolo_quantization_amd import binding, synth

pytestmark = pytest.mark.gpu


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


@pytest.fixture(scope="module", autouse=True)
def device():
    binding.init(0)


def _rand_layer(rng, n, c, k, m_lo=2.0 ** -11, m_hi=2.0 ** -7):
    K = c * k * k
    wq = rng.integers(0, 256, (n, K), dtype=np.uint8)
    zp_w = rng.integers(90, 166, n, dtype=np.uint8)
    bias = rng.integers(-20000, 20000, n).astype(np.int32)
    M = rng.uniform(m_lo, m_hi, n)
    # realistic decomposition M = M0*2^-31 * 2^-shift
    shift = np.floor(-np.log2(M)).astype(int)
    M0 = np.round(M * 2.0 ** shift * 2 ** 31)
    return wq, zp_w, bias, M0 * 2.0 ** -31, 2.0 ** -shift.astype(np.float64)


def _oracle_layer(x, wq, zp_w, k, zp_in, bias, mv, sv, zp_act, act, store, accum, stride=1):
    B = x.shape[0]
    accs, u8s = [], []
    for b in range(B):
        a = oracle.conv_acc(x[b], wq, zp_w, k, stride, k // 2, zp_in, accum)
        accs.append(a)
        u8s.append(oracle.requant(a, bias, mv, sv, zp_act, act, store))
    return np.stack(accs), np.stack(u8s)


CONV_CASES = [
    # (B, c, n, H, W, k, act, zp_in, zp_act)   -- ragged tiles, odd sizes, every chunk width, n not multiple of 16
    (1, 16, 32, 12, 12, 3, "leaky", 23, 23),
    (3, 32, 64, 7, 5, 3, "relu6", 0, 0),
    (2, 64, 128, 13, 13, 3, "leaky", 23, 23),
    (2, 128, 30, 13, 13, 1, "linear", 23, 128),
    (1, 384, 256, 26, 26, 3, "leaky", 23, 23),
    (5, 256, 255, 13, 13, 1, "linear", 0, 128),
    (2, 48, 40, 9, 11, 3, "relu", 7, 0),
    (1, 16, 16, 40, 300, 3, "leaky", 200, 23),   # wide rows: halo >> tile
    (1, 1024, 256, 13, 13, 1, "relu6", 0, 0),
    (1, 512, 1024, 13, 13, 3, "leaky", 23, 23),  # K = 4608: the deep layer of the net
]


@pytest.mark.parametrize("case", CONV_CASES, ids=lambda c: "B%d_c%d_n%d_%dx%d_k%d_%s" % c[:7])
@pytest.mark.parametrize("store", [binding.STORE_WRAP, binding.STORE_SATURATE], ids=["wrap", "saturate"])
def test_conv_mfma_bit_exact(case, store):
    B, c, n, H, W, k, act, zp_in, zp_act = case
    rng = np.random.default_rng(sum(v for v in case if isinstance(v, int)))
    x = rng.integers(0, 256, (B, c, H, W), dtype=np.uint8)
    wq, zp_w, bias, mv, sv = _rand_layer(rng, n, c, k)
    xt = binding.DevTensor.from_nchw(x, zp_in)
    got = binding.conv_forward(xt, wq, zp_w, k, bias, mv, sv, zp_in, zp_act, 0.05, binding.ACT[act], store,
                               binding.ACC_EXACT, want_acc=True, want_f32=True)
    acc, u8 = _oracle_layer(x, wq, zp_w, k, zp_in, bias, mv, sv, zp_act, oracle.ACT[act], store, oracle.ACC_EXACT)
    assert np.array_equal(got["int32"], acc), "int32 pre-requant accumulators"
    assert np.array_equal(got["u8"].reshape(B, n, H * W), u8), "uint8 activations"
    f32 = oracle.dequant(u8, zp_act, np.float32(0.05))
    assert np.array_equal(got["f32"], f32), "quant_stop float tail"
    if store == binding.STORE_WRAP:
        sat = np.stack([oracle.requant(acc[b], bias, mv, sv, zp_act, oracle.ACT[act], oracle.STORE_SATURATE) for b in range(B)])
        assert (sat != u8).any(), "case should exercise out-of-range (wrap != saturate) elements"
    # the throughput path: no dumps requested -> the kernels take their specialised (branch-free) epilogue
    fast = binding.conv_forward(xt, wq, zp_w, k, bias, mv, sv, zp_in, zp_act, 0.05, binding.ACT[act], store,
                                binding.ACC_EXACT, want_acc=False, want_f32=True)
    assert np.array_equal(fast["u8"].reshape(B, n, H * W), u8), "uint8 activations, fast epilogue"
    assert np.array_equal(fast["f32"], f32), "quant_stop float tail, fast epilogue"


@pytest.mark.parametrize("store", [binding.STORE_WRAP, binding.STORE_SATURATE], ids=["wrap", "saturate"])
@pytest.mark.parametrize("c,n,k,H,W", [(64, 128, 3, 13, 13), (256, 64, 1, 26, 26), (32, 32, 3, 20, 20), (3, 16, 3, 24, 24)])
def test_conv_fast_epilogue_huge_requantised_values(c, n, k, H, W, store):
    """Multipliers close to 1: |q| reaches 10^5..10^6, far outside the range of the 24-bit leaky shortcut, so the
    wave-uniform exact fallback of the fast epilogue has to produce the reference's bytes."""
    rng = np.random.default_rng(c + n + k)
    x = rng.integers(0, 256, (2, c, H, W), dtype=np.uint8)
    wq, zp_w, bias, mv, sv = _rand_layer(rng, n, c, k, 2.0 ** -3, 2.0 ** -1)
    xt = binding.DevTensor.from_nchw(x, 17)
    got = binding.conv_forward(xt, wq, zp_w, k, bias, mv, sv, 17, 23, 1.0, binding.ACT["leaky"], store, want_acc=False)
    acc, u8 = _oracle_layer(x, wq, zp_w, k, 17, bias, mv, sv, 23, oracle.LEAKY, store, oracle.ACC_EXACT)
    assert np.abs(acc).max() * mv.max() * sv.max() > 45000
    assert np.array_equal(got["u8"].reshape(2, n, H * W), u8)


@pytest.mark.parametrize("B,c,n,H,W,k", [(2, 32, 64, 40, 38, 3), (1, 64, 128, 76, 76, 3), (2, 128, 256, 19, 21, 3),
                                          (1, 16, 32, 31, 50, 3), (1, 256, 512, 38, 38, 3), (3, 48, 40, 9, 11, 3),
                                          # even maps with 16 / 32 channels: the weights-stationary kernel's stride-2 mode
                                          (2, 16, 32, 24, 300, 3), (1, 32, 32, 64, 64, 3), (2, 16, 64, 18, 22, 3), (1, 32, 64, 304, 304, 3),
                                          (2, 64, 96, 30, 46, 3), (1, 64, 128, 152, 152, 3)])
@pytest.mark.parametrize("store", [binding.STORE_WRAP, binding.STORE_SATURATE], ids=["wrap", "saturate"])
def test_conv_stride2(B, c, n, H, W, k, store):
    """Stride-2 convolutions (the downsampling layers of full YOLOv3, BASELINE config[4]; ref: the same
    forward_convolutional_layer_quant_inputi_outputi, im2col with stride 2): accumulators, bytes and the float tail
    against the oracle, odd and even maps, dump path and fast epilogue."""
    rng = np.random.default_rng(B + c + n + H + W)
    x = rng.integers(0, 256, (B, c, H, W), dtype=np.uint8)
    wq, zp_w, bias, mv, sv = _rand_layer(rng, n, c, k)
    xt = binding.DevTensor.from_nchw(x, 11)
    got = binding.conv_forward(xt, wq, zp_w, k, bias, mv, sv, 11, 23, 0.05, binding.ACT["leaky"], store, binding.ACC_EXACT,
                               want_acc=True, want_f32=True, stride=2)
    acc, u8 = _oracle_layer(x, wq, zp_w, k, 11, bias, mv, sv, 23, oracle.LEAKY, store, oracle.ACC_EXACT, stride=2)
    OH, OW = (H + 2 * (k // 2) - k) // 2 + 1, (W + 2 * (k // 2) - k) // 2 + 1
    assert acc.shape == (B, n, OH * OW)
    assert np.array_equal(got["int32"], acc), "int32 pre-requant accumulators"
    assert np.array_equal(got["u8"].reshape(B, n, OH * OW), u8), "uint8 activations"
    assert np.array_equal(got["f32"], oracle.dequant(u8, 23, np.float32(0.05)))
    fast = binding.conv_forward(xt, wq, zp_w, k, bias, mv, sv, 11, 23, 0.05, binding.ACT["leaky"], store, binding.ACC_EXACT,
                                want_acc=False, stride=2)
    assert np.array_equal(fast["u8"].reshape(B, n, OH * OW), u8), "uint8 activations, fast epilogue"


@pytest.mark.parametrize("B,c,n,H,W,act", [(1, 128, 30, 13, 13, "linear"), (3, 256, 96, 26, 26, "leaky"), (2, 512, 255, 13, 13, "linear"),
                                           (1, 1024, 256, 13, 13, "leaky"), (64, 256, 128, 13, 13, "leaky"), (2, 128, 33, 5, 7, "relu6"),
                                           (2, 64, 32, 76, 76, "leaky"), (1, 64, 200, 9, 31, "linear"),
                                           (1, 1024, 1, 9, 9, "relu")])
@pytest.mark.parametrize("store", [binding.STORE_WRAP, binding.STORE_SATURATE], ids=["wrap", "saturate"])
def test_conv1x1_weights_stationary_kernel(B, c, n, H, W, act, store):
    """The 1x1 kernel for the network's neck and heads (conv1x1.hip: weights held in registers, one balanced pixel
    tile per workgroup) against the oracle, and against the row-image kernel it replaces (debug flag 8192 routes the
    same call there): every channel depth it is compiled for, ragged output-channel counts, tiles that end mid-image."""
    rng = np.random.default_rng(B + c + n + H + W)
    x = rng.integers(0, 256, (B, c, H, W), dtype=np.uint8)
    wq, zp_w, bias, mv, sv = _rand_layer(rng, n, c, 1)
    xt = binding.DevTensor.from_nchw(x, 23)
    args = (xt, wq, zp_w, 1, bias, mv, sv, 23, 128, 0.05, binding.ACT[act], store, binding.ACC_EXACT)
    got = binding.conv_forward(*args, want_acc=False, want_f32=True)
    assert binding.shim().mi355_last_conv_kernel() == 3, "the call should be served by conv1x1.hip"
    if B <= 3:
        _, u8 = _oracle_layer(x, wq, zp_w, 1, 23, bias, mv, sv, 128, oracle.ACT[act], store, oracle.ACC_EXACT)
        assert np.array_equal(got["u8"].reshape(B, n, H * W), u8)
        assert np.array_equal(got["f32"], oracle.dequant(u8, 128, np.float32(0.05)))
    binding.shim().mi355_debug_flags(8192)
    try:
        rows = binding.conv_forward(*args, want_acc=False, want_f32=True)
    finally:
        binding.shim().mi355_debug_flags(0)
    assert np.array_equal(got["u8"], rows["u8"])
    assert np.array_equal(got["f32"], rows["f32"])


@pytest.mark.parametrize("B,c,n,H,W,act", [(64, 128, 256, 26, 26, "leaky"),    # layer 8 of yolov3-tiny: whole K per wave, 8 filter quads
                                           (64, 256, 512, 13, 13, "leaky"),    # layers 10 / 14: two K parts, one image per tile
                                           (30, 128, 64, 26, 26, "relu6"),     # 2 quads x 4 wave sets, 3 groups: one set idle
                                           (60, 256, 64, 20, 17, "linear"),    # 2 quads x 2 K parts x 2 sets, ragged tiles across images
                                           (40, 128, 512, 26, 26, "leaky"),    # 2 filter tiles, 7 groups
                                           (120, 256, 256, 13, 13, "relu"),    # odd group count: K part 0 owns one more
                                           (16, 256, 256, 38, 38, "leaky"),    # persistent workgroups: several tiles each (40 cells in a 64-slot row image)
                                           (12, 128, 256, 76, 76, "leaky"),    # whole K per wave, several tiles per workgroup (map wider than 62)
                                           (40, 256, 512, 19, 19, "relu6")])   # tiles that straddle images, ragged last tile
@pytest.mark.parametrize("store", [binding.STORE_WRAP, binding.STORE_SATURATE], ids=["wrap", "saturate"])
def test_conv_ws3_weights_stationary_kernel(B, c, n, H, W, act, store):
    """The 3x3 kernel for the middle of the net (conv_ws3.hip: 36 K-steps of weights per wave in registers, K parts
    chained through LDS, one tile of consecutive pixels per workgroup) against the oracle on the first / last images and
    against the row-image kernel on every byte (debug flag 16384 routes the same call there)."""
    rng = np.random.default_rng(B + c + n + H + W)
    x = rng.integers(0, 256, (B, c, H, W), dtype=np.uint8)
    wq, zp_w, bias, mv, sv = _rand_layer(rng, n, c, 3, 2.0 ** -13, 2.0 ** -9)
    xt = binding.DevTensor.from_nchw(x, 23)
    args = (xt, wq, zp_w, 3, bias, mv, sv, 23, 23, 1.0, binding.ACT[act], store, binding.ACC_EXACT)
    got = binding.conv_forward(*args, want_acc=False)
    assert binding.shim().mi355_last_conv_kernel() == 4, "the call should be served by conv_ws3.hip"
    sel = [0, B // 2, B - 1]
    _, u8 = _oracle_layer(x[sel], wq, zp_w, 3, 23, bias, mv, sv, 23, oracle.ACT[act], store, oracle.ACC_EXACT)
    assert np.array_equal(got["u8"][sel].reshape(3, n, H * W), u8)
    binding.shim().mi355_debug_flags(16384)
    try:
        rows = binding.conv_forward(*args, want_acc=False)
        assert binding.shim().mi355_last_conv_kernel() == 5
    finally:
        binding.shim().mi355_debug_flags(0)
    assert np.array_equal(got["u8"], rows["u8"])


def _fuzz_cases():
    rng = np.random.default_rng(20260928)
    cases = []
    for i in range(36):
        k = int(rng.choice([1, 3, 3]))
        c = int(rng.choice([16, 32, 48, 64, 128, 192, 256, 384]))
        n = int(rng.integers(1, 9)) * int(rng.choice([1, 4, 16, 32]))
        H, W = int(rng.integers(1, 40)), int(rng.integers(1, 40))
        B = int(rng.integers(1, 5))
        stride = 2 if (k == 3 and rng.random() < 0.25 and H > 2 and W > 2) else 1
        while n * c * k * k * B * H * W > 6e8:  # keep the oracle fast
            H, W = max(1, H // 2), max(1, W // 2)
        cases.append((B, c, n, H, W, k, stride, str(rng.choice(["leaky", "relu6", "linear", "relu"])),
                      int(rng.integers(0, 256)), int(rng.integers(0, 256)), int(rng.integers(0, 2)), i))
    return cases


@pytest.mark.parametrize("case", _fuzz_cases(), ids=lambda c: "f%d_B%d_c%d_n%d_%dx%d_k%d_s%d_%s" % (c[-1], *c[:8]))
def test_conv_random_shapes_vs_oracle(case):
    """Seeded random shapes over the whole supported domain (1x1 / 3x3, stride 1 / 2, any map size down to 1x1, ragged
    channel counts, every activation, arbitrary zero points, both store modes): accumulators, bytes and the float tail
    against the oracle through the dump path, bytes again through the throughput path."""
    B, c, n, H, W, k, stride, act, zp_in, zp_act, sat, seed = case
    store = binding.STORE_SATURATE if sat else binding.STORE_WRAP
    rng = np.random.default_rng(1000 + seed)
    x = rng.integers(0, 256, (B, c, H, W), dtype=np.uint8)
    wq, zp_w, bias, mv, sv = _rand_layer(rng, n, c, k)
    zp_w[rng.integers(0, n)] = rng.choice([0, 255])  # extreme weight zero points: dz = 128 and -127
    xt = binding.DevTensor.from_nchw(x, zp_in)
    args = (xt, wq, zp_w, k, bias, mv, sv, zp_in, zp_act, 0.03, binding.ACT[act], store, binding.ACC_EXACT)
    got = binding.conv_forward(*args, want_acc=True, want_f32=True, stride=stride)
    acc, u8 = _oracle_layer(x, wq, zp_w, k, zp_in, bias, mv, sv, zp_act, oracle.ACT[act], store, oracle.ACC_EXACT, stride=stride)
    OH, OW = (H + 2 * (k // 2) - k) // stride + 1, (W + 2 * (k // 2) - k) // stride + 1
    assert np.array_equal(got["int32"], acc)
    assert np.array_equal(got["u8"].reshape(B, n, OH * OW), u8)
    assert np.array_equal(got["f32"], oracle.dequant(u8, zp_act, np.float32(0.03)))
    fast = binding.conv_forward(*args, want_acc=False, want_f32=True, stride=stride)
    assert np.array_equal(fast["u8"].reshape(B, n, OH * OW), u8)
    assert np.array_equal(fast["f32"], got["f32"])


@pytest.mark.parametrize("B,c,n,H,W,act", [(16, 128, 256, 76, 76, "leaky"),   # one tile per workgroup
                                           (8, 256, 512, 76, 76, "leaky"),    # two K parts, persistent workgroups (YOLOv3's 256->512 s2)
                                           (40, 128, 128, 52, 36, "relu6"),   # non-square, ragged tiles, two wave sets
                                           (6, 128, 256, 152, 152, "linear")])       # 154-cell rows: tiles of whole output rows
def test_conv_ws3_stride2(B, c, n, H, W, act):
    """Stride-2 3x3 convolutions with 128 / 256 input channels in the weights-stationary kernel (output pixel (y, x) reads
    the input around (2y, 2x): only the pixel -> image-cell tables differ from stride 1): oracle on two images, every
    byte against the generic implicit GEMM (debug flag 16384)."""
    rng = np.random.default_rng(B + c + n + H + W)
    x = rng.integers(0, 256, (B, c, H, W), dtype=np.uint8)
    wq, zp_w, bias, mv, sv = _rand_layer(rng, n, c, 3, 2.0 ** -13, 2.0 ** -9)
    xt = binding.DevTensor.from_nchw(x, 23)
    args = (xt, wq, zp_w, 3, bias, mv, sv, 23, 23, 1.0, binding.ACT[act], binding.STORE_WRAP, binding.ACC_EXACT)
    got = binding.conv_forward(*args, want_acc=False, stride=2)
    assert binding.shim().mi355_last_conv_kernel() == 4, "the call should be served by conv_ws3.hip"
    sel = [0, B - 1]
    _, u8 = _oracle_layer(x[sel], wq, zp_w, 3, 23, bias, mv, sv, 23, oracle.ACT[act], oracle.STORE_WRAP, oracle.ACC_EXACT, stride=2)
    assert np.array_equal(got["u8"][sel].reshape(2, n, (H // 2) * (W // 2)), u8)
    binding.shim().mi355_debug_flags(16384)
    try:
        gen = binding.conv_forward(*args, want_acc=False, stride=2)
        assert binding.shim().mi355_last_conv_kernel() == 5
    finally:
        binding.shim().mi355_debug_flags(0)
    assert np.array_equal(got["u8"], gen["u8"])


@pytest.mark.parametrize("bm,bn,nt", [(128, 256, 0), (128, 128, 0), (64, 256, 0), (64, 128, 0), (32, 256, 0), (32, 128, 0),
                                      (128, 384, 0), (128, 384, 3), (128, 384, 7), (128, 256, 5), (64, 128, 13)])
def test_conv_every_tile_config(bm, bn, nt):
    """Force each compiled tile configuration (and, for the row-image kernel, uneven N-tile counts: tiles narrower
    than their capacity leave whole 32-column sub-tiles idle) on a shape it supports."""
    n = {128: 256, 64: 64, 32: 32}[bm]
    B, c, H, W, k = 2, 64, 20, 26, 3   # W >= 24: the widest row-image tiles refuse narrower maps
    bm |= nt << 16
    rng = np.random.default_rng((bm & 0xFFFF) * 7 + bn + nt)
    x = rng.integers(0, 256, (B, c, H, W), dtype=np.uint8)
    wq, zp_w, bias, mv, sv = _rand_layer(rng, n, c, k)
    xt = binding.DevTensor.from_nchw(x, 23)
    binding.shim().mi355_conv_set_tile(bm, bn)
    try:
        got = binding.conv_forward(xt, wq, zp_w, k, bias, mv, sv, 23, 23, 1.0, binding.ACT["leaky"])
    finally:
        binding.shim().mi355_conv_set_tile(0, 0)
    acc, u8 = _oracle_layer(x, wq, zp_w, k, 23, bias, mv, sv, 23, oracle.LEAKY, oracle.STORE_WRAP, oracle.ACC_EXACT)
    assert np.array_equal(got["int32"], acc)
    assert np.array_equal(got["u8"].reshape(B, n, H * W), u8)


@pytest.mark.parametrize("n", [16, 32])
def test_conv_first_layer_kernel(n):
    rng = np.random.default_rng(n)
    B, H, W = 2, 33, 47
    x = rng.integers(0, 256, (B, 3, H, W), dtype=np.uint8)
    wq, zp_w, bias, mv, sv = _rand_layer(rng, n, 3, 3, 2.0 ** -9, 2.0 ** -6)
    xt = binding.DevTensor.from_nchw(x, 5)
    assert xt.t.cs == 4
    got = binding.conv_forward(xt, wq, zp_w, 3, bias, mv, sv, 5, 23, 1.0, binding.ACT["leaky"])
    acc, u8 = _oracle_layer(x, wq, zp_w, 3, 5, bias, mv, sv, 23, oracle.LEAKY, oracle.STORE_WRAP, oracle.ACC_EXACT)
    assert np.array_equal(got["int32"], acc)
    assert np.array_equal(got["u8"].reshape(B, n, H * W), u8)


@pytest.mark.parametrize("n,act", [(16, "leaky"), (32, "leaky"), (32, "relu6"), (16, "linear")])
@pytest.mark.parametrize("store", [binding.STORE_WRAP, binding.STORE_SATURATE], ids=["wrap", "saturate"])
def test_conv_first_layer_mfma_without_pool(n, act, store):
    """The first layer of the non-tiny nets (3 -> 32 at full resolution, no maxpool behind it) on the matrix pipe
    (conv_first_mfma_kernel): even maps, ragged patches, weight zero points 0 / 255 (dz = 128 / -127), against the
    oracle and against the VALU kernel (debug flag 1024)."""
    rng = np.random.default_rng(n + len(act))
    B, H, W = 3, 38, 70
    x = rng.integers(0, 256, (B, 3, H, W), dtype=np.uint8)
    wq, zp_w, bias, mv, sv = _rand_layer(rng, n, 3, 3, 2.0 ** -9, 2.0 ** -6)
    zp_w[1], zp_w[n - 2] = 0, 255
    xt = binding.DevTensor.from_nchw(x, 9)
    args = (xt, wq, zp_w, 3, bias, mv, sv, 9, 23, 1.0, binding.ACT[act], store, binding.ACC_EXACT)
    got = binding.conv_forward(*args, want_acc=False)
    _, u8 = _oracle_layer(x, wq, zp_w, 3, 9, bias, mv, sv, 23, oracle.ACT[act], store, oracle.ACC_EXACT)
    assert np.array_equal(got["u8"].reshape(B, n, H * W), u8)
    binding.shim().mi355_debug_flags(1024)
    try:
        valu = binding.conv_forward(*args, want_acc=False)
    finally:
        binding.shim().mi355_debug_flags(0)
    assert np.array_equal(got["u8"], valu["u8"])


def test_conv_ref_f32_mode_reproduces_fp32_rounding():
    """accum_mode REF_F32 == the oracle's bit-faithful restatement of src/gemm.c:279-299, on data whose running
    sums exceed 2^24 (so it differs from exact integers)."""
    rng = np.random.default_rng(5)
    B, c, n, H, W, k = 1, 256, 32, 6, 6, 3
    x = rng.integers(150, 256, (B, c, H, W), dtype=np.uint8)
    wq, zp_w, bias, mv, sv = _rand_layer(rng, n, c, k)
    wq = np.maximum(wq, 160)
    xt = binding.DevTensor.from_nchw(x, 23)
    got = binding.conv_forward(xt, wq, zp_w, k, bias, mv, sv, 23, 23, 1.0, binding.ACT["leaky"],
                               accum=binding.ACC_REF_F32)
    acc, u8 = _oracle_layer(x, wq, zp_w, k, 23, bias, mv, sv, 23, oracle.LEAKY, oracle.STORE_WRAP, oracle.ACC_REF_F32)
    exact, _ = _oracle_layer(x, wq, zp_w, k, 23, bias, mv, sv, 23, oracle.LEAKY, oracle.STORE_WRAP, oracle.ACC_EXACT)
    assert (acc != exact).any(), "test data must enter the fp32-rounding regime"
    assert np.array_equal(got["int32"], acc)
    assert np.array_equal(got["u8"].reshape(B, n, H * W), u8)


@pytest.mark.parametrize("size,stride,H,W", [(2, 2, 12, 12), (2, 2, 13, 9), (2, 1, 13, 13), (3, 2, 11, 14)])
def test_maxpool(size, stride, H, W):
    rng = np.random.default_rng(size * 10 + stride)
    B, c = 3, 32
    pad = size - 1
    x = rng.integers(0, 256, (B, c, H, W), dtype=np.uint8)
    xt = binding.DevTensor.from_nchw(x, 23)
    oh, ow = (H + pad - size) // stride + 1, (W + pad - size) // stride + 1
    y = binding.DevTensor(B, oh, ow, c, 23)
    binding.check(binding.shim().mi355_maxpool_forward(xt.ref(), y.ref(), size, stride, pad, None), "maxpool")
    want = np.stack([oracle.maxpool_u8(x[b], size, stride, pad) for b in range(B)])
    assert np.array_equal(y.to_nchw(), want)


def test_upsample_and_route():
    rng = np.random.default_rng(9)
    B = 2
    a = rng.integers(0, 256, (B, 32, 5, 7), dtype=np.uint8)
    b = rng.integers(0, 256, (B, 48, 10, 14), dtype=np.uint8)
    at, bt = binding.DevTensor.from_nchw(a, 3), binding.DevTensor.from_nchw(b, 3)
    up = binding.DevTensor(B, 10, 14, 32, 3)
    binding.check(binding.shim().mi355_upsample_forward(at.ref(), up.ref(), 2, None), "upsample")
    want_up = np.stack([oracle.upsample_u8(a[i], 2) for i in range(B)])
    assert np.array_equal(up.to_nchw(), want_up)
    out = binding.DevTensor(B, 10, 14, 80, 3)
    import ctypes as C
    arr = (C.POINTER(binding.Tensor) * 2)(C.pointer(up.t), C.pointer(bt.t))
    binding.check(binding.shim().mi355_route_forward(arr, 2, out.ref(), None), "route")
    assert np.array_equal(out.to_nchw(), np.concatenate([want_up, b], axis=1))


def test_layout_roundtrip_and_pads():
    rng = np.random.default_rng(11)
    for shape in [(2, 3, 9, 13), (3, 16, 4, 6), (1, 30, 13, 13)]:
        x = rng.integers(0, 256, shape, dtype=np.uint8)
        t = binding.DevTensor.from_nchw(x, 77)
        assert np.array_equal(t.to_nchw(), x)
        raw = t.buf.to_numpy(np.uint8, t.buf.nbytes).reshape(-1, t.t.cs)
        H, W = shape[2], shape[3]
        # pad cells keep the (biased) zero point
        padcell = raw[t.t.lead + W]  # column W of row 0 block
        if t.t.cs == 4:
            assert list(padcell) == [77, 77, 77, 0]
        else:
            assert (padcell == (77 ^ 0x80)).all()


@pytest.mark.parametrize("c,n,H,W,act", [(3, 16, 20, 36, "leaky"), (16, 32, 24, 40, "leaky"), (32, 64, 16, 16, "relu6"),
                                         (48, 32, 10, 34, "linear"), (3, 32, 34, 70, "relu6"), (3, 16, 18, 30, "linear")])
@pytest.mark.parametrize("store", [binding.STORE_WRAP, binding.STORE_SATURATE], ids=["wrap", "saturate"])
@pytest.mark.parametrize("ept", [False, True], ids=["derive-in-kernel", "epilogue-table"])
def test_conv_fused_maxpool_equals_conv_then_pool(c, n, H, W, act, store, ept):
    """mi355_conv_pool_forward == conv + requant + 2x2/2 maxpool of the oracle (pre-pool tensor too), including
    wrap-on-store values inside pooling windows (max is taken AFTER the uint8 wrap, as the reference does).
    ept: the blob carries the host-derived epilogue table (mi355_conv_pack_epilogue) or the workgroups derive it."""
    import ctypes as C
    rng = np.random.default_rng(c * 100 + n + H)
    B = 3
    x = rng.integers(0, 256, (B, c, H, W), dtype=np.uint8)
    wq, zp_w, bias, mv, sv = _rand_layer(rng, n, c, 3, 2.0 ** -9 if c == 3 else 2.0 ** -11, 2.0 ** -6 if c == 3 else 2.0 ** -7)
    zp_in, zp_act = 9, 23
    xt = binding.DevTensor.from_nchw(x, zp_in)
    blob = binding.DevBuf.from_numpy(binding.conv_pack(wq, zp_w, c, 3, bias, mv, sv, *((binding.ACT[act], zp_act) if ept else ())))
    y = binding.DevTensor(B, H, W, n, zp_act)
    yp = binding.DevTensor(B, H // 2, W // 2, n, zp_act)
    d = binding.ConvDesc(n, c, 3, 1, 1, binding.ACT[act], store, binding.ACC_EXACT, zp_in, zp_act, 1.0)
    binding.check(binding.shim().mi355_conv_pool_forward(C.byref(d), xt.ref(), blob.ptr, y.ref(), yp.ref(), None), "conv_pool")
    acc, u8 = _oracle_layer(x, wq, zp_w, 3, zp_in, bias, mv, sv, zp_act, oracle.ACT[act], store, oracle.ACC_EXACT)
    u8 = u8.reshape(B, n, H, W)
    want_pool = np.stack([oracle.maxpool_u8(u8[b], 2, 2, 1) for b in range(B)])
    assert np.array_equal(y.to_nchw(), u8), "pre-pool tensor"
    assert np.array_equal(yp.to_nchw(), want_pool), "pooled tensor"
    # and without the pre-pool store
    yp2 = binding.DevTensor(B, H // 2, W // 2, n, zp_act)
    binding.check(binding.shim().mi355_conv_pool_forward(C.byref(d), xt.ref(), blob.ptr, None, yp2.ref(), None), "conv_pool")
    assert np.array_equal(yp2.to_nchw(), want_pool)


@pytest.mark.parametrize("c,n,H,W,act", [(16, 32, 22, 64, "leaky"), (32, 64, 14, 70, "leaky"), (16, 64, 12, 62, "relu6"),
                                         (32, 32, 18, 66, "linear"), (16, 32, 208, 208, "leaky"),
                                         (64, 128, 52, 52, "leaky"), (64, 96, 14, 30, "relu6"), (64, 64, 20, 132, "linear"),
                                         (64, 128, 6, 6, "leaky")])
@pytest.mark.parametrize("store", [binding.STORE_WRAP, binding.STORE_SATURATE], ids=["wrap", "saturate"])
@pytest.mark.parametrize("gain", ["no-wrap", "some-wrap", "much-wrap"])
@pytest.mark.parametrize("ept", [False, True], ids=["derive-in-kernel", "epilogue-table"])
def test_conv_small_channel_pool_kernel(c, n, H, W, act, store, gain, ept):
    """The weights-stationary few-channel kernels (conv_small.hip: 3x3, c 16|32 with n 32|64, c 64 with n 64..128 split
    over the waves; pooled output only).  It requantises only the maximum accumulator of a 2x2 window when no accumulator of the window can
    wrap on store, and falls back to the reference's order (wrap, then max) per wave otherwise: all three regimes --
    no wave wraps, a few do, most do -- must give the oracle's conv -> requant -> maxpool bytes.  Tiles are runs of 128
    pooled pixels that cross rows and images (B = 3, odd pooled widths)."""
    import ctypes as C
    rng = np.random.default_rng(c * 1000 + n * 10 + H + len(gain))
    B = 3 if H < 100 else 1
    x = rng.integers(0, 256, (B, c, H, W), dtype=np.uint8)
    lo, hi = {"no-wrap": (2.0 ** -17, 2.0 ** -16), "some-wrap": (2.0 ** -14, 2.0 ** -12), "much-wrap": (2.0 ** -11, 2.0 ** -7)}[gain]
    if c == 64:  # accumulators grow with K: keep the three regimes where they are
        lo, hi = lo / 4, hi / 4
    wq, zp_w, bias, mv, sv = _rand_layer(rng, n, c, 3, lo, hi)
    zp_w[0], zp_w[1], zp_w[2] = 0, 255, 1   # 128 - zp_w = 128 does not fit the int8 operand of the correction MFMA (c = 16)
    if gain != "much-wrap":
        bias = (bias // 16).astype(np.int32)
    zp_in, zp_act = 9, (23 if act != "linear" else 128)
    xt = binding.DevTensor.from_nchw(x, zp_in)
    # (ept with a zero point other than the launch's: the key does not match and the kernel must derive the constants itself)
    blob = binding.DevBuf.from_numpy(binding.conv_pack(wq, zp_w, c, 3, bias, mv, sv, *((binding.ACT[act], zp_act if n != 96 else zp_act + 1) if ept else ())))
    yp = binding.DevTensor(B, H // 2, W // 2, n, zp_act)
    d = binding.ConvDesc(n, c, 3, 1, 1, binding.ACT[act], store, binding.ACC_EXACT, zp_in, zp_act, 1.0)
    binding.check(binding.shim().mi355_conv_pool_forward(C.byref(d), xt.ref(), blob.ptr, None, yp.ref(), None), "conv_pool")
    acc, u8 = _oracle_layer(x, wq, zp_w, 3, zp_in, bias, mv, sv, zp_act, oracle.ACT[act], store, oracle.ACC_EXACT)
    u8 = u8.reshape(B, n, H, W)
    want_pool = np.stack([oracle.maxpool_u8(u8[b], 2, 2, 1) for b in range(B)])
    sat = np.stack([oracle.requant(acc[b], bias, mv, sv, zp_act, oracle.ACT[act], oracle.STORE_SATURATE) for b in range(B)])
    wrap = np.stack([oracle.requant(acc[b], bias, mv, sv, zp_act, oracle.ACT[act], oracle.STORE_WRAP) for b in range(B)])
    frac = float((sat != wrap).mean())
    if gain == "no-wrap":
        assert frac == 0.0
    elif gain == "much-wrap":
        assert frac > 0.05
    assert np.array_equal(yp.to_nchw(), want_pool), f"pooled tensor (wrap fraction {frac:.4f})"
    # the generic kernel on the same call (debug switch 1024 routes around conv_small.hip) agrees as well
    binding.shim().mi355_debug_flags(1024)
    try:
        yp2 = binding.DevTensor(B, H // 2, W // 2, n, zp_act)
        binding.check(binding.shim().mi355_conv_pool_forward(C.byref(d), xt.ref(), blob.ptr, None, yp2.ref(), None), "conv_pool")
    finally:
        binding.shim().mi355_debug_flags(0)
    assert np.array_equal(yp2.to_nchw(), want_pool)
    if c == 64:
        # round 6: under the throughput plan the 64-channel kernel runs as HALF workgroups (one wave set, tiles of whole pixel groups, unpadded
        # LDS pitch: conv_small.hip conv_small_pool_launch) -- same bytes
        d.plan = 1
        yp3 = binding.DevTensor(B, H // 2, W // 2, n, zp_act)
        binding.check(binding.shim().mi355_conv_pool_forward(C.byref(d), xt.ref(), blob.ptr, None, yp3.ref(), None), "conv_pool (throughput plan)")
        assert binding.shim().mi355_last_conv_kernel() == 2
        assert np.array_equal(yp3.to_nchw(), want_pool), "throughput plan (half workgroups)"


@pytest.mark.parametrize("B,H,W,act", [(3, 14, 70, "leaky"), (2, 34, 34, "linear"), (1, 104, 104, "leaky"), (5, 6, 10, "relu6"), (1, 40, 300, "leaky")])
@pytest.mark.parametrize("store", [binding.STORE_WRAP, binding.STORE_SATURATE], ids=["wrap", "saturate"])
@pytest.mark.parametrize("gain", ["no-wrap", "some-wrap", "much-wrap"])
@pytest.mark.parametrize("table", ["none", "match", "shift-not-pow2"])
def test_conv_small32_kernel(B, H, W, act, store, gain, table):
    """conv_small32.hip (round 6: 32 -> 64 channels + maxpool, conv_small.hip's tile on eight waves of one m-tile each, the 2x2 window in
    two halves): the oracle's conv -> requant -> maxpool bytes in the three wrap regimes -- a half that fails its range test leaves a BYTE
    maximum behind, a clean one an accumulator maximum, and all four combinations of the two halves must agree with the reference's order --
    on flat tiles that cross rows and images, 8 x 16 patches (wide maps), ragged ends; with the constants derived in the kernel, from the
    packed table, from a table made for another zero point (key mismatch), and with shifts that are no powers of two (two-step form, every
    value); `no-wrap` multipliers are small enough that the integer requantisation does not qualify (the FP64-of-maximum fast path, which conv_small.hip's
    32- / 64-channel kernels take as well since round 6 instead of the exact path for every window).  Debug bit 4096 selects the kernel (it measured
    slower than conv_small.hip and is not in the default path); the same call without the bit gives the same bytes."""
    import ctypes as C
    c, n = 32, 64
    rng = np.random.default_rng(H * 1000 + W + len(gain) + len(table))
    x = rng.integers(0, 256, (B, c, H, W), dtype=np.uint8)
    lo, hi = {"no-wrap": (2.0 ** -17, 2.0 ** -16), "some-wrap": (2.0 ** -14, 2.0 ** -12), "much-wrap": (2.0 ** -11, 2.0 ** -7)}[gain]
    wq, zp_w, bias, mv, sv = _rand_layer(rng, n, c, 3, lo, hi)
    zp_w[0], zp_w[1], zp_w[2] = 0, 255, 1
    if gain != "much-wrap":
        bias = (bias // 16).astype(np.int32)
    if table == "shift-not-pow2":
        sv = sv * 0.75
    zp_in, zp_act = 9, (23 if act != "linear" else 128)
    xt = binding.DevTensor.from_nchw(x, zp_in)
    ept = () if table == "none" else (binding.ACT[act], zp_act + (1 if (table == "match" and H == 34) else 0))  # (34 x 34: a table made for another zero point)
    blob = binding.DevBuf.from_numpy(binding.conv_pack(wq, zp_w, c, 3, bias, mv, sv, *ept))
    acc, u8 = _oracle_layer(x, wq, zp_w, 3, zp_in, bias, mv, sv, zp_act, oracle.ACT[act], store, oracle.ACC_EXACT)
    want_pool = np.stack([oracle.maxpool_u8(u8.reshape(B, n, H, W)[b], 2, 2, 1) for b in range(B)])
    d = binding.ConvDesc(n, c, 3, 1, 1, binding.ACT[act], store, binding.ACC_EXACT, zp_in, zp_act, 1.0)
    for flags, kernel in ((4096, 8), (0, 2)):
        yp = binding.DevTensor(B, H // 2, W // 2, n, zp_act)
        binding.shim().mi355_debug_flags(flags)
        try:
            binding.check(binding.shim().mi355_conv_pool_forward(C.byref(d), xt.ref(), blob.ptr, None, yp.ref(), None), "conv_pool")
            assert binding.shim().mi355_last_conv_kernel() == kernel
        finally:
            binding.shim().mi355_debug_flags(0)
        got = yp.to_nchw()
        assert np.array_equal(got, want_pool), f"kernel {kernel}: {int((got != want_pool).sum())} of {got.size} pooled bytes differ"
        raw = yp.buf.to_numpy(np.uint8, yp.buf.nbytes).reshape(-1, yp.t.cs)
        OW = W // 2
        assert (raw[yp.t.lead + OW] == (zp_act ^ 0x80)).all() and (raw[yp.t.lead + (OW + 1) + OW] == (zp_act ^ 0x80)).all()  # pad cells untouched


@pytest.mark.parametrize("B,c,n,H,W,act", [(3, 16, 32, 22, 64, "leaky"), (3, 32, 64, 14, 70, "leaky"), (2, 16, 32, 18, 30, "relu6"),
                                           (2, 32, 64, 34, 34, "linear"), (1, 16, 32, 208, 208, "leaky"), (1, 32, 64, 104, 104, "leaky"),
                                           (5, 16, 32, 2, 2, "leaky"), (2, 32, 64, 16, 32, "relu6"), (1, 16, 32, 40, 300, "linear")])
@pytest.mark.parametrize("store", [binding.STORE_WRAP, binding.STORE_SATURATE], ids=["wrap", "saturate"])
@pytest.mark.parametrize("gain", ["no-wrap", "some-wrap", "much-wrap"])
@pytest.mark.parametrize("table", ["match", "other-zero-point", "shift-not-pow2"])
def test_conv_pool16_kernel(B, c, n, H, W, act, store, gain, table):
    """conv_pool16.hip (16 -> 32 channels + maxpool on 16 x 16 x 64 tiles, chosen when the desc says the blob carries the
    epilogue table; the 32 -> 64 cases check that the hint changes nothing for shapes outside its domain): the oracle's conv -> requant -> maxpool bytes in the three wrap regimes (margin test + per-wave redo in the
    reference's order), ragged 8 x 16 pooled patches, maps smaller than a patch, weight zero points 0 / 255; a table made for another zero
    point (key mismatch: every window takes the exact path); the same call without the hint is served by conv_small.hip with the same bytes."""
    import ctypes as C
    rng = np.random.default_rng(c * 1000 + n * 10 + H + W + len(gain))
    x = rng.integers(0, 256, (B, c, H, W), dtype=np.uint8)
    lo, hi = {"no-wrap": (2.0 ** -17, 2.0 ** -16), "some-wrap": (2.0 ** -14, 2.0 ** -12), "much-wrap": (2.0 ** -11, 2.0 ** -7)}[gain]
    wq, zp_w, bias, mv, sv = _rand_layer(rng, n, c, 3, lo, hi)
    zp_w[0], zp_w[1], zp_w[2] = 0, 255, 1
    if gain != "much-wrap":
        bias = (bias // 16).astype(np.int32)
    if table == "shift-not-pow2":  # the reference's two-step form has no one-multiplier equivalent: the exact path, every window
        sv = sv * 0.75
    zp_in, zp_act = 9, (23 if act != "linear" else 128)
    xt = binding.DevTensor.from_nchw(x, zp_in)
    blob = binding.DevBuf.from_numpy(binding.conv_pack(wq, zp_w, c, 3, bias, mv, sv, binding.ACT[act], zp_act + (table == "other-zero-point")))
    acc, u8 = _oracle_layer(x, wq, zp_w, 3, zp_in, bias, mv, sv, zp_act, oracle.ACT[act], store, oracle.ACC_EXACT)
    want_pool = np.stack([oracle.maxpool_u8(u8.reshape(B, n, H, W)[b], 2, 2, 1) for b in range(B)])
    for hint, kernel in ((1, 7), (0, 2)):
        yp = binding.DevTensor(B, H // 2, W // 2, n, zp_act)
        d = binding.ConvDesc(n, c, 3, 1, 1, binding.ACT[act], store, binding.ACC_EXACT, zp_in, zp_act, 1.0)
        d.epilogue_packed = hint
        binding.check(binding.shim().mi355_conv_pool_forward(C.byref(d), xt.ref(), blob.ptr, None, yp.ref(), None), "conv_pool")
        want_kernel = 7 if hint and c == 16 else (2 if min(H, W) >= 4 else 5)  # (32 -> 64 stays with conv_small.hip, which leaves 2 x 2 maps to the generic kernel)
        assert binding.shim().mi355_last_conv_kernel() == want_kernel
        got = yp.to_nchw()
        assert np.array_equal(got, want_pool), f"hint {hint}: {int((got != want_pool).sum())} of {got.size} pooled bytes differ"
        # pad cells of the pooled tensor are untouched (ragged patches mask their stores)
        raw = yp.buf.to_numpy(np.uint8, yp.buf.nbytes).reshape(-1, yp.t.cs)
        OW = W // 2
        assert (raw[yp.t.lead + OW] == (zp_act ^ 0x80)).all() and (raw[yp.t.lead + (OW + 1) + OW] == (zp_act ^ 0x80)).all()


@pytest.mark.parametrize("B,c,n,H,W,act", [(2, 32, 64, 76, 76, "leaky"),      # flat tiles of 128 window blocks
                                           (1, 32, 64, 152, 152, "leaky"),    # 8 x 16 patches (wide map), YOLOv3's 32->64 layer shape
                                           (1, 16, 32, 40, 300, "relu6"),     # very wide rows, ragged patches
                                           (3, 16, 64, 30, 34, "linear"),     # two m-tiles per wave
                                           (2, 32, 32, 18, 22, "leaky"),
                                           (1, 64, 128, 152, 152, "leaky"),   # 64 channels (conv_mid): YOLOv3's 64->128 layer shape
                                           (3, 64, 64, 26, 38, "relu6"), (2, 64, 96, 52, 52, "linear")])
@pytest.mark.parametrize("store", [binding.STORE_WRAP, binding.STORE_SATURATE], ids=["wrap", "saturate"])
def test_conv_small_channel_kernel_without_pool(B, c, n, H, W, act, store):
    """The 16 / 32-channel weights-stationary kernel without a pool behind it (conv_small.hip, POOL = false: the four
    window positions of a lane are four output pixels): bytes against the oracle and against the generic implicit GEMM
    (debug flag 1024), extreme weight zero points included."""
    rng = np.random.default_rng(B + c + n + H + W)
    x = rng.integers(0, 256, (B, c, H, W), dtype=np.uint8)
    wq, zp_w, bias, mv, sv = _rand_layer(rng, n, c, 3)
    zp_w[0], zp_w[n - 1] = 0, 255
    xt = binding.DevTensor.from_nchw(x, 23)
    args = (xt, wq, zp_w, 3, bias, mv, sv, 23, 23, 1.0, binding.ACT[act], store, binding.ACC_EXACT)
    got = binding.conv_forward(*args, want_acc=False)
    assert binding.shim().mi355_last_conv_kernel() == 2, "the call should be served by conv_small.hip"
    _, u8 = _oracle_layer(x, wq, zp_w, 3, 23, bias, mv, sv, 23, oracle.ACT[act], store, oracle.ACC_EXACT)
    assert np.array_equal(got["u8"].reshape(B, n, H * W), u8)
    binding.shim().mi355_debug_flags(1024)
    try:
        gen = binding.conv_forward(*args, want_acc=False)
        assert binding.shim().mi355_last_conv_kernel() == 5
    finally:
        binding.shim().mi355_debug_flags(0)
    assert np.array_equal(got["u8"], gen["u8"])


@pytest.mark.parametrize("gain", ["no-wrap", "much-wrap"])
def test_first_layer_mfma_pool_extreme_weight_zero_points(gain):
    """First-layer MFMA kernel: weight zero points 0 and 255 (128 - zp_w = 128 does not fit the int8 operand of the
    correction MFMA and is split in two), every batch image, ragged 8x16 pooled patches."""
    import ctypes as C
    rng = np.random.default_rng(5 + len(gain))
    B, c, n, H, W = 3, 3, 16, 38, 46
    x = rng.integers(0, 256, (B, c, H, W), dtype=np.uint8)
    lo, hi = (2.0 ** -12, 2.0 ** -11) if gain == "no-wrap" else (2.0 ** -8, 2.0 ** -5)
    wq, zp_w, bias, mv, sv = _rand_layer(rng, n, c, 3, lo, hi)
    zp_w[0], zp_w[1], zp_w[2], zp_w[3] = 0, 255, 1, 128
    if gain == "no-wrap":
        bias = (bias // 64).astype(np.int32)
    xt = binding.DevTensor.from_nchw(x, 3)
    blob = binding.DevBuf.from_numpy(binding.conv_pack(wq, zp_w, c, 3, bias, mv, sv))
    for store in (binding.STORE_WRAP, binding.STORE_SATURATE):
        yp = binding.DevTensor(B, H // 2, W // 2, n, 23)
        d = binding.ConvDesc(n, c, 3, 1, 1, binding.ACT["leaky"], store, binding.ACC_EXACT, 3, 23, 1.0)
        binding.check(binding.shim().mi355_conv_pool_forward(C.byref(d), xt.ref(), blob.ptr, None, yp.ref(), None), "conv_pool")
        acc, u8 = _oracle_layer(x, wq, zp_w, 3, 3, bias, mv, sv, 23, oracle.LEAKY, store, oracle.ACC_EXACT)
        want = np.stack([oracle.maxpool_u8(u8.reshape(B, n, H, W)[b], 2, 2, 1) for b in range(B)])
        assert np.array_equal(yp.to_nchw(), want), store


def test_conv_upsample_and_conv_yolo_entry_points():
    """The two fused C-ABI entry points against their unfused definitions: mi355_conv_upsample_forward == conv_forward +
    upsample_forward (also into a channel window of a wider tensor, as the host's concat elimination uses it), and
    mi355_conv_yolo_forward == conv_forward(y_f32) + yolo_forward, bit for bit."""
    import ctypes as C
    S = binding.shim()
    rng = np.random.default_rng(77)
    # ---- conv 1x1 256 -> 128 on 13x13, stored 2x upsampled into channels [0, 128) of a 384-channel tensor
    B, c, n, H, W = 3, 256, 128, 13, 13
    x = rng.integers(0, 256, (B, c, H, W), dtype=np.uint8)
    wq, zp_w, bias, mv, sv = _rand_layer(rng, n, c, 1)
    xt = binding.DevTensor.from_nchw(x, 23)
    blob = binding.DevBuf.from_numpy(binding.conv_pack(wq, zp_w, c, 1, bias, mv, sv))
    d = binding.ConvDesc(n, c, 1, 1, 0, binding.ACT["leaky"], binding.STORE_WRAP, binding.ACC_EXACT, 23, 23, 1.0)
    _, u8 = _oracle_layer(x, wq, zp_w, 1, 23, bias, mv, sv, 23, oracle.LEAKY, oracle.STORE_WRAP, oracle.ACC_EXACT)
    want = np.stack([oracle.upsample_u8(u8.reshape(B, n, H, W)[b], 2) for b in range(B)])
    wide = binding.DevTensor(B, 2 * H, 2 * W, 384, 23)
    win = binding.Tensor(wide.t.data, B, 2 * H, 2 * W, n, wide.t.cs, wide.t.lead, wide.t.tail)  # channels 0..127 of `wide`
    binding.check(S.mi355_conv_upsample_forward(C.byref(d), xt.ref(), blob.ptr, C.byref(win), 2, None), "conv_upsample")
    binding.check(S.mi355_stream_sync(None), "sync")
    got = wide.to_nchw()
    assert np.array_equal(got[:, :n], want)
    assert (got[:, n:] == 23).all(), "the other channels of the wide tensor keep their fill"
    # ---- head: conv 1x1 512 -> 30 (3 anchors x (5 classes + 5)) + yolo
    c, n, classes = 512, 30, 5
    x = rng.integers(0, 256, (B, c, H, W), dtype=np.uint8)
    wq, zp_w, bias, mv, sv = _rand_layer(rng, n, c, 1, 2.0 ** -13, 2.0 ** -11)
    xt = binding.DevTensor.from_nchw(x, 23)
    blob = binding.DevBuf.from_numpy(binding.conv_pack(wq, zp_w, c, 1, bias, mv, sv))
    d = binding.ConvDesc(n, c, 1, 1, 0, binding.ACT["linear"], binding.STORE_WRAP, binding.ACC_EXACT, 23, 128, 0.0625)
    cnt = B * n * H * W
    y = binding.DevTensor(B, H, W, n, 128)
    f_fused, yo_fused = binding.DevBuf(cnt * 4), binding.DevBuf(cnt * 4)
    binding.check(S.mi355_conv_yolo_forward(C.byref(d), xt.ref(), blob.ptr, y.ref(), f_fused.ptr, yo_fused.ptr, classes, None), "conv_yolo")
    y2 = binding.DevTensor(B, H, W, n, 128)
    f_ref, yo_ref = binding.DevBuf(cnt * 4), binding.DevBuf(cnt * 4)
    binding.check(S.mi355_conv_forward(C.byref(d), xt.ref(), blob.ptr, None, None, y2.ref(), None, f_ref.ptr, None), "conv")
    binding.check(S.mi355_yolo_forward(f_ref.ptr, yo_ref.ptr, B, 3, classes, H, W, None), "yolo")
    binding.check(S.mi355_stream_sync(None), "sync")
    assert np.array_equal(y.to_nchw(), y2.to_nchw())
    assert np.array_equal(f_fused.to_numpy(np.float32, cnt), f_ref.to_numpy(np.float32, cnt))
    assert np.array_equal(yo_fused.to_numpy(np.float32, cnt), yo_ref.to_numpy(np.float32, cnt))
    _, u8 = _oracle_layer(x, wq, zp_w, 1, 23, bias, mv, sv, 128, oracle.LINEAR, oracle.STORE_WRAP, oracle.ACC_EXACT)
    assert np.array_equal(f_ref.to_numpy(np.float32, cnt), oracle.dequant(u8, 128, np.float32(0.0625)).ravel())


def test_fused_maxpool_net_equals_unfused(cfg_dir, tmp_path):
    """Whole yolov3-tiny, batch 3: the throughput configuration (conv+maxpool fused where possible, pre-pool tensors
    not stored, route inputs written straight into the route buffers) yields byte-identical tensors on every layer
    that is stored, compared with the plain configuration (every layer in its own buffer, nothing fused: the parity
    dump mode) and with fusion alone switched off."""
    cfg = os.path.join(cfg_dir, "yolov3-tiny_quant.cfg")
    wts = str(tmp_path / "w.weights")
    synth.synth_weights(cfg, wts, seed=1234)
    xs = synth.synth_image_u8(3, 416, 416, seed=21, batch=3)
    outs = {}
    for key, fuse, dump in (("plain", False, True), ("unfused", False, False), ("fast", True, False)):
        net = binding.Net(cfg, wts, batch=3, fuse_maxpool=fuse, dump_int32=dump)
        net.prepare_fixed(1.0 / 255.0, 0)
        net.push_input(xs)
        net.forward(); net.sync()
        outs[key] = [net.pull(i) for i in range(net.n)]
        info = net.info
        fused_flags = [net.is_fused(i) for i in range(net.n)]
        net.close()
    fused_convs = 0
    for i, inf in enumerate(info):
        nxt = info[i + 1] if i + 1 < len(info) else None
        skipped = (inf["type"] == binding.T_CONV and nxt and nxt["type"] == binding.T_MAXPOOL and inf["size"] == 3
                   and nxt["size"] == 2 and nxt["stride"] == 2
                   and (inf["c"] % 64 != 0 or (inf["c"] == 64 and inf["n"] % 32 == 0 and 64 <= inf["n"] <= 128)))
        skipped = bool(skipped or (inf["type"] == binding.T_CONV and nxt and nxt["type"] == binding.T_UPSAMPLE
                                   and inf["c"] % 64 == 0))  # conv + upsample: only the upsampled tensor is stored
        assert skipped == fused_flags[i]
        for k in outs["plain"][i]:
            if k in outs["unfused"][i] and k != "int32":  # accumulators are only dumped in the plain configuration
                assert np.array_equal(outs["unfused"][i][k], outs["plain"][i][k]), (i, k, "unfused vs plain")
        if skipped:
            fused_convs += 1
            continue  # its own tensor is not stored in the fused configuration
        for k in outs["fast"][i]:
            if k != "int32":
                assert np.array_equal(outs["fast"][i][k], outs["plain"][i][k]), (i, k, "fast vs plain")
    assert fused_convs == 5


# ------------------------------------------------------------------------------------------------ whole networks
def _run_host_net(cfg, wts, x_u8_batch, accum, store=binding.STORE_WRAP, graph=False, dump_int32=True):
    B = x_u8_batch.shape[0]
    net = binding.Net(cfg, wts, batch=B, accum=accum, store=store, dump_int32=dump_int32, use_graph=graph)
    # the reference flow: float image -> dynamic layer-0 quantiser (identity on pinned full-range images)
    xq = net.prepare_from_float(synth.image_u8_to_float(x_u8_batch))
    assert np.array_equal(xq, x_u8_batch.ravel())
    net.forward()
    if graph:
        net.forward()
    net.sync()
    outs = [net.pull(i) for i in range(net.n)]
    info = [dict(inf, fused=net.is_fused(i), fuses_next=net.fuses_next(i), kernel=net.conv_kernel(i)) for i, inf in enumerate(net.info)]
    net.close()
    return outs, info


@pytest.mark.parametrize("name,accum", [("tiny_unit", binding.ACC_EXACT), ("tiny_unit", binding.ACC_REF_F32),
                                        ("s2_unit", binding.ACC_EXACT)], ids=["tiny-exact", "tiny-ref_f32", "s2-exact"])
@pytest.mark.parametrize("seed", [1, 2])
def test_tiny_unit_net_vs_reference_golden(golden_dir, cfg_dir, tmp_path, seed, accum, name):
    """Full-tensor equality with the tensors the reference itself produced (tests/golden/{tiny,s2}_unit_seed*.npz),
    through the plain-C darknet host (cfg parser, weights reader, prep, layer.forward_gpu loop).  s2_unit is a chain of
    stride-2 3x3 convolutions (exact mode only: the fp32-emulation kernel is stride 1)."""
    g = np.load(os.path.join(golden_dir, f"{name}_seed{seed}.npz"))
    cfg = os.path.join(cfg_dir, f"{name}.cfg")
    wts = str(tmp_path / "w.weights")
    meta = synth.synth_weights(cfg, wts, seed=seed, act_gain=float(g["act_gain"]))
    assert meta["sha256"] == str(g["weights_sha256"])
    x = g["input_u8"][None]
    outs, info = _run_host_net(cfg, wts, x, accum)
    for i, inf in enumerate(info):
        if inf["type"] == binding.T_CONV:
            assert np.array_equal(outs[i]["int32"], g[f"L{i}_int32"]), f"layer {i} int32"
        if inf["type"] != binding.T_YOLO:
            assert np.array_equal(outs[i]["u8"], g[f"L{i}_u8"]), f"layer {i} u8"
        if inf["quant_stop"]:
            assert np.array_equal(outs[i]["f32"], g[f"L{i}_f32"]), f"layer {i} f32"
        if inf["type"] == binding.T_YOLO:
            np.testing.assert_allclose(outs[i]["f32"], g[f"L{i}_f32"], rtol=0, atol=2e-7)  # float logistic: 1 ulp


@pytest.mark.parametrize("tag", ["leaky", "relu6"])
def test_yolov3_tiny_416_ref_f32_equals_reference_hashes(golden_dir, cfg_dir, tmp_path, tag):
    """BASELINE config[0]: whole yolov3-tiny @416, one image, bit-faithful mode: per-layer SHA-256 of the int32
    accumulators, uint8 activations and float heads equal the reference default build's."""
    g = json.load(open(os.path.join(golden_dir, f"yolov3_tiny_{tag}.json")))
    cfg = os.path.join(cfg_dir, g["cfg"])
    wts = str(tmp_path / "w.weights")
    assert synth.synth_weights(cfg, wts, seed=g["weight_seed"])["sha256"] == g["weights_sha256"]
    x = synth.synth_image_u8(3, 416, 416, seed=g["image_seed"])[None]
    outs, info = _run_host_net(cfg, wts, x, binding.ACC_REF_F32)
    for e in g["layers"]:
        i = e["i"]
        if "int32_sha256" in e:
            assert sha(outs[i]["int32"]) == e["int32_sha256"], f"layer {i} int32"
        if "u8_sha256" in e:
            assert sha(outs[i]["u8"]) == e["u8_sha256"], f"layer {i} u8"
        if "f32_sha256" in e and e["type"] == "conv":
            assert sha(outs[i]["f32"]) == e["f32_sha256"], f"layer {i} f32"


@pytest.mark.parametrize("tag", ["leaky", "relu6"])
def test_yolov3_tiny_416_exact_equals_oracle(golden_dir, cfg_dir, tmp_path, tag):
    """Production mode (exact int32 on MFMA): whole net equals the oracle in exact mode on every tensor, and equals
    the reference's hashes on every layer up to the first one where the reference's fp32 accumulation rounds."""
    g = json.load(open(os.path.join(golden_dir, f"yolov3_tiny_{tag}.json")))
    cfg = os.path.join(cfg_dir, g["cfg"])
    wts = str(tmp_path / "w.weights")
    synth.synth_weights(cfg, wts, seed=g["weight_seed"])
    x = synth.synth_image_u8(3, 416, 416, seed=g["image_seed"])
    outs, info = _run_host_net(cfg, wts, x[None], binding.ACC_EXACT)
    onet = oracle.OracleNet(cfg, wts)
    onet.prepare(np.float32(1.0 / 255.0), 0)
    want = onet.forward(x, accum=oracle.ACC_EXACT)
    first_diverged = None
    for i, inf in enumerate(info):
        if inf["type"] == binding.T_CONV:
            assert np.array_equal(outs[i]["int32"], want[i]["int32"].ravel()), f"layer {i} int32"
        if inf["type"] != binding.T_YOLO:
            assert np.array_equal(outs[i]["u8"], want[i]["u8"].ravel()), f"layer {i} u8"
            if first_diverged is None and sha(outs[i]["u8"]) != g["layers"][i]["u8_sha256"]:
                first_diverged = i
        if inf["quant_stop"]:
            assert np.array_equal(outs[i]["f32"], want[i]["f32"].ravel()), f"layer {i} f32"
    # layers 0..9 have K <= 1152: 1152*255^2 > 2^24 is possible in theory, but the reference stays exact there on
    # this model (tests/test_oracle_golden.py); the first rounding shows up at K >= 2304.
    assert first_diverged is None or first_diverged >= 10


def test_batch_is_per_image_independent(cfg_dir, tmp_path):
    """batch > 1 == the batch-1 function applied to each image: a batch of different images equals the images run
    alone, including ragged N tiles that straddle image boundaries."""
    cfg = os.path.join(cfg_dir, "tiny_unit.cfg")
    wts = str(tmp_path / "w.weights")
    synth.synth_weights(cfg, wts, seed=1)
    xs = synth.synth_image_u8(3, 12, 12, seed=99, batch=5)
    outs, info = _run_host_net(cfg, wts, xs, binding.ACC_EXACT)
    for b in range(5):
        o1, _ = _run_host_net(cfg, wts, xs[b:b + 1], binding.ACC_EXACT)
        for i, inf in enumerate(info):
            per = inf["outputs"]
            if inf["type"] != binding.T_YOLO:
                assert np.array_equal(outs[i]["u8"][b * per:(b + 1) * per], o1[i]["u8"]), (b, i)
            if inf["type"] == binding.T_CONV:
                assert np.array_equal(outs[i]["int32"][b * per:(b + 1) * per], o1[i]["int32"]), (b, i)


def test_yolov3_tiny_batch64_properties(cfg_dir, tmp_path):
    """BASELINE config[2] at full size (batch 64 @416): size-independent properties -- identical images give
    identical outputs in every batch slot (checksum of checksums), image 0 equals the batch-1 run, and hipGraph
    replay gives the same bytes as eager launches."""
    cfg = os.path.join(cfg_dir, "yolov3-tiny_quant.cfg")
    wts = str(tmp_path / "w.weights")
    synth.synth_weights(cfg, wts, seed=1234)
    x = synth.synth_image_u8(3, 416, 416, seed=7)
    xb = np.repeat(x[None], 64, axis=0)
    xb[1::2] = synth.synth_image_u8(3, 416, 416, seed=8)  # two distinct images interleaved
    outs, info = _run_host_net(cfg, wts, xb, binding.ACC_EXACT, graph=True, dump_int32=False)
    one, _ = _run_host_net(cfg, wts, x[None], binding.ACC_EXACT)
    # L0, L2, L4, L6, L10 (stride-1 pool) fused with their maxpools, L18 with its upsample: own tensor not stored; L8 + L9 fused too, but
    # L8's tensor is stored as well (the route to layer 20 reads it); the two heads carry their yolo layers
    assert sum(inf["fused"] for inf in info) == 6 and [i for i, inf in enumerate(info) if inf["fuses_next"]] == [0, 2, 4, 6, 8, 10, 15, 18, 22]
    # every specialised kernel is exercised by the batch-64 net: first layer, conv + pool (16 channels: conv_pool16 = 7, 32 / 64: conv_small = 2), the
    # weights-stationary 3x3 (conv_ws3) and 1x1 (conv1x1) kernels, the row-image kernel on the two deep 3x3 layers
    assert {i: inf["kernel"] for i, inf in enumerate(info) if inf["type"] == binding.T_CONV} == \
        {0: 1, 2: 7, 4: 2, 6: 2, 8: 4, 10: 4, 12: 5, 13: 3, 14: 4, 15: 3, 18: 3, 21: 5, 22: 3}
    for i, inf in enumerate(info):
        if inf["type"] == binding.T_YOLO or inf["fused"]:
            continue  # a fused conv's own (pre-pool) tensor is not stored
        per = inf["outputs"]
        u = outs[i]["u8"].reshape(64, per)
        assert np.array_equal(u[0], one[i]["u8"]), f"layer {i}: slot 0 != batch-1 run"
        hs = [sha(u[b]) for b in range(64)]
        assert len(set(hs[0::2])) == 1 and len(set(hs[1::2])) == 1, f"layer {i}: slots differ"
        assert hs[0] != hs[1]


def _device_quantize(x):
    S = binding.shim()
    x = np.ascontiguousarray(x, np.float32).ravel()
    xd = binding.DevBuf.from_numpy(x)
    mm = binding.DevBuf(16)
    binding.check(S.mi355_image_minmax(xd.ptr, x.size, mm.ptr, None), "image_minmax")
    mx, mn = mm.to_numpy(np.float32, 2)
    return xd, mx, mn


@pytest.mark.parametrize("case", ["qimg_signed", "qimg_unit", "image_like", "negatives", "odd_count", "ties", "tiny_values", "nan"])
def test_image_quantiser_on_device(golden_dir, case):
    """SURVEY 8(f) row 2, quantiser half: mi355_image_minmax / mi355_image_quantize against the oracle's layer-0
    quantiser (itself pinned on vectors the reference produced: funcs.npz qimg_*): min / max, then every byte."""
    rng = np.random.default_rng(len(case))
    if case.startswith("qimg_"):
        x = np.load(os.path.join(golden_dir, "funcs.npz"))[case + "_x"]
    elif case == "image_like":
        x = rng.random(3 * 416 * 416, dtype=np.float32)
    elif case == "negatives":
        x = rng.uniform(-0.7, 1.3, 70001).astype(np.float32)
    elif case == "odd_count":
        x = rng.random(1001, dtype=np.float32)
    elif case == "ties":  # x / scale lands on k + 0.5 exactly: round half away from zero
        x = (np.arange(0, 511, dtype=np.float32) * np.float32(0.5)) * np.float32(1.0)
        x = np.concatenate([x, [np.float32(255.0)]]).astype(np.float32)  # scale == 1
    elif case == "tiny_values":
        x = (rng.random(4099, dtype=np.float32) * np.float32(3e-30)).astype(np.float32)
    else:
        x = rng.random(5000, dtype=np.float32)
        x[::97] = np.nan  # comparisons with NaN are false in the reference's min / max loops: skipped
    want_u8, s, z = oracle.quantize_image(x)
    xd, mx, mn = _device_quantize(x)
    finite = x[np.isfinite(x)]
    assert mx == max(np.float32(0), finite.max()) and mn == min(np.float32(0), finite.min())
    out = binding.DevBuf(x.size + 16)
    binding.check(binding.shim().mi355_image_quantize(xd.ptr, x.size, s, int(z), out.ptr, None), "image_quantize")
    got = out.to_numpy(np.uint8, x.size)
    if case == "nan":  # (int)NaN is undefined in C: compare the defined elements
        keep = np.isfinite(x)
        assert np.array_equal(got[keep], want_u8.ravel()[keep])
    else:
        assert np.array_equal(got, want_u8.ravel())


@pytest.mark.parametrize("name", ["wide", "tall", "up", "same", "odd", "gray"])
def test_letterbox_on_device_vs_reference_vectors(golden_dir, name):
    """mi355_letterbox_forward against the images the reference's letterbox_image produced (funcs.npz lbx_*): bit for bit."""
    f = np.load(os.path.join(golden_dir, "funcs.npz"))
    im, want = f[f"lbx_{name}_im"], f[f"lbx_{name}_out"]
    c, imh, imw = im.shape
    _, h, w = want.shape
    src = binding.DevBuf.from_numpy(im)
    dst = binding.DevBuf(want.nbytes)
    binding.check(binding.shim().mi355_letterbox_forward(src.ptr, imw, imh, c, dst.ptr, w, h, None), "letterbox")
    got = dst.to_numpy(np.float32, want.size).reshape(want.shape)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))


@pytest.mark.parametrize("imh,imw", [(480, 640), (375, 500), (1080, 1920), (416, 416), (300, 200)])
def test_letterbox_416_vs_oracle(imh, imw):
    """camera-sized sources into the 416 x 416 network input, against the oracle restatement"""
    rng = np.random.default_rng(imh + imw)
    im = rng.random((3, imh, imw), dtype=np.float32)
    want = oracle.letterbox_image(im, 416, 416)
    src = binding.DevBuf.from_numpy(im)
    dst = binding.DevBuf(want.nbytes)
    binding.check(binding.shim().mi355_letterbox_forward(src.ptr, imw, imh, 3, dst.ptr, 416, 416, None), "letterbox")
    got = dst.to_numpy(np.float32, want.size).reshape(want.shape)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))


def test_net_device_input_path_equals_host_path(cfg_dir, tmp_path):
    """network_letterbox_input_gpu + network_quantize_input_gpu == oracle letterbox + host quantiser: same uint8 network
    input for a batch of differently sized images, same layer outputs."""
    cfg = os.path.join(cfg_dir, "tiny_unit.cfg")
    wts = str(tmp_path / "w.weights")
    synth.synth_weights(cfg, wts, seed=6)
    rng = np.random.default_rng(10)
    dev = binding.Net(cfg, wts, batch=3, accum=binding.ACC_EXACT, dump_int32=False)
    host = binding.Net(cfg, wts, batch=3, accum=binding.ACC_EXACT, dump_int32=False)
    c, h, w = dev.info[0]["c"], dev.info[0]["h"], dev.info[0]["w"]
    images = [rng.random((c, ih, iw), dtype=np.float32) for ih, iw in [(30, 40), (17, 9), (12, 12)]]
    boxed = np.stack([oracle.letterbox_image(im, h, w) for im in images])
    want_in = host.prepare_from_float(boxed)
    got_in = dev.prepare_from_images_gpu(images)
    assert np.array_equal(got_in, want_in)
    host.forward(); host.sync(); dev.forward(); dev.sync()
    for i in range(dev.n):
        if not dev.is_fused(i):
            a, b = dev.pull(i), host.pull(i)
            for k in a:
                assert np.array_equal(a[k], b[k]), (i, k)
    host.close(); dev.close()


def test_net_input_quantiser_on_device_equals_host(cfg_dir, tmp_path):
    """quantization_weights_and_activations_gpu == quantization_weights_and_activations: same uint8 input, same layer
    outputs; a second batch with a different dynamic range re-derives layer 0 in place (and refills the input pads)."""
    cfg = os.path.join(cfg_dir, "tiny_unit.cfg")
    wts = str(tmp_path / "w.weights")
    synth.synth_weights(cfg, wts, seed=5)
    rng = np.random.default_rng(9)
    dev = binding.Net(cfg, wts, batch=2, accum=binding.ACC_EXACT, dump_int32=False)
    n_in = dev.inputs
    for lo, hi in [(0.0, 1.0), (-0.4, 0.8), (0.0, 0.37)]:
        x = rng.uniform(lo, hi, 2 * n_in).astype(np.float32)
        host = binding.Net(cfg, wts, batch=2, accum=binding.ACC_EXACT, dump_int32=False)
        want_in = host.prepare_from_float(x)
        host.forward(); host.sync()
        got_in = dev.prepare_from_float_gpu(x)
        dev.forward(); dev.sync()
        assert np.array_equal(got_in, want_in), (lo, hi)
        for i in range(dev.n):
            if dev.is_fused(i):
                continue
            a, b = dev.pull(i), host.pull(i)
            for k in a:
                assert np.array_equal(a[k], b[k]), (lo, hi, i, k)
        host.close()
    dev.close()


@pytest.mark.parametrize("name", ["tiny_unit", "s2_unit"])
def test_yolo_detections_on_device_vs_reference(golden_dir, cfg_dir, tmp_path, name):
    """get_yolo_detections + correct_yolo_boxes on the device (network_yolo_detections_gpu ->
    mi355_yolo_detections) against the detections the reference produced on the same net (golden fixture), and with a
    record budget smaller than the number of detections."""
    from test_oracle_golden import DET_CALLS, assert_detections_match
    g = np.load(os.path.join(golden_dir, f"{name}_seed1.npz"))
    cfg = os.path.join(cfg_dir, f"{name}.cfg")
    wts = str(tmp_path / "w.weights")
    synth.synth_weights(cfg, wts, seed=1, act_gain=float(g["act_gain"]))
    x = g["input_u8"]
    net = binding.Net(cfg, wts, batch=3)
    net.prepare_fixed(1.0 / 255.0, 0)
    net.push_input(np.repeat(x[None], 3, axis=0))
    net.forward(); net.sync()
    ly = [i for i, inf in enumerate(net.info) if inf["type"] == binding.T_YOLO]
    for i in ly:
        inf = net.info[i]
        cand = len(g[f"L{i}_mask"]) * inf["out_h"] * inf["out_w"]
        classes = inf["outputs"] // cand - 5
        for k, (imw, imh, rel, th) in enumerate(DET_CALLS):
            counts, recs = net.detections(i, classes, imw, imh, th, rel, cand)
            for b in range(3):
                assert_detections_match(int(counts[b]), recs[b, :counts[b]], int(g[f"L{i}_det{k}_count"]), g[f"L{i}_det{k}_recs"])
        want = int(g[f"L{i}_det0_count"])
        if want > 4:  # budget smaller than the detections: the count is still reported, records are a subset
            counts, recs = net.detections(i, classes, *DET_CALLS[0][:2], DET_CALLS[0][3], DET_CALLS[0][2], 4)
            assert (counts == want).all()
            ranks = set(g[f"L{i}_det0_recs"][:, 0].tolist())
            assert all(r in ranks for r in recs[:, :, 0].ravel().tolist())
    net.close()


def test_yolo_detections_batch64_vs_oracle(cfg_dir, tmp_path):
    """Both heads of yolov3-tiny @416 at batch 8 (two distinct images): device box decode == oracle on the device's own
    yolo tensors, per image."""
    cfg = os.path.join(cfg_dir, "yolov3-tiny_quant.cfg")
    wts = str(tmp_path / "w.weights")
    synth.synth_weights(cfg, wts, seed=1234)
    xb = np.repeat(synth.synth_image_u8(3, 416, 416, seed=7)[None], 8, axis=0)
    xb[1::2] = synth.synth_image_u8(3, 416, 416, seed=8)
    net = binding.Net(cfg, wts, batch=8)
    net.prepare_fixed(1.0 / 255.0, 0)
    net.push_input(xb)
    net.forward(); net.sync()
    anchors = np.array([10, 14, 23, 27, 37, 58, 81, 82, 135, 169, 344, 319], np.float32)
    for i, mask in ((16, [3, 4, 5]), (23, [0, 1, 2])):
        inf = net.info[i]
        h, w = inf["out_h"], inf["out_w"]
        out = net.pull(i)["f32"].reshape(8, -1)
        counts, recs = net.detections(i, 5, 640, 424, 0.5, 1, 3 * h * w)
        for b in range(8):
            cnt, want = oracle.yolo_detections(out[b], 3, 5, h, w, anchors, mask, 416, 416, 640, 424, 0.5, 1)
            assert counts[b] == cnt and cnt > 0
            got = recs[b, :cnt]
            exact = [0, 1, 2] + list(range(5, got.shape[1]))
            assert np.array_equal(got[:, exact], want[:, exact])
            np.testing.assert_allclose(got[:, 3:5], want[:, 3:5], rtol=3e-7, atol=0)  # exp(): device libm vs glibc
    net.close()


def test_yolov3_chain_608_properties(cfg_dir, tmp_path):
    """BASELINE config[4] in spirit (full-YOLOv3 convolution shapes at 608x608: 378 MB activation tensors at batch 32
    would not fit the test time budget of the CPU oracle, so batch 4 and size-independent properties): slots holding the
    same image give the same bytes on every layer, slot 0 equals the batch-1 run, the 255-channel head's float tensors
    are finite, and the first three layers (first-layer kernel, stride-2 32->64, 1x1 64->32 at 304x304) equal the oracle."""
    cfg = os.path.join(cfg_dir, "yolov3_chain_quant.cfg")
    wts = str(tmp_path / "w.weights")
    synth.synth_weights(cfg, wts, seed=99)
    x = synth.synth_image_u8(3, 608, 608, seed=5)
    xb = np.repeat(x[None], 4, axis=0)
    xb[1::2] = synth.synth_image_u8(3, 608, 608, seed=6)
    outs, info = _run_host_net(cfg, wts, xb, binding.ACC_EXACT, dump_int32=False)
    one, _ = _run_host_net(cfg, wts, x[None], binding.ACC_EXACT, dump_int32=False)
    assert sum(1 for inf in info if inf["type"] == binding.T_CONV and inf["stride"] == 2) == 5
    for i, inf in enumerate(info):
        if inf["type"] == binding.T_YOLO or inf["fused"]:
            continue
        per = inf["outputs"]
        u = outs[i]["u8"].reshape(4, per)
        assert np.array_equal(u[0], one[i]["u8"]), f"layer {i}: slot 0 != batch-1 run"
        assert np.array_equal(u[0], u[2]) and np.array_equal(u[1], u[3]) and not np.array_equal(u[0], u[1]), f"layer {i}"
    head = len(info) - 2
    assert np.isfinite(outs[head]["f32"]).all() and np.isfinite(outs[head + 1]["f32"]).all()
    onet = oracle.OracleNet(cfg, wts)
    onet.layers = onet.layers[:3]; onet.w = onet.w[:3]
    onet.prepare(np.float32(1.0 / 255.0), 0)
    want = onet.forward(x, accum=oracle.ACC_EXACT)
    for i in range(3):
        assert np.array_equal(one[i]["u8"], want[i]["u8"].ravel()), f"layer {i} vs oracle"


def test_microbench_shape_vs_oracle_and_linearity():
    """BASELINE config[1]: 3x3 s1 conv 256->256, 52x52, batch 32.  Oracle on 2 of the 32 images (full tensors) plus
    properties at full size: duplicated images -> identical outputs; accumulators are linear in the weights:
    acc(w1) + acc(w2) - acc(zero-weights) == acc(w1 + w2 - 0) when w1 + w2 stays in uint8."""
    rng = np.random.default_rng(1)
    B, c, n, H, W, k = 32, 256, 256, 52, 52, 3
    x = rng.integers(0, 256, (B, c, H, W), dtype=np.uint8)
    x[5] = x[2]
    wq, zp_w, bias, mv, sv = _rand_layer(np.random.default_rng(2), n, c, k, 2.0 ** -10, 2.0 ** -9)
    zp_w = np.random.default_rng(3).integers(100, 157, n, dtype=np.uint8)
    xt = binding.DevTensor.from_nchw(x, 0)
    got = binding.conv_forward(xt, wq, zp_w, k, bias, mv, sv, 0, 23, 1.0, binding.ACT["leaky"])
    for b in (0, 31):
        acc = oracle.conv_acc(x[b], wq, zp_w, k, 1, 1, 0, oracle.ACC_EXACT)
        assert np.array_equal(got["int32"][b], acc), f"image {b} accumulators"
        u8 = oracle.requant(acc, bias, mv, sv, 23, oracle.LEAKY, oracle.STORE_WRAP)
        assert np.array_equal(got["u8"][b].reshape(n, H * W), u8), f"image {b} uint8"
    assert np.array_equal(got["int32"][5], got["int32"][2])
    # linearity in the weights (zero points fixed): acc is sum (w - zp) x
    w1 = (wq // 2).astype(np.uint8); w2 = (wq - w1).astype(np.uint8)
    z = np.zeros_like(wq)
    a1 = binding.conv_forward(xt, w1, zp_w, k, bias, mv, sv, 0, 23, 1.0, binding.ACT["leaky"])["int32"].astype(np.int64)
    a2 = binding.conv_forward(xt, w2, zp_w, k, bias, mv, sv, 0, 23, 1.0, binding.ACT["leaky"])["int32"].astype(np.int64)
    a0 = binding.conv_forward(xt, z, zp_w, k, bias, mv, sv, 0, 23, 1.0, binding.ACT["leaky"])["int32"].astype(np.int64)
    assert np.array_equal(a1 + a2 - a0, got["int32"].astype(np.int64))


@pytest.mark.parametrize("tag", ["leaky", "relu6"])
def test_yolov3_tiny_416_real_image_through_the_device_quantiser(golden_dir, cfg_dir, tmp_path, tag):
    """BASELINE config[0], real-image half.  The float image rebuilt from
    tests/golden/realimg_416.npz (the reference's test image after ITS load_image_color -> letterbox_image, as data) goes through the
    DEVICE quantiser (quantization_weights_and_activations_gpu: min / max reduce, scale / zero point, per-element quantise; ref
    src/blas.c:279) and the 24 layers:
      * MI355_ACC_REF_F32: every int32 / uint8 / float tensor has the hash the reference produced on that image;
      * MI355_ACC_EXACT (production kernels): equals the oracle's exact mode on every tensor; on every conv whose input still equals
        the reference's, the accumulators and bytes inside the committed fp32-exact mask hash like the reference's."""
    r = np.load(os.path.join(golden_dir, "realimg_416.npz"))
    g = json.load(open(os.path.join(golden_dir, f"yolov3_tiny_{tag}_realimg.json")))
    cfg = os.path.join(cfg_dir, g["cfg"])
    wts = str(tmp_path / "w.weights")
    assert synth.synth_weights(cfg, wts, seed=g["weight_seed"])["sha256"] == g["weights_sha256"]
    xf = synth.dequantized_float_image(r["input_u8"], r["scale"], r["zero_point"], r["fmin"], r["imin"], r["fmax"], r["imax"])
    x = r["input_u8"]
    outs = {}
    for accum in (binding.ACC_REF_F32, binding.ACC_EXACT):
        net = binding.Net(cfg, wts, batch=1, accum=accum, dump_int32=True)
        xq = net.prepare_from_float_gpu(xf)
        assert np.array_equal(xq, x.ravel()), "device quantiser bytes"
        p0 = net.prep(0)
        assert np.float32(p0["s_in"]) == r["scale"] and p0["zp_in"] == int(r["zero_point"])
        net.forward()
        net.sync()
        outs[accum] = [net.pull(i) for i in range(net.n)]
        info = net.info
        net.close()
    for e in g["layers"]:  # bit-faithful mode: the reference's hashes, every layer
        i, o = e["i"], outs[binding.ACC_REF_F32][e["i"]]
        if "int32_sha256" in e:
            assert sha(o["int32"]) == e["int32_sha256"], f"layer {i} int32"
        if "u8_sha256" in e:
            assert sha(o["u8"]) == e["u8_sha256"], f"layer {i} u8"
        if "f32_sha256" in e and e["type"] == "conv":
            assert sha(o["f32"]) == e["f32_sha256"], f"layer {i} f32"
    onet = oracle.OracleNet(cfg, wts)
    onet.prepare(np.float32(r["scale"]), int(r["zero_point"]))
    want = onet.forward(x, accum=oracle.ACC_EXACT, want_s1=True)
    same_input = True  # the exact net's input to layer i still equals the reference's
    checked = 0
    for i, inf in enumerate(info):
        o, e = outs[binding.ACC_EXACT][i], g["layers"][i]
        if inf["type"] == binding.T_CONV:
            assert np.array_equal(o["int32"], want[i]["int32"].ravel()), f"layer {i} int32 vs oracle"
            if same_input:
                ex, s1 = want[i]["int32"], want[i]["s1"]
                mask = (s1 <= 2 ** 24) & (np.abs(ex.astype(np.int64)) <= 2 ** 24)
                assert int(mask.sum()) == e["exact_mask_count"] and sha(np.packbits(mask.ravel())) == e["exact_mask_sha256"], f"layer {i} mask"
                assert sha(np.where(mask, o["int32"].reshape(ex.shape), 0).astype(np.int32)) == e["ref_int32_masked_sha256"], f"layer {i} masked int32"
                assert sha(np.where(mask, o["u8"].reshape(ex.shape), 0).astype(np.uint8)) == e["ref_u8_masked_sha256"], f"layer {i} masked u8"
                checked += 1
        if inf["type"] != binding.T_YOLO:
            assert np.array_equal(o["u8"], want[i]["u8"].ravel()), f"layer {i} u8 vs oracle"
            same_input = same_input and sha(o["u8"]) == e["u8_sha256"]
    assert checked >= 5  # layers 0..8 at least: K <= 1152 never leaves the fp32-exact regime on this image
