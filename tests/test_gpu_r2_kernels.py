"""GPU suite, part 5 (`-m gpu`): kernel variants added in round 2, each against the kernel / oracle it must equal."""
import os

import numpy as np
import pytest

import oracle
from yolo_quantization_amd import binding, synth

pytestmark = pytest.mark.gpu
C = binding.C
ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


@pytest.fixture(scope="module", autouse=True)
def device():
    binding.init(0)


def _rand_layer(rng, n, c, k, m_lo=2.0 ** -11, m_hi=2.0 ** -7):
    K = c * k * k
    wq = rng.integers(0, 256, (n, K), dtype=np.uint8)
    zp_w = rng.integers(90, 166, n, dtype=np.uint8)
    bias = rng.integers(-20000, 20000, n).astype(np.int32)
    M = rng.uniform(m_lo, m_hi, n)
    shift = np.floor(-np.log2(M)).astype(int)
    M0 = np.round(M * 2.0 ** shift * 2 ** 31)
    return wq, zp_w, bias, M0 * 2.0 ** -31, 2.0 ** -shift.astype(np.float64)


def _planar_tensor(x):
    """The reference's [B][3][H][W] bytes on the device, described in place (mi355_tensor_describe_nchw)."""
    S = binding.shim()
    S.mi355_tensor_describe_nchw.restype = C.c_size_t
    S.mi355_tensor_describe_nchw.argtypes = [C.POINTER(binding.Tensor), C.c_int, C.c_int, C.c_int, C.c_int]
    B, Cc, H, W = x.shape
    t = binding.Tensor()
    assert S.mi355_tensor_describe_nchw(C.byref(t), B, H, W, Cc) == x.size
    buf = binding.DevBuf.from_numpy(x)
    t.data = buf.ptr
    return t, buf


# (round 6, ADVICE r05: every activation x store instantiation of the 32-filter planar kernels -- the ones that spill -- is exercised, with and
# without the packed epilogue table; rounds 2-5 ran three of the twelve)
@pytest.mark.parametrize("n,act,store", [(16, "leaky", binding.STORE_WRAP), (32, "relu6", binding.STORE_SATURATE), (16, "linear", binding.STORE_WRAP),
                                         (32, "linear", binding.STORE_WRAP), (32, "leaky", binding.STORE_WRAP), (32, "leaky", binding.STORE_SATURATE),
                                         (32, "relu6", binding.STORE_WRAP), (32, "linear", binding.STORE_SATURATE), (16, "relu6", binding.STORE_WRAP),
                                         (16, "leaky", binding.STORE_SATURATE)])
@pytest.mark.parametrize("B,H,W,zp_in", [(2, 32, 64, 0), (1, 48, 100, 37), (3, 18, 36, 255), (1, 416, 416, 0)])
@pytest.mark.parametrize("ept", [False, True], ids=["derive", "table"])
def test_first_layer_reads_nchw_planes_in_place(n, act, store, B, H, W, zp_in, ept):
    """The first-layer MFMA kernels fed the reference's colour planes directly (no nchw -> 4-byte-cell conversion pass):
    with and without the fused 2x2/2 maxpool, bytes equal the oracle's conv (+ maxpool); ragged tile edges, non-zero input
    zero point in the pad, every W % 4 == 0."""
    rng = np.random.default_rng(n + B + H + W)
    x = rng.integers(0, 256, (B, 3, H, W), dtype=np.uint8)
    wq, zp_w, bias, mv, sv = _rand_layer(rng, n, 3, 3, 2.0 ** -8, 2.0 ** -5)
    blob = binding.DevBuf.from_numpy(binding.conv_pack(wq, zp_w, 3, 3, bias, mv, sv, *((binding.ACT[act], 23) if ept else ())))
    xt, keep = _planar_tensor(x)
    d = binding.ConvDesc(n, 3, 3, 1, 1, binding.ACT[act], store, binding.ACC_EXACT, zp_in, 23, 0.05)
    S = binding.shim()
    want = np.stack([oracle.requant(oracle.conv_acc(x[b], wq, zp_w, 3, 1, 1, zp_in), bias, mv, sv, 23, oracle.ACT[act], store).reshape(n, H, W)
                     for b in range(B)])
    y = binding.DevTensor(B, H, W, n, 23)
    binding.check(S.mi355_conv_forward(C.byref(d), C.byref(xt), blob.ptr, None, None, y.ref(), None, None, None), "conv (planar input)")
    assert S.mi355_last_conv_kernel() == 1
    assert np.array_equal(y.to_nchw(), want)
    yp = binding.DevTensor(B, H // 2, W // 2, n, 23)
    binding.check(S.mi355_conv_pool_forward(C.byref(d), C.byref(xt), blob.ptr, None, yp.ref(), None), "conv+pool (planar input)")
    pooled = np.stack([oracle.maxpool_u8(want[b], 2, 2, 1) for b in range(B)])
    assert np.array_equal(yp.to_nchw(), pooled)


def test_planar_input_is_refused_outside_its_domain():
    rng = np.random.default_rng(0)
    S = binding.shim()
    for (n, H, W) in ((20, 32, 32), (16, 32, 34), (16, 31, 32)):   # filters not 16 / 32; W % 4 != 0; odd map
        x = rng.integers(0, 256, (1, 3, H, W), dtype=np.uint8)
        wq, zp_w, bias, mv, sv = _rand_layer(rng, n, 3, 3)
        blob = binding.DevBuf.from_numpy(binding.conv_pack(wq, zp_w, 3, 3, bias, mv, sv))
        xt, keep = _planar_tensor(x)
        d = binding.ConvDesc(n, 3, 3, 1, 1, binding.ACT["leaky"], 0, 0, 0, 23, 0.05)
        y = binding.DevTensor(1, H, W, n, 23)
        assert S.mi355_conv_forward(C.byref(d), C.byref(xt), blob.ptr, None, None, y.ref(), None, None, None) == -22


def test_net_falls_back_to_conversion_when_layer0_cannot_read_planes(cfg_dir, tmp_path):
    """A net whose first layer is outside the in-place kernel's domain (24 filters) still runs: the host converts the input
    and clears the flag; bytes equal the oracle."""
    txt = open(os.path.join(cfg_dir, "tiny_unit.cfg")).read().replace("filters=16", "filters=48", 1)
    cfg = str(tmp_path / "t.cfg")
    open(cfg, "w").write(txt)
    wts = str(tmp_path / "w.weights")
    synth.synth_weights(cfg, wts, seed=2)
    x = synth.synth_image_u8(3, 12, 12, seed=5, batch=2)
    net = binding.Net(cfg, wts, batch=2)
    net.prepare_fixed(1.0 / 255.0, 0)
    net.push_input(x)
    net.forward(); net.forward(); net.sync()
    onet = oracle.OracleNet(cfg, wts)
    onet.prepare(np.float32(1.0 / 255.0), 0)
    for b in range(2):
        want = onet.forward(x[b])
        for i, inf in enumerate(net.info):
            if inf["type"] != binding.T_YOLO and not net.is_fused(i):
                per = inf["outputs"]
                assert np.array_equal(net.pull(i)["u8"][b * per:(b + 1) * per], want[i]["u8"].ravel()), (b, i)
    net.close()


@pytest.mark.parametrize("B,c,n,H,W,act,fam", [(24, 128, 256, 26, 26, "relu6", 4), (16, 256, 512, 20, 18, "leaky", 4), (4, 128, 256, 76, 76, "leaky", 4),
                                              (32, 256, 512, 38, 38, "leaky", 4)])
@pytest.mark.parametrize("store", [binding.STORE_WRAP, binding.STORE_SATURATE], ids=["wrap", "saturate"])
def test_conv_with_fused_residual_add(B, c, n, H, W, act, fam, store):
    """mi355_conv_shortcut_forward (the 3x3 conv of a residual block + the `[shortcut] quantized=1` after it in one kernel)
    equals conv -> requantise -> orc_shortcut_u8 of the oracle (conv_ws3.hip: the 128- / 256-channel 3x3 layers, 16 of YOLOv3's
    23 residual blocks; the other kernels refuse and the host runs the add on its own -- measured faster there)."""
    S = binding.shim()
    S.mi355_conv_shortcut_forward.argtypes = [C.POINTER(binding.ConvDesc), C.POINTER(binding.Tensor), C.c_void_p, C.POINTER(binding.Tensor),
                                              C.POINTER(binding.Tensor), C.c_int32, C.c_int32, C.c_uint8, C.c_uint8, C.c_void_p]
    rng = np.random.default_rng(B + c + n + H)
    x = rng.integers(0, 256, (B, c, H, W), dtype=np.uint8)
    r = rng.integers(0, 256, (B, n, H, W), dtype=np.uint8)
    wq, zp_w, bias, mv, sv = _rand_layer(rng, n, c, 3)
    zp_in, zp_act, zp_from, zp_out = 23, 23, 40, 31
    Ka = oracle.shortcut_multiplier(np.float32(6.6 / 255), np.float32(9.0 / 255)); Kb = oracle.shortcut_multiplier(np.float32(5.0 / 255), np.float32(9.0 / 255))
    xt = binding.DevTensor.from_nchw(x, zp_in); rt = binding.DevTensor.from_nchw(r, zp_from)
    yt = binding.DevTensor(B, H, W, n, zp_out)
    blob = binding.DevBuf.from_numpy(binding.conv_pack(wq, zp_w, c, 3, bias, mv, sv))
    d = binding.ConvDesc(n, c, 3, 1, 1, binding.ACT[act], store, binding.ACC_EXACT, zp_in, zp_act, 0.05)
    binding.check(S.mi355_conv_shortcut_forward(C.byref(d), xt.ref(), blob.ptr, rt.ref(), yt.ref(), Ka, Kb, zp_from, zp_out, None), "conv+shortcut")
    assert S.mi355_last_conv_kernel() == fam
    conv = np.stack([oracle.requant(oracle.conv_acc(x[b], wq, zp_w, 3, 1, 1, zp_in), bias, mv, sv, zp_act, oracle.ACT[act], store).reshape(n, H, W)
                     for b in range(B)])
    want = oracle.shortcut_u8(conv, r, Ka, Kb, zp_act, zp_from, zp_out)
    assert np.array_equal(yt.to_nchw(), want)
    assert (want == 255).any() and want.min() < 64


def test_fused_residual_add_is_refused_where_no_kernel_has_it():
    S = binding.shim()
    S.mi355_conv_shortcut_forward.argtypes = [C.POINTER(binding.ConvDesc), C.POINTER(binding.Tensor), C.c_void_p, C.POINTER(binding.Tensor),
                                              C.POINTER(binding.Tensor), C.c_int32, C.c_int32, C.c_uint8, C.c_uint8, C.c_void_p]
    rng = np.random.default_rng(1)
    for (c, n, k) in ((512, 1024, 3), (64, 32, 1), (32, 64, 3)):   # row-image kernel, 1x1 kernel, few-channel kernel
        x = rng.integers(0, 256, (1, c, 19, 19), dtype=np.uint8)
        wq, zp_w, bias, mv, sv = _rand_layer(rng, n, c, k)
        xt = binding.DevTensor.from_nchw(x, 0); rt = binding.DevTensor(1, 19, 19, n, 0); yt = binding.DevTensor(1, 19, 19, n, 0)
        blob = binding.DevBuf.from_numpy(binding.conv_pack(wq, zp_w, c, k, bias, mv, sv))
        d = binding.ConvDesc(n, c, k, 1, k // 2, binding.ACT["leaky"], 0, 0, 0, 23, 0.05)
        assert S.mi355_conv_shortcut_forward(C.byref(d), xt.ref(), blob.ptr, rt.ref(), yt.ref(), 40000, 40000, 0, 0, None) == -22


def test_route_with_channel_counts_not_multiple_of_16():
    """forward_route_layer_quant (ref: src/route_layer.c:107-130) concatenates any channel counts; the 16-byte-group copy kernel
    serves aligned offsets, a byte-granular one the rest."""
    rng = np.random.default_rng(8)
    S = binding.shim()
    for chans in ((30, 48, 20), (16, 30), (255, 255), (7,)):
        xs = [rng.integers(0, 256, (2, c, 5, 7), dtype=np.uint8) for c in chans]
        ts = [binding.DevTensor.from_nchw(x, 3) for x in xs]
        y = binding.DevTensor(2, 5, 7, sum(chans), 3)
        arr = (C.POINTER(binding.Tensor) * len(ts))(*[C.pointer(t.t) for t in ts])
        binding.check(S.mi355_route_forward(arr, len(ts), y.ref(), None), "route")
        assert np.array_equal(y.to_nchw(), np.concatenate(xs, axis=1)), chans


@pytest.mark.parametrize("B,c,n,H,W,act,mode", [(64, 128, 256, 26, 26, "leaky", "s2"),    # yolov3-tiny layers 8 + 9 at the bench's batch
                                               (24, 128, 256, 26, 26, "relu6", "s2"),    # tiles of 16 windows that straddle images
                                               (16, 256, 512, 20, 18, "leaky", "s2"),    # two K parts: the bytes are staged in the partial-sum slots
                                               (64, 256, 512, 13, 13, "leaky", "s1"),    # layers 10 + 11
                                               (40, 256, 512, 13, 13, "linear", "s1"),   # fewer images than workgroups
                                               (300, 128, 256, 9, 11, "relu6", "s1")])   # several whole-image tiles per (persistent) workgroup
@pytest.mark.parametrize("store", [binding.STORE_WRAP, binding.STORE_SATURATE], ids=["wrap", "saturate"])
@pytest.mark.parametrize("keep", [False, True], ids=["pooled-only", "both-tensors"])
def test_ws3_conv_with_fused_maxpool(B, c, n, H, W, act, mode, store, keep):
    """conv_ws3.hip + the maxpool behind it in one kernel (mi355_conv_pool_forward): the 2x2 / stride-2 window (ypool of half
    the size) and the reference's 2x2 / stride-1 window (pad = 1: ypool of the conv's own size, windows clipped at the right /
    lower border; ref src/maxpool_layer.c:109-146), with and without the conv's own tensor.  The kernel pools the STORED
    bytes, so wrapped values inside a window behave as in the reference.  Oracle: conv -> requantise -> orc_maxpool_u8 on the
    distinct images (the batch repeats four images)."""
    rng = np.random.default_rng(B + c + n + H)
    S = binding.shim()
    x4 = rng.integers(0, 256, (4, c, H, W), dtype=np.uint8)
    x = x4[np.arange(B) % 4]
    wq, zp_w, bias, mv, sv = _rand_layer(rng, n, c, 3, 2.0 ** -13, 2.0 ** -9)   # some accumulators leave 0..255: wrap != saturate
    zp_in, zp_act = 23, 31
    xt = binding.DevTensor.from_nchw(x, zp_in)
    blob = binding.DevBuf.from_numpy(binding.conv_pack(wq, zp_w, c, 3, bias, mv, sv))
    y = binding.DevTensor(B, H, W, n, zp_act) if keep else None
    yp = binding.DevTensor(B, H // 2, W // 2, n, zp_act) if mode == "s2" else binding.DevTensor(B, H, W, n, zp_act)
    d = binding.ConvDesc(n, c, 3, 1, 1, binding.ACT[act], store, binding.ACC_EXACT, zp_in, zp_act, 0.05)
    binding.check(S.mi355_conv_pool_forward(C.byref(d), xt.ref(), blob.ptr, y.ref() if keep else None, yp.ref(), None), "conv_pool")
    assert S.mi355_last_conv_kernel() == 4
    u8 = np.stack([oracle.requant(oracle.conv_acc(x4[b], wq, zp_w, 3, 1, 1, zp_in), bias, mv, sv, zp_act, oracle.ACT[act], store).reshape(n, H, W)
                   for b in range(4)])
    sat = np.stack([oracle.requant(oracle.conv_acc(x4[0], wq, zp_w, 3, 1, 1, zp_in), bias, mv, sv, zp_act, oracle.ACT[act], binding.STORE_SATURATE)])
    if store == binding.STORE_WRAP:
        assert (sat[0].reshape(n, H, W) != u8[0]).any(), "case should exercise wrapped bytes"
    pool = np.stack([oracle.maxpool_u8(u8[b], 2, 2 if mode == "s2" else 1, 1) for b in range(4)])
    got = yp.to_nchw()
    for b in range(B):
        assert np.array_equal(got[b], pool[b % 4]), f"pooled tensor, image {b}"
    if keep:
        goty = y.to_nchw()
        for b in range(B):
            assert np.array_equal(goty[b], u8[b % 4]), f"conv tensor, image {b}"


def test_ws3_fused_maxpool_is_refused_outside_its_domain():
    """MI355_EINVAL (nothing launched) where the weights-stationary kernel cannot fuse: too few pixels per workgroup, odd
    maps for the stride-2 window, the stride-1 window behind any other kernel family."""
    S = binding.shim()
    rng = np.random.default_rng(3)
    for (B, c, n, H, W, s1) in ((1, 128, 256, 26, 26, False), (8, 128, 256, 13, 13, False), (6, 256, 512, 13, 13, True), (8, 64, 128, 13, 13, True), (8, 512, 1024, 13, 13, True)):
        x = rng.integers(0, 256, (B, c, H, W), dtype=np.uint8)
        wq, zp_w, bias, mv, sv = _rand_layer(rng, n, c, 3)
        xt = binding.DevTensor.from_nchw(x, 0)
        yp = binding.DevTensor(B, H, W, n, 0) if s1 else binding.DevTensor(B, H // 2, W // 2, n, 0)
        blob = binding.DevBuf.from_numpy(binding.conv_pack(wq, zp_w, c, 3, bias, mv, sv))
        d = binding.ConvDesc(n, c, 3, 1, 1, binding.ACT["leaky"], 0, 0, 0, 23, 0.05)
        assert S.mi355_conv_pool_forward(C.byref(d), xt.ref(), blob.ptr, None, yp.ref(), None) == -22, (B, c, n, H, W, s1)


@pytest.mark.parametrize("B,c,n,H,W,act", [(64, 512, 1024, 13, 13, "leaky"),   # yolov3-tiny layer 12 at the bench's batch: 128 x 384 tiles
                                           (16, 384, 256, 26, 26, "relu6"),    # layer 21's shape
                                           (8, 256, 192, 19, 19, "linear"),    # ragged filter count, 19-wide map
                                           (4, 64, 64, 30, 30, "leaky"),       # 64-filter tiles
                                           (2, 128, 128, 52, 52, "leaky"),
                                           (32, 512, 1024, 19, 19, "leaky"),   # YOLOv3-608's 19-wide maps: the narrow-map variant (128 x 384 tiles, exact LDS rows, extra DMA slots)
                                           (24, 256, 128, 20, 17, "relu6")])   # 17 + 2 cells in a 32-slot row, tiles that straddle images
@pytest.mark.parametrize("store", [binding.STORE_WRAP, binding.STORE_SATURATE], ids=["wrap", "saturate"])
def test_row_image_kernel_on_16x16x64_mfma_equals_the_32x32x32_one(B, c, n, H, W, act, store):
    """conv_rows16.hip (V_MFMA_I32_16X16X64_I8, fragments prefetched in place) against conv_rows.hip (debug switch 2^20 routes the
    call back to it) on the same launch: identical bytes and float tails; image 0 against the oracle."""
    rng = np.random.default_rng(B + c + n + H)
    S = binding.shim()
    x = rng.integers(0, 256, (B, c, H, W), dtype=np.uint8)
    wq, zp_w, bias, mv, sv = _rand_layer(rng, n, c, 3)
    xt = binding.DevTensor.from_nchw(x, 23)
    S.mi355_debug_flags(16384)   # keep the weights-stationary kernel out of the way: both runs on the row-image family
    try:
        new = binding.conv_forward(xt, wq, zp_w, 3, bias, mv, sv, 23, 31, 0.05, binding.ACT[act], store, binding.ACC_EXACT, want_acc=False, want_f32=True)
        assert S.mi355_last_conv_kernel() == 5
        S.mi355_debug_flags(16384 | (1 << 20))
        old = binding.conv_forward(xt, wq, zp_w, 3, bias, mv, sv, 23, 31, 0.05, binding.ACT[act], store, binding.ACC_EXACT, want_acc=False, want_f32=True)
        assert S.mi355_last_conv_kernel() == 5
    finally:
        S.mi355_debug_flags(0)
    assert np.array_equal(new["u8"], old["u8"]) and np.array_equal(new["f32"], old["f32"])
    want = oracle.requant(oracle.conv_acc(x[0], wq, zp_w, 3, 1, 1, 23), bias, mv, sv, 31, oracle.ACT[act], store)
    assert np.array_equal(new["u8"][0].reshape(n, -1), want.reshape(n, -1))


def test_heads_without_their_own_float_tensor(cfg_dir, tmp_path):
    """The library's default: a quant_stop head conv fused with its yolo layer stores the yolo layer's l.output only (its own
    l.output is an intermediate nothing else reads; mi355_conv_yolo_forward with y_f32 == NULL).  Same yolo outputs and uint8
    tensors as the net that keeps them; the C-ABI entry refuses the NULL where no kernel has that form."""
    cfg = os.path.join(cfg_dir, "yolov3-tiny_quant.cfg")
    wts = str(tmp_path / "w.weights")
    synth.synth_weights(cfg, wts, seed=1234)
    xs = synth.synth_image_u8(3, 416, 416, seed=5, batch=2)
    outs = {}
    for keep in (True, False):
        net = binding.Net(cfg, wts, batch=2, keep_head_float=keep)
        net.prepare_fixed(1.0 / 255.0, 0)
        net.push_input(xs)
        net.forward(); net.sync()
        outs[keep] = [net.pull(i) for i in range(net.n)]
        heads = [i for i, inf in enumerate(net.info) if inf["type"] == binding.T_CONV and inf["quant_stop"]]
        assert heads == [15, 22] and all(net.fuses_next(i) for i in heads)
        net.close()
    for i in range(len(outs[True])):
        for k, v in outs[False][i].items():
            if k != "int32":
                assert np.array_equal(v, outs[True][i][k]), (i, k)
    assert "f32" in outs[True][15] and "f32" not in outs[False][15] and "f32" in outs[False][16]
    # a 3x3 head (row-image kernel) cannot drop the tensor: MI355_EINVAL, nothing launched
    S = binding.shim()
    S.mi355_conv_yolo_forward.argtypes = [C.POINTER(binding.ConvDesc), C.POINTER(binding.Tensor), C.c_void_p, C.POINTER(binding.Tensor), C.c_void_p,
                                          C.c_void_p, C.c_int, C.c_void_p]
    rng = np.random.default_rng(2)
    x = rng.integers(0, 256, (1, 64, 13, 13), dtype=np.uint8)
    wq, zp_w, bias, mv, sv = _rand_layer(rng, 30, 64, 3)
    xt = binding.DevTensor.from_nchw(x, 0); yt = binding.DevTensor(1, 13, 13, 30, 0)
    blob = binding.DevBuf.from_numpy(binding.conv_pack(wq, zp_w, 64, 3, bias, mv, sv))
    yo = binding.DevBuf(4 * 30 * 169)
    d = binding.ConvDesc(30, 64, 3, 1, 1, binding.ACT["linear"], 0, 0, 0, 23, 0.05)
    assert S.mi355_conv_yolo_forward(C.byref(d), xt.ref(), blob.ptr, yt.ref(), None, yo.ptr, 5, None) == -22


@pytest.mark.parametrize("B,c,n,H,W,act", [(32, 1024, 512, 19, 19, "leaky"),   # YOLOv3-608's necks: two filter tiles of 256
                                           (16, 256, 512, 13, 13, "relu6"),
                                           (8, 512, 768, 10, 10, "linear"),    # three filter tiles
                                           (4, 64, 1024, 9, 7, "leaky")])
@pytest.mark.parametrize("store", [binding.STORE_WRAP, binding.STORE_SATURATE], ids=["wrap", "saturate"])
def test_conv1x1_ws_with_more_than_256_filters(B, c, n, H, W, act, store):
    """conv1x1.hip with the filters split over workgroups (tiles of 256): the same bytes as the row-image kernel on the same call
    (debug switch 8192 routes around conv1x1.hip) and as the oracle on image 0."""
    rng = np.random.default_rng(B + c + n + H)
    S = binding.shim()
    x = rng.integers(0, 256, (B, c, H, W), dtype=np.uint8)
    wq, zp_w, bias, mv, sv = _rand_layer(rng, n, c, 1)
    zp_w[0], zp_w[n - 1] = 0, 255
    xt = binding.DevTensor.from_nchw(x, 23)
    args = (xt, wq, zp_w, 1, bias, mv, sv, 23, 31, 0.05, binding.ACT[act], store, binding.ACC_EXACT)
    new = binding.conv_forward(*args, want_acc=False)
    assert S.mi355_last_conv_kernel() == 3
    S.mi355_debug_flags(8192)
    try:
        old = binding.conv_forward(*args, want_acc=False)
        assert S.mi355_last_conv_kernel() == 5
    finally:
        S.mi355_debug_flags(0)
    assert np.array_equal(new["u8"], old["u8"])
    want = oracle.requant(oracle.conv_acc(x[0], wq, zp_w, 1, 1, 0, 23), bias, mv, sv, 31, oracle.ACT[act], store)
    assert np.array_equal(new["u8"][0].reshape(n, -1), want.reshape(n, -1))


def test_determinism_selfcheck(cfg_dir, tmp_path):
    """network_selfcheck: repeated passes over the same input give the same device-side checksums of the yolo outputs (the bench runs
    it before its warmup steps); the checksum itself separates different outputs."""
    cfg = os.path.join(cfg_dir, "yolov3-tiny_quant.cfg")
    wts = str(tmp_path / "w.weights")
    synth.synth_weights(cfg, wts, seed=1234)
    net = binding.Net(cfg, wts, batch=4, keep_head_float=False)
    net.prepare_fixed(1.0 / 255.0, 0)
    net.push_input(synth.synth_image_u8(3, 416, 416, seed=5, batch=4))
    assert net.selfcheck_result() == -1
    net.selfcheck(6)
    assert net.selfcheck_result() == 0 and net.selfcheck_result() == -1
    net.close()
    S = binding.shim()
    S.mi355_checksum_u32.argtypes = [C.c_void_p, C.c_long, C.c_void_p, C.c_void_p]
    a = np.arange(5000, dtype=np.uint32); b = a.copy(); b[4321] ^= 1
    sums = []
    for v in (a, a[::-1].copy(), b):
        dv = binding.DevBuf.from_numpy(v); ds = binding.DevBuf.from_numpy(np.zeros(1, np.uint64))
        binding.check(S.mi355_checksum_u32(dv.ptr, v.size, ds.ptr, None), "checksum")
        binding.check(S.mi355_stream_sync(None), "sync")
        sums.append(int(ds.to_numpy(np.uint64, 1)[0]))
    want = int((a.astype(object) * (2 * np.arange(5000).astype(object) + 1)).sum() % (1 << 64))
    assert sums[0] == want and sums[1] != sums[0] and sums[2] != sums[0]


def test_first_layer_tile_table_many_tiles_per_workgroup():
    """The pooled first-layer kernel keeps its workgroup's tiles one per LANE (<= 64); the launcher must widen the grid for launches with more
    tiles than 64 per workgroup.  A child process caps the persistent grid at 8 workgroups (MI355_L0_GRID, read once per process) so that a
    676-tile launch needs the widening (16 workgroups x 43 tiles, XCD-wise walk) and a 70-tile one walks 9 tiles per workgroup: bytes equal
    the oracle's conv -> requantise -> maxpool, with the host-derived epilogue table and without, on data with wrapping windows."""
    import subprocess
    import sys
    code = r'''
import ctypes as C, sys, os
sys.path.insert(0, os.path.join(%(root)r, "tests")); sys.path.insert(0, %(root)r)
import numpy as np
import oracle
from yolo_quantization_amd import binding
from test_gpu_r2_kernels import _planar_tensor, _rand_layer
binding.init(0)
S = binding.shim()
for (B, H, W, ept) in ((4, 208, 416, True), (4, 208, 416, False), (2, 40, 224, True)):
    rng = np.random.default_rng(B + H + W)
    x = rng.integers(0, 256, (B, 3, H, W), dtype=np.uint8)
    wq, zp_w, bias, mv, sv = _rand_layer(rng, 16, 3, 3, 2.0 ** -8, 2.0 ** -5)
    act, zp_act = "leaky", 23
    blob = binding.DevBuf.from_numpy(binding.conv_pack(wq, zp_w, 3, 3, bias, mv, sv, *((binding.ACT[act], zp_act) if ept else ())))
    xt, keep = _planar_tensor(x)
    d = binding.ConvDesc(16, 3, 3, 1, 1, binding.ACT[act], binding.STORE_WRAP, binding.ACC_EXACT, 7, zp_act, 0.05)
    want = np.stack([oracle.maxpool_u8(oracle.requant(oracle.conv_acc(x[b], wq, zp_w, 3, 1, 1, 7), bias, mv, sv, zp_act, oracle.ACT[act], binding.STORE_WRAP).reshape(16, H, W), 2, 2, 1)
                     for b in range(B)])
    yp = binding.DevTensor(B, H // 2, W // 2, 16, zp_act)
    binding.check(S.mi355_conv_pool_forward(C.byref(d), C.byref(xt), blob.ptr, None, yp.ref(), None), "conv+pool")
    assert S.mi355_last_conv_kernel() == 1
    assert np.array_equal(yp.to_nchw(), want), (B, H, W, ept)
print("child ok")
''' % {"root": ROOT}
    env = dict(os.environ, MI355_L0_GRID="8")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0 and "child ok" in r.stdout, r.stdout[-1500:] + r.stderr[-3000:]
