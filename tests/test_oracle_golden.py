"""CPU suite: the oracle (oracle/oracle.c restatement) against the committed golden fixtures, which were produced
by running the unmodified reference (tests/golden/make_golden.py).  No GPU, no /root/reference needed."""
import hashlib
import json
import os

import numpy as np
import pytest

import oracle
from yolo_quantization_amd import synth


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


@pytest.fixture(scope="module")
def funcs(golden_dir):
    return np.load(os.path.join(golden_dir, "funcs.npz"))


@pytest.mark.parametrize("name", ["small", "big"])
def test_gemm_known_answer(funcs, name):
    """src/gemm.c:279-299 incl. the fp32-rounding regime ('big': running sums > 2^24)."""
    A, B, Z = funcs[f"gemm_{name}_A"], funcs[f"gemm_{name}_B"], funcs[f"gemm_{name}_Z"]
    Cm = np.zeros((A.shape[0], B.shape[1]), np.int32)
    oracle.gemm_u8(A, B, 1.0, 0, Cm)
    assert np.array_equal(Cm, funcs[f"gemm_{name}_C1"])
    oracle.gemm_u8(Z, B, -1.0, 1, Cm)
    assert np.array_equal(Cm, funcs[f"gemm_{name}_C2"])
    exact = A.astype(np.int64) @ B.astype(np.int64) - Z.astype(np.int64) @ B.astype(np.int64)
    if name == "small":
        assert np.array_equal(Cm, exact)
    else:
        assert not np.array_equal(Cm, exact), "fixture is meant to show the reference's fp32 rounding"


@pytest.mark.parametrize("name,k,s,p,pv", [("k3s1", 3, 1, 1, 23), ("k3s2", 3, 2, 1, 128)])
def test_im2col_known_answer(funcs, name, k, s, p, pv):
    assert np.array_equal(oracle.im2col_u8(funcs["im2col_im"], k, s, p, pv), funcs[f"im2col_{name}"])


def test_quant_multiplier_known_answer(funcs):
    for m, m0, sh in zip(funcs["qm_M"], funcs["qm_M0"], funcs["qm_shift"]):
        rc, a, b = oracle.quant_multiplier(m)
        assert rc == 0 and (a, b) == (int(m0), int(sh)), (m, a, b, m0, sh)
    assert oracle.quant_multiplier(1.0)[0] != 0 and oracle.quant_multiplier(0.0)[0] != 0  # reference asserts


@pytest.mark.parametrize("name", ["signed", "unit"])
def test_quantize_image_known_answer(funcs, name):
    u8, s, z = oracle.quantize_image(funcs[f"qimg_{name}_x"])
    assert s == funcs[f"qimg_{name}_scale"] and z == funcs[f"qimg_{name}_zp"]
    assert np.array_equal(u8, funcs[f"qimg_{name}_u8"])


@pytest.mark.parametrize("name", ["wide", "tall", "up", "same", "odd", "gray"])
def test_letterbox_image_known_answer(funcs, name):
    """letterbox_image + resize_image (src/image.c:812-831, :1199-1242) against images the reference produced: every
    float bit for bit (the quirk of the last row -- only its first term, with (int)(r * h_scale) as it comes -- included)."""
    want = funcs[f"lbx_{name}_out"]
    got = oracle.letterbox_image(funcs[f"lbx_{name}_im"], want.shape[1], want.shape[2])
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))


@pytest.mark.parametrize("name", ["tiny_unit", "s2_unit"])
@pytest.mark.parametrize("seed", [1, 2])
@pytest.mark.parametrize("accum", [oracle.ACC_REF_F32, oracle.ACC_EXACT])
def test_tiny_unit_full_tensors(golden_dir, cfg_dir, tmp_path, seed, accum, name):
    """Every layer type of the path on a 12x12 net (tiny_unit) and a chain of stride-2 3x3 convolutions on 24x24
    (s2_unit): full-tensor equality with the reference, including the wrap-on-store cases of seed 2 (act_gain 8).
    K <= 1152 and small sums, so fp32 accumulation is exact: both modes must match."""
    g = np.load(os.path.join(golden_dir, f"{name}_seed{seed}.npz"))
    cfg = os.path.join(cfg_dir, f"{name}.cfg")
    wts = str(tmp_path / "w.weights")
    meta = synth.synth_weights(cfg, wts, seed=seed, act_gain=float(g["act_gain"]))
    assert meta["sha256"] == str(g["weights_sha256"]), "synthetic weight generator drifted from the fixture"
    net = oracle.OracleNet(cfg, wts)
    L0 = net.layers[0]
    x = synth.synth_image_u8(L0.c, L0.h, L0.w, seed=int(g["img_seed"]))
    assert np.array_equal(x, g["input_u8"])
    net.prepare(np.float32(1.0 / 255.0), 0)
    outs = net.forward(x, accum=accum, store=oracle.STORE_WRAP)
    nwrap = 0
    for i, L in enumerate(net.layers):
        if L.type == "conv":
            for k in ("biases_int32", "M_value", "shift_value", "M0", "shift"):
                assert np.array_equal(net.p[i][k], g[f"L{i}_{k}"]), (i, k)
            assert np.array_equal(outs[i]["int32"].ravel(), g[f"L{i}_int32"]), f"layer {i} accumulators"
            sat = oracle.requant(outs[i]["int32"], net.p[i]["biases_int32"], net.p[i]["M_value"],
                                 net.p[i]["shift_value"], net.w[i]["zp_act"], oracle.ACT[L.activation],
                                 oracle.STORE_SATURATE)
            nwrap += int((sat.ravel() != g[f"L{i}_u8"]).sum())
        if L.type != "yolo":
            assert np.array_equal(outs[i]["u8"].ravel(), g[f"L{i}_u8"]), f"layer {i} uint8"
        if L.quant_stop:
            assert np.array_equal(outs[i]["f32"].ravel(), g[f"L{i}_f32"]), f"layer {i} dequant"
        if L.type == "yolo":
            np.testing.assert_allclose(outs[i]["f32"].ravel(), g[f"L{i}_f32"], rtol=0, atol=1e-6)
    if seed == 2:
        assert nwrap > 50, "seed-2 fixture must contain wrap-on-store elements (saturate != reference)"


@pytest.mark.parametrize("tag", ["leaky", "relu6"])
def test_yolov3_tiny_416_hashes_ref_f32(golden_dir, cfg_dir, tmp_path, tag):
    """BASELINE config[0]: whole yolov3-tiny @416x416, one image: per-layer SHA-256 of int32 accumulators, uint8
    activations and float heads equal the reference default build's (oracle in bit-faithful ref-f32 mode)."""
    g = json.load(open(os.path.join(golden_dir, f"yolov3_tiny_{tag}.json")))
    cfg = os.path.join(cfg_dir, g["cfg"])
    wts = str(tmp_path / "w.weights")
    meta = synth.synth_weights(cfg, wts, seed=g["weight_seed"])
    assert meta["sha256"] == g["weights_sha256"]
    net = oracle.OracleNet(cfg, wts)
    x = synth.synth_image_u8(3, 416, 416, seed=g["image_seed"])
    assert sha(x) == g["input_sha256"]
    net.prepare(np.float32(1.0 / 255.0), 0)
    outs = net.forward(x, accum=oracle.ACC_REF_F32)
    for e in g["layers"]:
        i = e["i"]
        if "prep_sha256" in e:
            p = net.p[i]
            assert sha(np.concatenate([p["biases_int32"].view(np.uint8), p["M_value"].view(np.uint8),
                                       p["shift_value"].view(np.uint8)])) == e["prep_sha256"], f"prep {i}"
        if "int32_sha256" in e:
            assert sha(outs[i]["int32"]) == e["int32_sha256"], f"layer {i} int32"
        if "u8_sha256" in e:
            assert sha(outs[i]["u8"]) == e["u8_sha256"], f"layer {i} u8"
        if "f32_sha256" in e and e["type"] == "conv":
            assert sha(outs[i]["f32"]) == e["f32_sha256"], f"layer {i} f32"


def test_exact_mode_equals_ref_where_fp32_is_exact(golden_dir, cfg_dir, tmp_path):
    """Teacher-forced per-layer check on the leaky model: with the reference's own input to each conv, exact integer
    accumulation equals the fp32 path on every element whose pass-1 sum and result stay within 2^24, and differs
    somewhere on the deep layers (documenting SURVEY.md §0 fact 4)."""
    g = json.load(open(os.path.join(golden_dir, "yolov3_tiny_leaky.json")))
    cfg = os.path.join(cfg_dir, g["cfg"])
    wts = str(tmp_path / "w.weights")
    synth.synth_weights(cfg, wts, seed=g["weight_seed"])
    net = oracle.OracleNet(cfg, wts)
    x = synth.synth_image_u8(3, 416, 416, seed=g["image_seed"])
    net.prepare(np.float32(1.0 / 255.0), 0)
    ref = net.forward(x, accum=oracle.ACC_REF_F32)  # == reference (previous test)
    differs = {}
    for i, L in enumerate(net.layers):
        if L.type != "conv":
            continue
        xin = x if i == 0 else ref[i - 1]["u8"]
        d = net.w[i]
        acc, s1 = oracle.conv_acc(xin, d["wq"], d["zp_w"], L.size, L.stride, L.pad, net.p[i]["zp_in"],
                                  oracle.ACC_EXACT, want_s1=True)
        safe = (s1 <= 2 ** 24) & (np.abs(acc.astype(np.int64)) <= 2 ** 24)
        neq = acc != ref[i]["int32"]
        assert not (neq & safe).any(), f"layer {i}: exact != reference inside the fp32-exact regime"
        differs[i] = int(neq.sum())
        # the committed, reference-produced description of that regime (what tests/test_gpu_refpin.py holds the HIP kernels to)
        e = g["layers"][i]
        assert int(safe.sum()) == e["exact_mask_count"] and sha(np.packbits(safe.ravel())) == e["exact_mask_sha256"], f"layer {i} mask"
        assert sha(np.where(safe, acc, 0).astype(np.int32)) == e["ref_int32_masked_sha256"], f"layer {i} masked int32"
        assert differs[i] == e["exact_vs_ref_mismatch"], f"layer {i} mismatch count"
        K = L.c * L.size * L.size
        if K * 255 * 255 < 2 ** 24:
            assert safe.all() and differs[i] == 0
    assert differs[12] > 0, "L12 (K=4608) is expected to show the reference's fp32 rounding"


@pytest.mark.parametrize("tag", ["leaky", "relu6"])
def test_deep_layers_low_range_vectors(golden_dir, cfg_dir, tmp_path, tag):
    """The K >= 2304 layers on input bytes 0..7 (function-level vectors the reference produced with each layer's own forward
    pointer): the fp32 GEMM is exact there, so BOTH accumulate modes of the restatement must give the reference's hashes."""
    g = json.load(open(os.path.join(golden_dir, f"yolov3_tiny_{tag}.json")))
    cfg = os.path.join(cfg_dir, g["cfg"])
    wts = str(tmp_path / "w.weights")
    synth.synth_weights(cfg, wts, seed=g["weight_seed"])
    net = oracle.OracleNet(cfg, wts)
    net.prepare(np.float32(1.0 / 255.0), 0)
    n = 0
    for e in g["layers"]:
        lr = e.get("lowrange")
        if not lr:
            continue
        i = e["i"]; L = net.layers[i]; d = net.w[i]; p = net.p[i]
        xl = (synth.synth_image_u8(L.c, L.h, L.w, seed=lr["seed"]) >> lr["shift"]).astype(np.uint8)
        assert sha(xl) == lr["input_sha256"]
        for accum in (oracle.ACC_EXACT, oracle.ACC_REF_F32):
            acc = oracle.conv_acc(xl, d["wq"], d["zp_w"], L.size, L.stride, L.pad, p["zp_in"], accum)
            assert sha(acc) == lr["int32_sha256"], (i, accum)
            u8 = oracle.requant(acc, p["biases_int32"], p["M_value"], p["shift_value"], d["zp_act"], oracle.ACT[L.activation])
            assert sha(u8) == lr["u8_sha256"], (i, accum)
        n += 1
    assert n == 4


DET_CALLS = [(640, 480, 1, 0.5), (300, 500, 0, 0.3), (416, 416, 1, 0.6)]  # as in tests/golden/make_golden.py


def assert_detections_match(count, recs, g_count, g_recs):
    """Detections against the reference's get_yolo_detections: same count, same records in the same order; cell / anchor
    rank, box centre, objectness and class scores bit for bit, box width / height to 4 ulp -- the reference binary is
    built with -Ofast, whose exp() of a float is not a correctly rounded libm call (the centre, a plain add and divide,
    is reproducible)."""
    assert count == g_count
    assert recs.shape == g_recs.shape
    exact = [0, 1, 2] + list(range(5, recs.shape[1]))
    assert np.array_equal(recs[:, exact], g_recs[:, exact])
    np.testing.assert_allclose(recs[:, 3:5], g_recs[:, 3:5], rtol=5e-7, atol=0)


@pytest.mark.parametrize("name", ["tiny_unit", "s2_unit"])
@pytest.mark.parametrize("seed", [1, 2])
def test_yolo_detections_vs_reference(golden_dir, name, seed):
    """orc_yolo_detections (get_yolo_detections + correct_yolo_boxes, src/yolo_layer.c:246-277,316-345) on the yolo
    tensors the reference produced, against the detections the reference produced from them."""
    g = np.load(os.path.join(golden_dir, f"{name}_seed{seed}.npz"))
    ly = [int(k[1:].split("_")[0]) for k in g.files if k.endswith("_anchors")]
    assert ly
    for i in ly:
        out = g[f"L{i}_f32"]
        mask = g[f"L{i}_mask"]; anchors = g[f"L{i}_anchors"]
        n = len(mask)
        side = {"tiny_unit": (12, 12, 12), "s2_unit": (2, 2, 24)}[name]  # head h, w, net size
        classes = out.size // (n * side[0] * side[1]) - 5
        for k, (imw, imh, rel, th) in enumerate(DET_CALLS):
            cnt, recs = oracle.yolo_detections(out, n, classes, side[0], side[1], anchors, mask, side[2], side[2], imw, imh, th, rel)
            assert_detections_match(cnt, recs, int(g[f"L{i}_det{k}_count"]), g[f"L{i}_det{k}_recs"])
            assert cnt > 0 or name == "s2_unit"


@pytest.mark.parametrize("tag", ["leaky", "relu6"])
def test_yolov3_tiny_416_real_image_hashes_ref_f32(golden_dir, cfg_dir, tmp_path, tag):
    """BASELINE config[0], real-image half: the reference's test image as its own load_image_color ->
    letterbox_image -> layer-0 dynamic quantiser saw it (tests/golden/realimg_416.npz, data only).  The oracle's quantiser on the
    rebuilt float image returns the committed bytes / scale / zero point, and the oracle net prepared with THAT scale reproduces
    the reference's per-layer hashes of all 24 layers (bit-faithful ref-f32 accumulation)."""
    r = np.load(os.path.join(golden_dir, "realimg_416.npz"))
    g = json.load(open(os.path.join(golden_dir, f"yolov3_tiny_{tag}_realimg.json")))
    xf = synth.dequantized_float_image(r["input_u8"], r["scale"], r["zero_point"], r["fmin"], r["imin"], r["fmax"], r["imax"])
    u8, s, zp = oracle.quantize_image(xf)
    assert np.array_equal(u8.ravel(), r["input_u8"].ravel()) and np.float32(s) == r["scale"] and int(zp) == int(r["zero_point"])
    assert sha(r["input_u8"]) == g["input_sha256"]
    cfg = os.path.join(cfg_dir, g["cfg"])
    wts = str(tmp_path / "w.weights")
    assert synth.synth_weights(cfg, wts, seed=g["weight_seed"])["sha256"] == g["weights_sha256"]
    net = oracle.OracleNet(cfg, wts)
    net.prepare(np.float32(s), int(zp))
    outs = net.forward(r["input_u8"], accum=oracle.ACC_REF_F32)
    for e in g["layers"]:
        i = e["i"]
        if "prep_sha256" in e:
            p = net.p[i]
            assert sha(np.concatenate([p["biases_int32"].view(np.uint8), p["M_value"].view(np.uint8),
                                       p["shift_value"].view(np.uint8)])) == e["prep_sha256"], f"prep {i}"
        if "int32_sha256" in e:
            assert sha(outs[i]["int32"]) == e["int32_sha256"], f"layer {i} int32"
        if "u8_sha256" in e:
            assert sha(outs[i]["u8"]) == e["u8_sha256"], f"layer {i} u8"
        if "f32_sha256" in e and e["type"] == "conv":
            assert sha(outs[i]["f32"]) == e["f32_sha256"], f"layer {i} f32"
