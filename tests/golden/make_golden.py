#!/usr/bin/env python3
"""Generate the committed golden fixtures by RUNNING THE REFERENCE ITSELF (oracle/_ref = the unmodified
/root/reference sources compiled by oracle/build_ref.sh, Makefile-default flags: GPU=0 QUANTIZATION=1 -Ofast).

  python tests/golden/make_golden.py            # rewrites tests/golden/*.npz, *.json

Fixtures are data only (inputs + the reference's outputs):
  tiny_unit_seed{1,2}.npz   full per-layer tensors of cfg/tiny_unit.cfg (seed 2 = act_gain 8: wrap-on-store cases)
  s2_unit_seed{1,2}.npz     the same for cfg/s2_unit.cfg (stride-2 3x3 convolutions)
  funcs.npz                 known-answer vectors for gemm_nn_uint8_int32_te / im2col_cpu_uint8 /
                            quant_multi_smaller_than_one_to_scale_and_shift / quant_weights_with_min_max_channel
  nms.npz                   do_nms_sort (src/box.c:58-89) on clustered random detections
  realimg_416.npz           the reference's test image as the NETWORK sees it: quantised uint8 input + scale / zero point of the
                            layer-0 dynamic quantiser (+ the two extreme floats): data, no image file
  yolov3_tiny_{leaky,relu6}_realimg.json   the per-layer hashes below for that input
  yolov3_tiny_{leaky,relu6}.json   per-layer SHA-256 of output_int32 / output_uint8_final / output (f32) of the
                            24-layer net @416x416 on the seeded synthetic model + seeded uint8 image, the host-prep
                            arrays' SHA-256, and the full head tensors' uint8 bytes (L15, L22) as hex.
"""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.normpath(os.path.join(HERE, "..", ".."))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from yolo_quantization_amd import synth  # noqa: E402
import refdrv  # noqa: E402
import oracle  # noqa: E402  (only for the fp32-exact-regime mask: pass-1 sums of the reference's own layer inputs)

WEIGHT_SEED, IMAGE_SEED = 1234, 7
LOWRANGE_SEED, LOWRANGE_SHIFT = 500, 5   # function-level vectors of the deep layers: input bytes in 0..7


DET_CALLS = [(640, 480, 1, 0.5), (300, 500, 0, 0.3), (416, 416, 1, 0.6)]  # (image w, h, relative, thresh)


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def run_ref(cfg, wts, img_seed):
    net = refdrv.RefNet(cfg, wts)
    _, layers = synth.layer_shapes(synth.read_cfg(cfg))
    L0 = layers[0]
    x = synth.synth_image_u8(L0.c, L0.h, L0.w, seed=img_seed)
    xq = net.prepare(synth.image_u8_to_float(x))
    assert np.array_equal(xq, x.ravel()), "layer-0 quantiser is expected to be the identity on pinned images"
    net.forward()
    return net, layers, x


def tiny_unit(seed, act_gain, name="tiny_unit"):
    cfg = os.path.join(ROOT, "cfg", f"{name}.cfg")
    wts = f"/tmp/golden_{name}_{seed}.weights"
    meta = synth.synth_weights(cfg, wts, seed=seed, act_gain=act_gain)
    net, layers, x = run_ref(cfg, wts, img_seed=100 + seed)
    d = {"input_u8": x, "weights_sha256": np.array(meta["sha256"]), "seed": np.array(seed),
         "act_gain": np.array(act_gain), "img_seed": np.array(100 + seed)}
    for i, L in enumerate(layers):
        if L.type == "conv":
            d[f"L{i}_int32"] = net.layer_int32(i)
            p = net.prep(i)
            for k in ("biases_int32", "M_value", "shift_value", "M0", "shift"):
                d[f"L{i}_{k}"] = p[k]
        if L.type != "yolo":
            d[f"L{i}_u8"] = net.layer_u8(i)
        if L.quant_stop or L.type == "yolo":
            d[f"L{i}_f32"] = net.layer_f32(i)
    for i, L in enumerate(layers):  # the reference's get_yolo_detections on every yolo layer, three call shapes
        if L.type == "yolo":
            b, m = net.yolo_params(i)
            d[f"L{i}_anchors"] = b; d[f"L{i}_mask"] = m
            for k, (imw, imh, rel, th) in enumerate(DET_CALLS):
                cnt, recs = net.yolo_detections(i, L.c // L.n - 5, imw, imh, th, rel, L.n * L.h * L.w)
                d[f"L{i}_det{k}_count"] = np.array(cnt); d[f"L{i}_det{k}_recs"] = recs
    np.savez_compressed(os.path.join(HERE, f"{name}_seed{seed}.npz"), **d)
    print(f"{name} seed {seed}: wrote {len(d)} arrays")


LBX_CASES = {"wide": (3, 37, 53, 48, 48), "tall": (3, 64, 20, 32, 32), "up": (3, 9, 7, 40, 40), "same": (3, 24, 24, 24, 24),
             "odd": (3, 31, 45, 26, 38), "gray": (1, 50, 33, 20, 28)}


def funcs():
    L = refdrv.lib()
    rng = np.random.default_rng(42)
    d = {}
    # GEMM: a small exact case and a case whose running sums exceed 2^24 (fp32 rounding visible)
    for name, (M, N, K, lo, hi) in {"small": (5, 7, 33, 0, 256), "big": (3, 11, 700, 180, 256)}.items():
        A = rng.integers(lo, hi, (M, K), dtype=np.uint8)
        B = rng.integers(lo, hi, (K, N), dtype=np.uint8)
        Z = np.repeat(rng.integers(100, 157, (M, 1), dtype=np.uint8), K, axis=1)
        Cm = np.zeros((M, N), np.int32)
        L.refdrv_gemm_u8(M, N, K, 1.0, A.ctypes.data, K, B.ctypes.data, N, 0, Cm.ctypes.data, N)
        C1 = Cm.copy()
        L.refdrv_gemm_u8(M, N, K, -1.0, Z.ctypes.data, K, B.ctypes.data, N, 1, Cm.ctypes.data, N)
        d[f"gemm_{name}_A"] = A; d[f"gemm_{name}_B"] = B; d[f"gemm_{name}_Z"] = Z
        d[f"gemm_{name}_C1"] = C1; d[f"gemm_{name}_C2"] = Cm
    # im2col 3x3 pad 1 with non-zero pad value, and stride 2
    im = rng.integers(0, 256, (3, 5, 6), dtype=np.uint8)
    for name, (k, s, p, pv) in {"k3s1": (3, 1, 1, 23), "k3s2": (3, 2, 1, 128)}.items():
        oh = (5 + 2 * p - k) // s + 1; ow = (6 + 2 * p - k) // s + 1
        col = np.zeros((3 * k * k, oh * ow), np.uint8)
        L.refdrv_im2col_u8(im.ctypes.data, 3, 5, 6, k, s, p, col.ctypes.data, pv)
        d[f"im2col_{name}"] = col
    d["im2col_im"] = im
    # multiplier decomposition
    ms = np.concatenate([rng.uniform(1e-6, 0.999, 200), [0.5, 0.25, 0.99999994, 0.49999997, 1e-7, 0.75]]).astype(np.float32)
    m0 = np.zeros(ms.size, np.int32); sh = np.zeros(ms.size, np.int32)
    import ctypes as C
    for i, m in enumerate(ms):
        a = C.c_int32(); b = C.c_int()
        L.refdrv_quant_multiplier(float(m), C.byref(a), C.byref(b))
        m0[i] = a.value; sh[i] = b.value
    d["qm_M"] = ms; d["qm_M0"] = m0; d["qm_shift"] = sh
    # image quantiser: a signed-range float image (non-trivial zero point) and a [0,1] image
    for name, x in {"signed": rng.normal(0.2, 0.7, 500).astype(np.float32),
                    "unit": rng.uniform(0, 1, 500).astype(np.float32)}.items():
        out = np.zeros(x.size, np.uint8); s = C.c_float(); z = C.c_uint8()
        xx = x.copy()
        L.refdrv_quantize_image(xx.ctypes.data, x.size, out.ctypes.data, C.byref(s), C.byref(z))
        d[f"qimg_{name}_x"] = x; d[f"qimg_{name}_u8"] = out
        d[f"qimg_{name}_scale"] = np.float32(s.value); d[f"qimg_{name}_zp"] = np.uint8(z.value)
    # letterbox_image (bilinear resize + centre on 0.5 grey): wide, tall, up-scaled, same-size and odd-sized sources
    L.refdrv_letterbox.argtypes = [C.c_void_p] + [C.c_int] * 5 + [C.c_void_p]
    for name, (c, imh, imw, h, w) in LBX_CASES.items():
        im = rng.uniform(0, 1, (c, imh, imw)).astype(np.float32)
        out = np.zeros((c, h, w), np.float32)
        src = im.copy()
        L.refdrv_letterbox(src.ctypes.data, imw, imh, c, w, h, out.ctypes.data)
        d[f"lbx_{name}_im"] = im; d[f"lbx_{name}_out"] = out
    np.savez_compressed(os.path.join(HERE, "funcs.npz"), **d)
    print("funcs: wrote", len(d), "arrays")


def nms():
    """do_nms_sort (src/box.c:58-89) known answers: clustered random boxes (many overlaps), a few zero-objectness rows."""
    import ctypes as C
    L = refdrv.lib()
    L.refdrv_nms_sort.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_float]
    rng = np.random.default_rng(77)
    d = {}
    for name, (n, classes, thresh) in {"a": (60, 5, 0.45), "b": (200, 3, 0.3), "c": (7, 80, 0.45), "empty": (0, 4, 0.45)}.items():
        centres = rng.uniform(0.2, 0.8, (max(n // 6, 1), 2))
        idx = rng.integers(0, len(centres), n)
        boxes = np.concatenate([centres[idx] + rng.normal(0, 0.03, (n, 2)), rng.uniform(0.05, 0.3, (n, 2))], axis=1).astype(np.float32)
        obj = rng.uniform(0.3, 1.0, n).astype(np.float32)
        if n > 10:
            obj[rng.integers(0, n, n // 10)] = 0
        probs = (obj[:, None] * rng.uniform(0, 1, (n, classes))).astype(np.float32)
        probs[probs < 0.25] = 0
        out = probs.copy()
        L.refdrv_nms_sort(boxes.ctypes.data, out.ctypes.data, obj.ctypes.data, n, classes, thresh)
        d[f"{name}_boxes"] = boxes; d[f"{name}_obj"] = obj; d[f"{name}_probs"] = probs; d[f"{name}_out"] = out
        d[f"{name}_thresh"] = np.float32(thresh)
        print(f"nms {name}: {int((probs > 0).sum())} scores in, {int((out > 0).sum())} kept")
    np.savez_compressed(os.path.join(HERE, "nms.npz"), **d)


REAL_IMAGE = "/root/reference/test_image/000044.jpg"


def real_image():
    """BASELINE config[0]'s real-image half: the reference's own test image through the reference's own
    load_image_color -> letterbox_image(416, 416) (ref examples/detector.c:903-904) and its layer-0 dynamic quantiser
    (ref src/blas.c:279 -> :108-168).  Committed as DATA: the quantised uint8 network input, the quantiser's scale / zero point, and the
    two extreme floats of the letterboxed image with their positions -- enough to rebuild a float image on which the reference's
    quantiser (checked here) returns the same scale, zero point and bytes: synth.dequantized_float_image.  No image file is stored."""
    import ctypes as C
    im = refdrv.load_image_color(REAL_IMAGE)
    lb = refdrv.letterbox(im, 416, 416)
    L = refdrv.lib()

    def quantize(xf):
        xx = np.ascontiguousarray(xf, np.float32).ravel().copy()
        out = np.empty(xx.size, np.uint8)
        s_, z_ = C.c_float(), C.c_uint8()
        L.refdrv_quantize_image(xx.ctypes.data, xx.size, out.ctypes.data, C.byref(s_), C.byref(z_))
        return out, np.float32(s_.value), int(z_.value)
    u8, s_, z_ = quantize(lb)
    imin, imax = int(np.argmin(lb)), int(np.argmax(lb))
    d = {"input_u8": u8.reshape(3, 416, 416), "scale": np.float32(s_), "zero_point": np.uint8(z_), "fmin": np.float32(lb.ravel()[imin]),
         "fmax": np.float32(lb.ravel()[imax]), "imin": np.int64(imin), "imax": np.int64(imax), "source_w": np.int32(im.shape[2]),
         "source_h": np.int32(im.shape[1]), "letterbox_sha256": np.array(sha(lb))}
    xr = synth.dequantized_float_image(d["input_u8"], d["scale"], d["zero_point"], d["fmin"], d["imin"], d["fmax"], d["imax"])
    u2, s2, z2 = quantize(xr)
    assert np.array_equal(u2, u8) and s2 == s_ and z2 == z_, "the rebuilt float image must quantise like the letterboxed one"
    np.savez_compressed(os.path.join(HERE, "realimg_416.npz"), **d)
    print(f"realimg_416: {im.shape[2]}x{im.shape[1]} -> 416x416, scale {s_!r}, zero point {z_}, bytes {u8.min()}..{u8.max()}")
    return lb, d


def yolov3_tiny(tag, cfgname, real=None):
    cfg = os.path.join(ROOT, "cfg", cfgname)
    wts = f"/tmp/golden_{tag}.weights"
    meta = synth.synth_weights(cfg, wts, seed=WEIGHT_SEED)
    if real is None:
        net, layers, x = run_ref(cfg, wts, img_seed=IMAGE_SEED)
    else:  # the letterboxed real image through the reference's dynamic layer-0 quantiser (test_detector's order: prep, then predict)
        lb, rd = real
        net = refdrv.RefNet(cfg, wts)
        _, layers = synth.layer_shapes(synth.read_cfg(cfg))
        xq = net.prepare(lb)
        assert np.array_equal(xq, rd["input_u8"].ravel())
        p0 = net.prep(0)
        assert p0["zp_in"] == int(rd["zero_point"])
        net.forward()
        x = rd["input_u8"]
        tag = tag + "_realimg"
    wrec = oracle.read_weights(wts, layers)
    out = {"cfg": cfgname, "weight_seed": WEIGHT_SEED, "image_seed": IMAGE_SEED if real is None else None, "weights_sha256": meta["sha256"],
           "input_sha256": sha(x), "oracle": "reference default build (GPU=0 QUANTIZATION=1 -Ofast)", "layers": []}
    if real is not None:
        out["image"] = "ref test_image/000044.jpg -> load_image_color -> letterbox_image(416,416) -> layer-0 dynamic quantiser; data in realimg_416.npz"
        out["input_scale"] = float(real[1]["scale"]); out["input_zero_point"] = int(real[1]["zero_point"])
    for i, L in enumerate(layers):
        e = {"i": i, "type": L.type}
        if L.type == "conv":
            a = net.layer_int32(i)
            e["int32_sha256"] = sha(a)
            e["int32_min"] = int(a.min()); e["int32_max"] = int(a.max())
            p = net.prep(i)
            e["prep_sha256"] = sha(np.concatenate([p["biases_int32"].view(np.uint8), p["M_value"].view(np.uint8),
                                                   p["shift_value"].view(np.uint8)]))
            # the "fp32-exact regime" of the reference's accumulators.  With the reference's OWN input to
            # this layer (teacher forcing), an element is provably exact in src/gemm.c:279-299 when its pass-1 sum
            # sum_k w_u8*x_u8 (all terms >= 0, so the final sum bounds every prefix) and its result stay within 2^24.
            # Committed: the mask's hash and size, the reference's int32 / uint8 tensors with the elements outside the
            # mask zeroed (hashes), and how many elements of the exact-integer result differ from the reference at all.
            xin = x if i == 0 else net.layer_u8(i - 1).reshape(L.c, L.h, L.w)
            d = wrec[i]
            ex, s1 = oracle.conv_acc(xin, d["wq"], d["zp_w"], L.size, L.stride, L.pad, p["zp_in"], oracle.ACC_EXACT, want_s1=True)
            mask = (s1 <= 2 ** 24) & (np.abs(ex.astype(np.int64)) <= 2 ** 24)
            ref = a.reshape(ex.shape)
            assert not ((ex != ref) & mask).any(), f"layer {i}: exact != reference inside the fp32-exact regime"
            e["exact_mask_count"] = int(mask.sum())
            e["exact_mask_sha256"] = sha(np.packbits(mask.ravel()))
            e["ref_int32_masked_sha256"] = sha(np.where(mask, ref, 0).astype(np.int32))
            e["ref_u8_masked_sha256"] = sha(np.where(mask, net.layer_u8(i).reshape(ex.shape), 0).astype(np.uint8))
            e["exact_vs_ref_mismatch"] = int((ex != ref).sum())
            e["exact_vs_ref_max_abs_diff"] = int(np.abs(ex.astype(np.int64) - ref).max())
            e["M0_lut0"], e["M0_right_shift_lut0"] = net.leaky_lut(i)   # src/blas.c:318-323 (the MKL path's LEAKY multiplier)
        if L.type != "yolo":
            u = net.layer_u8(i)
            e["u8_sha256"] = sha(u)
            e["u8_head"] = u[:16].tolist()
        if L.quant_stop or L.type == "yolo":
            e["f32_sha256"] = sha(net.layer_f32(i))
        if L.type == "conv" and L.quant_stop:
            e["u8_hex"] = net.layer_u8(i).tobytes().hex()
        out["layers"].append(e)
    # Function-level known answers for the deep layers (K >= 2304), whose accumulators the fp32 GEMM rounds on the net's own
    # activations: the same layers on a LOW-RANGE input (seeded bytes >> LOWRANGE_SHIFT), where every pass-1 sum stays
    # below 2^24, so the reference's tensors are exact integers and can pin the exact-integer kernels on every element.
    # Run after the whole-net hashes above (a conv's forward only overwrites that layer's own outputs).
    for i, L in enumerate(layers):
        if L.type != "conv" or L.c * L.size * L.size < 2304 or real is not None:
            continue
        xl = (synth.synth_image_u8(L.c, L.h, L.w, seed=LOWRANGE_SEED + i) >> LOWRANGE_SHIFT).astype(np.uint8)
        net.forward_layer(i, xl)
        a = net.layer_int32(i); u = net.layer_u8(i)
        p = net.prep(i); d = wrec[i]
        ex, s1 = oracle.conv_acc(xl, d["wq"], d["zp_w"], L.size, L.stride, L.pad, p["zp_in"], oracle.ACC_EXACT, want_s1=True)
        mask = (s1 <= 2 ** 24) & (np.abs(ex.astype(np.int64)) <= 2 ** 24)
        assert mask.all() and np.array_equal(ex.ravel(), a), f"layer {i}: low-range input is meant to keep the fp32 GEMM exact"
        out["layers"][i]["lowrange"] = {"seed": LOWRANGE_SEED + i, "shift": LOWRANGE_SHIFT, "input_sha256": sha(xl),
                                        "int32_sha256": sha(a), "u8_sha256": sha(u), "int32_min": int(a.min()),
                                        "int32_max": int(a.max()), "s1_max": int(s1.max())}
    json.dump(out, open(os.path.join(HERE, f"yolov3_tiny_{tag}.json"), "w"), indent=1)
    print(f"yolov3_tiny_{tag}: wrote {len(out['layers'])} layers")


if __name__ == "__main__":
    assert refdrv.available(), "run oracle/build_ref.sh first (needs /root/reference)"
    only = sys.argv[1:]   # e.g. `make_golden.py yolov3_tiny` rewrites the two JSON files only

    def want(name):
        return not only or name in only
    if want("tiny_unit"):
        tiny_unit(1, 1.0)
        tiny_unit(2, 8.0)
    if want("s2_unit"):
        tiny_unit(1, 1.0, "s2_unit")
        tiny_unit(2, 8.0, "s2_unit")
    if want("funcs"):
        funcs()
    if want("nms"):
        nms()
    if want("yolov3_tiny"):
        yolov3_tiny("leaky", "yolov3-tiny_quant.cfg")
        yolov3_tiny("relu6", "yolov3-tiny_quant_relu6.cfg")
    if want("realimg"):
        ri = real_image()
        yolov3_tiny("leaky", "yolov3-tiny_quant.cfg", real=ri)
        yolov3_tiny("relu6", "yolov3-tiny_quant_relu6.cfg", real=ri)
