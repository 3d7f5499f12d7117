"""GPU suite, part 4 (`-m gpu`): the drop-in boundary exercised from both sides.

  * the REFERENCE's own structs, cfg parser, weights loader and host prep with its layer.forward_gpu pointers bound to
    libmi355yolo.so by integration/mi355_glue.c (oracle/_ref/libdarknet_ref_mi355.so: unmodified reference objects + the
    glue, built by oracle/build_ref.sh) -- every layer compared with the reference's CPU forward of the same network;
  * `bin/darknet detector test <data> <cfg> <weights> <image>` on an image that is NOT network-sized: device letterbox,
    device quantiser, device box decode, host NMS, the reference's console output -- against the oracle pipeline;
  * the C-ABI RCCL broadcast (mi355_comm_* / mi355_bcast_blob / network_bcast_packed) as a world-1 self-broadcast.
"""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

import oracle
import refdrv
from yolo_quantization_amd import binding, synth

pytestmark = pytest.mark.gpu
ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


@pytest.fixture(scope="module", autouse=True)
def device():
    binding.init(0)


def _layer_tensors(r):
    out = []
    for i, inf in enumerate(r.info):
        d = {}
        if inf["type"] != refdrv.T_YOLO:
            d["u8"] = r.layer_u8(i)
        if inf["type"] == refdrv.T_CONV:
            d["int32"] = r.layer_int32(i)
        if inf["type"] == refdrv.T_YOLO or inf["quant_stop"]:
            d["f32"] = r.layer_f32(i)
        out.append(d)
    return out


@pytest.mark.parametrize("name,size,accum", [("tiny_unit", 12, binding.ACC_EXACT), ("tiny_unit", 12, binding.ACC_REF_F32),
                                             ("yolov3-tiny_quant", 416, binding.ACC_REF_F32), ("yolov3-tiny_quant", 416, binding.ACC_EXACT)],
                         ids=["tiny-exact", "tiny-ref_f32", "yolov3tiny-ref_f32", "yolov3tiny-exact"])
def test_reference_network_with_mi355_forward_gpu(cfg_dir, tmp_path, name, size, accum):
    if not refdrv.available("mi355"):
        pytest.skip("oracle/_ref/libdarknet_ref_mi355.so was not built (needs /root/reference at build time)")
    cfg = os.path.join(cfg_dir, f"{name}.cfg")
    wts = str(tmp_path / "w.weights")
    synth.synth_weights(cfg, wts, seed=1234 if size == 416 else 1)
    x = synth.synth_image_u8(3, size, size, seed=7)
    r = refdrv.RefNet(cfg, wts, omp="mi355")
    r.prepare(synth.image_u8_to_float(x))
    r.forward()                      # the reference's CPU forward (its own l.forward pointers)
    cpu = _layer_tensors(r)
    r.mi355_bind(0, accum, binding.STORE_WRAP)
    r.forward_mi355(pull_all=True)   # the same structs through l.forward_gpu -> libmi355yolo.so
    gpu = _layer_tensors(r)
    r.mi355_unbind()
    first_diff = None
    for i, (a, b) in enumerate(zip(cpu, gpu)):
        for k in a:
            same = np.array_equal(a[k], b[k]) if not (k == "f32" and r.info[i]["type"] == refdrv.T_YOLO) \
                else np.allclose(a[k], b[k], rtol=0, atol=2e-7)
            if not same and first_diff is None:
                first_diff = (i, k)
    if accum == binding.ACC_REF_F32 or size == 12:
        assert first_diff is None, f"first differing tensor: layer {first_diff}"
    else:   # exact integers vs the reference's fp32-rounded accumulators: identical until the first K >= 2304 layer
        assert first_diff is not None and first_diff[0] >= 10, first_diff


def _write_ppm(path, rgb_hwc):
    h, w, _ = rgb_hwc.shape
    with open(path, "wb") as f:
        f.write(b"P6\n%d %d\n255\n" % (w, h))
        f.write(np.ascontiguousarray(rgb_hwc, np.uint8).tobytes())


def test_cli_detector_test_on_a_non_network_sized_image(cfg_dir, tmp_path):
    """`darknet detector test` end to end on a 53x37 PPM into a 12x12 net (and a 640x480 one into yolov3-tiny @416 as a
    smoke run): the printed boxes equal letterbox -> quantise -> forward -> get_yolo_detections -> do_nms_sort of the
    oracle pipeline (NMS: the host's, which test_host_cpu pins against the reference's)."""
    exe = os.path.join(ROOT, "yolo_quantization_amd", "bin", "darknet")
    cfg = os.path.join(cfg_dir, "tiny_unit.cfg")
    wts = str(tmp_path / "w.weights")
    synth.synth_weights(cfg, wts, seed=1)
    rng = np.random.default_rng(4)
    rgb = rng.integers(0, 256, (37, 53, 3), dtype=np.uint8)
    ppm = str(tmp_path / "im.ppm")
    _write_ppm(ppm, rgb)
    names = str(tmp_path / "x.names")
    open(names, "w").write("\n".join(["ant", "bee", "cat", "dog", "eel"]) + "\n")
    data = str(tmp_path / "x.data")
    open(data, "w").write(f"classes= 5\nnames = {names}\n")
    thresh = 0.3
    r = subprocess.run([exe, "detector", "test", data, cfg, wts, ppm, "-thresh", str(thresh), "-boxes"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    assert "Predicted in" in r.stdout
    got = [(int(m.group(1)), float(m.group(2)), [float(m.group(k)) for k in range(3, 7)])
           for m in re.finditer(r"box: class (\d+) prob (\S+) x (\S+) y (\S+) w (\S+) h (\S+)", r.stdout)]
    # oracle pipeline
    im = (rgb.astype(np.float32) / np.float32(255.0)).transpose(2, 0, 1).copy()
    lb = oracle.letterbox_image(im, 12, 12)
    xq, s, zp = oracle.quantize_image(lb)
    onet = oracle.OracleNet(cfg, wts)
    onet.prepare(s, zp)
    outs = onet.forward(xq.reshape(3, 12, 12))
    boxes, probs, objs = [], [], []
    for i, L in enumerate(onet.layers):
        if L.type != "yolo":
            continue
        sec = onet.sections[i]
        classes = int(sec["classes"])
        anchors = np.array([float(v) for v in sec["anchors"].split(",")], np.float32)
        mask = np.array([int(v) for v in sec["mask"].split(",")], np.int32)
        cnt, recs = oracle.yolo_detections(outs[i]["f32"], L.n, classes, L.h, L.w, anchors, mask, 12, 12, 53, 37, thresh, 1)
        boxes.append(recs[:, 1:5]); objs.append(recs[:, 5]); probs.append(recs[:, 6:])
    boxes = np.concatenate(boxes).astype(np.float32); objs = np.concatenate(objs).astype(np.float32)
    probs = np.ascontiguousarray(np.concatenate(probs), np.float32)
    assert f"\n{len(boxes)}\n" in r.stdout, "nboxes line (examples/detector.c:927)"
    H = binding.host()
    H.do_nms_sort_arrays.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_float]
    H.do_nms_sort_arrays(np.ascontiguousarray(boxes).ctypes.data, probs.ctypes.data, objs.ctypes.data, len(boxes), probs.shape[1], 0.45)
    want = sorted((int(j), float(probs[i, j]), [float(v) for v in boxes[i]]) for i in range(len(boxes)) for j in range(probs.shape[1]) if probs[i, j] > thresh)
    got = sorted(got)
    assert len(got) == len(want) and len(want) > 0, (len(got), len(want))
    for (gc, gp, gb), (wc, wp, wb) in zip(got, want):
        assert gc == wc and gp == pytest.approx(wp, rel=1e-6)
        assert gb[:2] == pytest.approx(wb[:2], rel=1e-6) and gb[2:] == pytest.approx(wb[2:], rel=2e-6)   # w / h: exp(), few ulp
    for cls, p, _ in want:   # draw_detections' console lines (src/image.c:255)
        assert f"{['ant', 'bee', 'cat', 'dog', 'eel'][cls]}: {p * 100:.0f}%" in r.stdout
    # packed file round trip through the CLI: same output from -packed as from the weights file
    pk = str(tmp_path / "m.packed")
    r1 = subprocess.run([exe, "detector", "test", data, cfg, wts, ppm, "-thresh", str(thresh), "-boxes", "-save_packed", pk], capture_output=True, text=True, timeout=300)
    r2 = subprocess.run([exe, "detector", "test", data, cfg, "none", ppm, "-thresh", str(thresh), "-boxes", "-packed", pk], capture_output=True, text=True, timeout=300)
    assert r1.returncode == 0 and r2.returncode == 0, r1.stderr + r2.stderr
    assert [l for l in r1.stdout.splitlines() if l.startswith("box:")] == [l for l in r2.stdout.splitlines() if l.startswith("box:")] != []
    # -gpus 0 -bcast: the threaded multi-device path with the RCCL start-up broadcast, on the one device of this box
    r3 = subprocess.run([exe, "detector", "test", data, cfg, wts, ppm, "-thresh", str(thresh), "-boxes", "-gpus", "0", "-bcast", "-batch", "4"],
                        capture_output=True, text=True, timeout=300)
    assert r3.returncode == 0, r3.stderr
    assert [l for l in r3.stdout.splitlines() if l.startswith("box:")] == [l for l in r1.stdout.splitlines() if l.startswith("box:")]
    assert "1 GPUs x batch 4" in r3.stdout
    # full-size smoke: camera-sized image into yolov3-tiny @416
    cfg2 = os.path.join(cfg_dir, "yolov3-tiny_quant.cfg")
    wts2 = str(tmp_path / "w2.weights")
    synth.synth_weights(cfg2, wts2, seed=1234)
    ppm2 = str(tmp_path / "cam.ppm")
    _write_ppm(ppm2, rng.integers(0, 256, (480, 640, 3), dtype=np.uint8))
    r4 = subprocess.run([exe, "detector", "test", data, cfg2, wts2, ppm2, "-thresh", "0.5"], capture_output=True, text=True, timeout=300)
    assert r4.returncode == 0 and "Predicted in" in r4.stdout, r4.stderr
    # several batches in flight from plain C: four executors (network_replica), same detections as the single pass
    r5 = subprocess.run([exe, "detector", "test", data, cfg2, wts2, ppm2, "-thresh", "0.5", "-boxes", "-batch", "8", "-n", "3", "-inflight", "4"],
                        capture_output=True, text=True, timeout=300)
    assert r5.returncode == 0 and "4 batches in flight" in r5.stdout, r5.stderr
    r6 = subprocess.run([exe, "detector", "test", data, cfg2, wts2, ppm2, "-thresh", "0.5", "-boxes", "-batch", "8"], capture_output=True, text=True, timeout=300)
    assert [l for l in r5.stdout.splitlines() if l.startswith("box:")] == [l for l in r6.stdout.splitlines() if l.startswith("box:")]


def test_c_abi_rccl_broadcast_world1(cfg_dir, tmp_path):
    """The rank != root start-up path behind the C ABI on one device: unique id -> communicator (1 rank) -> the root's
    packed bytes broadcast in place in HBM -> a SECOND network (cfg only) imports them from HBM; every layer equals the
    weights-file network.  N > 1 is unmeasured on hardware (1-GPU boxes)."""
    S, H = binding.shim(), binding.host()
    S.mi355_comm_unique_id.argtypes = [C.c_void_p]
    S.mi355_comm_init.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_void_p, C.c_int]
    S.mi355_bcast_blob.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
    S.mi355_comm_destroy.argtypes = [C.c_void_p]
    S.mi355_comm_last_error.restype = C.c_char_p
    H.network_bcast_packed.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int]
    cfg = os.path.join(cfg_dir, "tiny_unit.cfg")
    wts = str(tmp_path / "w.weights")
    synth.synth_weights(cfg, wts, seed=3)
    x = synth.synth_image_u8(3, 12, 12, seed=9, batch=2)
    ident = (C.c_char * 128)()
    assert S.mi355_comm_unique_id(ident) == 0, S.mi355_comm_last_error()
    comm = C.c_void_p()
    assert S.mi355_comm_init(C.byref(comm), 1, ident, 0) == 0, S.mi355_comm_last_error()
    root = binding.Net(cfg, wts, batch=2)
    root.prepare_fixed(1.0 / 255.0, 0)
    packed = root.export_packed()
    dev = binding.DevBuf.from_numpy(packed)
    assert S.mi355_bcast_blob(comm, dev.ptr, packed.nbytes, 0, None) == 0, S.mi355_comm_last_error()
    binding.check(S.mi355_stream_sync(None), "sync")
    assert np.array_equal(dev.to_numpy(np.uint8, packed.nbytes), packed)
    peer = binding.Net(cfg, None, batch=2)
    peer.import_packed_gpu(dev.ptr, packed.nbytes)
    H.network_bcast_packed(root.h, comm, 0, 0)   # the host-level wrapper on the root (exports, uploads, broadcasts)
    for n in (root, peer):
        n.push_input(x); n.forward(); n.sync()
    for i in range(root.n):
        a, b = root.pull(i), peer.pull(i)
        for k in a:
            if k != "int32":
                assert np.array_equal(a[k], b[k]), (i, k)
    assert S.mi355_comm_destroy(comm) == 0
    root.close(); peer.close()
