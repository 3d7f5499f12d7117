"""CPU suite for the product's host side: the C-ABI library loads and exports every declared symbol, the packer and the
plain-C darknet host (cfg parser, .weights reader, integer prep) agree with the oracle / golden fixtures.
No compute kernels are launched here (no GPU in this container)."""
import ctypes as C
import hashlib
import json
import os
import re

import numpy as np
import pytest

import oracle
from yolo_quantization_amd import binding, synth

ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def _declared(header, prefix_re):
    txt = open(os.path.join(ROOT, "include", header)).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(" + prefix_re + r")\s*\(", txt)))


def test_shim_exports_every_declared_symbol():
    names = _declared("mi355_yolo_int8.h", r"mi355_\w+")
    assert len(names) >= 30
    L = binding.shim()
    missing = [n for n in names if not hasattr(L, n)]
    assert not missing, missing


def test_host_exports_every_declared_symbol():
    names = _declared("darknet_q.h", r"[a-z_]+\w*")
    names = [n for n in names if n not in ("defined", "sizeof", "void") and not n.startswith("mi355")]
    L = binding.host()
    missing = [n for n in names if not hasattr(L, n)]
    assert not missing, missing
    assert "forward_network_gpu" in names and "quantization_weights_and_activations" in names


def test_no_device_fails_loudly():
    """On a box without a gfx950 the product must refuse, not fall back."""
    L = binding.shim()
    if L.mi355_device_count() > 0:
        pytest.skip("a GPU is present")
    assert L.mi355_init(0) < 0
    assert b"no HIP device" in L.mi355_last_error() or b"gfx950" in L.mi355_last_error()


def test_quant_multiplier_matches_golden(golden_dir):
    f = np.load(os.path.join(golden_dir, "funcs.npz"))
    H = binding.host()
    for m, m0, sh in zip(f["qm_M"], f["qm_M0"], f["qm_shift"]):
        a = C.c_int32(); b = C.c_int()
        H.quant_multi_smaller_than_one_to_scale_and_shift(float(m), C.byref(a), C.byref(b))
        assert (a.value, b.value) == (int(m0), int(sh))


@pytest.mark.parametrize("name", ["signed", "unit"])
def test_image_quantiser_matches_golden(golden_dir, name):
    f = np.load(os.path.join(golden_dir, "funcs.npz"))
    x = np.ascontiguousarray(f[f"qimg_{name}_x"])
    out = np.zeros(x.size, np.uint8); s = C.c_float(); z = C.c_uint8()
    binding.host().quant_image_with_min_max(x.size, x.ctypes.data, out.ctypes.data, C.byref(s), C.byref(z))
    assert np.float32(s.value) == f[f"qimg_{name}_scale"] and z.value == f[f"qimg_{name}_zp"]
    assert np.array_equal(out, f[f"qimg_{name}_u8"])


@pytest.mark.parametrize("tag", ["leaky", "relu6"])
def test_host_prep_matches_reference_hashes(golden_dir, cfg_dir, tmp_path, tag):
    """cfg parser + .weights reader + integer prep of the plain-C host == the reference's prep (golden SHA-256)."""
    g = json.load(open(os.path.join(golden_dir, f"yolov3_tiny_{tag}.json")))
    cfg = os.path.join(cfg_dir, g["cfg"])
    wts = str(tmp_path / "w.weights")
    synth.synth_weights(cfg, wts, seed=g["weight_seed"])
    net = binding.Net(cfg, wts)
    net.prepare_host_only(1.0 / 255.0, 0)
    _, shapes = synth.layer_shapes(synth.read_cfg(cfg))
    assert net.n == len(shapes)
    for i, (L, e) in enumerate(zip(shapes, g["layers"])):
        inf = net.info[i]
        assert (inf["out_c"], inf["out_h"], inf["out_w"]) == (L.out_c, L.out_h, L.out_w), i
        if "prep_sha256" in e:
            p = net.prep(i)
            h = hashlib.sha256(np.concatenate([p["biases_int32"].view(np.uint8), p["M_value"].view(np.uint8),
                                               p["shift_value"].view(np.uint8)]).tobytes()).hexdigest()
            assert h == e["prep_sha256"], f"layer {i}"
    net.close()


def _unpack_check(blob, wq, zp_w, c, ksize):
    """Decode the packed blob with an independent numpy reading of the documented layout and rebuild the GEMM
    operands; returns (W' [n,Kp] int8 in unit order, cw, dzp)."""
    hdr = np.frombuffer(blob, dtype=np.int32, count=11, offset=4)
    n, cc, ks, mpad, cb, nchunks, upc, spc, ksteps, ktrue, first = [int(v) for v in hdr[:11]]
    offs = np.frombuffer(blob, dtype=np.uint64, count=7, offset=48)  # magic + 11 ints = 48 bytes
    return dict(n=n, c=cc, ksize=ks, mpad=mpad, cb=cb, nchunks=nchunks, upc=upc, spc=spc, ksteps=ksteps, ktrue=ktrue,
                first=first, offs=[int(o) for o in offs])


@pytest.mark.parametrize("n,c,k", [(32, 16, 3), (30, 256, 1), (128, 64, 3), (256, 384, 3), (20, 48, 3), (16, 3, 3)])
def test_pack_blob_reconstructs_exact_accumulators(n, c, k):
    """Algebra check of the offset decomposition on the packed operands, CPU only:
    sum w'x' + d*sum x' + cw  ==  sum (w_u8 - zp_w) x_u8  for random data, via the blob's own contents."""
    rng = np.random.default_rng(n * 1000 + c + k)
    K = c * k * k
    wq = rng.integers(0, 256, (n, K), dtype=np.uint8)
    zp_w = rng.integers(0, 256, n, dtype=np.uint8)
    blob = binding.conv_pack(wq, zp_w, c, k, np.arange(n, dtype=np.int32), np.full(n, 0.5), np.full(n, 2.0 ** -9))
    h = _unpack_check(blob.tobytes(), wq, zp_w, c, k)
    assert (h["n"], h["c"], h["ksize"], h["ktrue"]) == (n, c, k, K)
    off_wp, off_cw, off_dzp, off_bias, off_mval, off_sval, total = h["offs"]
    assert total == blob.size
    cw = np.frombuffer(blob, np.int32, h["mpad"], off_cw)
    dzp = np.frombuffer(blob, np.int32, h["mpad"], off_dzp)
    assert np.array_equal(np.frombuffer(blob, np.int32, n, off_bias), np.arange(n))
    x = rng.integers(0, 256, K, dtype=np.uint8)  # one im2col column in the reference's (ci,ky,kx) order
    want = (wq.astype(np.int64) - zp_w[:, None].astype(np.int64)) @ x.astype(np.int64)
    if h["first"]:
        wp = np.frombuffer(blob, np.uint32, n * 9, off_wp).reshape(n, 9)
        xs = x.reshape(c, 9).astype(np.int64)
        got = np.zeros(n, np.int64)
        for t in range(9):
            for ci in range(3):
                got += ((wp[:, t] >> (8 * ci)) & 0xFF).astype(np.int64) * xs[ci, t]
        got -= (128 - dzp[:n].astype(np.int64)) * x.astype(np.int64).sum()
        assert np.array_equal(got, want)
        return
    cb, bpc = h["cb"], h["cb"] // 16
    wp = np.frombuffer(blob, np.int8, h["mpad"] * h["ksteps"] * 64, off_wp).reshape(h["mpad"] // 16, h["ksteps"], 4, 16, 16)
    xs = x.reshape(c, k * k).astype(np.int64) - 128  # x' per (ci, tap)
    mf = np.zeros(n, np.int64); sx = 0
    for chunk in range(h["nchunks"]):
        for s in range(h["spc"]):
            g = chunk * h["spc"] + s
            for kg in range(4):
                u = 4 * s + kg
                if u >= h["upc"]:
                    assert not wp[:, g, kg].any()
                    continue
                tap, blk = u // bpc, u % bpc
                ci0 = chunk * cb + blk * 16
                xv = xs[ci0:ci0 + 16, tap]
                wv = wp[:, g, kg].reshape(h["mpad"], 16)[:n].astype(np.int64)
                mf += wv @ xv
                sx += xv.sum()
    got = mf + dzp[:n].astype(np.int64) * sx + cw[:n].astype(np.int64)
    assert np.array_equal(got, want)
    assert np.abs(got).max() < 2 ** 31
