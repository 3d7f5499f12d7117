"""Synthetic model / input generator in the reference's on-disk formats.

The reference ships neither weights nor its `yolov3-tiny_quant.cfg`, so tests and the
benchmark synthesise a seeded model and write it in the reference's `.weights` layout
(`/root/reference/src/parser.c:884-929` writer, `:1124-1199` reader; header `:970-976`, `:1219-1225`):

    int32 major=0, minor=2, revision=0; uint64 seen
    per [convolutional]: f32 biases[n]; if batch_normalize: f32 scales[n], rolling_mean[n], rolling_variance[n];
                         f32 in_scale; u8 in_zp; f32 act_scale; u8 act_zp; f32 w_scale[n]; u8 w_zp[n];
                         u8 weights_uint8[n*c*k*k]; f32 weights[n*c*k*k]
    per [maxpool] (always), per quantized [route] with >1 input, per quantized [upsample]: f32 act_scale; u8 act_zp
    per [shortcut] (quantized residual add -- this build's own op, the reference has none): f32 act_scale; u8 act_zp

Recipe: He-normal float weights, BN scale U[.8,1.2], mean U[-.05,.05], var U[.5,1.5], bias
U[-.1,.1]; per-channel min/max (incl. 0) uint8 quantisation of the BN-folded weights; activation (scale, zp) =
(6/255, 0) relu6, (6.6/255, 23) leaky, (16/255, 128) linear; pools / routes / upsamples inherit their input's.
"""
from __future__ import annotations

import hashlib
import struct
from dataclasses import dataclass, field

import numpy as np

ACT_QPARAMS = {
    "relu6": (np.float32(6.0 / 255.0), 0),
    "relu": (np.float32(6.0 / 255.0), 0),
    "leaky": (np.float32(6.6 / 255.0), 23),
    "linear": (np.float32(16.0 / 255.0), 128),
}


# ----------------------------------------------------------------------------------------------- cfg reading
def read_cfg(path: str) -> list[dict]:
    """Minimal INI reader with the semantics of the reference's read_cfg (src/parser.c:817-860):
    '[section]' starts a section, 'key=value' lines, '#', ';' and blank lines skipped, spaces stripped."""
    sections: list[dict] = []
    with open(path) as f:
        for raw in f:
            line = raw.strip().replace(" ", "")
            if not line or line[0] in "#;":
                continue
            if line[0] == "[":
                sections.append({"type": line})
            else:
                k, _, v = line.partition("=")
                sections[-1][k] = v
    return sections


@dataclass
class LayerShape:
    type: str
    c: int = 0
    h: int = 0
    w: int = 0
    out_c: int = 0
    out_h: int = 0
    out_w: int = 0
    n: int = 0
    size: int = 0
    stride: int = 1
    pad: int = 0
    activation: str = ""
    batch_normalize: int = 0
    quantized: int = 0
    quant_stop: int = 0
    inputs: list[int] = field(default_factory=list)  # route sources (absolute indices)


def layer_shapes(sections: list[dict]) -> tuple[dict, list[LayerShape]]:
    """Shape inference following parse_network_cfg (src/parser.c:682-815)."""
    net = sections[0]
    assert net["type"] in ("[net]", "[network]")
    h, w, c = int(net.get("height", 0)), int(net.get("width", 0)), int(net.get("channels", 0))
    layers: list[LayerShape] = []
    for idx, s in enumerate(sections[1:]):
        t = s["type"]
        q = int(s.get("quantized", 0))
        qs = int(s.get("quant_stop", 0))
        if t == "[convolutional]":
            n = int(s.get("filters", 1)); size = int(s.get("size", 1)); stride = int(s.get("stride", 1))
            pad = int(s.get("pad", 0)); padding = int(s.get("padding", 0))
            if pad:
                padding = size // 2
            oh = (h + 2 * padding - size) // stride + 1
            ow = (w + 2 * padding - size) // stride + 1
            L = LayerShape("conv", c, h, w, n, oh, ow, n, size, stride, padding, s.get("activation", "logistic"),
                           int(s.get("batch_normalize", 0)), q, qs)
        elif t == "[maxpool]":
            stride = int(s.get("stride", 1)); size = int(s.get("size", stride))
            padding = int(s.get("padding", size - 1))
            oh = (h + padding - size) // stride + 1
            ow = (w + padding - size) // stride + 1
            L = LayerShape("maxpool", c, h, w, c, oh, ow, 0, size, stride, padding, "", 0, q, qs)
        elif t == "[upsample]":
            stride = int(s.get("stride", 2))
            L = LayerShape("upsample", c, h, w, c, h * stride, w * stride, 0, 0, stride, 0, "", 0, q, qs)
        elif t == "[route]":
            srcs = [int(x) for x in s["layers"].split(",")]
            srcs = [x if x >= 0 else idx + x for x in srcs]
            f = layers[srcs[0]]
            oc = sum(layers[i].out_c for i in srcs)
            L = LayerShape("route", 0, 0, 0, oc, f.out_h, f.out_w, len(srcs), 0, 1, 0, "", 0, q, qs, srcs)
        elif t == "[shortcut]":
            frm = int(s["from"])
            frm = frm if frm >= 0 else idx + frm
            if q and int(s.get("first_time", 0)):  # same rule as host/parser.c parse_shortcut: the sum has no record to inherit
                raise ValueError("[shortcut] quantized=1 first_time=1: the residual add needs its own activation record")
            L = LayerShape("shortcut", c, h, w, c, h, w, 0, 0, 1, 0, s.get("activation", "linear"), 0, q, qs, [idx - 1, frm])
            assert (layers[frm].out_c, layers[frm].out_h, layers[frm].out_w) == (c, h, w), "shortcut inputs must share dims"
        elif t == "[yolo]":
            L = LayerShape("yolo", c, h, w, c, h, w)
            L.n = len(s.get("mask", "0").split(","))
        else:
            raise ValueError(f"layer type {t} is outside the INT8 hot path (SURVEY.md §2 row 20)")
        layers.append(L)
        h, w, c = L.out_h, L.out_w, L.out_c
    return net, layers


# ------------------------------------------------------------------------------------------- model synthesis
def _quantize_per_channel(wf: np.ndarray) -> tuple[np.ndarray, np.ndarray, np.ndarray]:
    """min/max (including 0) uint8 quantisation per output channel; wf is [n, K] float32.
    Same nudging as quant_weights_with_min_max_channel (src/blas.c:108-168), vectorised in float32."""
    mn = np.minimum(wf.min(axis=1), np.float32(0))
    mx = np.maximum(wf.max(axis=1), np.float32(0))
    scale = ((mx - mn) / np.float32(255.0)).astype(np.float32)
    izp = np.float64(0.0) - mn.astype(np.float64) / scale.astype(np.float64)
    zp = np.where(izp < 0, 0, np.where(izp > 255, 255, np.floor(np.abs(izp) + 0.5) * np.sign(izp))).astype(np.uint8)
    t = np.floor(np.abs(wf / scale[:, None]) + np.float32(0.5)) * np.sign(wf) + zp[:, None].astype(np.float32)
    q = np.clip(t, 0, 255).astype(np.uint8)
    return q, scale, zp


def synth_weights(cfg_path: str, out_path: str, seed: int = 1234, act_gain: float = 1.0, small_m_channels: int = 0) -> dict:
    """Write a seeded synthetic `.weights` file for `cfg_path`. Returns {'sha256': ..., 'layers': [...]}.
    act_gain > 1 divides every activation scale by that factor so that requantised values overflow 0..255 and
    exercise the reference's wrap-on-store behaviour (src/convolutional_layer.c:737-749).
    small_m_channels > 0 shrinks the float weights of the first that many filters of every convolution by 64: their per-channel weight
    scale -- and with it the requantisation multiplier M = s_in * s_w / s_out (src/blas.c:313) -- falls to ~2e-5, below what the kernels'
    integer requantisation accepts (common.h intrq_make): the 'unfriendly model' of bench.py --small-m-channels."""
    rng = np.random.default_rng(seed)
    _, layers = layer_shapes(read_cfg(cfg_path))
    act_q: list[tuple[np.float32, int]] = []
    blob = bytearray()
    blob += struct.pack("<iiiQ", 0, 2, 0, 0)
    info = []
    for i, L in enumerate(layers):
        if L.type == "conv":
            K = L.c * L.size * L.size
            w = (rng.standard_normal((L.n, K)) * np.sqrt(2.0 / K)).astype(np.float32)
            if small_m_channels > 0:
                w[:min(small_m_channels, L.n)] *= np.float32(1.0 / 64.0)
            bias = rng.uniform(-0.1, 0.1, L.n).astype(np.float32)
            blob += bias.tobytes()
            if L.batch_normalize:
                sc = rng.uniform(0.8, 1.2, L.n).astype(np.float32)
                mean = rng.uniform(-0.05, 0.05, L.n).astype(np.float32)
                var = rng.uniform(0.5, 1.5, L.n).astype(np.float32)
                blob += sc.tobytes() + mean.tobytes() + var.tobytes()
                wf = (w * sc[:, None] / (np.sqrt(var)[:, None] + np.float32(1e-6))).astype(np.float32)
            else:
                wf = w
            q, wscale, wzp = _quantize_per_channel(wf)
            a_s, a_zp = ACT_QPARAMS[L.activation]
            a_s = np.float32(a_s / np.float32(act_gain))
            in_s, in_zp = act_q[i - 1] if i > 0 else (np.float32(1.0 / 255.0), 0)
            blob += struct.pack("<fBfB", float(in_s), int(in_zp), float(a_s), int(a_zp))
            blob += wscale.tobytes() + wzp.tobytes() + q.tobytes() + w.tobytes()
            act_q.append((a_s, a_zp))
            info.append({"i": i, "type": "conv", "K": K, "n": L.n})
        elif L.type == "maxpool":
            a = act_q[i - 1]
            blob += struct.pack("<fB", float(a[0]), int(a[1]))
            act_q.append(a)
        elif L.type == "upsample":
            a = act_q[i - 1]
            if L.quantized:
                blob += struct.pack("<fB", float(a[0]), int(a[1]))
            act_q.append(a)
        elif L.type == "route":
            a = act_q[L.inputs[0]]
            if L.quantized and len(L.inputs) > 1:
                blob += struct.pack("<fB", float(a[0]), int(a[1]))
            act_q.append(a)
        elif L.type == "shortcut":
            # the sum's own record: zero point of the full real range of a + b, scale = 0.6 of that range / 255 (a calibrated
            # scale is tighter than the worst case; this also makes both ends saturate in the tests)
            (sa, za), (sb, zb) = act_q[L.inputs[0]], act_q[L.inputs[1]]
            lo = -(np.float64(sa) * za + np.float64(sb) * zb)
            hi = np.float64(sa) * (255 - za) + np.float64(sb) * (255 - zb)
            zp = int(min(255, max(0, np.floor(-lo / ((hi - lo) / 255.0) + 0.5))))
            a = (np.float32((hi - lo) * 0.6 / 255.0), zp)
            blob += struct.pack("<fB", float(a[0]), int(a[1]))
            act_q.append(a)
        else:  # yolo
            act_q.append(act_q[i - 1])
    with open(out_path, "wb") as f:
        f.write(blob)
    return {"sha256": hashlib.sha256(blob).hexdigest(), "layers": info, "bytes": len(blob)}


def synth_image_u8(c: int, h: int, w: int, seed: int = 7, batch: int | None = None) -> np.ndarray:
    """Seeded U{0..255} uint8 image(s), NCHW (CHW when batch is None). Pixel [0,0,0]=0 and [0,0,1]=255 are pinned
    so that the reference's dynamic layer-0 quantiser (src/blas.c:279, :115-150) sees min=0,max=1 on x/255 and
    reproduces the bytes exactly (scale 1/255, zero point 0)."""
    rng = np.random.default_rng(seed)
    shape = (c, h, w) if batch is None else (batch, c, h, w)
    x = rng.integers(0, 256, size=shape, dtype=np.uint8)
    flat = x.reshape(-1, c * h * w) if batch is not None else x.reshape(1, -1)
    flat[:, 0] = 0
    flat[:, 1] = 255
    return x


def image_u8_to_float(x: np.ndarray) -> np.ndarray:
    return (x.astype(np.float32) / np.float32(255.0)).astype(np.float32)


def dequantized_float_image(u8, scale, zero_point, fmin, imin, fmax, imax) -> np.ndarray:
    """A float image on which the reference's layer-0 dynamic quantiser (src/blas.c:108-168) returns exactly (u8, scale, zero_point):
    every element is its own dequantised value (u8 - zp) * scale, except the two positions that carried the original image's extreme
    floats (the quantiser's scale / zero point are functions of min and max alone).  tests/golden/realimg_416.npz holds the arguments;
    tests/golden/make_golden.py checks the claim against the compiled reference when it writes them."""
    x = ((np.asarray(u8).astype(np.int32).ravel() - int(zero_point)).astype(np.float32) * np.float32(scale)).astype(np.float32)
    x[int(imin)] = np.float32(fmin)
    x[int(imax)] = np.float32(fmax)
    return x.reshape(np.asarray(u8).shape)
