"""ctypes bindings over the two product libraries:

  lib/libmi355yolo.so   HIP kernels + C-ABI   (include/mi355_yolo_int8.h)
  lib/libdarknet_q.so   plain-C darknet host   (include/darknet_q.h + host/capi.c accessors)

There is no fallback: if a library is missing or the device is not a gfx950, loading / init raises.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_DIR = os.environ.get("MI355_LIB_DIR") or os.path.join(_HERE, "lib")   # override: A/B runs against another build of the libraries

ACT = {"relu": 1, "linear": 3, "relu6": 8, "leaky": 9}
STORE_WRAP, STORE_SATURATE = 0, 1
ACC_EXACT, ACC_REF_F32 = 0, 1


class MI355Error(RuntimeError):
    pass


class Tensor(C.Structure):
    _fields_ = [("data", C.c_void_p), ("B", C.c_int), ("H", C.c_int), ("W", C.c_int), ("C", C.c_int),
                ("cs", C.c_int), ("lead", C.c_int), ("tail", C.c_int)]


class ConvDesc(C.Structure):
    _fields_ = [("n", C.c_int), ("c", C.c_int), ("ksize", C.c_int), ("stride", C.c_int), ("pad", C.c_int),
                ("activation", C.c_int), ("store_mode", C.c_int), ("accum_mode", C.c_int),
                ("zp_in", C.c_uint8), ("zp_act", C.c_uint8), ("s_act", C.c_float), ("plan", C.c_int), ("epilogue_packed", C.c_int)]


_shim = None
_host = None


def shim():
    global _shim
    if _shim is None:
        path = os.path.join(LIB_DIR, "libmi355yolo.so")
        if not os.path.exists(path):
            raise MI355Error(f"{path} is missing: build it with yolo_quantization_amd/csrc/build.sh "
                             "(__graft_entry__.build()); there is no CPU fallback")
        L = C.CDLL(path, mode=C.RTLD_GLOBAL)
        vp, ci, sz = C.c_void_p, C.c_int, C.c_size_t
        L.mi355_last_error.restype = C.c_char_p
        L.mi355_alloc.argtypes = [C.POINTER(vp), sz]
        L.mi355_free.argtypes = [vp]
        L.mi355_memset.argtypes = [vp, ci, sz, vp]
        for n in ("mi355_h2d", "mi355_d2h", "mi355_d2d"):
            getattr(L, n).argtypes = [vp, vp, sz, vp]
        L.mi355_stream_create.argtypes = [C.POINTER(vp)]
        L.mi355_stream_acquire.argtypes = [C.POINTER(vp)]
        L.mi355_stream_release.argtypes = [vp]
        L.mi355_stream_destroy.argtypes = [vp]
        L.mi355_stream_sync.argtypes = [vp]
        L.mi355_event_create.argtypes = [C.POINTER(vp)]
        L.mi355_event_destroy.argtypes = [vp]
        L.mi355_event_record.argtypes = [vp, vp]
        L.mi355_event_elapsed_ms.argtypes = [vp, vp, C.POINTER(C.c_float)]
        L.mi355_graph_begin.argtypes = [vp]
        L.mi355_graph_end.argtypes = [vp, C.POINTER(vp)]
        L.mi355_graph_launch.argtypes = [vp, vp]
        L.mi355_graph_destroy.argtypes = [vp]
        L.mi355_tensor_describe.restype = sz
        L.mi355_tensor_describe.argtypes = [C.POINTER(Tensor), ci, ci, ci, ci]
        L.mi355_tensor_fill.argtypes = [C.POINTER(Tensor), C.c_uint8, vp]
        L.mi355_nchw_to_tensor.argtypes = [vp, C.POINTER(Tensor), vp]
        L.mi355_tensor_to_nchw.argtypes = [C.POINTER(Tensor), vp, vp]
        L.mi355_conv_pack_size.restype = sz
        L.mi355_conv_pack_size.argtypes = [ci, ci, ci]
        L.mi355_conv_pack.argtypes = [ci, ci, ci, vp, vp, vp, vp, vp, vp]
        if hasattr(L, "mi355_conv_pack_epilogue"):  # (absent from older A/B builds under build_ab/; conv_pack(..., activation) then raises)
            L.mi355_conv_pack_epilogue.argtypes = [ci, ci, ci, ci, ci, vp]
        L.mi355_conv_forward.argtypes = [C.POINTER(ConvDesc), C.POINTER(Tensor), vp, vp, vp, C.POINTER(Tensor), vp, vp, vp]
        L.mi355_conv_set_tile.argtypes = [ci, ci]
        L.mi355_conv_pool_forward.argtypes = [C.POINTER(ConvDesc), C.POINTER(Tensor), vp, C.POINTER(Tensor), C.POINTER(Tensor), vp]
        L.mi355_debug_flags.argtypes = [ci]
        L.mi355_last_conv_kernel.argtypes = []
        L.mi355_last_conv_kernel.restype = ci
        L.mi355_conv_yolo_forward.argtypes = [C.POINTER(ConvDesc), C.POINTER(Tensor), vp, C.POINTER(Tensor), vp, vp, ci, vp]
        L.mi355_conv_upsample_forward.argtypes = [C.POINTER(ConvDesc), C.POINTER(Tensor), vp, C.POINTER(Tensor), ci, vp]
        L.mi355_maxpool_forward.argtypes = [C.POINTER(Tensor), C.POINTER(Tensor), ci, ci, ci, vp]
        L.mi355_upsample_forward.argtypes = [C.POINTER(Tensor), C.POINTER(Tensor), ci, vp]
        L.mi355_route_forward.argtypes = [C.POINTER(C.POINTER(Tensor)), ci, C.POINTER(Tensor), vp]
        L.mi355_yolo_forward.argtypes = [vp, vp, ci, ci, ci, ci, ci, vp]
        L.mi355_dequant_forward.argtypes = [C.POINTER(Tensor), ci, ci, C.c_uint8, C.c_float, vp, ci, ci, vp]
        L.mi355_shortcut_multiplier.argtypes = [C.c_float, C.c_float, vp]
        L.mi355_shortcut_forward.argtypes = [C.POINTER(Tensor), C.POINTER(Tensor), C.POINTER(Tensor), C.c_int32, C.c_int32,
                                             C.c_uint8, C.c_uint8, C.c_uint8, vp]
        L.mi355_image_minmax.argtypes = [vp, C.c_long, vp, vp]
        L.mi355_letterbox_forward.argtypes = [vp, ci, ci, ci, vp, ci, ci, vp]
        L.mi355_image_quantize.argtypes = [vp, C.c_long, C.c_float, ci, vp, vp]
        _shim = L
    return _shim


def check(rc, what=""):
    if rc != 0:
        raise MI355Error(f"{what}: code {rc}: {shim().mi355_last_error().decode()}")


ABI_VERSION = 6  # include/mi355_yolo_int8.h MI355_ABI_VERSION: the struct mirrors above are this version's


def init(device=0):
    got = shim().mi355_abi_version()
    if got != ABI_VERSION:
        raise MI355Error(f"libmi355yolo.so speaks ABI {got}, binding.py mirrors ABI {ABI_VERSION}")
    check(shim().mi355_init(device), "mi355_init")


# ------------------------------------------------------------------------------------- thin device helpers
class DevBuf:
    """Owned device allocation (shim allocator)."""

    def __init__(self, nbytes):
        self.ptr = C.c_void_p()
        self.nbytes = int(nbytes)
        check(shim().mi355_alloc(C.byref(self.ptr), self.nbytes), "alloc")

    @classmethod
    def from_numpy(cls, a):
        a = np.ascontiguousarray(a)
        b = cls(max(a.nbytes, 16))
        check(shim().mi355_h2d(b.ptr, a.ctypes.data, a.nbytes, None), "h2d")
        check(shim().mi355_stream_sync(None), "sync")
        return b

    def to_numpy(self, dtype, count):
        out = np.empty(count, dtype=dtype)
        check(shim().mi355_stream_sync(None), "sync")
        check(shim().mi355_d2h(out.ctypes.data, self.ptr, out.nbytes, None), "d2h")
        check(shim().mi355_stream_sync(None), "sync")
        return out

    def free(self):
        if self.ptr:
            shim().mi355_free(self.ptr)
            self.ptr = C.c_void_p()

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class DevTensor:
    """PHWC activation tensor on the device."""

    def __init__(self, B, H, W, Cc, zero_point=0):
        self.t = Tensor()
        nbytes = shim().mi355_tensor_describe(C.byref(self.t), B, H, W, Cc)
        if not nbytes:
            raise MI355Error("bad tensor dims")
        self.buf = DevBuf(nbytes)
        self.t.data = self.buf.ptr
        check(shim().mi355_tensor_fill(C.byref(self.t), zero_point, None), "fill")

    @classmethod
    def from_nchw(cls, x, zero_point=0):
        x = np.ascontiguousarray(x, np.uint8)
        B, Cc, H, W = x.shape
        t = cls(B, H, W, Cc, zero_point)
        src = DevBuf.from_numpy(x)
        check(shim().mi355_nchw_to_tensor(src.ptr, C.byref(t.t), None), "nchw_to_tensor")
        check(shim().mi355_stream_sync(None), "sync")
        return t

    def to_nchw(self):
        n = self.t.B * self.t.C * self.t.H * self.t.W
        dst = DevBuf(n)
        check(shim().mi355_tensor_to_nchw(C.byref(self.t), dst.ptr, None), "tensor_to_nchw")
        return dst.to_numpy(np.uint8, n).reshape(self.t.B, self.t.C, self.t.H, self.t.W)

    def ref(self):
        return C.byref(self.t)


def conv_pack(wq, zp_w, c, ksize, biases_int32, M_value, shift_value, activation=None, zp_act=None):
    """Host-side packing -> numpy uint8 blob.  With activation / zp_act the blob also gets the conv + maxpool kernels' epilogue
    table (mi355_conv_pack_epilogue); without, those kernels derive the constants per workgroup (same bytes)."""
    wq = np.ascontiguousarray(wq, np.uint8)
    n = wq.shape[0]
    sz = shim().mi355_conv_pack_size(n, c, ksize)
    if not sz:
        raise MI355Error(f"unsupported conv shape n={n} c={c} k={ksize}")
    blob = np.zeros(sz, np.uint8)
    zp_w = np.ascontiguousarray(zp_w, np.uint8)
    b = np.ascontiguousarray(biases_int32, np.int32)
    mv = np.ascontiguousarray(M_value, np.float64)
    sv = np.ascontiguousarray(shift_value, np.float64)
    check(shim().mi355_conv_pack(n, c, ksize, wq.ctypes.data, zp_w.ctypes.data, b.ctypes.data, mv.ctypes.data,
                                 sv.ctypes.data, blob.ctypes.data), "conv_pack")
    if activation is not None:
        check(shim().mi355_conv_pack_epilogue(n, c, ksize, int(activation), int(zp_act), blob.ctypes.data), "conv_pack_epilogue")
    return blob


def conv_forward(x: DevTensor, wq, zp_w, ksize, biases_int32, M_value, shift_value, zp_in, zp_act, s_act,
                 activation, store=STORE_WRAP, accum=ACC_EXACT, want_acc=True, want_f32=False, stride=1):
    """One quantized conv layer through the C-ABI. Returns dict(u8 NCHW, int32 [B,n,HW], f32)."""
    n = wq.shape[0]
    c = x.t.C
    blob = DevBuf.from_numpy(conv_pack(wq, zp_w, c, ksize, biases_int32, M_value, shift_value))
    wraw = DevBuf.from_numpy(np.ascontiguousarray(wq, np.uint8))
    zraw = DevBuf.from_numpy(np.ascontiguousarray(zp_w, np.uint8))
    pad = ksize // 2
    OH, OW = (x.t.H + 2 * pad - ksize) // stride + 1, (x.t.W + 2 * pad - ksize) // stride + 1
    y = DevTensor(x.t.B, OH, OW, n, zp_act)
    cnt = x.t.B * n * OH * OW
    acc = DevBuf(cnt * 4) if want_acc else None
    f32 = DevBuf(cnt * 4) if want_f32 else None
    d = ConvDesc(n, c, ksize, stride, pad, activation, store, accum, zp_in, zp_act, float(s_act))
    check(shim().mi355_conv_forward(C.byref(d), x.ref(), blob.ptr, wraw.ptr, zraw.ptr, y.ref(),
                                    acc.ptr if acc else None, f32.ptr if f32 else None, None), "conv_forward")
    check(shim().mi355_stream_sync(None), "sync")
    out = {"u8": y.to_nchw(), "tensor": y}
    if acc:
        out["int32"] = acc.to_numpy(np.int32, cnt).reshape(x.t.B, n, OH * OW)
    if f32:
        out["f32"] = f32.to_numpy(np.float32, cnt).reshape(x.t.B, n, OH * OW)
    return out


# ------------------------------------------------------------------------------------------- darknet host
def host():
    global _host
    if _host is None:
        shim()  # resolve the dependency first, RTLD_GLOBAL
        path = os.path.join(LIB_DIR, "libdarknet_q.so")
        if not os.path.exists(path):
            raise MI355Error(f"{path} is missing: run `make -C yolo_quantization_amd/host`")
        L = C.CDLL(path)
        vp, ci = C.c_void_p, C.c_int
        L.load_network.restype = vp
        L.load_network.argtypes = [C.c_char_p, C.c_char_p, ci]
        L.parse_network_cfg.restype = vp
        L.parse_network_cfg.argtypes = [C.c_char_p, ci]
        L.set_batch_network.argtypes = [vp, ci]
        L.network_replica.restype = vp
        L.network_replica.argtypes = [vp]
        L.free_network.argtypes = [vp]
        L.quantization_weights_and_activations.argtypes = [vp]
        L.quantization_weights_and_activations_fixed_input.argtypes = [vp, C.c_float, C.c_uint8]
        L.quantization_weights_and_activations_gpu.argtypes = [vp, vp]
        L.network_letterbox_input_gpu.argtypes = [vp, ci, vp, ci, ci]
        L.network_quantize_input_gpu.argtypes = [vp]
        L.quantization_prep_host.argtypes = [vp, C.c_float, C.c_uint8]
        L.forward_network_gpu.argtypes = [vp]
        L.network_predict.restype = vp
        L.network_predict.argtypes = [vp, vp]
        L.push_network_input_uint8.argtypes = [vp, vp]
        L.pull_layer_output.argtypes = [vp, ci]
        L.network_profile_begin.argtypes = [vp, ci]
        L.network_profile_read.argtypes = [vp, vp]
        L.network_yolo_detections_gpu.argtypes = [vp, ci, ci, ci, C.c_float, ci, vp, ci, vp]
        L.network_profile_set_stride.argtypes = [vp, ci]
        L.network_profile_set_phase.argtypes = [vp, ci]
        L.network_selfcheck.argtypes = [vp, ci]
        L.network_selfcheck_result.argtypes = [vp]
        L.network_packed_size.restype = C.c_size_t
        L.network_packed_size.argtypes = [vp]
        L.network_export_packed.argtypes = [vp, vp]
        L.network_import_packed.argtypes = [vp, vp, C.c_size_t]
        L.network_import_packed_gpu.argtypes = [vp, vp, C.c_size_t]
        L.network_import_packed_host.argtypes = [vp, vp, C.c_size_t]
        L.quant_multi_smaller_than_one_to_scale_and_shift.argtypes = [C.c_float, vp, vp]
        L.quant_image_with_min_max.argtypes = [ci, vp, vp, vp, vp]
        for n in ("dnq_net_n", "dnq_net_batch", "dnq_net_inputs"):
            getattr(L, n).argtypes = [vp]
        for n in ("dnq_net_stream", "dnq_net_input_gpu", "dnq_net_input_host", "dnq_net_input_float"):
            getattr(L, n).restype = vp
            getattr(L, n).argtypes = [vp]
        L.dnq_net_set.argtypes = [vp, C.c_char_p, ci]
        L.dnq_layer_info.argtypes = [vp, ci, vp]
        for n in ("dnq_layer_u8", "dnq_layer_int32", "dnq_layer_f32", "dnq_layer_f32_gpu"):
            getattr(L, n).restype = vp
            getattr(L, n).argtypes = [vp, ci]
        L.dnq_layer_prep.argtypes = [vp, ci] + [vp] * 6
        L.dnq_layer_is_fused.argtypes = [vp, ci]
        L.dnq_layer_fuses_next.argtypes = [vp, ci]
        L.dnq_layer_shortcut.argtypes = [vp, ci, vp]
        L.network_save_packed.argtypes = [vp, C.c_char_p]
        L.network_load_packed.argtypes = [vp, C.c_char_p]
        L.dnq_layer_conv_kernel.argtypes = [vp, ci]
        L.dnq_layer_conv_kernel.restype = ci
        _host = L
    return _host


INFO_KEYS = ["type", "out_c", "out_h", "out_w", "c", "h", "w", "n", "size", "stride", "pad", "activation",
             "batch_normalize", "quantized", "quant_stop", "outputs"]
T_CONV, T_MAXPOOL, T_ROUTE, T_SHORTCUT, T_YOLO, T_UPSAMPLE = 0, 3, 8, 13, 23, 26


def _as(ptr, n, dt):
    return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(dt)), shape=(n,))


class Net:
    """The darknet host network (libdarknet_q.so): load_network -> prep -> forward_network_gpu."""

    def __init__(self, cfg, weights=None, batch=1, gpu=0, accum=ACC_EXACT, store=STORE_WRAP, dump_int32=False,
                 use_graph=False, fuse_maxpool=True, keep_head_float=True):
        """keep_head_float: the head convs fused with their yolo layers also store their own float tensors (the library's default is
        not to: nothing but the yolo layer reads them; parity pulls want them)"""
        H = host()
        self.H = H
        self.h = H.load_network(cfg.encode(), weights.encode() if weights else None, 0)
        H.dnq_net_set(self.h, b"gpu_index", gpu)
        H.dnq_net_set(self.h, b"accum_mode", accum)
        H.dnq_net_set(self.h, b"store_mode", store)
        H.dnq_net_set(self.h, b"dump_int32", int(dump_int32))
        H.dnq_net_set(self.h, b"use_graph", int(use_graph))
        H.dnq_net_set(self.h, b"fuse_maxpool", int(fuse_maxpool))
        H.dnq_net_set(self.h, b"keep_head_float", int(keep_head_float))
        self.keep_head_float = bool(keep_head_float)
        if batch != H.dnq_net_batch(self.h):
            H.set_batch_network(self.h, batch)
        self.n = H.dnq_net_n(self.h)
        self.batch = batch
        self.inputs = H.dnq_net_inputs(self.h)
        self.info = []
        for i in range(self.n):
            a = (C.c_int * 16)()
            H.dnq_layer_info(self.h, i, a)
            self.info.append(dict(zip(INFO_KEYS, list(a))))

    def set(self, key, val):
        assert self.H.dnq_net_set(self.h, key.encode(), int(val)) == 0

    def replica(self, default_stream=False):
        """network_replica: a second executor of this prepared model on the same device (own activations, input and HIP
        stream; this network's packed weights).  This network must outlive it.  default_stream: the replica launches on the
        device's default stream (HIP's fourth hardware queue: darknet_q.h replica_default_stream)."""
        if default_stream:
            self.set("replica_default_stream", 1)
        r = object.__new__(Net)
        r.H = self.H
        r.h = self.H.network_replica(self.h)
        r.keep_head_float = self.keep_head_float
        r.n, r.batch, r.inputs, r.info = self.n, self.batch, self.inputs, self.info
        r._parent = self  # keeps the parent alive
        return r

    def prepare_fixed(self, in_scale=1.0 / 255.0, in_zp=0):
        self.H.quantization_weights_and_activations_fixed_input(self.h, np.float32(in_scale), in_zp)

    def prepare_host_only(self, in_scale=1.0 / 255.0, in_zp=0):
        self.H.quantization_prep_host(self.h, np.float32(in_scale), in_zp)

    def prepare_from_float(self, x_float):
        """Reference flow: dynamic layer-0 quantiser on the float image(s) (src/blas.c:279)."""
        x = np.ascontiguousarray(x_float, np.float32).ravel()
        assert x.size == self.batch * self.inputs
        C.memmove(self.H.dnq_net_input_float(self.h), x.ctypes.data, x.nbytes)
        self.H.quantization_weights_and_activations(self.h)
        return _as(self.H.dnq_net_input_host(self.h), self.batch * self.inputs, C.c_uint8).copy()

    def prepare_from_float_gpu(self, x_float):
        """The same with the quantiser on the device: the floats are uploaded as they are, min / max and the per-element
        quantiser run in HBM (quantization_weights_and_activations_gpu).  Returns the uint8 input the device produced."""
        x = np.ascontiguousarray(x_float, np.float32).ravel()
        assert x.size == self.batch * self.inputs
        buf = DevBuf.from_numpy(x)
        self.H.quantization_weights_and_activations_gpu(self.h, buf.ptr)
        self.sync()
        out = np.empty(x.size, np.uint8)
        check(shim().mi355_d2h(out.ctypes.data, self.input_gpu_ptr(), out.nbytes, None), "d2h")
        check(shim().mi355_stream_sync(None), "sync")
        buf.free()
        return out

    def prepare_from_images_gpu(self, images):
        """Device input path: every image (float32 [c][h][w], any size) is uploaded, letterboxed into its batch slot and
        the batch quantised on the device.  Returns the uint8 network input the device produced."""
        assert len(images) == self.batch
        bufs = []
        for slot, im in enumerate(images):
            im = np.ascontiguousarray(im, np.float32)
            b = DevBuf.from_numpy(im)
            bufs.append(b)
            self.H.network_letterbox_input_gpu(self.h, slot, b.ptr, im.shape[2], im.shape[1])
        self.H.network_quantize_input_gpu(self.h)
        self.sync()
        out = np.empty(self.batch * self.inputs, np.uint8)
        check(shim().mi355_d2h(out.ctypes.data, self.input_gpu_ptr(), out.nbytes, None), "d2h")
        check(shim().mi355_stream_sync(None), "sync")
        for b in bufs:
            b.free()
        return out

    def push_input(self, x_u8):
        x = np.ascontiguousarray(x_u8, np.uint8).ravel()
        assert x.size == self.batch * self.inputs
        self._keep = x
        self.H.push_network_input_uint8(self.h, x.ctypes.data)

    def input_gpu_ptr(self):
        return self.H.dnq_net_input_gpu(self.h)

    def stream(self):
        return self.H.dnq_net_stream(self.h)

    def forward(self):
        self.H.forward_network_gpu(self.h)

    def sync(self):
        check(shim().mi355_stream_sync(self.stream()), "sync")

    def pull(self, i):
        self.H.pull_layer_output(self.h, i)
        cnt = self.batch * self.info[i]["outputs"]
        out = {}
        ty = self.info[i]["type"]
        if ty != T_YOLO:
            out["u8"] = _as(self.H.dnq_layer_u8(self.h, i), cnt, C.c_uint8).copy()
        if ty == T_CONV:
            out["int32"] = _as(self.H.dnq_layer_int32(self.h, i), cnt, C.c_int32).copy()
        if ty == T_YOLO or (self.info[i]["quant_stop"] and (self.keep_head_float or ty != T_CONV or not self.fuses_next(i))):
            out["f32"] = _as(self.H.dnq_layer_f32(self.h, i), cnt, C.c_float).copy()
        return out

    def conv_kernel(self, i):
        """mi355_last_conv_kernel code of the kernel that served conv layer i in the last forward pass (5 = conv_rows / conv_igemm)"""
        return int(self.H.dnq_layer_conv_kernel(self.h, i))

    def is_fused(self, i):
        """conv i runs fused with the layer after it and its own uint8 tensor is not stored."""
        return bool(self.H.dnq_layer_is_fused(self.h, i))

    def fuses_next(self, i):
        """layer i + 1 runs inside conv i's kernel (conv i's own tensor may be stored too: a conv + pool whose output a route reads)."""
        return bool(self.H.dnq_layer_fuses_next(self.h, i))

    def prep(self, i):
        n = max(self.info[i]["n"], 1)
        b = np.zeros(n, np.int32); mv = np.zeros(n, np.float64); sv = np.zeros(n, np.float64)
        m0 = np.zeros(n, np.int32); sh = np.zeros(n, np.int32); q = np.zeros(4, np.float32)
        self.H.dnq_layer_prep(self.h, i, b.ctypes.data, mv.ctypes.data, sv.ctypes.data, m0.ctypes.data,
                              sh.ctypes.data, q.ctypes.data)
        return dict(biases_int32=b, M_value=mv, shift_value=sv, M0=m0, shift=sh, s_in=q[0], zp_in=int(q[1]),
                    s_act=q[2], zp_act=int(q[3]))

    def detections(self, i, classes, imw, imh, thresh, relative, max_recs):
        """Box decode of yolo layer i on the device: (counts [B], records [B][max_recs][6 + classes], reference order)."""
        recs = np.zeros((self.batch, max_recs, 6 + classes), np.float32)
        counts = np.zeros(self.batch, np.int32)
        self.H.network_yolo_detections_gpu(self.h, i, imw, imh, C.c_float(thresh), int(relative), recs.ctypes.data, max_recs,
                                           counts.ctypes.data)
        return counts, recs

    def selfcheck(self, passes):
        """queue `passes` forward passes over the resident input with a device-side checksum of the yolo outputs after each
        (nothing is synchronised); selfcheck_result() -> number of passes that differ from the first"""
        self.H.network_selfcheck(self.h, passes)

    def selfcheck_result(self):
        return int(self.H.network_selfcheck_result(self.h))

    def profile_begin(self, max_steps, stride=1, phase=0):
        """record per-layer events on the forward passes whose index (from now) % stride == phase, at most max_steps of them"""
        self.H.network_profile_set_stride(self.h, stride)
        self.H.network_profile_set_phase(self.h, phase)
        self.H.network_profile_begin(self.h, max_steps)

    def profile_read(self):
        ms = np.zeros(self.n + 1, np.float32)
        steps = self.H.network_profile_read(self.h, ms.ctypes.data)
        return steps, ms

    def packed_size(self):
        return self.H.network_packed_size(self.h)

    def export_packed(self):
        buf = np.zeros(self.packed_size(), np.uint8)
        self.H.network_export_packed(self.h, buf.ctypes.data)
        return buf

    def save_packed(self, path):
        self.H.network_save_packed(self.h, path.encode())

    def load_packed(self, path):
        """Packed file -> host blobs -> device (the on-disk twin of import_packed)."""
        self.H.network_load_packed(self.h, path.encode())

    def shortcut_multipliers(self, i):
        k = (C.c_int32 * 3)()
        assert self.H.dnq_layer_shortcut(self.h, i, k) == 0
        return int(k[0]), int(k[1])

    def shortcut_from(self, i):
        k = (C.c_int32 * 3)()
        assert self.H.dnq_layer_shortcut(self.h, i, k) == 0
        return int(k[2])

    def import_packed(self, buf):
        buf = np.ascontiguousarray(buf, np.uint8)
        self.H.network_import_packed(self.h, buf.ctypes.data, buf.nbytes)

    def import_packed_host(self, buf):
        buf = np.ascontiguousarray(buf, np.uint8)
        self.H.network_import_packed_host(self.h, buf.ctypes.data, buf.nbytes)

    def import_packed_gpu(self, dev_ptr, nbytes):
        self.H.network_import_packed_gpu(self.h, dev_ptr, nbytes)

    def close(self):
        if self.h:
            self.H.free_network(self.h)
            self.h = None
