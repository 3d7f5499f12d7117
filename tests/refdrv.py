"""ctypes binding to oracle/_ref/libdarknet_ref*.so (the unmodified reference + oracle/ref_driver.c).
TEST INFRASTRUCTURE: only tests/, bench.py's cpu_baseline leg and the golden generator may import this."""
import ctypes as C
import os
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
REF_DIR = os.path.join(_HERE, "..", "oracle", "_ref")


_NAMES = {False: "libdarknet_ref.so", True: "libdarknet_ref_omp.so", "mi355": "libdarknet_ref_mi355.so"}


def available(omp=False):
    return os.path.exists(os.path.join(REF_DIR, _NAMES[omp]))


_libs = {}


def lib(omp=False):
    if omp not in _libs:
        L = C.CDLL(os.path.join(REF_DIR, _NAMES[omp]))
        if omp == "mi355":
            L.refdrv_mi355_bind.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int]
            L.refdrv_forward_mi355.argtypes = [C.c_void_p, C.c_int]
            L.refdrv_mi355_unbind.argtypes = [C.c_void_p]
        L.refdrv_load.restype = C.c_void_p
        L.refdrv_load.argtypes = [C.c_char_p, C.c_char_p]
        L.refdrv_nlayers.argtypes = [C.c_void_p]
        L.refdrv_layer_info.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.refdrv_prepare.argtypes = [C.c_void_p, C.c_void_p]
        L.refdrv_set_input_u8.argtypes = [C.c_void_p, C.c_void_p]
        L.refdrv_input_u8.restype = C.c_void_p
        L.refdrv_input_u8.argtypes = [C.c_void_p]
        L.refdrv_forward.argtypes = [C.c_void_p, C.c_void_p]
        L.refdrv_network_predict.restype = C.c_double
        L.refdrv_network_predict.argtypes = [C.c_void_p, C.c_void_p]
        for n in ("int32", "u8", "f32"):
            f = getattr(L, "refdrv_layer_" + n)
            f.restype = C.c_void_p
            f.argtypes = [C.c_void_p, C.c_int]
        L.refdrv_layer_prep.argtypes = [C.c_void_p, C.c_int] + [C.c_void_p] * 6
        L.refdrv_gemm_u8.argtypes = [C.c_int, C.c_int, C.c_int, C.c_float, C.c_void_p, C.c_int, C.c_void_p, C.c_int,
                                     C.c_int, C.c_void_p, C.c_int]
        L.refdrv_im2col_u8.argtypes = [C.c_void_p] + [C.c_int] * 6 + [C.c_void_p, C.c_uint8]
        L.refdrv_quant_multiplier.argtypes = [C.c_float, C.c_void_p, C.c_void_p]
        L.refdrv_quantize_image.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.refdrv_yolo_detections.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int, C.c_void_p, C.c_int]
        L.refdrv_yolo_params.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        if hasattr(L, "refdrv_forward_layer"):
            L.refdrv_forward_layer.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        if hasattr(L, "refdrv_leaky_lut"):
            L.refdrv_leaky_lut.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        _libs[omp] = L
    return _libs[omp]


INFO_KEYS = ["type", "out_c", "out_h", "out_w", "c", "h", "w", "n", "size", "stride", "pad", "activation",
             "batch_normalize", "quantized", "quant_stop", "outputs"]
# LAYER_TYPE enum values of the reference (include/darknet.h:99-130)
T_CONV, T_MAXPOOL, T_ROUTE, T_UPSAMPLE, T_YOLO = 0, 3, 8, 26, 23


def _as(ptr, n, dt):
    return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(dt)), shape=(n,))


def load_image_color(path, omp=False):
    """the reference's own load_image_color (stb_image decode) -> float32 [c][h][w] in 0..1 (only where the reference is built)"""
    L = lib(omp)
    L.refdrv_load_image_color.restype = C.POINTER(C.c_float)
    L.refdrv_load_image_color.argtypes = [C.c_char_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.refdrv_free_floats.argtypes = [C.POINTER(C.c_float)]
    w, h, c = C.c_int(), C.c_int(), C.c_int()
    p = L.refdrv_load_image_color(path.encode(), C.byref(w), C.byref(h), C.byref(c))
    assert p, f"load_image_color({path}) failed"
    a = np.ctypeslib.as_array(p, shape=(c.value, h.value, w.value)).copy()
    L.refdrv_free_floats(p)
    return a


def letterbox(im_chw, w, h, omp=False):
    """the reference's own letterbox_image (src/image.c:812-831)"""
    L = lib(omp)
    im = np.ascontiguousarray(im_chw, np.float32)
    out = np.empty((im.shape[0], h, w), np.float32)
    L.refdrv_letterbox.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
    L.refdrv_letterbox(im.ctypes.data, im.shape[2], im.shape[1], im.shape[0], w, h, out.ctypes.data)
    return out


class RefNet:
    def __init__(self, cfg, weights, omp=False):
        self.L = lib(omp)
        self.h = self.L.refdrv_load(cfg.encode(), weights.encode())
        self.n = self.L.refdrv_nlayers(self.h)
        self.info = []
        for i in range(self.n):
            a = (C.c_int * 16)()
            self.L.refdrv_layer_info(self.h, i, a)
            self.info.append(dict(zip(INFO_KEYS, list(a))))

    def prepare(self, x_float_chw):
        x = np.ascontiguousarray(x_float_chw, dtype=np.float32)
        self._keep = x
        assert self.L.refdrv_prepare(self.h, x.ctypes.data) == 0
        n = x.size
        return _as(self.L.refdrv_input_u8(self.h), n, C.c_uint8).copy()

    def set_input_u8(self, x):
        x = np.ascontiguousarray(x, dtype=np.uint8)
        self.L.refdrv_set_input_u8(self.h, x.ctypes.data)

    def forward(self):
        t = np.zeros(self.n, dtype=np.float64)
        assert self.L.refdrv_forward(self.h, t.ctypes.data) == 0
        return t

    def layer_int32(self, i):
        return _as(self.L.refdrv_layer_int32(self.h, i), self.info[i]["outputs"], C.c_int32).copy()

    def layer_u8(self, i):
        return _as(self.L.refdrv_layer_u8(self.h, i), self.info[i]["outputs"], C.c_uint8).copy()

    def layer_f32(self, i):
        return _as(self.L.refdrv_layer_f32(self.h, i), self.info[i]["outputs"], C.c_float).copy()

    def yolo_params(self, i):
        """(anchors [2*total] f32, mask [n] i32) of yolo layer i."""
        b = np.zeros(64, np.float32); m = np.zeros(16, np.int32); t = C.c_int()
        n = self.L.refdrv_yolo_params(self.h, i, b.ctypes.data, m.ctypes.data, C.byref(t))
        assert n > 0
        return b[:2 * t.value].copy(), m[:n].copy()

    def yolo_detections(self, i, classes, imw, imh, thresh, relative, cap):
        """The reference's get_yolo_detections on yolo layer i: (count, records [count][6 + classes])."""
        recs = np.zeros((cap, 6 + classes), np.float32)
        cnt = self.L.refdrv_yolo_detections(self.h, i, imw, imh, C.c_float(thresh), int(relative), recs.ctypes.data, cap)
        assert cnt >= 0
        return cnt, recs[:min(cnt, cap)]

    def mi355_bind(self, gpu=0, accum=0, store=0):
        """Point the reference's layer.forward_gpu pointers at libmi355yolo.so (integration/mi355_glue.c)."""
        assert self.L.refdrv_mi355_bind(self.h, gpu, accum, store) == 0

    def forward_mi355(self, pull_all=True):
        assert self.L.refdrv_forward_mi355(self.h, int(pull_all)) == 0

    def mi355_unbind(self):
        self.L.refdrv_mi355_unbind(self.h)

    def forward_layer(self, i, x_u8):
        """Layer i alone on the given uint8 input (its own forward pointer); outputs via layer_int32 / layer_u8 / layer_f32."""
        x = np.ascontiguousarray(x_u8, dtype=np.uint8)
        assert x.size == self.info[i]["c"] * self.info[i]["h"] * self.info[i]["w"]
        assert self.L.refdrv_forward_layer(self.h, i, x.ctypes.data) == 0

    def leaky_lut(self, i):
        """(M0_lut0, M0_right_shift_lut0) of conv layer i: the quantised 0.1 of the MKL path's LEAKY (src/blas.c:318-323)."""
        a = C.c_int32(); b = C.c_int()
        self.L.refdrv_leaky_lut(self.h, i, C.byref(a), C.byref(b))
        return a.value, b.value

    def prep(self, i):
        n = max(self.info[i]["n"], 1)
        b = np.zeros(n, np.int32); mv = np.zeros(n, np.float64); sv = np.zeros(n, np.float64)
        m0 = np.zeros(n, np.int32); sh = np.zeros(n, np.int32); q = np.zeros(4, np.float32)
        self.L.refdrv_layer_prep(self.h, i, b.ctypes.data, mv.ctypes.data, sv.ctypes.data, m0.ctypes.data,
                                 sh.ctypes.data, q.ctypes.data)
        return dict(biases_int32=b, M_value=mv, shift_value=sv, M0=m0, shift=sh, s_in=q[0], zp_in=int(q[1]),
                    s_act=q[2], zp_act=int(q[3]))
