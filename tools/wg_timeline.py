#!/usr/bin/env python3
"""Per-workgroup phase timeline of ONE layer group of the bench net, as the net launches it (plan, fused pool, epilogue table), from the wall-clock
stamps the -DMI355_ABLATE build leaves per workgroup (MI355_LIB_DIR=build_ab/libablate):
  tools/wg_timeline.py --layer 6 --kernel mid      conv_mid_pool_kernel  (64 -> 128 @52 + pool)
  tools/wg_timeline.py --layer 8 --kernel rows     conv_rows_i8_kernel   (throughput plan: 128 x 128 tiles)
  tools/wg_timeline.py --layer 12 --kernel rows16  conv_rows16_i8_kernel
--inflight N: N instances launch the layer at once (the stamps are those of the LAST launch that wrote them)."""
import argparse
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, ROOT)
from yolo_quantization_amd import binding, synth  # noqa: E402

PH = {"mid": ("mi355_debug_read_tsm", ["geometry, image DMA issued", "parameters to LDS, A fragments issued, pixel tables", "image / fragments landed, barrier",
                                       "cell sums, box sums (two barriers)", "groups: MFMA chains + epilogues + stores issued"]),
      "rows": ("mi355_debug_read_ts", ["setup (index math, accumulator seeds, parameters)", "first DMA wait", "K loop", "epilogue: box sums + requantise", "copy-out"]),
      "rows16": ("mi355_debug_read_ts16", ["index math, DMA tables, prologue loads issued", "accumulator seeds, parameters, first image landed", "K loop",
                                           "epilogue: box sums + requantise", "copy-out"])}
ap = argparse.ArgumentParser()
ap.add_argument("--layer", type=int, required=True)
ap.add_argument("--kernel", choices=list(PH), required=True)
ap.add_argument("--inflight", type=int, default=1)
ap.add_argument("--plan", type=int, default=1)
ap.add_argument("--batch", type=int, default=64)
a = ap.parse_args()
binding.init(0)
cfg = os.path.join(ROOT, "cfg", "yolov3-tiny_quant.cfg")
wts = f"/tmp/wgtl_{os.getpid()}.weights"
synth.synth_weights(cfg, wts, seed=1234)
net = binding.Net(cfg, wts, batch=a.batch, keep_head_float=False)
net.prepare_fixed(1.0 / 255.0, 0)
nets = [net] + [net.replica(default_stream=(k == 3)) for k in range(1, a.inflight)]
convs = [i for i, inf in enumerate(net.info) if inf["type"] == binding.T_CONV]
hi = min([c for c in convs if c > a.layer] + [len(net.info)])
for k, nk in enumerate(nets):
    nk.set("plan", a.plan)
    nk.push_input(synth.synth_image_u8(3, 416, 416, seed=100 + k, batch=a.batch))
    nk.forward()
    nk.sync()
    nk.set("range_lo", a.layer); nk.set("range_hi", hi)
for _ in range(10):
    for nk in nets:
        nk.forward()
for nk in nets:
    nk.sync()
for nk in nets:  # the launch whose stamps are read
    nk.forward()
for nk in nets:
    nk.sync()
S = binding.shim()
fn, names = PH[a.kernel]
ts = np.zeros((6, 4096), np.int64)
getattr(S, fn).argtypes = [C.c_void_p]
assert getattr(S, fn)(ts.ctypes.data) == 0
nb = int((ts[0] > 0).sum())
t = ts[:, :nb].astype(np.float64) / 100.0
t0 = t[0].min()
inf = net.info[a.layer]
print(f"layer {a.layer}..{hi - 1}: {inf['size']}x{inf['size']} {inf['c']}->{inf['n']} @{inf['out_h']}, plan {a.plan}, {a.inflight} in flight: {nb} workgroups; "
      f"span first start .. last end {t[5].max() - t0:.2f} us; starts p50 {np.median(t[0]) - t0:.2f}, p90 {np.percentile(t[0], 90) - t0:.2f}, max {t[0].max() - t0:.2f}")
for i, nm in enumerate(names):
    d = t[i + 1] - t[i]
    print(f"  {nm:56s} p50 {np.median(d):6.2f}  min {d.min():6.2f}  max {d.max():6.2f} us")
d = t[5] - t[0]
print(f"  {'whole workgroup':56s} p50 {np.median(d):6.2f}  min {d.min():6.2f}  max {d.max():6.2f} us")
for nk in reversed(nets):
    nk.close()
os.remove(wts)
