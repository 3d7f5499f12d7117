#!/usr/bin/env python3
"""Where a wave of the first-layer kernel spends its shader clocks (needs the -DMI355_ABLATE build: MI355_LIB_DIR=build_ab/<name>).
usage: tools/l0_phases.py [--inflight 1|4] [--plan 0|1]"""
import argparse
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, ROOT)
from yolo_quantization_amd import binding, synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--inflight", type=int, default=1)
ap.add_argument("--plan", type=int, default=1)
ap.add_argument("--batch", type=int, default=64)
a = ap.parse_args()
binding.init(0)
cfg = os.path.join(ROOT, "cfg", "yolov3-tiny_quant.cfg")
wts = f"/tmp/l0ph_{os.getpid()}.weights"
synth.synth_weights(cfg, wts, seed=1234)
net = binding.Net(cfg, wts, batch=a.batch, keep_head_float=False)
net.prepare_fixed(1.0 / 255.0, 0)
nets = [net] + [net.replica(default_stream=(k == 3)) for k in range(1, a.inflight)]
for k, nk in enumerate(nets):
    nk.set("plan", a.plan)
    nk.push_input(synth.synth_image_u8(3, 416, 416, seed=100 + k, batch=a.batch))
    nk.set("range_lo", 0); nk.set("range_hi", 2)
for _ in range(20):
    for nk in nets:
        nk.forward()
for nk in nets:
    nk.sync()
S = binding.shim()
ph = np.zeros((4096, 4, 12), np.int64)
S.mi355_debug_read_l0ph.argtypes = [C.c_void_p]
assert S.mi355_debug_read_l0ph(ph.ctypes.data) == 0
nb = int((ph[:, 0, 7] > 0).sum())
w = ph[:nb].astype(np.float64)
tiles = w[:, :, 7].mean()
tot = (w[:, :, :7].sum(axis=2) + w[:, :, 8:10].sum(axis=2)).mean()
names = ["barrier", "deferred stores + prefetch issue", "row 0: B reads + MFMA chain", "row 0: epilogue", "row 1: B reads + MFMA chain",
         "row 1: epilogue", "bookkeeping + loop", None, "wait: prefetch + deferred stores landed", "staging: permutes + LDS writes"]
print(f"workgroups {nb}, tiles per workgroup {tiles:.1f}, shader clocks per wave {tot:.0f} = {tot / tiles:.0f} per tile ({a.inflight} in flight)")
print(f"  prologue (kernel entry -> tile loop) {w[:, :, 10].mean():.0f} clocks per wave; whole wave {w[:, :, 11].mean():.0f} clocks; tile loop share {100 * tot / w[:, :, 11].mean():.1f} %")
for k, nm in enumerate(names):
    if nm is None:
        continue
    v = w[:, :, k].mean()
    print(f"  {nm:36s} {v / tiles:8.0f} clocks per tile  {100 * v / tot:5.1f} %   by wave {[int(x / tiles) for x in w[:, :, k].mean(axis=0)]}")
for nk in reversed(nets):
    nk.close()
os.remove(wts)
