#!/usr/bin/env python3
"""Experiment: lock the phase between the four in-flight instances with events so that an instance's first layers (L0 .. L10: instruction-
issue bound) always start when another instance starts its deep layers (L12 ..: matrix pipe / power bound) -- tools/mix_flood.py measured
L0 beside L12 / L21 at -8..-11 % of their combined time.  Instance k's step waits for instance (k + 2) % 4 to reach layer `--split`;
instances 0 / 1 and 2 / 3 then run in antiphase.  Events through the HIP runtime directly (experiment only).
usage: tools/phase_lock.py [--split 12] [--steps 400]"""
import argparse
import ctypes as C
import os
import sys
import time

ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, ROOT)
from yolo_quantization_amd import binding, synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--cfg", default=os.path.join(ROOT, "cfg", "yolov3-tiny_quant.cfg"))
ap.add_argument("--batch", type=int, default=64)
ap.add_argument("--split", default="12,8,21")
ap.add_argument("--steps", type=int, default=400)
a = ap.parse_args()
binding.init(0)
hip = C.CDLL("libamdhip64.so")
hip.hipEventCreateWithFlags.argtypes = [C.POINTER(C.c_void_p), C.c_uint]
hip.hipEventRecord.argtypes = [C.c_void_p, C.c_void_p]
hip.hipStreamWaitEvent.argtypes = [C.c_void_p, C.c_void_p, C.c_uint]
wts = f"/tmp/pl_{os.getpid()}.weights"
synth.synth_weights(a.cfg, wts, seed=1234)
net = binding.Net(a.cfg, wts, batch=a.batch, keep_head_float=False)
net.prepare_fixed(1.0 / 255.0, 0)
nets = [net] + [net.replica(default_stream=(k == 3)) for k in range(1, 4)]
info = net.info
nlayers = len(info)
for k, nk in enumerate(nets):
    nk.set("plan", 1)
    nk.push_input(synth.synth_image_u8(info[0]["c"], info[0]["h"], info[0]["w"], seed=100 + k, batch=a.batch))
for _ in range(200):
    for nk in nets:
        nk.forward()
for nk in nets:
    nk.sync()
ev = []
for _ in range(4):
    e = C.c_void_p()
    assert hip.hipEventCreateWithFlags(C.byref(e), 2) == 0  # hipEventDisableTiming
    ev.append(e)
streams = [C.c_void_p(nk.stream()) for nk in nets]


def plain(steps):
    for nk in nets:
        nk.sync()
    t0 = time.perf_counter()
    for i in range(steps):
        nets[i % 4].forward()
    for nk in nets:
        nk.sync()
    return (time.perf_counter() - t0) / steps * 1e3


def split_only(steps, split):
    """two launches of the executor per step, no events: what the split itself costs"""
    for nk in nets:
        nk.sync()
    t0 = time.perf_counter()
    for i in range(steps):
        nk = nets[i % 4]
        nk.set("range_lo", 0); nk.set("range_hi", split); nk.forward()
        nk.set("range_lo", split); nk.set("range_hi", nlayers); nk.forward()
    for nk in nets:
        nk.sync()
    for nk in nets:
        nk.set("range_lo", 0); nk.set("range_hi", 0)
    return (time.perf_counter() - t0) / steps * 1e3


def locked(steps, split, partner):
    for nk in nets:
        nk.sync()
    t0 = time.perf_counter()
    for i in range(steps):
        k = i % 4
        nk = nets[k]
        if i >= 2:
            assert hip.hipStreamWaitEvent(streams[k], ev[(k + partner) % 4], 0) == 0
        nk.set("range_lo", 0); nk.set("range_hi", split); nk.forward()
        assert hip.hipEventRecord(ev[k], streams[k]) == 0
        nk.set("range_lo", split); nk.set("range_hi", nlayers); nk.forward()
    for nk in nets:
        nk.sync()
    for nk in nets:
        nk.set("range_lo", 0); nk.set("range_hi", 0)
    return (time.perf_counter() - t0) / steps * 1e3


print(f"plain round-robin: {plain(a.steps):.4f} ms per step")
for sp in (int(v) for v in a.split.split(",")):
    print(f"split at layer {sp}: two executor calls per step, no events {split_only(a.steps, sp):.4f};  "
          f"locked to instance k + 2: {locked(a.steps, sp, 2):.4f};  locked to k + 1: {locked(a.steps, sp, 1):.4f} ms per step", flush=True)
print(f"plain round-robin: {plain(a.steps):.4f} ms per step")
for nk in reversed(nets):
    nk.close()
os.remove(wts)
