#!/usr/bin/env python3
"""Per-step wall time of the first N forward passes of a fresh network (is the start slow, and for how long?)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from yolo_quantization_amd import binding, synth
cfg = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "cfg", "yolov3-tiny_quant.cfg")
wts = "/tmp/curve.weights"
synth.synth_weights(cfg, wts, seed=1234)
binding.init(0)
net = binding.Net(cfg, wts, batch=64)
net.prepare_fixed(1.0 / 255.0, 0)
x = synth.synth_image_u8(3, 416, 416, seed=7, batch=64)
net.push_input(x)
idle = float(sys.argv[1]) if len(sys.argv) > 1 else 0.0
time.sleep(idle)
ts = []
group = 5
for i in range(60):
    t0 = time.perf_counter()
    for _ in range(group):
        net.forward()
    net.sync()
    ts.append((time.perf_counter() - t0) / group * 1e3)
print("ms/step in groups of %d:" % group, " ".join("%.3f" % t for t in ts[:12]), "...")
for gap in (0.002, 0.02, 0.2):
    time.sleep(gap)
    ts = []
    for i in range(8):
        t0 = time.perf_counter()
        for _ in range(group):
            net.forward()
        net.sync()
        ts.append((time.perf_counter() - t0) / group * 1e3)
    print("after %.0f ms idle:" % (gap * 1e3), " ".join("%.3f" % t for t in ts))
