#!/usr/bin/env python3
"""Does the PHASE between the in-flight network instances matter?  Four instances run whole steps back to back on their streams.  Started
together they run the same layer at about the same time (the step then costs the sum of the layers' flood figures, tools/layer_flood.py);
tools/mix_flood.py says the first layer shares the chip 8-11 % better with the deep 3x3 layers than with itself.  Here every instance first runs
a partial pass (layers [0, X_k): real kernels, results overwritten by the full passes behind them) so that instance k lags instance k - 1 by
about `--lag` of a step, then `--steps` full steps are issued round-robin and timed as bench.py does.
usage: tools/stagger.py [--lags 0,0.125,0.25,0.375] [--steps 400]"""
import argparse
import os
import sys
import time

ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, ROOT)
from yolo_quantization_amd import binding, synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--cfg", default=os.path.join(ROOT, "cfg", "yolov3-tiny_quant.cfg"))
ap.add_argument("--batch", type=int, default=64)
ap.add_argument("--lags", default="0,0.25,0,0.125,0.25,0.375,0")
ap.add_argument("--steps", type=int, default=400)
a = ap.parse_args()
binding.init(0)
wts = f"/tmp/stg_{os.getpid()}.weights"
synth.synth_weights(a.cfg, wts, seed=1234)
net = binding.Net(a.cfg, wts, batch=a.batch, keep_head_float=False)
net.prepare_fixed(1.0 / 255.0, 0)
nets = [net] + [net.replica(default_stream=(k == 3)) for k in range(1, 4)]
info = net.info
for k, nk in enumerate(nets):
    nk.set("plan", 1)
    nk.push_input(synth.synth_image_u8(info[0]["c"], info[0]["h"], info[0]["w"], seed=100 + k, batch=a.batch))
for _ in range(200):
    for nk in nets:
        nk.forward()
for nk in nets:
    nk.sync()
# cumulative share of a step at the start of every conv group (flood us of r05_v3, good enough for a lag)
FLOOD = {0: 36.5, 2: 26.4, 4: 17.8, 6: 18.6, 8: 17.3, 10: 14.4, 12: 46.6, 13: 8.6, 14: 13.6, 15: 3.8, 18: 4.0, 21: 37.3, 22: 6.7}
tot = sum(FLOOD.values())
cum, acc = [], 0.0
for lo in sorted(FLOOD):
    cum.append((lo, acc / tot)); acc += FLOOD[lo]


def layer_at(frac):
    """first layer of the conv group whose start is nearest to `frac` of a step"""
    return min(cum, key=lambda c: abs(c[1] - frac))[0]


for lag in (float(v) for v in a.lags.split(",")):
    for nk in nets:
        nk.sync()
    t0 = time.perf_counter()
    # instance k is to lag instance 0 by k * lag of a step: it runs the FIRST (1 - k * lag) of a step less, i.e. it starts with a partial pass
    # over the layers from the one at (k * lag) .. hmm: simpler -- instance k runs layers [0, X) with X at k * lag of a step before its full steps
    for k, nk in enumerate(nets):
        x = layer_at(k * lag) if lag > 0 else 0
        if x > 0:
            nk.set("range_lo", 0); nk.set("range_hi", x)
            nk.forward()
            nk.set("range_lo", 0); nk.set("range_hi", 0)
    for i in range(a.steps):
        nets[i % 4].forward()
    for nk in nets:
        nk.sync()
    dt = time.perf_counter() - t0
    print(f"lag {lag:5.3f} of a step between instances: {dt / a.steps * 1e3:.4f} ms per step ({a.batch * a.steps / dt:.0f} images/s; the partial passes are inside the time)", flush=True)
for nk in reversed(nets):
    nk.close()
os.remove(wts)
