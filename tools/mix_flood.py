#!/usr/bin/env python3
"""Do two DIFFERENT layers share the chip better than a layer shares it with itself?  Four network instances (streams); a pair (A, B) of layer
groups: first A on all four, then B on all four (tools/layer_flood.py's flood figure), then A on two instances and B on the other two at once,
launch counts chosen so that both kinds are busy for about the same time.  gain = 1 - t_mixed / (t_A + t_B).
usage: tools/mix_flood.py [--pairs 12:0,12:2,21:0,...] [--reps 80]"""
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
ap.add_argument("--pairs", default="12:0,12:2,12:4,12:6,21:0,21:2,12:21,0:2,8:0,13:0")
ap.add_argument("--reps", type=int, default=80)
a = ap.parse_args()
binding.init(0)
wts = f"/tmp/mix_{os.getpid()}.weights"
synth.synth_weights(a.cfg, wts, seed=1234)
net = binding.Net(a.cfg, wts, batch=a.batch, keep_head_float=False)
net.prepare_fixed(1.0 / 255.0, 0)
nets = [net] + [net.replica(default_stream=(k == 3)) for k in range(1, 4)]
info = net.info
for k, nk in enumerate(nets):
    nk.set("plan", 1)
    nk.push_input(synth.synth_image_u8(info[0]["c"], info[0]["h"], info[0]["w"], seed=100 + k, batch=a.batch))
    for _ in range(3):
        nk.forward()
    nk.sync()
convs = [i for i, inf in enumerate(info) if inf["type"] == binding.T_CONV]
group_of = {c: (c, convs[k + 1] if k + 1 < len(convs) else len(info)) for k, c in enumerate(convs)}


def set_range(nk, lo):
    g = group_of[lo]
    nk.set("range_lo", g[0]); nk.set("range_hi", g[1])


def run(counts):
    """counts[k] launches on instance k, issued round-robin; wall time in us"""
    for nk in nets:
        nk.sync()
    left = list(counts)
    t0 = time.perf_counter()
    while any(left):
        for k, nk in enumerate(nets):
            if left[k]:
                nk.forward(); left[k] -= 1
    for nk in nets:
        nk.sync()
    return (time.perf_counter() - t0) * 1e6


print("| A | B | flood A us | flood B us | launches A : B | A then B us | A with B us | gain |")
print("|---|---|---|---|---|---|---|---|")
for pair in a.pairs.split(","):
    la, lb = (int(v) for v in pair.split(":"))
    for nk in nets:
        set_range(nk, la)
    run([5] * 4); fa = run([a.reps] * 4) / (4 * a.reps)
    for nk in nets:
        set_range(nk, lb)
    run([5] * 4); fb = run([a.reps] * 4) / (4 * a.reps)
    # both kinds busy for about the same time: nA fA = nB fB, nA + nB launches per instance pair
    na = 2 * a.reps
    nb = max(2, int(round(na * fa / fb / 2)) * 2)
    for nk in nets:
        set_range(nk, la)
    ta = run([na // 4] * 4)
    for nk in nets:
        set_range(nk, lb)
    tb = run([nb // 4] * 4)
    for k, nk in enumerate(nets):
        set_range(nk, la if k < 2 else lb)
    run([3] * 4)
    tm = run([na // 2, na // 2, nb // 2, nb // 2])
    # and interleaved the other way round (instances 0 / 2 run A): which hardware queue a stream sits on should not matter
    for k, nk in enumerate(nets):
        set_range(nk, la if k % 2 == 0 else lb)
    run([3] * 4)
    tm2 = run([na // 2, nb // 2, na // 2, nb // 2])
    print(f"| L{la} | L{lb} | {fa:.1f} | {fb:.1f} | {na} : {nb} | {ta + tb:.0f} | {tm:.0f} / {tm2:.0f} | {1 - min(tm, tm2) / (ta + tb):+.3f} |", flush=True)
for nk in nets:
    nk.set("range_lo", 0); nk.set("range_hi", 0)
for nk in reversed(nets):
    nk.close()
os.remove(wts)
