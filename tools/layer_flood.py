#!/usr/bin/env python3
"""What does a layer cost with batches in flight?  For every conv layer of the net (with the layers fused into it):
  serial  -- the launch repeated back to back on ONE stream (each waits for the one before: launch gap, pipeline fill and tail
             are paid every time),
  flood   -- the same launch repeated on N network instances / streams at once (independent batches): time per launch when the
             device may overlap the launches with each other.
The sum of the flood column is what a step would cost if mixing different layers bought nothing beyond overlapping a layer
with itself; the in-flight step of bench.py sits next to it.
usage: tools/layer_flood.py [--cfg ...] [--batch 64] [--inflight 4] [--plan 1] [--reps 60]"""
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
ap.add_argument("--inflight", type=int, default=4)
ap.add_argument("--plan", type=int, default=1)
ap.add_argument("--reps", type=int, default=60)
ap.add_argument("--only", default="", help="comma-separated first layers of the groups to measure (default: every group)")
ap.add_argument("--by-shape", action="store_true", help="one row per distinct conv shape (sums over its layer groups): for the deep nets")
ap.add_argument("--power", action="store_true", help="sample the device's socket power and shader clock (hwmon) during every flood phase (use --reps 2000+)")
a = ap.parse_args()
binding.init(0)
if os.environ.get("BENCH_FORCE_TILE"):  # "bm,bn[,nt]" for every conv_rows launch
    t = [int(v) for v in os.environ["BENCH_FORCE_TILE"].split(",")]
    binding.shim().mi355_conv_set_tile(t[0] | ((t[2] if len(t) > 2 else 0) << 16), t[1])
if os.environ.get("BENCH_DEBUG_FLAGS"):
    binding.shim().mi355_debug_flags(int(os.environ["BENCH_DEBUG_FLAGS"]))
wts = f"/tmp/flood_{os.getpid()}.weights"
synth.synth_weights(a.cfg, wts, seed=1234)
net = binding.Net(a.cfg, wts, batch=a.batch, keep_head_float=False)
net.prepare_fixed(1.0 / 255.0, 0)
nets = [net] + [net.replica(default_stream=(k == 3)) for k in range(1, a.inflight)]  # the fourth instance on the default stream, as in bench.py
info = net.info
for k, nk in enumerate(nets):
    nk.set("plan", a.plan)
    nk.push_input(synth.synth_image_u8(info[0]["c"], info[0]["h"], info[0]["w"], seed=100 + k, batch=a.batch))
    for _ in range(3):
        nk.forward()
    nk.sync()


from yolo_quantization_amd.hwmon import Sampler as _Sampler  # noqa: E402


class Sampler(_Sampler):
    def run(self, fn):
        r, pw, ck, _ = super().run(fn)
        return r, pw, ck


def timed(ns, reps):
    for nk in ns:
        nk.sync()
    t0 = time.perf_counter()
    for _ in range(reps):
        for nk in ns:
            nk.forward()
    for nk in ns:
        nk.sync()
    return (time.perf_counter() - t0) / (reps * len(ns)) * 1e6


# layer groups: a conv and everything up to the next conv (fused pools / yolo / upsample, elided routes)
convs = [i for i, inf in enumerate(info) if inf["type"] == binding.T_CONV]
groups = [(c, (convs[k + 1] if k + 1 < len(convs) else len(info))) for k, c in enumerate(convs)]
tot_s = tot_f = 0.0
shapes = {}
sampler = Sampler() if a.power else None
print(f"plan {a.plan}, {a.inflight} instances, batch {a.batch}")
print("| layers | conv | serial us | flood us per launch | flood / serial |")
print("|---|---|---|---|---|")
only = {int(v) for v in a.only.split(",") if v}
for lo, hi in groups:
    if only and lo not in only:
        continue
    for nk in nets:
        nk.set("range_lo", lo); nk.set("range_hi", hi)
    timed(nets, 5)
    ts = timed(nets[:1], min(a.reps, 200))
    extra = ""
    if sampler:
        tf, pw, ck = sampler.run(lambda: timed(nets, a.reps))
        ts, pws, cks = sampler.run(lambda: timed(nets[:1], a.reps))  # the same launch back to back on ONE instance: power / clock of the kernel alone
        extra = (f" flood {pw:.0f} W, {ck:.0f} MHz | serial {pws:.0f} W, {cks:.0f} MHz |" if pw is not None and pws is not None else "")
    else:
        tf = timed(nets, a.reps)
    tot_s += ts; tot_f += tf
    inf = info[lo]
    key = (inf["size"], inf["c"], inf["n"], inf["out_h"], hi - lo)
    e = shapes.setdefault(key, [0, 0.0, 0.0]); e[0] += 1; e[1] += ts; e[2] += tf
    if not a.by_shape:
        print(f"| {lo}..{hi - 1} | {inf['size']}x{inf['size']} {inf['c']}->{inf['n']} @{inf['out_h']} | {ts:.1f} | {tf:.1f} | {tf / ts:.2f} |" + extra)
if a.by_shape:
    for (k, c, n, hw, span), (cnt, ts, tf) in sorted(shapes.items(), key=lambda kv: -kv[1][2]):
        ops = 2.0 * n * c * k * k * hw * hw * a.batch
        print(f"| {cnt} x ({span} layers) | {k}x{k} {c}->{n} @{hw} | {ts:.1f} | {tf:.1f} | {tf / ts:.2f} | {ops / (tf / cnt) / 1e6:.0f} TOP/s sustained |")
for nk in nets:
    nk.set("range_lo", 0); nk.set("range_hi", 0)
timed(nets, 10)
step_s = timed(nets[:1], a.reps)
if sampler:
    step_f, pw, ck = sampler.run(lambda: timed(nets, a.reps))
    print(f"whole step in flight: {pw:.0f} W, {ck:.0f} MHz" if pw is not None else "no power data")
    _, pw1, ck1 = sampler.run(lambda: timed(nets[:1], a.reps))
    print(f"whole step, one batch at a time: {pw1:.0f} W, {ck1:.0f} MHz" if pw1 is not None else "")
else:
    step_f = timed(nets, a.reps)
print(f"| sum | | {tot_s:.1f} | {tot_f:.1f} | {tot_f / tot_s:.2f} |")
print(f"| whole step | | {step_s:.1f} | {step_f:.1f} | {step_f / step_s:.2f} |")
for nk in reversed(nets):
    nk.close()
os.remove(wts)
