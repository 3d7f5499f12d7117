#!/usr/bin/env python3
"""Per-layer roofline table (markdown) from bench.py --layers output (gpurun_out/bench_layers_n1.json).
Times are HIP-event intervals on the launch stream of the profiled steps, net of the cost of recording an event
(calibrated by bench.py on the intervals that contain no launch);
ops / bytes are the algorithmic figures of SURVEY.md section 8 times the batch.

usage: tools/layer_table.py [bench_layers.json] [batch]"""
import json
import sys

PEAK_TOPS = 5033.2   # 256 CU x 4 SIMD x 2048 op/clk x 2.4 GHz
PEAK_GBS = 8000.0
TYPES = {0: "conv", 3: "maxpool", 8: "route", 23: "yolo", 26: "upsample"}

path = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/bench_layers_n1.json"
d = json.load(open(path))
print(f"step {d['ms_per_step'] * 1e3:.1f} us (unprofiled steps included); per-layer event intervals of the profiled steps:\n")
print("| L | type | shape | us | TOP/s | % of 5 033 | GB/s (algorithmic) | % of 8 TB/s | bound |")
print("|---|---|---|---|---|---|---|---|---|")
for r in d["layers"]:
    t = TYPES.get(r["type"], str(r["type"]))
    us = r["ms"] * 1e3
    if "tops" in r:
        shape = f"{r['k']}x{r['k']} {r['c']}->{r['n']} @{r['hw']}"
        bound = "MFMA" if r["c"] % 64 == 0 and r["k"] == 3 else ("VALU/HBM" if r["c"] < 64 else "launch/latency")
        print(f"| {r['i']} | {t} | {shape} | {us:.1f} | {r['tops']:.0f} | {100 * r['tops'] / PEAK_TOPS:.1f} | {r['gbs']:.0f} | "
              f"{100 * r['gbs'] / PEAK_GBS:.1f} | {bound} |")
    else:
        print(f"| {r['i']} | {t} | | {us:.1f} | | | | | {'fused / elided' if us < 1.5 else 'HBM / launch'} |")
