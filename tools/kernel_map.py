#!/usr/bin/env python3
"""The table DESIGN.md section 3 opens with: which kernel serves which layer under which plan, at what cost.

  tools/kernel_map.py <bench_layers_throughput_plan.json> <bench_layers_latency_plan.json> <bench line json with roofline.sustained>

Inputs are what tools/gpu_r6.sh leaves in gpurun_out/r06 (bench.py --layers under each plan, and the default bench line whose flood leg holds
the in-flight cost of the 3x3 layers); layer_flood's table (all layers) is used instead when given as a fourth argument."""
import json
import re
import sys

PEAK = 5033.2
FAM = {1: "conv_first_mfma_pool (conv_aux.hip)", 2: "conv_small_pool / conv_mid_pool (conv_small.hip)", 3: "conv1x1_ws (conv1x1.hip)", 4: "conv_ws3 (conv_ws3.hip)",
       5: "conv_rows / conv_rows16 (row image)", 6: "conv_ref_f32", 7: "conv_pool16 (conv_pool16.hip)", 8: "conv_small32 (conv_small32.hip)"}
TYPES = {0: "conv", 3: "maxpool", 8: "route", 23: "yolo", 26: "upsample"}


def rows_of(path):
    return {r["i"]: r for r in json.load(open(path))["layers"]}


tp, lp = rows_of(sys.argv[1]), rows_of(sys.argv[2])
line = json.loads(open(sys.argv[3]).read().strip().splitlines()[-1])
flood = {}
for key in ("sustained",):
    for r in (line["roofline"].get("conv3x3_s1_aggregate", {}).get(key, {}) or {}).get("launches", []):
        flood[r["layer"]] = r["us_per_launch"]
if len(sys.argv) > 4:  # tools/layer_flood.py markdown: | L.. | serial | flood |
    for ln in open(sys.argv[4]):
        m = re.match(r"\|\s*(\d+)\.\.\d+\s*\|[^|]*\|\s*([\d.]+)\s*\|\s*([\d.]+)", ln)
        if m:
            flood.setdefault(int(m.group(1)), float(m.group(3)))
print("| L | layer | kernel, throughput plan (four batches in flight: what `value` runs) | us alone | us per launch in flight | of 5 033 TOP/s in flight | kernel, latency plan (one batch at a time) | us alone |")
print("|---|---|---|---|---|---|---|---|")
for i in sorted(tp):
    r, q = tp[i], lp.get(i, {})
    t = TYPES.get(r["type"], str(r["type"]))
    if t == "conv":
        shape = f"{r['k']}x{r['k']} {r['c']}->{r['n']} @{r['hw']}" + (" + next layer fused" if r.get("fused_next") else "")
        fl = flood.get(i)
        frac = f"{r['ops'] / fl / 1e6 / PEAK:.3f}" if fl and "ops" in r else ("%.3f" % (r["tops"] * r["ms"] * 1e3 / fl / PEAK * 1e0) if fl else "")
        print(f"| {i} | {shape} | {FAM.get(r.get('kernel_family'), '?')} | {r['ms'] * 1e3:.1f} | {fl if fl else ''} | {frac} | {FAM.get(q.get('kernel_family'), '?')} | {q.get('ms', 0) * 1e3:.1f} |")
    elif t == "maxpool":
        # a pool runs as a launch of its own unless the conv in front of it fused it (an event interval with no launch in it still reads ~2 us)
        own_t = not tp.get(i - 1, {}).get("fused_next", False)
        own_l = not lp.get(i - 1, {}).get("fused_next", False)
        if own_t or own_l:
            print(f"| {i} | maxpool | {'maxpool_u8 (glue.hip), own launch' if own_t else 'fused into the conv above'} | {r['ms'] * 1e3 if own_t else 0:.1f} | | | "
                  f"{'maxpool_u8 (glue.hip), own launch' if own_l else 'fused into the conv above'} | {q.get('ms', 0) * 1e3 if own_l else 0:.1f} |")
print("\n(routes, upsample and yolo layers launch nothing: concatenations are views, the upsample and the yolo activations are written by the 1x1 conv in front of them.)")
