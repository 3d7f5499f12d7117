#!/usr/bin/env python3
"""Aggregate a rocprofv3 counter_collection.csv: per kernel name, mean counter value per dispatch."""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
agg = defaultdict(lambda: defaultdict(list))
for r in rows:
    k = r.get("Kernel_Name", "?").split("(")[0][:70]
    agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in agg.items():
    if "conv" not in k and "maxpool" not in k:
        continue
    print(k)
    for c, v in sorted(cs.items()):
        print(f"   {c:28s} n={len(v):3d} mean={sum(v) / len(v):.4g}")
