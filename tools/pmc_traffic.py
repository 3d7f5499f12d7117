#!/usr/bin/env python3
"""Per-kernel HBM traffic from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE) over bench.py.

Units / corrections (MI355X_MICROARCH.md, HBM section): FETCH_SIZE and WRITE_SIZE are in KiB-like units of 1024 B
derived from TCC_EA0_RDREQ / WRREQ; on gfx950 FETCH_SIZE reports exactly half of the bytes of a wide coalesced
streaming read (128-B requests tallied at 64 B), so the read side is doubled.  WRITE_SIZE is uncalibrated there; it is
reported as counted.  Infinity-Cache hits are included in both (they are fabric-side counters).

usage: tools/pmc_traffic.py <dir FETCH_SIZE pass> <dir WRITE_SIZE pass> <out.json>"""
import csv
import glob
import json
import sys
from collections import defaultdict


def load(d, counter):
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
    agg = defaultdict(list)
    if not f:
        return agg
    for r in csv.DictReader(open(f[0])):
        if r["Counter_Name"] != counter:
            continue
        k = r["Kernel_Name"].split("(")[0]
        agg[(k, int(r["Grid_Size"]) if "Grid_Size" in r else 0)].append(float(r["Counter_Value"]))
    return agg


fe = load(sys.argv[1], "FETCH_SIZE")
wr = load(sys.argv[2], "WRITE_SIZE")
out = []
for key in sorted(set(fe) | set(wr), key=lambda k: -sum(fe.get(k, [0])) - sum(wr.get(k, [0]))):
    f = fe.get(key, []); w = wr.get(key, [])
    fetch_b = 2.0 * 1024.0 * (sum(f) / len(f)) if f else None   # x2: gfx950 FETCH_SIZE correction
    write_b = 1024.0 * (sum(w) / len(w)) if w else None
    out.append({"kernel": key[0], "grid_threads": key[1], "launches": max(len(f), len(w)),
                "hbm_read_bytes_per_launch": fetch_b, "hbm_write_bytes_per_launch": write_b})
json.dump(out, open(sys.argv[3], "w"), indent=1)
for o in out[:25]:
    print(f"{o['kernel'][:60]:60s} grid={o['grid_threads']:>9d} n={o['launches']:3d} "
          f"read={0 if o['hbm_read_bytes_per_launch'] is None else o['hbm_read_bytes_per_launch'] / 1e6:9.2f} MB "
          f"write={0 if o['hbm_write_bytes_per_launch'] is None else o['hbm_write_bytes_per_launch'] / 1e6:9.2f} MB")
