#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd SQLite result (kernel trace) as a per-kernel stats table (name, calls, total/avg/min/max
ns, %), plus per-(kernel, grid) rows -- the same content as rocprofv3's kernel_stats.csv.
usage: tools/rocpd_summary.py <results.db> [out.md]"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    cur = db.cursor()
    rows = cur.execute("select name, grid_x, workgroup_x, lds_size, vgpr_count, accum_vgpr_count, duration from kernels").fetchall()
    tot = sum(r[6] for r in rows)
    agg = {}
    for name, gx, wx, lds, vg, ag, d in rows:
        short = name.split("(")[0]
        for key in ((short, None), (short, (gx, wx, lds, vg, ag))):
            a = agg.setdefault(key, [0, 0, 1 << 62, 0])
            a[0] += 1; a[1] += d; a[2] = min(a[2], d); a[3] = max(a[3], d)
    out = ["| kernel | grid | wg | LDS B | VGPR+AGPR | calls | total ns | avg ns | min ns | max ns | % |", "|---|---|---|---|---|---|---|---|---|---|---|"]
    for (short, shape), (n, t, mn, mx) in sorted(agg.items(), key=lambda kv: (-agg[(kv[0][0], None)][1], kv[0][1] is not None, -kv[1][1])):
        if shape is None:
            out.append(f"| **{short}** | | | | | {n} | {t} | {t // n} | {mn} | {mx} | {100.0 * t / tot:.1f} |")
        else:
            gx, wx, lds, vg, ag = shape
            out.append(f"| &nbsp;&nbsp;{short} | {gx // wx} | {wx} | {lds} | {vg}+{ag} | {n} | {t} | {t // n} | {mn} | {mx} | {100.0 * t / tot:.1f} |")
    text = "\n".join(out)
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(text + "\n")
    print(text)


if __name__ == "__main__":
    main()
