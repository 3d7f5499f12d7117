#!/usr/bin/env python3
"""Overlap analysis of a rocprofv3 kernel trace (rocpd SQLite) taken with several batches in flight: per kernel its
average duration, how much of the traced span has 0 / 1 / 2 / 3+ kernels running, and for every pair of kernels the time
they ran side by side.  usage: tools/rocpd_overlap.py <results.db> [out.md] [--last N]  (the last N dispatches; default all)"""
import sqlite3
import sys


def short(n):
    n = n.split("(")[0].replace("void ", "")
    return n


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    last = 0
    if "--last" in sys.argv:
        last = int(sys.argv[sys.argv.index("--last") + 1]); args = [a for a in args if a != str(last)]
    db = sqlite3.connect(args[0])
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)").fetchall()]
    scol = "start" if "start" in cols else "start_timestamp"
    ecol = "end" if "end" in cols else "end_timestamp"
    qcol = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else None)
    q = f"select name, {scol}, {ecol}, {qcol or 0}, lds_size, vgpr_count from kernels order by {scol}"
    rows = cur.execute(q).fetchall()
    if last:
        rows = rows[-last:]
    rows = [(short(n), s, e, qq, l, v) for n, s, e, qq, l, v in rows if "rocclr" not in n and "fill_u32" not in n and "checksum" not in n]
    t0 = min(r[1] for r in rows); t1 = max(r[2] for r in rows)
    ev = []
    for i, (n, s, e, qq, l, v) in enumerate(rows):
        ev.append((s, 1, i)); ev.append((e, -1, i))
    ev.sort()
    active = set()
    prev = t0
    conc = {}
    pair = {}
    alone = {}
    for t, kind, i in ev:
        dt = t - prev
        if dt > 0:
            k = len(active)
            conc[k] = conc.get(k, 0) + dt
            names = sorted(rows[j][0] for j in active)
            if k == 1:
                alone[names[0]] = alone.get(names[0], 0) + dt
            for x in range(len(names)):
                for y in range(x + 1, len(names)):
                    pair[(names[x], names[y])] = pair.get((names[x], names[y]), 0) + dt
        prev = t
        if kind == 1:
            active.add(i)
        else:
            active.discard(i)
    span = t1 - t0
    out = [f"span {span / 1e3:.1f} us, {len(rows)} dispatches, queues {sorted(set(r[3] for r in rows))}", "",
           "| kernels running | us | % of span |", "|---|---|---|"]
    for k in sorted(conc):
        out.append(f"| {k} | {conc[k] / 1e3:.1f} | {100.0 * conc[k] / span:.1f} |")
    agg = {}
    for n, s, e, qq, l, v in rows:
        a = agg.setdefault((n, l, v), [0, 0])
        a[0] += 1; a[1] += e - s
    out += ["", "| kernel | LDS B | VGPR | calls | avg us | alone us (total) |", "|---|---|---|---|---|---|"]
    for (n, l, v), (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        out.append(f"| {n} | {l} | {v} | {c} | {t / c / 1e3:.1f} | {alone.get(n, 0) / 1e3:.1f} |")
    out += ["", "| pair running side by side | us |", "|---|---|"]
    for (x, y), t in sorted(pair.items(), key=lambda kv: -kv[1])[:30]:
        out.append(f"| {x} + {y} | {t / 1e3:.1f} |")
    text = "\n".join(out)
    if len(args) > 1:
        open(args[1], "w").write(text + "\n")
    print(text)


if __name__ == "__main__":
    main()
