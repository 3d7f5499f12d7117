#!/usr/bin/env python3
"""Same-box A/B of kernel variants: every variant is run in its own process (the libraries load once per process), the variants are
interleaved and the whole round is repeated, because box-to-box and minute-to-minute drift on this pool is larger than most kernel changes.

  tools/ab.py flood  [--layers 0,2,4] A B ...     per layer group: serial / flood us per launch (tools/layer_flood.py), + the whole step
  tools/ab.py bench  A B ...                      bench.py long run: in-flight ms per step, serial ms, roofline fractions
  tools/ab.py micro "c n hw k batch" A B ...      one conv launch (tools/conv_microbench.py)

A variant is  name[:key=value,...]  with keys  lib=<dir under build_ab/ or path>  flags=<mi355_debug_flags>  plan=<0|1>  inflight=<n>
tile=<bm,bn[,nt]>; `cur` alone = the in-tree build with defaults.   Example:  tools/ab.py flood --layers 0,2 cur new:lib=new
"""
import argparse
import json
import os
import re
import subprocess
import sys

ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def parse_variant(v):
    name, _, rest = v.partition(":")
    kv = dict(x.split("=", 1) for x in rest.split(",") if x) if rest else {}
    env = dict(os.environ)
    if "lib" in kv:
        d = kv["lib"] if os.path.isabs(kv["lib"]) else os.path.join(ROOT, "build_ab", kv["lib"])
        env["MI355_LIB_DIR"] = d
    if "flags" in kv:
        env["BENCH_DEBUG_FLAGS"] = kv["flags"]
    if "tile" in kv:
        env["BENCH_FORCE_TILE"] = kv["tile"]
    if "plan" in kv:
        env["BENCH_PLAN"] = kv["plan"]
    return name, kv, env


def run(cmd, env, timeout=900):
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout, cwd=ROOT)
    if r.returncode:
        print("  FAILED:", " ".join(cmd), r.stderr[-800:], file=sys.stderr)
    return r.stdout, r.stderr


ap = argparse.ArgumentParser()
ap.add_argument("what", choices=["flood", "bench", "micro"])
ap.add_argument("rest", nargs="+")
ap.add_argument("--layers", default="", help="flood: first layers of the groups to measure (default: all)")
ap.add_argument("--rounds", type=int, default=2)
ap.add_argument("--reps", type=int, default=60)
ap.add_argument("--steps", type=int, default=300)
ap.add_argument("--cfg", default="")
ap.add_argument("--batch", type=int, default=0)
a = ap.parse_args()
shape = None
if a.what == "micro":
    shape, a.rest = a.rest[0].split(), a.rest[1:]
variants = [parse_variant(v) for v in a.rest]
extra = (["--cfg", a.cfg] if a.cfg else []) + (["--batch", str(a.batch)] if a.batch else [])
for rnd in range(a.rounds):
    for name, kv, env in variants:
        if a.what == "flood":
            cmd = [sys.executable, "tools/layer_flood.py", "--reps", str(a.reps), "--plan", kv.get("plan", "1"), "--inflight", kv.get("inflight", "4")] + extra
            if a.layers:
                cmd += ["--only", a.layers]
            out, _ = run(cmd, env)
            rows = [ln for ln in out.splitlines() if ln.startswith("|") and "->" in ln or ln.startswith("| whole") or ln.startswith("| sum")]
            cells = []
            for ln in rows:
                c = [x.strip() for x in ln.strip("|").split("|")]
                cells.append(f"{c[0]} s {c[2]} f {c[3]}")
            print(f"round {rnd} {name:12s} " + "  ".join(cells), flush=True)
        elif a.what == "bench":
            cmd = [sys.executable, "bench.py", "--steps", str(a.steps), "--warmup", "30", "--no-cpu-baseline", "--no-ref-f32", "--inflight", kv.get("inflight", "4")] + extra
            out, _ = run(cmd, env)
            try:
                d = json.loads([ln for ln in out.splitlines() if ln.startswith("{")][-1])
                r = d.get("roofline") or {}
                print(f"round {rnd} {name:12s} ms/step {d['ms_per_step']} img/s {d['value']} serial {(d.get('serial') or {}).get('ms_per_step')} frac {r.get('frac')} "
                      f"s33 {(r.get('conv3x3_s1_aggregate') or {}).get('frac')} sustained {(r.get('sustained') or {}).get('frac')} lat {(r.get('latency_plan') or {}).get('frac')} "
                      f"s33-sustained {((r.get('conv3x3_s1_aggregate') or {}).get('sustained') or {}).get('frac')} "
                      + " ".join(f"L{x['layer']}:{x['us_per_launch']}" for x in ((r.get('conv3x3_s1_aggregate') or {}).get('sustained') or {}).get('launches', [])), flush=True)
            except Exception as e:  # noqa: BLE001
                print(f"round {rnd} {name}: no result ({e})")
        else:
            cmd = [sys.executable, "tools/conv_microbench.py", "--c", shape[0], "--n", shape[1], "--hw", shape[2], "--k", shape[3], "--batch", shape[4], "--iters", str(a.steps)]
            if "tile" in kv:
                t = kv["tile"].split(",")
                cmd += ["--tile", t[0], t[1]]
            out, _ = run(cmd, env)
            m = re.search(r'"us": ([0-9.]+)', out)
            print(f"round {rnd} {name:12s} {m.group(1) if m else '?'} us", flush=True)
