#!/bin/bash
# A/B: forced conv_rows tiles (L12 / L21 of yolov3-tiny) x batches in flight.  usage: tile_ab.sh "flags:tile:inflight ..."
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
for c in ${1:-"0:-:1 0:-:3"}; do
  IFS=: read f t n <<< "$c"
  env=""; [[ $t != - ]] && export BENCH_FORCE_TILE=$t || unset BENCH_FORCE_TILE
  BENCH_DEBUG_FLAGS=$f python bench.py --steps ${STEPS:-300} --warmup 30 --no-cpu-baseline --no-ref-f32 --inflight $n 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readlines()[-1])
r = d.get('roofline') or {}
print('flags $f tile $t inflight $n', 'ms/step', d['ms_per_step'], 'img/s', d['value'], 'serial', (d.get('serial') or {}).get('ms_per_step'), 'rows ms', r.get('ms_per_launch_avg'), 'frac', r.get('frac'))"
done 2>&1 | tee gpurun_out/tile_ab.log
