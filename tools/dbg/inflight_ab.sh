#!/bin/bash
# A/B: batches in flight per GPU (bench.py --inflight): 1 = one stream, step after step; n = consecutive steps on
# n streams / activation sets, kernels of neighbouring steps overlap on the device.
# usage: tools/dbg/inflight_ab.sh "64:1 64:2 64:3 32:4 ..."   (batch:inflight[:g for --graph])
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
combos=${1:-"64:1 64:2 64:3 64:1 64:2 64:3"}
for c in $combos; do
  b=${c%%:*}; r=${c#*:}; n=${r%%:*}; g=""; [[ $r == *:g ]] && g="--graph"
  steps=$(( ${STEPS:-300} * 64 / b ))
  python bench.py --batch $b --steps $steps --warmup 30 --no-cpu-baseline --no-ref-f32 --inflight $n $g 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readlines()[-1])
print('batch', $b, 'inflight', $n, '$g', 'ms/step', d['ms_per_step'], 'img/s', d['value'], 'rows frac', (d.get('roofline') or {}).get('frac'))"
done 2>&1 | tee gpurun_out/inflight_ab.log
