#!/bin/bash
# A/B: kernel plan x batches in flight.  usage: plan_ab.sh "plan:inflight[:flags] ..."   (plan - = automatic)
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
for c in ${1:-"-:1 -:3 0:3 -:3 0:3"}; do
  IFS=: read p n f <<< "$c"
  [[ $p != - ]] && export BENCH_PLAN=$p || unset BENCH_PLAN
  BENCH_DEBUG_FLAGS=${f:-0} python bench.py --steps ${STEPS:-300} --warmup 30 --no-cpu-baseline --no-ref-f32 --inflight $n ${EXTRA:-} 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readlines()[-1])
r = d.get('roofline') or {}
print('plan $p inflight $n flags ${f:-0}', 'ms/step', d['ms_per_step'], 'img/s', d['value'], 'serial', (d.get('serial') or {}).get('ms_per_step'), 'rows ms', r.get('ms_per_launch_avg'), 'frac', r.get('frac'), 's33', (r.get('conv3x3_s1_aggregate') or {}).get('frac'))"
done 2>&1 | tee gpurun_out/plan_ab.log
