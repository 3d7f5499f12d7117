#!/bin/bash
# YOLOv3-608 batch 32 (BASELINE config[4]): batches in flight x kernel plan.  usage: v3_ab.sh "plan:inflight ..."
cd "$(dirname "$0")/../.."
for c in ${1:-"-:1 -:3 0:3"}; do
  IFS=: read p n <<< "$c"
  [[ $p != - ]] && export BENCH_PLAN=$p || unset BENCH_PLAN
  python bench.py --cfg cfg/yolov3_quant.cfg --batch 32 --steps 30 --warmup 3 --no-cpu-baseline --no-ref-f32 --selfcheck-passes 4 --serial-steps 8 --inflight $n 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readlines()[-1])
print('yolov3-608 plan $p inflight $n', 'ms/step', d['ms_per_step'], 'img/s', d['value'], 'serial', (d.get('serial') or {}).get('ms_per_step'))"
done 2>&1 | tee gpurun_out/v3_ab.log
