#!/bin/bash
cd "$(dirname "$0")/../.."
for i in 1 2; do
for v in 0 1; do
  echo "== BENCH_NO_DIRECT_INPUT=$v tiny"; BENCH_NO_DIRECT_INPUT=$v python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-ref-f32 --layers 2>&1 | grep -E '"i": 0,|ms_per_step' | python -c "
import sys,json,re
for l in sys.stdin:
    if l.startswith('[layer]'): print('  L0', json.loads(l.split('[layer] ')[1])['ms'])
    elif l.startswith('{'): d=json.loads(l); print('  step', d['ms_per_step'], 'layout', d['roofline']['input_layout_ms'])
"
done; done
for v in 0 1; do
  echo "== BENCH_NO_DIRECT_INPUT=$v yolov3-608"; BENCH_NO_DIRECT_INPUT=$v python bench.py --cfg cfg/yolov3_quant.cfg --batch 32 --steps 20 --warmup 3 --no-cpu-baseline --layers 2>&1 | grep -E '"i": 0,|ms_per_step' | python -c "
import sys,json,re
for l in sys.stdin:
    if l.startswith('[layer]'): print('  L0', json.loads(l.split('[layer] ')[1])['ms'])
    elif l.startswith('{'): d=json.loads(l); print('  step', d['ms_per_step'], 'layout', d['roofline']['input_layout_ms'])
"
done
