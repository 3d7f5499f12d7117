#!/bin/bash
# A/B of two builds of the libraries inside one box: lib/ (current) against build_ab/libold
cd "$(dirname "$0")/../.."
for lib in "" "$PWD/build_ab/libold" "" "$PWD/build_ab/libold"; do
  echo "== bench lib=${lib:-current}"; MI355_LIB_DIR=$lib python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-ref-f32 --layers 2>&1 | grep -E "\"i\": (${AB_LAYERS:-0|2|4|6}),|ms_per_step" | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('[layer]'): r=json.loads(l.split('[layer] ')[1]); print('  L%d %.2f us' % (r['i'], r['ms']*1000))
    elif l.startswith('{'): d=json.loads(l); print('  step', d['ms_per_step'], 'agg3x3', d['roofline']['conv3x3_s1_aggregate']['frac'])
"
done
