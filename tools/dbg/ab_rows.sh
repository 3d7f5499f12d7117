#!/bin/bash
# A/B of a mi355_debug_flags bit inside one box, conv_rows layers (L12, L21) + the step
cd "$(dirname "$0")/../.."
for f in 0 $1 0 $1; do
  echo "== flags=$f"; BENCH_DEBUG_FLAGS=$f python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-ref-f32 --layers 2>&1 | grep -E '"i": (8|10|12|14|21),|ms_per_step' | python -c "
import sys,json
o=[]
for l in sys.stdin:
    if l.startswith('[layer]'): r=json.loads(l.split('[layer] ')[1]); o.append('L%d %.1f' % (r['i'], r['ms']*1000))
    elif l.startswith('{'): d=json.loads(l); o.append('step %.4f frac %.4f agg3x3 %.4f' % (d['ms_per_step'], d['roofline']['frac'], d['roofline']['conv3x3_s1_aggregate']['frac']))
print('  '+'  '.join(o))
"
done
