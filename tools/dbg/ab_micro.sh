#!/bin/bash
# A/B/C of library builds inside one box on single conv shapes (tools/conv_microbench.py), interleaved and repeated:
#   tools/dbg/ab_micro.sh "<c> <n> <hw> <k> <batch>" dirA dirB ...   ("" = the in-tree lib/)
cd "$(dirname "$0")/../.."
shape=($1); shift
for rep in 1 2 3; do
  for lib in "$@"; do
    r=$(MI355_LIB_DIR=${lib:+$PWD/$lib} python tools/conv_microbench.py --c ${shape[0]} --n ${shape[1]} --hw ${shape[2]} --k ${shape[3]} --batch ${shape[4]} --iters ${ITERS:-300} | python -c "import sys,json; print(json.loads(sys.stdin.readline())['us'])")
    echo "rep $rep  ${lib:-current}  $r us"
  done
done
