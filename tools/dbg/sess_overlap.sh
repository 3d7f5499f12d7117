#!/bin/bash
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
mkdir -p gpurun_out
tools/dbg/tile_ab.sh "0:-:3 4194304:128,128:3 12582912:128,128:3 79691776:128,128:3 79691776:128,128:4 0:128,128:3 4194304:-:3 0:-:3" 
for tag in default small; do
  if [[ $tag == small ]]; then export BENCH_DEBUG_FLAGS=79691776 BENCH_FORCE_TILE=128,128; fi
  rm -rf gpurun_out/prof_$tag
  ( cd /tmp && timeout 600 rocprofv3 --kernel-trace -d "$OLDPWD/gpurun_out/prof_$tag" -o t -- \
      python "$OLDPWD/bench.py" --steps 60 --warmup 6 --no-cpu-baseline --no-ref-f32 --selfcheck-passes 0 --serial-steps 2 > "$OLDPWD/gpurun_out/prof_$tag.json" 2> "$OLDPWD/gpurun_out/prof_$tag.err" )
  db=$(find gpurun_out/prof_$tag -name "*results.db" | head -1)
  [ -n "$db" ] && python tools/rocpd_overlap.py "$db" gpurun_out/overlap_$tag.md --last 600 | head -70
done
