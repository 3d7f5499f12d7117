#!/bin/bash
# rocprofv3 kernel trace of the bench with batches in flight -> overlap table (tools/rocpd_overlap.py)
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
mkdir -p gpurun_out
tag=${1:-flight}
rm -rf gpurun_out/prof_$tag
( cd /tmp && timeout 600 rocprofv3 --kernel-trace -d "$OLDPWD/gpurun_out/prof_$tag" -o t -- \
    python "$OLDPWD/bench.py" --steps 60 --warmup 6 --no-cpu-baseline --no-ref-f32 --selfcheck-passes 0 --serial-steps 2 ${EXTRA:-} > "$OLDPWD/gpurun_out/prof_$tag.json" 2> "$OLDPWD/gpurun_out/prof_$tag.err" )
db=$(find gpurun_out/prof_$tag -name "*results.db" | head -1)
[ -n "$db" ] && python tools/rocpd_overlap.py "$db" gpurun_out/overlap_$tag.md --last 700 | head -${LINES_OUT:-40}
rm -rf gpurun_out/prof_$tag
