#!/bin/bash
# One GPU-box session: parity tests, smoke, bench, rocprofv3 kernel trace.  Everything lands in gpurun_out/.
# usage: tools/gpu_session.sh [tests|bench|prof|all]
set -uo pipefail
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
what=${1:-all}
export TMPDIR=/tmp
if [[ $what == all || $what == tests ]]; then
  timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -40 | tee gpurun_out/pytest_gpu.log
  timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -5 | tee gpurun_out/smoke.log
fi
if [[ $what == all || $what == bench ]]; then
  timeout 900 python bench.py --steps 200 --warmup 20 --layers 2>gpurun_out/bench.err | tee gpurun_out/bench.json
  tail -40 gpurun_out/bench.err
  cp gpurun_out/bench_layers_n1.json gpurun_out/bench_layers_clean.json   # the rocprofv3 legs below rewrite the former
fi
if [[ $what == all || $what == prof ]]; then
  rm -rf gpurun_out/prof
  ( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d "$OLDPWD/gpurun_out/prof" -o r01 -- \
      python "$OLDPWD/bench.py" --steps 10 --warmup 2 --no-cpu-baseline > "$OLDPWD/gpurun_out/prof_bench.json" 2> "$OLDPWD/gpurun_out/prof.err" )
  db=$(find gpurun_out/prof -name "*results.db" | head -1)
  [ -n "$db" ] && python tools/rocpd_summary.py "$db" gpurun_out/kernel_stats.md | head -40
fi
if [[ $what == all || $what == pmc ]]; then
  # HBM traffic of every kernel of one bench step: separate --pmc passes (FETCH_SIZE and WRITE_SIZE do not fit one
  # pass), kernel-trace only.
  for ctr in FETCH_SIZE WRITE_SIZE; do
    rm -rf gpurun_out/pmc_$ctr
    ( cd /tmp && timeout 900 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d "$OLDPWD/gpurun_out/pmc_$ctr" -o p -- \
        python "$OLDPWD/bench.py" --steps 3 --warmup 1 --no-cpu-baseline > "$OLDPWD/gpurun_out/pmc_$ctr.json" 2> "$OLDPWD/gpurun_out/pmc_$ctr.err" )
  done
  python tools/pmc_traffic.py gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE gpurun_out/pmc_traffic.json | head -30
fi
