#!/bin/bash
# Round-2 GPU session: tests, smoke, bench (tiny + yolov3), optional profile.  Everything lands in gpurun_out/.
# usage: tools/gpu_r2.sh [tests] [bench] [bench608] [prof] [pmc]
set -uo pipefail
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
for what in "$@"; do
case $what in
tests)
  timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 | tee gpurun_out/pytest_gpu.log; echo
  timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -5 | tee gpurun_out/smoke.log ;;
newtests)
  timeout 2400 python -m pytest tests/test_gpu_refpin.py tests/test_gpu_residual.py -m gpu -x -q 2>&1 | tail -15 | tee gpurun_out/pytest_gpu_new.log; echo ;;
k2tests)
  timeout 2400 python -m pytest tests/test_gpu_r2_kernels.py -m gpu -x -q 2>&1 | tail -15 | tee gpurun_out/pytest_gpu_k2.log; echo ;;
quicktests)
  timeout 2400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_r2_kernels.py -m gpu -x -q 2>&1 | tail -30 | tee gpurun_out/pytest_gpu_quick.log ;;
bench)
  timeout 900 python bench.py --steps 200 --warmup 20 --layers 2>gpurun_out/bench.err | tee gpurun_out/bench.json
  tail -40 gpurun_out/bench.err
  cp gpurun_out/bench_layers_n1.json gpurun_out/bench_layers_clean.json ;;
bench608)
  timeout 900 python bench.py --cfg cfg/yolov3_quant.cfg --batch 32 --steps 20 --warmup 3 --layers 2>gpurun_out/bench608.err | tee gpurun_out/bench608.json
  grep "\[layer\]" gpurun_out/bench608.err > gpurun_out/bench608_layers.log; tail -5 gpurun_out/bench608.err ;;
prof)
  rm -rf gpurun_out/prof
  ( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d "$OLDPWD/gpurun_out/prof" -o r02 -- \
      python "$OLDPWD/bench.py" --steps 10 --warmup 2 --no-cpu-baseline --no-ref-f32 > "$OLDPWD/gpurun_out/prof_bench.json" 2> "$OLDPWD/gpurun_out/prof.err" )
  db=$(find gpurun_out/prof -name "*results.db" | head -1)
  [ -n "$db" ] && python tools/rocpd_summary.py "$db" gpurun_out/kernel_stats.md | head -40 ;;
pmc)
  for ctr in FETCH_SIZE WRITE_SIZE; do
    rm -rf gpurun_out/pmc_$ctr
    ( cd /tmp && timeout 900 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d "$OLDPWD/gpurun_out/pmc_$ctr" -o p -- \
        python "$OLDPWD/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --no-ref-f32 > "$OLDPWD/gpurun_out/pmc_$ctr.json" 2> "$OLDPWD/gpurun_out/pmc_$ctr.err" )
  done
  python tools/pmc_traffic.py gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE gpurun_out/pmc_traffic.json | head -30 ;;
sq)
  bash tools/pmc_bench.sh 2>&1 | tail -5 ;;
esac
done
