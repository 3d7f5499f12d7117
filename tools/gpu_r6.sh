#!/bin/bash
# Round-6 evidence session: everything lands in gpurun_out/r06/, to be copied into profiles/r06_<tag>_*.
# usage: tools/gpu_r6.sh [tests] [bench] [sq] [flood] [prof] [pmc] [bench608] [dist]
set -uo pipefail
cd "$(dirname "$0")/.."
O=gpurun_out/r06; mkdir -p $O
export TMPDIR=/tmp
R=$PWD
for what in "$@"; do
case $what in
tests)
  timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 | tee $O/pytest_gpu.log
  timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 | tee $O/smoke.log ;;
bench)
  # the driver's command, then a long run with the per-layer table; one batch at a time under the latency plan for the round 1-2 kernels
  timeout 900 python bench.py --steps 20 --warmup 5 2>$O/bench_driver_cmd.err | tee $O/bench_driver_cmd.json | cut -c1-400
  timeout 900 python bench.py --steps 300 --warmup 30 --layers 2>$O/bench.err | tee $O/bench.json | cut -c1-300
  cp gpurun_out/bench_layers_n1.json $O/bench_layers_throughput_plan.json
  timeout 600 python bench.py --steps 300 --warmup 30 --inflight 1 --layers --no-cpu-baseline --no-ref-f32 2>$O/bench_if1.err | tee $O/bench_inflight1.json | cut -c1-300
  cp gpurun_out/bench_layers_n1.json $O/bench_layers_latency_plan.json ;;
sq)
  # SQ counters per kernel (instruction counts, busy cycles): two passes, one batch at a time under each plan's kernels
  for pass in sq1 sq2; do
    ctrs="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_WAVES SQ_INSTS_SALU SQ_ACTIVE_INST_VALU"
    [[ $pass == sq2 ]] && ctrs="SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_I8 SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS"
    for plan in 0 1; do
      rm -rf $O/pmc_$pass
      ( cd /tmp && BENCH_PLAN=$plan timeout 600 rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d "$R/$O/pmc_$pass" -o p -- \
          python "$R/bench.py" --steps 3 --warmup 1 --inflight 1 --no-cpu-baseline --no-ref-f32 --selfcheck-passes 0 --no-preroll-leg > "$R/$O/pmc_$pass.out" 2> "$R/$O/pmc_$pass.err" )
      f=$(find $O/pmc_$pass -name "*counter_collection.csv" | head -1)
      [ -n "$f" ] && python tools/pmc_summary.py "$f" > $O/pmc_${pass}_plan$plan.txt
      rm -rf $O/pmc_$pass
    done
  done
  grep -A8 "conv_first_mfma_pool\|conv_small_pool" $O/pmc_sq1_plan0.txt | grep "kernel\|INSTS_VALU\|INSTS_SALU\|^void" | head -20 ;;
flood)
  timeout 900 python tools/layer_flood.py --plan 1 --inflight 4 --reps 60 | tee $O/layer_flood_throughput_plan.md | tail -16 ;;
prof)
  for tag in default inflight1_latency_plan inflight1_throughput_plan; do
    extra=""; unset BENCH_PLAN
    [[ $tag == inflight1_latency_plan ]] && extra="--inflight 1"
    [[ $tag == inflight1_throughput_plan ]] && extra="--inflight 1" && export BENCH_PLAN=1
    rm -rf $O/prof_$tag
    ( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d "$R/$O/prof_$tag" -o r06 -- \
        python "$R/bench.py" --steps 20 --warmup 5 --no-cpu-baseline --no-ref-f32 $extra > "$R/$O/prof_$tag.json" 2> "$R/$O/prof_$tag.err" )
    db=$(find $O/prof_$tag -name "*results.db" | head -1)
    [ -n "$db" ] && python tools/rocpd_summary.py "$db" $O/kernel_stats_$tag.md | grep "^| \*\*" | head -8
    rm -rf $O/prof_$tag
  done
  unset BENCH_PLAN ;;
pmc)
  for ctr in FETCH_SIZE WRITE_SIZE; do
    rm -rf $O/pmc_$ctr
    ( cd /tmp && timeout 900 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d "$R/$O/pmc_$ctr" -o p -- \
        python "$R/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --no-ref-f32 --selfcheck-passes 0 --serial-steps 2 --no-extra-legs --no-preroll-leg > "$R/$O/pmc_$ctr.json" 2> "$R/$O/pmc_$ctr.err" )
  done
  python tools/pmc_traffic.py $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $O/pmc_traffic.json | head -12
  rm -rf $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE ;;
bench608)
  timeout 900 python bench.py --cfg cfg/yolov3_quant.cfg --batch 32 --steps 30 --warmup 3 --layers --selfcheck-passes 4 --serial-steps 8 2>$O/bench608.err > $O/bench608.json; python -c "import json; d=json.load(open('$O/bench608.json')); print('yolov3-608', d['value'], d['ms_per_step'], d['serial'])"
  grep "\[layer\]" $O/bench608.err > $O/bench608_layers.log; tail -3 $O/bench608.err ;;
dist)
  BENCH_FORCE_DIST=1 timeout 600 python bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-ref-f32 2>$O/bench_dist1.err > $O/bench_dist1.json; python -c "import json; d=json.loads(open('$O/bench_dist1.json').read().strip().splitlines()[-1]); print('force-dist', d['value'], d['config']['weight_broadcast_ms'])" ;;
esac
done
