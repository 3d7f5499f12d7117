#!/bin/bash
# kernel-trace durations and SQ instruction counters of the first-layer kernel, product build vs build_ab/base, layer 0 flooded on 4 instances
cd "$(dirname "$0")/.."
O=gpurun_out/r05; mkdir -p $O
export TMPDIR=/tmp
R=$PWD
for v in base cur; do
  L=""; [[ $v == base ]] && L="$R/build_ab/base"
  rm -rf $O/prof_l0_$v
  ( cd /tmp && MI355_LIB_DIR=$L timeout 600 rocprofv3 --kernel-trace --stats -d "$R/$O/prof_l0_$v" -o t -- python "$R/tools/layer_flood.py" --only 0 --reps 200 > "$R/$O/prof_l0_$v.out" 2>&1 )
  db=$(find $O/prof_l0_$v -name "*results.db" | head -1)
  [ -n "$db" ] && python tools/rocpd_summary.py "$db" $O/l0_kernel_stats_$v.md | grep "first" | head -3
  grep "first" $O/l0_kernel_stats_$v.md | head -3
  rm -rf $O/prof_l0_$v
  for ctrs in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAVES SQ_BUSY_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_I8 SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"; do
    rm -rf $O/pmc_l0
    ( cd /tmp && MI355_LIB_DIR=$L timeout 600 rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d "$R/$O/pmc_l0" -o p -- python "$R/tools/layer_flood.py" --only 0 --reps 5 --inflight 1 > /dev/null 2>&1 )
    f=$(find $O/pmc_l0 -name "*counter_collection.csv" | head -1)
    [ -n "$f" ] && python tools/pmc_summary.py "$f" | grep -A8 "first_mfma_pool" | head -12 | tee -a $O/l0_pmc_$v.txt
    rm -rf $O/pmc_l0
  done
done
