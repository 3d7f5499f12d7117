#!/bin/bash
# PMC counter passes (separate runs, kernel-trace only -- no sys/hip/hsa tracing) on the conv microbench.
# usage: tools/pmc_session.sh "<microbench args>" <tag> [pass ...]     pass = name:CTR1,CTR2,...
set -uo pipefail
cd "$(dirname "$0")/.."
ARGS=${1:-"--c 512 --n 1024 --hw 13 --batch 64"}
TAG=${2:-pmc}
shift 2 || true
mkdir -p gpurun_out/$TAG
export TMPDIR=/tmp
R=$PWD
run_pass() { # $1 = pass name, rest = counters
  local name=$1; shift
  ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$R/gpurun_out/$TAG/$name" -o p -- \
      python "$R/tools/conv_microbench.py" $ARGS --iters 3 > "$R/gpurun_out/$TAG/$name.out" 2> "$R/gpurun_out/$TAG/$name.err" )
  local f=$(find gpurun_out/$TAG/$name -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then python tools/pmc_summary.py "$f" | tee gpurun_out/$TAG/$name.summary; else echo "no counter csv for $name"; tail -5 gpurun_out/$TAG/$name.err; fi
}
if [ $# -eq 0 ]; then
  set -- "sq1:SQ_WAVE_CYCLES,SQ_BUSY_CYCLES,SQ_WAIT_ANY,SQ_WAIT_INST_ANY,SQ_ACTIVE_INST_ANY,SQ_VALU_MFMA_BUSY_CYCLES,SQ_INSTS_VALU,SQ_WAVES" \
         "sq2:SQ_LDS_BANK_CONFLICT,SQ_LDS_IDX_ACTIVE,SQ_INSTS_LDS,SQ_WAIT_INST_LDS,SQ_ACTIVE_INST_LDS,SQ_ACTIVE_INST_VALU,SQ_INSTS_SALU,SQ_INSTS_VMEM" \
         "sq3:SQ_INST_LEVEL_VMEM,SQ_INST_LEVEL_LDS,SQ_INSTS_VALU_MFMA_I8,SQ_VALU_MFMA_COEXEC_CYCLES,SQ_ACTIVE_INST_MISC,SQ_ACTIVE_INST_SCA,SQ_INSTS_SMEM,SQ_IFETCH" \
         "tcc1:TCC_HIT_sum,TCC_MISS_sum" "tcc2:FETCH_SIZE" "tcc3:WRITE_SIZE"
fi
for p in "$@"; do
  name=${p%%:*}; ctrs=${p#*:}
  run_pass $name ${ctrs//,/ }
done
