set -uo pipefail
cd "$(dirname "$0")/.."; mkdir -p gpurun_out/pmcb; export TMPDIR=/tmp; R=$PWD
run_pass() { local name=$1; shift
  ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$R/gpurun_out/pmcb/$name" -o p -- python "$R/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --no-ref-f32 > "$R/gpurun_out/pmcb/$name.out" 2> "$R/gpurun_out/pmcb/$name.err" )
  local f=$(find gpurun_out/pmcb/$name -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python tools/pmc_summary.py "$f" | tee gpurun_out/pmcb/$name.summary
}
run_pass sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_WAVES SQ_INSTS_SALU SQ_ACTIVE_INST_VALU
run_pass sq2 SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_I8 SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS
