#!/bin/bash
# quick GPU check: parity subset (-k expr in $1), then serial / in-flight bench and the serial per-layer table
cd "$(dirname "$0")/../.."
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_r2_kernels.py tests/test_gpu_replica.py -m gpu -x -q -k "${1:-first or tiny or net or whole}" 2>&1 | tail -3
tools/dbg/plan_ab.sh "${2:--:1 -:3 -:3}"
python - <<PY
import json
d=json.load(open("gpurun_out/bench_layers_n1.json"))
print("serial", d["ms_per_step"], [(r["i"], round(r["ms"]*1e3,1)) for r in d["layers"] if r["type"]==0])
PY
