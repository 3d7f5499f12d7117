#!/bin/bash
# Rebuild libmi355yolo.so with the timing-ablation switches and phase timestamps compiled in (-DMI355_ABLATE).
# Only for tools/conv_microbench.py --ablate / --timeline on a scratch GPU box: results of ablated launches are wrong.
set -euo pipefail
cd "$(dirname "$0")/../yolo_quantization_amd/csrc"
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -DMI355_ABLATE -Wno-unused-result -shared \
  conv_igemm.hip conv_rows.hip conv_rows_k1.hip conv_small.hip conv1x1.hip conv_ws3.hip conv_aux.hip glue.hip shim.hip -o ../lib/libmi355yolo.so 2>&1 | grep -E "error" -A5 || true
echo "ablate build done"
