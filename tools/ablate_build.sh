#!/bin/bash
# Build libmi355yolo.so with the timing-ablation switches and phase timestamps compiled in (-DMI355_ABLATE) into
# build_ab/libablate (select it with MI355_LIB_DIR).  Only for tools/conv_microbench.py --ablate / --timeline / --waveprof
# on a scratch GPU box: results of ablated launches are wrong.
set -euo pipefail
cd "$(dirname "$0")/.."
OUT=$PWD/build_ab/libablate
mkdir -p "$OUT"
MI355_BUILD_OUT=$OUT EXTRA_HIPCC_FLAGS="-DMI355_ABLATE -Wno-unused-result" bash yolo_quantization_amd/csrc/build.sh
cp yolo_quantization_amd/lib/libdarknet_q.so "$OUT"/
echo "ablate build done: $OUT"
