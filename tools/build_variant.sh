#!/bin/bash
# Build the product libraries into build_ab/<name> (git-ignored, travels to the GPU box) so that tools/ab.py can run two builds against
# each other inside one box:   tools/build_variant.sh <name> [extra hipcc flags, e.g. -DSOME_SWITCH]
# "tools/build_variant.sh base" before editing a kernel keeps the old build around as the A side.
set -euo pipefail
cd "$(dirname "$0")/.."
OUT=$PWD/build_ab/$1; shift
mkdir -p "$OUT"
MI355_BUILD_OUT=$OUT EXTRA_HIPCC_FLAGS="$*" bash yolo_quantization_amd/csrc/build.sh
make -s -C yolo_quantization_amd/host
cp yolo_quantization_amd/lib/libdarknet_q.so "$OUT"/
echo "variant build: $OUT"
