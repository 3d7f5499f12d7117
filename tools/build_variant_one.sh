#!/bin/bash
# Variant build that differs from the in-tree build in ONE translation unit:  tools/build_variant_one.sh <name> <unit> [hipcc flags]
# copies the in-tree objects to build_ab/<name>, recompiles <unit>.hip with the extra flags and links (seconds instead of minutes).
set -euo pipefail
cd "$(dirname "$0")/.."
NAME=$1; UNIT=$2; shift 2
OUT=$PWD/build_ab/$NAME; SRC=$PWD/yolo_quantization_amd/csrc; LIB=$PWD/yolo_quantization_amd/lib
mkdir -p "$OUT"; cp "$LIB"/*.o "$OUT"/
X=""; [ "$UNIT" = conv_rows16 ] && X="-mllvm -pragma-unroll-threshold=1000000"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-function -Wno-inline-asm $X "$@" -c "$SRC/$UNIT.hip" -o "$OUT/$UNIT.o"
[ "$UNIT" = conv_rows ] && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-function -Wno-inline-asm "$@" -c "$SRC/conv_rows_k1.hip" -o "$OUT/conv_rows_k1.o"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$OUT/libmi355yolo.so" "$OUT"/conv_igemm.o "$OUT"/conv_rows.o "$OUT"/conv_rows16.o "$OUT"/conv_rows_k1.o "$OUT"/conv_small.o "$OUT"/conv_small32.o "$OUT"/conv_pool16.o "$OUT"/conv1x1.o "$OUT"/conv_ws3.o "$OUT"/conv_aux.o "$OUT"/glue.o "$OUT"/comm.o "$OUT"/shim.o -ldl
cp "$LIB"/libdarknet_q.so "$OUT"/
echo "variant build: $OUT"
