#!/bin/bash
# Build libmi355yolo.so for gfx950 (cross-compiles without a GPU).  -ffp-contract=off: the requantise epilogue
# must execute the reference's double operations one by one (no FMA contraction).
set -euo pipefail
HERE="$(cd "$(dirname "$0")" && pwd)"
OUT="${MI355_BUILD_OUT:-$HERE/../lib}"
mkdir -p "$OUT"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-function -Wno-inline-asm"
# (-Wno-inline-asm: the LDS-DMA macros declare the m0 they write as clobbered -- ADVICE r05 -- and clang warns that m0 is a reserved register on every expansion)
pids=()
for f in conv_igemm conv_rows conv_rows16 conv_rows_k1 conv_small conv_small32 conv_pool16 conv1x1 conv_ws3 conv_aux glue comm shim; do
  if [ ! -f "$OUT/$f.o" ] || [ "$HERE/$f.hip" -nt "$OUT/$f.o" ] || [ "$HERE/conv_rows.hip" -nt "$OUT/$f.o" -a "$f" = conv_rows_k1 ] || [ "$HERE/kargs.h" -nt "$OUT/$f.o" ] || \
     [ "$HERE/common.h" -nt "$OUT/$f.o" ] || [ "$HERE/../../include/mi355_yolo_int8.h" -nt "$OUT/$f.o" ]; then
    # conv_rows16: its 128-row wave tiles unroll past clang's default pragma-unroll budget; a loop left rolled indexes the accumulator
    # array dynamically, which then lives in scratch memory (one scratch store behind every MFMA)
    X=""; [ "$f" = conv_rows16 ] && X="-mllvm -pragma-unroll-threshold=1000000"
    $HIPCC $FLAGS $X ${EXTRA_HIPCC_FLAGS:-} -c "$HERE/$f.hip" -o "$OUT/$f.o" &
    pids+=($!)
  fi
done
fail=0
for p in "${pids[@]:-}"; do if [ -n "$p" ]; then wait "$p" || fail=1; fi; done
if [ "$fail" != 0 ]; then echo "build.sh: a translation unit failed to compile" >&2; exit 1; fi
$HIPCC --offload-arch=gfx950 -shared -fPIC -o "$OUT/libmi355yolo.so" "$OUT"/conv_igemm.o "$OUT"/conv_rows.o "$OUT"/conv_rows16.o "$OUT"/conv_rows_k1.o "$OUT"/conv_small.o "$OUT"/conv_small32.o "$OUT"/conv_pool16.o "$OUT"/conv1x1.o "$OUT"/conv_ws3.o "$OUT"/conv_aux.o "$OUT"/glue.o "$OUT"/comm.o "$OUT"/shim.o -ldl
echo "built $OUT/libmi355yolo.so"
