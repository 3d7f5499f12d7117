#!/bin/bash
# TEST INFRASTRUCTURE.  Builds the *unmodified* reference (ArtyZe/yolo_quantization) from the sources where
# they lie under $REF (default /root/reference) into oracle/_ref/ -- outputs only, never sources.
#
#   oracle/_ref/libdarknet_ref.so      Makefile-default flags  (GPU=0 QUANTIZATION=1 -Ofast), 1 thread
#   oracle/_ref/libdarknet_ref_omp.so  MULTI_CORE=1 flavour    (-fopenmp; `#pragma omp parallel for` in gemm)
#
# We do not run the reference's own Makefile: the flags below restate Makefile:33-37,88-96 and the object list
# Makefile:98.  The only deviation is `-idirafter include` instead of `-Iinclude`: the reference ships a
# Windows `include/unistd.h` shim (`#include <io.h>`) that shadows the system header under -I.
# The MKL flavour (OPENBLAS=1, ref Makefile:56-61) needs mkl.h / mkl_cblas.h: the image carries libmkl_rt.so (/opt/conda/lib) but NOT the
# headers, and stand-in headers are not allowed -> unbuildable here (its epilogue is restated in oracle.c:orc_requant_mkl, unpinned).
set -euo pipefail
REF=${REF:-/root/reference}
HERE="$(cd "$(dirname "$0")" && pwd)"
OUT="$HERE/_ref"
if [ ! -d "$REF/src" ]; then
  echo "build_ref: $REF not present (GPU box?) -- keeping prebuilt $OUT" >&2
  exit 0
fi
SRCS="gemm utils cuda deconvolutional_layer convolutional_layer image activations im2col col2im blas crop_layer
maxpool_layer softmax_layer data matrix network connected_layer parser option_list detection_layer route_layer
upsample_layer box normalization_layer avgpool_layer layer local_layer shortcut_layer logistic_layer
activation_layer batchnorm_layer region_layer reorg_layer tree yolo_layer list"
COMMON="-idirafter $REF/include -I$REF/src -DQUANTIZATION -Wall -Wno-unused-result -Wno-unknown-pragmas -w -fPIC -Ofast"
build_flavour() {   # $1 = obj dir suffix, $2 = extra cflags, $3 = output .so, $4 = extra ldflags
  local od="$OUT/obj$1"; mkdir -p "$od"
  local pids=()
  for s in $SRCS; do
    if [ ! -f "$od/$s.o" ] || [ "$REF/src/$s.c" -nt "$od/$s.o" ]; then
      gcc $COMMON $2 -c "$REF/src/$s.c" -o "$od/$s.o" &
      pids+=($!)
    fi
  done
  for p in "${pids[@]:-}"; do [ -n "$p" ] && wait "$p"; done
  gcc $COMMON $2 -c "$HERE/ref_driver.c" -o "$od/ref_driver.o"
  gcc -shared -o "$OUT/$3" "$od"/*.o -lm -pthread $4
}
build_flavour ""     ""         libdarknet_ref.so     ""
build_flavour "_omp" "-fopenmp" libdarknet_ref_omp.so "-lgomp"
# The reference bound to the MI355X kernels: the SAME unmodified reference objects + integration/mi355_glue.c (the file a
# maintainer adds as src/mi355_glue.c, compiled against the reference's own include/darknet.h) + the driver's -DMI355 hooks,
# linked against the product's C-ABI library.  Proves the drop-in boundary: tests/test_gpu_refbind.py runs it on the GPU box.
SHIM="$HERE/../yolo_quantization_amd/lib"
if [ -f "$SHIM/libmi355yolo.so" ]; then
  od="$OUT/obj"
  gcc $COMMON -DMI355 -I"$HERE/../include" -I"$HERE/../integration" -c "$HERE/../integration/mi355_glue.c" -o "$OUT/mi355_glue.o"
  gcc $COMMON -DMI355 -I"$HERE/../include" -I"$HERE/../integration" -c "$HERE/ref_driver.c" -o "$OUT/ref_driver_mi355.o"
  objs=$(ls "$od"/*.o | grep -v ref_driver.o)
  gcc -shared -o "$OUT/libdarknet_ref_mi355.so" $objs "$OUT/mi355_glue.o" "$OUT/ref_driver_mi355.o" -lm -pthread \
      -L"$SHIM" -lmi355yolo -Wl,-rpath,'$ORIGIN/../../yolo_quantization_amd/lib'
fi
echo "build_ref: OK -> $OUT"
