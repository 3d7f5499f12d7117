#!/bin/bash
# round 5: first-layer kernel A/B against the HEAD-of-round build (build_ab/base), same box
cd "$(dirname "$0")/.."
O=gpurun_out/r05; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_r2_kernels.py -m gpu -x -q -k "fused_maxpool or small_channel_pool or first or planar or tiny_unit or direct" 2>&1 | tail -4 | tee $O/pytest_l0.log
python tools/ab.py flood --layers 0,2 --rounds 2 base:lib=base cur 2>&1 | tee $O/l0_ab_flood.log
MI355_LIB_DIR=build_ab/libablate python tools/l0_phases.py --inflight 4 2>&1 | tee $O/l0_phases.log
python tools/ab.py bench --rounds 2 base:lib=base cur 2>&1 | tee $O/l0_ab_bench.log
