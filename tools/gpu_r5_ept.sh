#!/bin/bash
# round 5, first A/B: epilogue table (host-derived per-channel constants) against the HEAD build, same box
cd "$(dirname "$0")/.."
O=gpurun_out/r05; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "fused_maxpool or small_channel_pool" 2>&1 | tail -4 | tee $O/pytest_ept.log
python tools/ab.py flood --layers 0,2,4,6 --rounds 2 base:lib=base cur 2>&1 | tee $O/ept_ab_flood.log
MI355_LIB_DIR=build_ab/libablate python tools/l0_phases.py --inflight 4 2>&1 | head -3
python tools/ab.py bench --rounds 2 base:lib=base cur 2>&1 | tee $O/ept_ab_bench.log
