#!/bin/bash
# round 5: conv_small<32> / conv_mid read a channel group's constants one group ahead -- parity, then same-box A/B
cd "$(dirname "$0")/.."
O=gpurun_out/r05; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "small_channel or fused_maxpool or pool16" 2>&1 | tail -5 | tee $O/pytest_ahead.log
timeout 900 python tools/ab.py flood --layers 4,6 --rounds 3 base:lib=base cur 2>&1 | tee $O/ahead_ab_flood.log
timeout 900 python tools/ab.py bench --rounds 2 base:lib=base cur 2>&1 | tee $O/ahead_ab_bench.log
