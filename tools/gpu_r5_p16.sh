#!/bin/bash
# round 5: conv_pool16.hip (layers 2 / 4 on 16 x 16 x 64 tiles) -- parity, then same-box A/B against conv_small.hip (debug flag 2^30)
cd "$(dirname "$0")/.."
O=gpurun_out/r05; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "pool16" 2>&1 | tail -15 | tee $O/pytest_p16.log
timeout 900 python tools/ab.py flood --layers 2,4 --rounds 2 small:flags=1073741824 v1:lib=p16v1 cur 2>&1 | tee $O/p16_ab_flood.log
