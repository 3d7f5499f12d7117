mkdir -p gpurun_out/s1
( timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "small_channel or fused_maxpool or stride2 or tiny_416_exact or chain_608" 2>&1 | tail -8 ) > gpurun_out/s1/pytest.log 2>&1
( timeout 900 python tools/ab.py flood --layers 2,4,6 --rounds 3 base:lib=base cur ) > gpurun_out/s1/ab_flood.log 2>&1
( timeout 600 python tools/ab.py flood --layers 2,4,6 --rounds 2 base:lib=base,plan=0,inflight=1 cur:plan=0,inflight=1 ) > gpurun_out/s1/ab_flood_lat.log 2>&1
( timeout 300 python tools/ab.py micro "256 256 52 3 32" --rounds 1 cur cur128:tile=128,128 cur256:tile=128,256 ) > gpurun_out/s1/cfg1.log 2>&1
cat gpurun_out/s1/*.log
