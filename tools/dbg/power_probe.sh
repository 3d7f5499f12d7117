#!/bin/bash
# shader clock / power while bench.py runs (rocm-smi sampled every 0.2 s).  usage: power_probe.sh "<bench args>"
cd "$(dirname "$0")/../.."
( for i in $(seq 1 60); do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power" | tr '\n' ' '; echo; sleep 0.2; done ) > gpurun_out/smi_probe.log &
SM=$!
sleep 1
python bench.py --steps 6000 --warmup 30 --no-cpu-baseline --no-ref-f32 --no-extra-legs --serial-steps 3000 $1 2>/dev/null | tail -1 | cut -c1-120
wait $SM
grep -o "sclk clock level: [0-9]*: ([0-9]*Mhz)\|Power (W): [0-9.]*\|Socket Graphics Package Power (W): [0-9.]*" gpurun_out/smi_probe.log | paste - - | sort | uniq -c | sort -rn | head -12
