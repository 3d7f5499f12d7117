#!/bin/bash
cd "$(dirname "$0")/../.."
for p in 0 1; do
BENCH_PLAN=$p python bench.py --cfg cfg/yolov3_quant.cfg --batch 32 --steps 24 --warmup 3 --no-cpu-baseline --no-ref-f32 --selfcheck-passes 0 --inflight 1 --layers >/dev/null 2>gpurun_out/v3_layers_$p.err
cp gpurun_out/bench_layers_yolov3_quant_n1.json gpurun_out/v3_layers_plan$p.json
done
python - <<PY
import json
a=json.load(open("gpurun_out/v3_layers_plan0.json")); b=json.load(open("gpurun_out/v3_layers_plan1.json"))
print("plan0", a["ms_per_step"], "plan1", b["ms_per_step"])
agg={}
for x,y in zip(a["layers"], b["layers"]):
    if x["type"]!=0: 
        k=("type",x["type"])
    else:
        k=(x["k"],x["c"],x["n"],x["hw"])
    e=agg.setdefault(k,[0,0,0]); e[0]+=1; e[1]+=x["ms"]*1e3; e[2]+=y["ms"]*1e3
for k,(n,t0,t1) in sorted(agg.items(), key=lambda kv:-abs(kv[1][2]-kv[1][1])):
    print(k, n, "plan0 %.1f us  plan1 %.1f us  per-launch %.1f -> %.1f" % (t0, t1, t0/n, t1/n))
PY
