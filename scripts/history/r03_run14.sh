#!/bin/bash
# whole-step A/B of the fused GroupNorm -> conv route
O=gpurun_out; mkdir -p $O
for v in 1 0 1 0; do
  timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-train-leg --no-latency-leg --set-option fused_norm=$v > $O/r03p_bench_fused_norm_$v.json 2>/dev/null
  python - <<PY
import json
j=json.loads(open("gpurun_out/r03p_bench_fused_norm_$v.json").read().strip().splitlines()[-1])
r=j["roofline"]
print("fused_norm=$v", round(j["value"],2), round(j["ms_per_step"],2), round(r["frac"],4), round(r["kernel_ms_per_step"],2), {k:round(v["ms_per_step"],2) for k,v in r["other_kernels"].items()})
PY
done
