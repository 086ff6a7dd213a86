#!/bin/bash
# round 5, call 3: fp32 fused attention (attn32.hip), the phases on igemm2, the 4x4 dgrad — tests first, then what they buy in the training legs; loader throughput
TAG=${1:-r05c}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out
mkdir -p $O
timeout 900 python -m pytest tests/test_attn32_gpu.py tests/test_upconv_phases_gpu.py -q --maxfail=40 2>&1 | tail -80 > $O/${TAG}_new_tests.log
timeout 1200 python -m pytest tests/test_bwd_gpu.py tests/test_ops_gpu.py tests/test_train_gpu.py tests/test_model_gpu.py tests/test_fullsize_parity_gpu.py tests/test_reference_callers_gpu.py -q --maxfail=30 2>&1 | tail -60 > $O/${TAG}_affected_tests.log
for a in "16 5 5184 5 fp32" "16 5 5184 3 fp32 bwd" "16 10 1296 10 fp32" "16 10 1296 5 fp32 bwd" "8 5 9216 3 fp32"; do python scripts/attn_bench.py $a; done > $O/${TAG}_attn32_bench.txt 2>&1
timeout 400 python bench.py --train --dtype fp32 --steps 3 --warmup 1 --detail $O/${TAG}_bench_train_fp32_per_shape.tsv > $O/${TAG}_bench_train_fp32.json 2>/dev/null
timeout 400 python bench.py --train --dtype fp32 --steps 3 --warmup 1 --round4-paths > $O/${TAG}_bench_train_fp32_round4_paths.json 2>/dev/null
timeout 300 python bench.py --train --dtype bf16 --steps 6 --warmup 2 --detail $O/${TAG}_bench_train_bf16_per_shape.tsv > $O/${TAG}_bench_train_bf16.json 2>/dev/null
timeout 300 python bench.py --train --dtype bf16 --steps 6 --warmup 2 --round4-paths > $O/${TAG}_bench_train_bf16_round4_paths.json 2>/dev/null
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-train-leg --no-latency-leg --no-geowizard-leg > $O/${TAG}_bench_inference.json 2>/dev/null
timeout 400 python scripts/loader_bench.py --samples 24 --workers 32 --batch 8 --epochs 3 > $O/${TAG}_loader_bench.json 2> $O/${TAG}_loader_bench.err
tail -30 $O/${TAG}_new_tests.log; tail -12 $O/${TAG}_affected_tests.log; grep -v amdgpu.ids $O/${TAG}_attn32_bench.txt
python - <<PY
import json
for n in ("train_fp32","train_fp32_round4_paths","train_bf16","train_bf16_round4_paths"):
    try:
        j=json.load(open("gpurun_out/${TAG}_bench_%s.json"%n)); print(n, j["value"], j.get("median_ms_per_step"), j["roofline"]["frac"], {k: round(v["ms_per_step"],1) for k,v in j["roofline"]["other_kernels"].items() if v["ms_per_step"]>5})
    except Exception as e: print(n, "failed", e)
try:
    j=json.load(open("gpurun_out/${TAG}_bench_inference.json")); print("inference", j["value"], j["ms_per_step"], j.get("stages",{}).get("ms_per_step"))
except Exception as e: print("inference failed", e)
try: print(open("gpurun_out/${TAG}_loader_bench.json").read()[:1500])
except Exception as e: print(e)
PY
