#!/bin/bash
# round 5, call 6: the folded two-token cross-attention — its tests, everything that runs the UNet, the step with and without it
TAG=${1:-r05h}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out
mkdir -p $O
timeout 600 python -m pytest tests/test_cross_attn_fold_gpu.py -q --maxfail=20 2>&1 | tail -60 > $O/${TAG}_fold_tests.log
timeout 1500 python -m pytest tests/test_model_gpu.py tests/test_fullsize_gpu.py tests/test_fullsize_parity_gpu.py tests/test_benchmarked_configs_gpu.py tests/test_reference_callers_gpu.py tests/test_prepost_gpu.py tests/test_ops_gpu.py tests/test_clip_gpu.py -q --maxfail=30 2>&1 | tail -60 > $O/${TAG}_affected_tests.log
for rep in 1 2; do
  timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-train-leg --no-geowizard-leg > $O/${TAG}_bench_fold_on_$rep.json 2>/dev/null
  timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-train-leg --no-geowizard-leg --no-cross-attn-fold > $O/${TAG}_bench_fold_off_$rep.json 2>/dev/null
done
tail -25 $O/${TAG}_fold_tests.log; tail -8 $O/${TAG}_affected_tests.log
python - <<PY
import json
for n in ("fold_on_1","fold_off_1","fold_on_2","fold_off_2"):
    try:
        j=json.load(open("gpurun_out/${TAG}_bench_%s.json"%n)); r=j["roofline"]
        print(n, round(j["value"],2), round(j["ms_per_step"],2), "unet", round(j["stages"]["ms_per_step"]["unet"],2), "latency", round(j["latency_b1_576x768"]["value"],2), "attn", {k:(round(v["ms_per_step"],2), round(v.get("tflops",0))) for k,v in r["other_kernels"].items() if k=="attn"}, "frac", round(r["frac"],4), round(r["frac_reference_formulation"],4))
    except Exception as e: print(n, "failed", e)
PY
