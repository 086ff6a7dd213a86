#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out
timeout 600 python -m pytest tests/test_two_ranks_one_gpu.py tests/test_attn512_gpu.py -q 2>&1 | tail -8 > $O/r03c_new_tests.log
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_clip_gpu.py tests/test_model_gpu.py tests/test_train_gpu.py tests/test_bwd_gpu.py tests/test_reference_callers_gpu.py -q 2>&1 | tail -25 > $O/r03c_attn_tests.log
timeout 600 python -m pytest tests/test_fullsize_parity_gpu.py tests/test_benchmarked_configs_gpu.py -q -s -k "attention or geowizard or config1_768" 2>&1 | grep -v "^$" | tail -20 > $O/r03c_fullsize_attn.log
timeout 600 python bench.py --steps 15 --warmup 4 --no-train-leg --no-cpu-baseline --detail $O/r03c_bench_per_shape.tsv > $O/r03c_bench.json 2> $O/r03c_bench.err
tail -8 $O/r03c_new_tests.log; tail -12 $O/r03c_attn_tests.log; tail -12 $O/r03c_fullsize_attn.log
python - <<'PY'
import json
j=json.load(open("gpurun_out/r03c_bench.json"))
print(j["value"], j["ms_per_step"], j["roofline"]["frac"], json.dumps(j["roofline"]["other_kernels"]), j.get("latency_b1_576x768",{}).get("value"))
PY
grep "^attn" $O/r03c_bench_per_shape.tsv
