#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out
timeout 600 python -m pytest tests/test_wgrad_gpu.py -q -x --tb=short 2>&1 | grep -v "^$" | tail -30 > $O/r03f_wgrad_tests.log
timeout 900 python -m pytest tests/test_train_gpu.py tests/test_bwd_gpu.py tests/test_reference_callers_gpu.py tests/test_two_ranks_one_gpu.py -q 2>&1 | tail -12 > $O/r03f_train_tests.log
timeout 600 python bench.py --train --steps 3 --warmup 1 --detail $O/r03f_bench_train_per_shape.tsv > $O/r03f_bench_train_bf16.json 2> $O/r03f_bench_train.err
tail -20 $O/r03f_wgrad_tests.log; tail -6 $O/r03f_train_tests.log
python - <<'PY'
import json
j=json.load(open("gpurun_out/r03f_bench_train_bf16.json"))
print(j["value"], j["images_per_s"], j["roofline"]["frac"], j["peak_mem_gib"], json.dumps({k: round(v["ms_per_step"],2) for k,v in j["roofline"]["other_kernels"].items()}))
PY
