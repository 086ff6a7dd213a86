#!/bin/bash
O=gpurun_out; mkdir -p $O
timeout 600 python -m pytest tests/test_wgrad_gpu.py -x -q > $O/r03l_wgrad_tests.log 2>&1; tail -3 $O/r03l_wgrad_tests.log
grep -q " passed" $O/r03l_wgrad_tests.log || exit 1
timeout 900 python -m pytest tests/test_train_gpu.py tests/test_bwd_gpu.py -x -q > $O/r03l_train_tests.log 2>&1; tail -3 $O/r03l_train_tests.log
timeout 600 python bench.py --train --steps 3 --warmup 1 --detail $O/r03l_bench_train_per_shape.tsv > $O/r03l_bench_train_bf16.json 2> $O/r03l_bench_train.err
python - <<'PY'
import json
j=json.loads(open("gpurun_out/r03l_bench_train_bf16.json").read().strip().splitlines()[-1])
print(j["value"], j["roofline"]["frac"], {k:round(v["ms_per_step"],1) for k,v in j["roofline"]["other_kernels"].items()})
PY
grep wgrad $O/r03l_bench_train_per_shape.tsv | sort -t$'\t' -k5 -n -r | head -12
