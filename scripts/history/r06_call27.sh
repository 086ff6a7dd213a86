#!/bin/bash
# round 6, call 27: the split weight gradient with one partial buffer and one reduction for its three launches (host-side cost): the test file, the fp32 parity tests, the strict-fp32 step
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests/test_f32split_gpu.py -x -q -m gpu -s 2>&1 | tail -50 > $O/r06A_f32split_test.log
timeout 1800 python -m pytest tests/test_fullsize_parity_gpu.py tests/test_config3_c4_gpu.py tests/test_config2_batch32_gpu.py tests/test_bwd_gpu.py tests/test_model_gpu.py tests/test_train_gpu.py tests/test_fullsize_gpu.py tests/test_ops_gpu.py tests/test_reference_callers_gpu.py -q -m gpu -k "fp32 or float32 or config2 or config3 or train or micro or gemm or linear or replay" 2>&1 | tail -15 > $O/r06A_fp32_parity_tests.log
timeout 600 python bench.py --train --dtype fp32 --steps 4 --warmup 1 --detail $O/r06A_bench_train_fp32_per_shape.tsv > $O/r06A_bench_train_fp32.json 2> $O/r06A_bench_train_fp32.err
cat $O/r06A_f32split_test.log | cut -c1-330; tail -8 $O/r06A_fp32_parity_tests.log | cut -c1-300
python - <<PY
import json
j=json.load(open("gpurun_out/r06A_bench_train_fp32.json")); print("train fp32", j["value"], j.get("median_ms_per_step"), j.get("final_loss"), j.get("peak_mem_gib"), j["roofline"]["frac"])
PY
tail -5 $O/r06A_bench_train_fp32.err
