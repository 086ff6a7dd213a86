#!/bin/bash
# round 6, call 10: f16-split fp32 convolutions, second batch: igemm5 route (any filter / stride, widths that are not multiples of 32), GroupNorm -> planes fused
# (no_grad and _NormConvSplitFn), weights split on the device; test, fp32 parity tests, A/B timing, the strict-fp32 training step
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests/test_f32split_gpu.py -x -q -m gpu -s 2>&1 | tail -45 > $O/r06j_f32split_test.log
timeout 1800 python -m pytest tests/test_fullsize_parity_gpu.py tests/test_config3_c4_gpu.py tests/test_config2_batch32_gpu.py tests/test_bwd_gpu.py tests/test_model_gpu.py tests/test_train_gpu.py tests/test_fullsize_gpu.py -q -m gpu -s -k "fp32 or float32 or config2 or config3 or train or micro" 2>&1 | tail -60 > $O/r06j_fp32_parity_tests.log
{
for o in 1 0; do
timeout 200 python scripts/conv_bench.py 16 576 576 128 128 3 10 fp32 0 f32_split=$o
timeout 200 python scripts/conv_bench.py 16 144 144 512 512 3 10 fp32 0 f32_split=$o
timeout 200 python scripts/conv_bench.py 16 72 72 512 512 3 10 fp32 0 f32_split=$o
timeout 200 python scripts/conv_bench.py 16 72 72 320 320 3 10 fp32 0 f32_split=$o
done
} > $O/r06j_conv_ab.txt 2>&1
timeout 600 python bench.py --train --dtype fp32 --steps 4 --warmup 1 --detail $O/r06j_bench_train_fp32_per_shape.tsv > $O/r06j_bench_train_fp32.json 2> $O/r06j_bench_train_fp32.err
cat $O/r06j_f32split_test.log | cut -c1-330; tail -25 $O/r06j_fp32_parity_tests.log | cut -c1-300; grep conv $O/r06j_conv_ab.txt
python - <<PY
import json
for n in ("",):
    try:
        j=json.load(open("gpurun_out/r06j_bench_train_fp32%s.json"%n)); print("train fp32", n, j["value"], j.get("median_ms_per_step"), j.get("final_loss"), j.get("peak_mem_gib"))
    except Exception as e: print("failed", n, e)
PY
tail -5 $O/r06j_bench_train_fp32.err
