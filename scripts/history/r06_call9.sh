#!/bin/bash
# round 6, call 9: fp32 convolutions from f16 splits (csrc/f32split.hip): the new test, the fp32 parity tests that now route through it, A/B timing of one layer,
# the strict-fp32 training step with the route on and off
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests/test_f32split_gpu.py -x -q -m gpu -s 2>&1 | tail -40 > $O/r06i_f32split_test.log
timeout 1500 python -m pytest tests/test_fullsize_parity_gpu.py tests/test_config3_c4_gpu.py tests/test_config2_batch32_gpu.py tests/test_bwd_gpu.py tests/test_model_gpu.py -q -m gpu -s -k "fp32 or float32 or config2 or config3" 2>&1 | tail -60 > $O/r06i_fp32_parity_tests.log
{
for o in 1 0; do
timeout 200 python scripts/conv_bench.py 16 576 576 128 128 3 10 fp32 0 f32_split=$o
timeout 200 python scripts/conv_bench.py 16 576 576 128 128 3 10 fp32 1 f32_split=$o
timeout 200 python scripts/conv_bench.py 16 288 288 256 256 3 10 fp32 0 f32_split=$o
timeout 200 python scripts/conv_bench.py 16 144 144 512 512 3 10 fp32 0 f32_split=$o
done
} > $O/r06i_conv_ab.txt 2>&1
timeout 600 python bench.py --train --dtype fp32 --steps 4 --warmup 1 --detail $O/r06i_bench_train_fp32_per_shape.tsv > $O/r06i_bench_train_fp32.json 2> $O/r06i_bench_train_fp32.err
timeout 600 python bench.py --train --dtype fp32 --steps 4 --warmup 1 --set-option f32_split=0 > $O/r06i_bench_train_fp32_off.json 2>/dev/null
cat $O/r06i_f32split_test.log | cut -c1-400; tail -25 $O/r06i_fp32_parity_tests.log | cut -c1-300; cat $O/r06i_conv_ab.txt
python - <<PY
import json
for n in ("", "_off"):
    try:
        j=json.load(open("gpurun_out/r06i_bench_train_fp32%s.json"%n)); print("train fp32", n, j["value"], j.get("median_ms_per_step"), j.get("final_loss"), j.get("peak_mem_gib"))
    except Exception as e: print("failed", n, e)
PY
tail -5 $O/r06i_bench_train_fp32.err
