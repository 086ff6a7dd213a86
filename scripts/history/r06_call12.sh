#!/bin/bash
# round 6, call 12: f16-split fp32, third batch: upsampler phases, weight gradients (three 16-bit launches on the split planes), one atomic per workgroup in the maximum pass;
# the test file, the fp32 parity tests, the strict-fp32 training step (plain and with recompute = the reference recipe), the whole GPU suite
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests/test_f32split_gpu.py -x -q -m gpu -s 2>&1 | tail -45 > $O/r06l_f32split_test.log
timeout 600 python bench.py --train --dtype fp32 --steps 4 --warmup 1 --detail $O/r06l_bench_train_fp32_per_shape.tsv > $O/r06l_bench_train_fp32.json 2> $O/r06l_bench_train_fp32.err
timeout 600 python bench.py --train --dtype fp32 --steps 4 --warmup 1 --grad-ckpt > $O/r06l_bench_train_fp32_ckpt.json 2>/dev/null
timeout 600 python bench.py --train --dtype fp32 --steps 4 --warmup 1 --set-option f32_split=0 > $O/r06l_bench_train_fp32_off.json 2>/dev/null
timeout 2700 python -m pytest tests -q -m gpu 2>&1 | tail -15 > $O/r06l_gpu_tests.log
cat $O/r06l_f32split_test.log | cut -c1-330; tail -8 $O/r06l_gpu_tests.log | cut -c1-300
python - <<PY
import json
for n in ("", "_ckpt", "_off"):
    try:
        j=json.load(open("gpurun_out/r06l_bench_train_fp32%s.json"%n)); print("train fp32", n, j["value"], j.get("median_ms_per_step"), j.get("final_loss"), j.get("peak_mem_gib"), j["roofline"].get("frac"), j["roofline"].get("pipes"))
    except Exception as e: print("failed", n, e)
PY
tail -5 $O/r06l_bench_train_fp32.err
