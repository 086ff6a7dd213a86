#!/bin/bash
# round 6, call 5: strict-fp32 weight gradients straight from the NHWC tensors (wgrad32_kernel) and the fp32 form of the narrow conv_out kernel: tests, micro-benchmarks,
# the fp32 training step with and without them
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests/test_wgrad_gpu.py tests/test_ops_gpu.py tests/test_bwd_gpu.py -x -q -m gpu 2>&1 | tail -8 > $O/r06e_fp32_wgrad_tests.log
timeout 900 python -m pytest tests/test_train_gpu.py "tests/test_fullsize_parity_gpu.py::test_config2_576_fp32_micro_step_gradients" tests/test_config3_c4_gpu.py -x -q -m gpu 2>&1 | tail -8 >> $O/r06e_fp32_wgrad_tests.log
{
for shp in "16 72 72 320 320 3" "16 36 36 640 640 3" "16 18 18 1280 1280 3" "16 72 72 320 2560 1" "16 72 72 1280 320 1" "16 72 72 640 320 3"; do
  timeout 120 python scripts/wgrad_bench.py $shp 5 fp32
done
} > $O/r06e_wgrad32_bench.txt 2>&1
timeout 900 python bench.py --train --dtype fp32 --steps 4 --warmup 1 --detail $O/r06e_bench_train_fp32_per_shape.tsv > $O/r06e_bench_train_fp32.json 2> $O/r06e_bench_train_fp32.err
cat $O/r06e_fp32_wgrad_tests.log; cat $O/r06e_wgrad32_bench.txt
python -c "
import json; j=json.load(open('gpurun_out/r06e_bench_train_fp32.json')); print('fp32 step', j['value'], j.get('median_ms_per_step'), j.get('peak_mem_gib'), (j.get('roofline') or {}).get('frac'))"
