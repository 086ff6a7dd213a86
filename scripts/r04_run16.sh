#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out
timeout 1500 python -m pytest tests/test_train_gpu.py tests/test_reference_callers_gpu.py tests/test_two_ranks_one_gpu.py tests/test_bwd_gpu.py tests/test_wgrad_gpu.py tests/test_config2_batch32_gpu.py -q -m gpu -x 2>&1 | tail -8
python bench.py --train --steps 3 --warmup 1 --detail $O/r04c_bench_train_per_shape.tsv > $O/r04c_bench_train_bf16.json 2>/dev/null
python -c "
import json; j=json.load(open('gpurun_out/r04c_bench_train_bf16.json')); print('train bf16', j['value'], j['images_per_s'], j['peak_mem_gib'], j['roofline']['frac'])"
