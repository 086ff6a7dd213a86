#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out
timeout 900 python -m pytest tests/test_train_gpu.py tests/test_two_ranks_one_gpu.py tests/test_reference_callers_gpu.py tests/test_bwd_gpu.py -q -m gpu -x 2>&1 | tail -15
python bench.py --train --steps 3 --warmup 1 > $O/r04g_bench_train_bf16.json 2>/dev/null
python -c "
import json; j=json.load(open('$O/r04g_bench_train_bf16.json')); print('train', j['value'], j['final_loss'])"
