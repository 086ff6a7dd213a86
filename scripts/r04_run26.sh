#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_train_gpu.py tests/test_bwd_gpu.py tests/test_two_ranks_one_gpu.py tests/test_reference_callers_gpu.py -q -m gpu 2>&1 | tail -6 > gpurun_out/r04i_train_tests.log
cat gpurun_out/r04i_train_tests.log | head -5
python bench.py --train --dtype bf16 --grad-ckpt --steps 2 --warmup 1 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('ckpt', j['value'], j['final_loss'], j['peak_mem_gib'])"
