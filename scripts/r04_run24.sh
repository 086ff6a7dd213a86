#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out
timeout 600 python -m pytest tests/test_train_gpu.py -q -m gpu -x 2>&1 | tail -3
for i in 1 2; do
python bench.py --train --steps 3 --warmup 1 --no-direct-grads 2>/dev/null | python -c "import json,sys; print('memset + accumulate', json.loads(sys.stdin.read())['value'])"
python bench.py --train --steps 3 --warmup 1 2>/dev/null | python -c "import json,sys; print('direct grads       ', json.loads(sys.stdin.read())['value'])"
done
