#!/bin/bash
# round 6, call 19: the split route inside a hipGraph capture (no host read anywhere on it)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests/test_f32split_gpu.py -x -q -m gpu -s 2>&1 | tail -12 > $O/r06s_f32split_test.log
cat $O/r06s_f32split_test.log | cut -c1-300
