#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out
timeout 2400 python -m pytest tests -q -m gpu 2>&1 | tail -15 > $O/r04b_gpu_tests.log
tail -8 $O/r04b_gpu_tests.log
