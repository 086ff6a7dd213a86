#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out
timeout 1500 python -m pytest tests -q -m gpu --deselect tests/test_benchmarked_configs_gpu.py --deselect tests/test_fullsize_parity_gpu.py 2>&1 | tail -25 > $O/r03e_gpu_tests.log
tail -25 $O/r03e_gpu_tests.log
