#!/bin/bash
O=gpurun_out; mkdir -p $O
timeout 600 python -m pytest tests/test_fused_norm_conv_gpu.py -x -q -s > $O/r03q_fused_norm_tests.log 2>&1; tail -3 $O/r03q_fused_norm_tests.log
grep -q "1 passed" $O/r03q_fused_norm_tests.log || { grep -v "^$" $O/r03q_fused_norm_tests.log | tail -30; exit 1; }
timeout 300 python scripts/norm_conv_bench.py 8 768 768 128 3
