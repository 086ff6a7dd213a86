#!/bin/bash
O=gpurun_out; mkdir -p $O
timeout 600 python -m pytest tests/test_fused_norm_conv_gpu.py -x -q -s > $O/r03o_fused_norm_tests.log 2>&1; tail -3 $O/r03o_fused_norm_tests.log
grep -q "1 passed" $O/r03o_fused_norm_tests.log || { grep -v "^$" $O/r03o_fused_norm_tests.log | tail -40; exit 1; }
timeout 300 python scripts/norm_conv_bench.py 8 768 768 128 128 > $O/r03o_norm_conv_ab.txt 2>&1; cat $O/r03o_norm_conv_ab.txt
timeout 300 python scripts/norm_conv_bench.py 8 768 768 256 128 >> $O/r03o_norm_conv_ab.txt 2>&1; tail -4 $O/r03o_norm_conv_ab.txt
