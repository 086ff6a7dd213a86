#!/bin/bash
O=gpurun_out; mkdir -p $O
timeout 600 python -m pytest tests/test_thin_conv_gpu.py -x -q -s > $O/r03m_thin_tests.log 2>&1; tail -4 $O/r03m_thin_tests.log
grep -q "2 passed" $O/r03m_thin_tests.log || { grep -v "^$" $O/r03m_thin_tests.log | head -40; exit 1; }
for tc in 1 0; do timeout 120 python scripts/conv_bench.py 8 768 768 8 128 3 30 fp16 0 thin_input_conv=$tc 2>&1 | tail -1; done
timeout 900 python -m pytest tests/test_patch_conv_gpu.py tests/test_persistent_gpu.py -x -q > $O/r03m_persistent_tests.log 2>&1; tail -2 $O/r03m_persistent_tests.log
