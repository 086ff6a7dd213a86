#!/bin/bash
O=gpurun_out; mkdir -p $O
timeout 600 python -m pytest tests/test_patch_conv_gpu.py -x -q > $O/r03n_patch_tests.log 2>&1; tail -2 $O/r03n_patch_tests.log
grep -q "2 passed" $O/r03n_patch_tests.log || exit 1
for shp in "8 768 768 128 128" "8 384 384 256 256" "8 192 192 512 512"; do timeout 120 python scripts/conv_bench.py $shp 3 30 fp16 0 2>&1 | tail -1; done
timeout 120 python scripts/conv_bench.py 8 768 768 128 128 3 30 fp16 1 2>&1 | tail -1
