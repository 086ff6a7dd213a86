#!/bin/bash
# halo-patch conv (igemm6): parity tests under a timeout, then A/B against igemm5 on the three dominant shapes
O=gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests/test_patch_conv_gpu.py -x -q -s > $O/r03h_patch_tests.log 2>&1; echo "tests rc=$?" >> $O/r03h_patch_tests.log
tail -5 $O/r03h_patch_tests.log
if grep -q "2 passed" $O/r03h_patch_tests.log; then
  for shp in "8 768 768 128 128" "8 384 384 256 256" "8 192 192 512 512" "8 96 96 512 512" "8 768 768 256 128" "8 96 96 320 320"; do
    for pc in 1 0; do
      echo "== $shp patch_conv=$pc" >> $O/r03h_patch_ab.txt
      timeout 120 python scripts/conv_bench.py $shp 3 30 fp16 0 patch_conv=$pc 2>&1 | tail -1 >> $O/r03h_patch_ab.txt
    done
  done
  echo "== with residual" >> $O/r03h_patch_ab.txt
  for pc in 1 0; do timeout 120 python scripts/conv_bench.py 8 768 768 128 128 3 30 fp16 1 patch_conv=$pc 2>&1 | tail -1 >> $O/r03h_patch_ab.txt; done
  cat $O/r03h_patch_ab.txt
fi
