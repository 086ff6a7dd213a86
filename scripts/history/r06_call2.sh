#!/bin/bash
# round 6, call 2: the 2x2-tap variant of igemm6 — its tests, the upconv tests both ways, A/B against igemm5 on the three decoder upsampler shapes
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests/test_patch_conv_2x2_gpu.py tests/test_upconv_phases_gpu.py tests/test_patch_conv_gpu.py tests/test_fused_norm_conv_gpu.py -x -q -m gpu 2>&1 | tail -15 > $O/r06b_patch2x2_tests.log
{
for shp in "8 384 384 256 256" "8 192 192 512 512" "8 96 96 512 512"; do
  for rep in 1 2; do
    timeout 120 python scripts/upconv_bench.py $shp 10 fp16 patch_conv_2x2=0 2>&1 | grep "four 2x2" | sed 's/$/   [igemm5]/'
    timeout 120 python scripts/upconv_bench.py $shp 10 fp16 patch_conv_2x2=1 2>&1 | grep "four 2x2" | sed 's/$/   [igemm6 2x2]/'
  done
done
} > $O/r06b_upconv_ab.txt 2>&1
cat $O/r06b_patch2x2_tests.log; cat $O/r06b_upconv_ab.txt
