#!/bin/bash
# round 6, call 3: two sweeps — (1) the persistent kernels' minimum tile count (two rounds of the machine vs half a round) on the UNet's mid-size layers;
# (2) pixels per thread of the GroupNorm apply pass on the mid-size / small tensors
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out; mkdir -p $O
{
for q in 8 4 2; do
  for shp in "8 24 24 1280 1280" "8 48 48 640 640" "8 24 24 2560 1280" "8 48 48 1920 640" "8 48 48 1280 640" "8 96 96 320 320"; do
    timeout 100 python scripts/conv_bench.py $shp 3 20 fp16 0 persistent_min_qrounds=$q 2>/dev/null | sed "s/\$/   [min_qrounds $q]/"
  done
  for shp in "18432 640 640 30 fp16 1" "4608 1280 1280 30 fp16 1" "18432 640 2560 30 fp16 1" "4608 1280 5120 30 fp16 1" "73728 320 320 30 fp16 1" "18432 1920 640 30 fp16 0" "4608 3840 1280 30 fp16 0"; do
    timeout 100 python scripts/gemm_bench.py $shp persistent_min_qrounds=$q 2>/dev/null | sed "s/\$/   [min_qrounds $q]/"
  done
done
} > $O/r06c_min_rounds_sweep.txt 2>&1
{
for it in 1 2 4 8; do
  for shp in "8 96 96 512" "8 96 96 320" "8 192 192 512" "8 384 384 256" "8 48 48 640" "8 24 24 1280" "8 768 768 128"; do
    timeout 100 python scripts/gn_bench.py $shp 1 gn_apply_iters=$it 2>&1 | grep "^gn"
  done
done
} > $O/r06c_gn_apply_sweep.txt 2>&1
cat $O/r06c_min_rounds_sweep.txt; cat $O/r06c_gn_apply_sweep.txt
