#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out
timeout 900 python -m pytest tests/test_fused_norm_conv_gpu.py tests/test_patch_conv_gpu.py -q -m gpu 2>&1 | tail -2
for shape in "8 768 768 128 128" "8 768 768 256 128"; do python scripts/norm_conv_bench.py $shape 20 2>&1 | tail -4; done | tee $O/r04_norm_conv_ab_b.txt
