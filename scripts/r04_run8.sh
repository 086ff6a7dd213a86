#!/bin/bash
# round 4, GPU call 8: attention ceiling probes; igemm6 NORM with the arithmetic in the MFMA shadow (tests + A/B)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out
for n in 0 1 2 3 4; do scripts/bin/attn_probe_$n 8 5 9216 20; done > $O/r04_attn_probes.txt 2>&1
cat $O/r04_attn_probes.txt
for sh in "8 5 9216" "8 10 2304" "8 20 576" "2 5 9216"; do for v in v1 v2; do echo "== $v $sh"; scripts/bin/attn_$v $sh 20; done; done > $O/r04_attn_v1_v2.txt 2>&1
cat $O/r04_attn_v1_v2.txt
timeout 900 python -m pytest tests/test_fused_norm_conv_gpu.py tests/test_patch_conv_gpu.py -q -m gpu 2>&1 | tail -3
for shape in "8 768 768 128 128" "8 768 768 256 128"; do python scripts/norm_conv_bench.py $shape 20 2>&1 | tail -4; done | tee $O/r04_norm_conv_ab.txt
BF16_INSTANCE=gpuseeded BF16_NO_REF2=1 timeout 600 python scripts/bf16_localise.py cpu 7 576 $O/r04c_gpuseeded_bf16_localise > $O/r04c_cpu.log 2>&1
grep "draw\|run" $O/r04c_cpu.log | tail -16
