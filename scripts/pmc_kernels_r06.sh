#!/bin/bash
# SQ counters of the kernels that carry the step and of the round's new ones, with the two-pass collector (scripts/pmc_passes.sh: the LDS counter pair in its own pass,
# demangling name filter): igemm6 plain / NORM on conv 128->128 @768^2, igemm6 on conv 512->512 @192^2, the 2x2-tap variant on an upsampler, attention d = 64 (DMA),
# the fp32 weight-gradient kernel.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
{
bash scripts/pmc_passes.sh "conv 128->128 @768^2 B8 fp16" "igemm6_kernel<_Float16, false, false, 3>" python scripts/conv_bench.py 8 768 768 128 128 3 5 fp16
bash scripts/pmc_passes.sh "conv 512->512 @192^2 B8 fp16" "igemm6_kernel<_Float16, false, false, 3>" python scripts/conv_bench.py 8 192 192 512 512 3 5 fp16
bash scripts/pmc_passes.sh "GroupNorm+SiLU -> conv 128->128 @768^2 B8 fp16 (fused)" "igemm6_kernel<_Float16, false, true, 3>" python scripts/norm_conv_bench.py 8 768 768 128 128 5
bash scripts/pmc_passes.sh "upconv2x phases 256->256 384^2 -> 768^2 B8 fp16 on igemm6 2x2" "igemm6_kernel<_Float16, false, false, 2>" python scripts/upconv_bench.py 8 384 384 256 256 3 fp16
bash scripts/pmc_passes.sh "attention d=64 B8 h5 N9216 fp16" "attn_fwd_dma_kernel" python scripts/attn_bench.py 8 5 9216 5
bash scripts/pmc_passes.sh "fp32 weight gradient 3x3 320->320 B16 @72^2" "wgrad32_kernel" python scripts/wgrad_bench.py 16 72 72 320 320 3 3 fp32
} 2>&1 | grep -v amdgpu.ids | tee $O/r06_pmc_kernels.txt
