#!/bin/bash
# round 6, call 22 (no library change): SQ counters of the fp32-split kernels; the whole GPU suite with E2EFT_TEST_PERSISTENT_GRID=8 (every eligible SMALL problem through the
# persistent kernels — for the fp32 tests that means through the F32O variants; not what the driver runs)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
{
bash scripts/pmc_passes.sh "fp32 conv 128->128 @576^2 B16 from f16 split planes (igemm6 F32O)" "igemm6_kernel<_Float16, true, false, 3, true>" python scripts/conv_bench.py 16 576 576 128 128 3 5 fp32
bash scripts/pmc_passes.sh "fp32 conv 512->512 @144^2 B16 from f16 split planes (igemm5 F32O)" "igemm5_kernel<_Float16, 1, true, true>" python scripts/conv_bench.py 16 144 144 512 512 3 5 fp32
bash scripts/pmc_passes.sh "fp32 conv 128->128 @576^2 B16 on the fp32 instruction (igemm2)" "igemm2_kernel<float, 1, true, 8>" python scripts/conv_bench.py 16 576 576 128 128 3 5 fp32 0 f32_split=0
} 2>&1 | grep -v amdgpu.ids > $O/r06_pmc_kernels_f32split.txt
E2EFT_TEST_PERSISTENT_GRID=8 timeout 3000 python -m pytest tests -q -m gpu 2>&1 | tail -25 > $O/r06v_gpu_tests_persist_grid8.log
tail -40 $O/r06_pmc_kernels_f32split.txt | cut -c1-250; tail -14 $O/r06v_gpu_tests_persist_grid8.log | cut -c1-250
