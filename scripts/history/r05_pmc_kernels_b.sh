#!/bin/bash
# the two igemm5 rows of scripts/r05_pmc_kernels.sh (the pattern there did not match the mangled symbol), plus the fused-upsample igemm6 launch of the same layer for comparison
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out
C="SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE"
{
for sh in "8 384 384 256 256" "8 192 192 512 512"; do
  rm -rf /tmp/pmc_k; rocprofv3 --kernel-trace --pmc $C --output-format csv -d /tmp/pmc_k -o p -- python scripts/upconv_bench.py $sh 3 fp16 > /tmp/pmc_k.log 2>&1
  python scripts/pmc_kernel_counters.py /tmp/pmc_k igemm5_kernel "upconv2x: the four phase launches on igemm5, B H W Cin Cout = $sh (low-resolution input), fp16"
  python scripts/pmc_kernel_counters.py /tmp/pmc_k igemm6_kernel "the same layer as ONE fused-upsample 3x3 launch on igemm6, $sh"
done
} 2>&1 | grep -v amdgpu.ids | tee $O/r05_pmc_kernels_upconv.txt
