#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out
C="SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE"
rm -rf /tmp/pmc_k; rocprofv3 --kernel-trace --pmc $C --output-format csv -d /tmp/pmc_k -o p -- python scripts/norm_conv_bench.py 8 768 768 128 128 5 > /tmp/pmc_k.log 2>&1
python scripts/pmc_kernel_counters.py /tmp/pmc_k "igemm6_kernelIDF16_Lb0ELb1" "igemm6 NORM (GroupNorm+SiLU applied to the fetched patch) 128->128 @768^2 B8 fp16" | tee $O/r04_pmc_kernels_norm.txt
for shape in "8 768 768 128 128 3 9000" "8 192 192 512 512 3 10000"; do
  echo "== igemm6 conv $shape"; bash scripts/smi_power_during_conv.sh "$shape" 2>&1 | grep -v amdgpu.ids
done | cut -c1-400 > $O/r04_smi_power_clock_during_igemm6.txt
cat $O/r04_smi_power_clock_during_igemm6.txt | cut -c1-330
