#!/bin/bash
# MFMA / VALU utilisation counters of the kernels that carry the step: one rocprofv3 --pmc pass per micro-benchmark (--kernel-trace only), aggregated by
# scripts/pmc_kernel_counters.py -> gpurun_out/r04_pmc_kernels.txt; then the SMI power / clock trace under the dominant kernel
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out
C="SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE"
run() {   # label, kernel substring, command...
  local label="$1" pat="$2"; shift 2
  rm -rf /tmp/pmc_k; rocprofv3 --kernel-trace --pmc $C --output-format csv -d /tmp/pmc_k -o p -- "$@" > /tmp/pmc_k.log 2>&1
  python scripts/pmc_kernel_counters.py /tmp/pmc_k "$pat" "$label" || tail -3 /tmp/pmc_k.log
}
{
run "igemm6 conv3x3 128->128 @768^2 B8 fp16 (plain)" igemm6_kernel python scripts/conv_bench.py 8 768 768 128 128 3 5
run "igemm6 conv3x3 512->512 @192^2 B8 fp16" igemm6_kernel python scripts/conv_bench.py 8 192 192 512 512 3 5
run "igemm6 NORM (GroupNorm+SiLU inside) 128->128 @768^2 B8 fp16" "igemm6_kernel<_Float16, false, true>" python scripts/norm_conv_bench.py 8 768 768 128 128 5
run "attn_fwd d=64 B8 h5 N9216 fp16" attn_fwd_kernel python scripts/attn_bench.py 8 5 9216 5
run "attn512_fwd B8 N9216 fp16" attn512_fwd_kernel python scripts/attn512_bench.py 8 9216 5
run "wgrad 3x3 320->320 B32 72^2 bf16" wgrad_kernel python scripts/wgrad_bench.py 32 72 72 320 320 3 5
run "wgrad 3x3 128->128 B8 576^2 bf16" wgrad_kernel python scripts/wgrad_bench.py 8 576 576 128 128 3 5
} 2>&1 | grep -v amdgpu.ids | tee $O/r04_pmc_kernels.txt
for shape in "8 768 768 128 128 3 300" "8 192 192 512 512 3 300"; do
  echo "== igemm6 conv $shape"; bash scripts/smi_power_during_conv.sh "$shape" 2>&1 | grep -v amdgpu.ids
done | cut -c1-400 > $O/r04_smi_power_clock_during_igemm6.txt
tail -12 $O/r04_smi_power_clock_during_igemm6.txt | cut -c1-250
