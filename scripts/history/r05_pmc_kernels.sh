#!/bin/bash
# round 5, measurement only (no code change after the final suite): SQ counters of the kernels that are new this round — the fp32 attention (forward, dK/dV, dQ), the
# LDS-DMA d = 64 attention, the four phase launches of an upsampler convolution on igemm5 — one rocprofv3 --pmc pass per micro-benchmark (--kernel-trace only)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out
C="SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE"
run() {   # label, kernel substring, command...
  local label="$1" pat="$2"; shift 2
  rm -rf /tmp/pmc_k; rocprofv3 --kernel-trace --pmc $C --output-format csv -d /tmp/pmc_k -o p -- "$@" > /tmp/pmc_k.log 2>&1
  python scripts/pmc_kernel_counters.py /tmp/pmc_k "$pat" "$label" || tail -3 /tmp/pmc_k.log
}
{
rm -rf /tmp/pmc_k; rocprofv3 --kernel-trace --pmc $C --output-format csv -d /tmp/pmc_k -o p -- python scripts/attn_bench.py 16 5 5184 3 fp32 bwd > /tmp/pmc_k.log 2>&1
python scripts/pmc_kernel_counters.py /tmp/pmc_k attn32_fwd_kernel "attn32 forward B16 h5 N5184 fp32"
python scripts/pmc_kernel_counters.py /tmp/pmc_k attn32_bwd_dkdv_kernel "attn32 backward dK/dV B16 h5 N5184 fp32"
python scripts/pmc_kernel_counters.py /tmp/pmc_k attn32_bwd_dq_kernel "attn32 backward dQ B16 h5 N5184 fp32"
run "attn_fwd_dma d=64 B8 h5 N9216 fp16 (default since round 5)" attn_fwd_dma_kernel python scripts/attn_bench.py 8 5 9216 5
run "upconv2x phases 256->256 384^2 -> 768^2 B8 fp16 on igemm5 (K = 1024)" "igemm5_kernel<_Float16, 1, false>" python scripts/upconv_bench.py 8 384 384 256 256 3 fp16
run "upconv2x phases 512->512 192^2 -> 384^2 B8 fp16 on igemm5 (K = 2048)" "igemm5_kernel<_Float16, 1, false>" python scripts/upconv_bench.py 8 192 192 512 512 3 fp16
} 2>&1 | grep -v amdgpu.ids | tee $O/r05_pmc_kernels.txt
for a in "16 5 5184 5 fp32" "16 5 5184 3 fp32 bwd" "8 5 9216 20 fp16"; do python scripts/attn_bench.py $a; done 2>&1 | grep -v amdgpu.ids | tee $O/r05_attn_final_bench.txt
