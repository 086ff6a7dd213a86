#!/bin/bash
O=gpurun_out; F=$O/r03t_bf16_grad_bisect.txt; : > $F
for opts in "fused_norm=0" "thin_input_conv=0" "igemm2_waves=4" "patch_conv=0"; do
  echo "== $opts" >> $F
  E2EFT_TEST_OPTIONS=$opts timeout 600 python -m pytest tests/test_fullsize_parity_gpu.py -q -x -s -k "bf16_compute_micro_step" 2>&1 | grep "576^2 bf16\|passed\|failed" | cut -c1-400 >> $F
done
cat $F
