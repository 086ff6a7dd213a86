#!/bin/bash
O=gpurun_out; F=$O/r03t_bf16_grad_cos.txt; : > $F
for opts in "" "thin_input_conv=0"; do
  echo "== [$opts]" >> $F
  E2EFT_TEST_OPTIONS=$opts timeout 600 python -m pytest tests/test_fullsize_parity_gpu.py -q -x -s -k "bf16_compute_micro_step" 2>&1 | grep "^576^2 bf16\|^\.576^2\|passed\|failed" >> $F
done
cat $F
