#!/bin/bash
# which kernel variant the host should pick for the quantisation-limited launches: every top non-igemm6 shape under each forced option
O=gpurun_out; mkdir -p $O; F=$O/r03k_variant_sweep.txt; : > $F
g() { for opt in "" "igemm2_waves=4" "igemm2_waves=8" "persistent=0" "persistent=0 igemm2_waves=4"; do echo -n "gemm $* [$opt]: " >> $F; timeout 60 python scripts/gemm_bench.py $* $opt 2>&1 | tail -1 >> $F; done; }
c() { for opt in "" "igemm2_waves=4" "igemm2_waves=8" "persistent=0" "patch_conv=0"; do echo -n "conv $* [$opt]: " >> $F; timeout 60 python scripts/conv_bench.py $* $opt 2>&1 | tail -1 >> $F; done; }
g 18432 640 640 50 fp16 1
g 4608 1280 1280 50 fp16 1
g 73728 320 320 50 fp16 1
g 73728 320 320 50 fp16 0
g 73728 2560 320 30 fp16 0
g 18432 5120 640 30 fp16 0
g 73728 320 1280 30 fp16 1
g 18432 640 2560 30 fp16 1
g 4608 1280 5120 30 fp16 1
g 4608 10240 1280 30 fp16 0
c 8 24 24 1280 1280 3 30 fp16 1
c 8 48 48 640 640 3 30 fp16 1
c 8 12 12 1280 1280 3 30 fp16 1
c 8 24 24 2560 1280 3 30 fp16 1
cat $F
