mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_strip_conv_gpu.py -x -q -m gpu 2>&1 | tail -3 > gpurun_out/strip_test.txt
for rep in 1 2; do
  for shp in "8 768 768 128 128" "8 384 384 256 256" "8 192 192 512 512" "8 96 96 512 512"; do
    echo -n "igemm2        " ; E2EFT_PERSIST=0 E2EFT_STRIP=0 timeout 120 python scripts/conv_bench.py $shp 3 30 fp16 2>&1 | tail -1
    echo -n "igemm4 hoist  " ; E2EFT_PERSIST=0 E2EFT_STRIP=1 timeout 120 python scripts/conv_bench.py $shp 3 30 fp16 2>&1 | tail -1
    echo -n "igemm5        " ; E2EFT_PERSIST=1 E2EFT_STRIP=0 timeout 120 python scripts/conv_bench.py $shp 3 30 fp16 2>&1 | tail -1
  done
done > gpurun_out/strip_ab.txt 2>&1
cat gpurun_out/strip_test.txt gpurun_out/strip_ab.txt
