mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_persistent_gpu.py tests/test_train_gpu.py -x -q -m gpu 2>&1 | tail -5 > gpurun_out/pers_test.txt
L=diffusion-e2e-ft_amd/lib/libe2eft_stamps.so
{
E2EFT_LIB=$L timeout 120 python scripts/stamp5_bench.py 8 768 768 128 128 3 0 0
E2EFT_LIB=$L timeout 120 python scripts/stamp5_bench.py 8 768 768 128 128 3 1 1
E2EFT_LIB=$L timeout 120 python scripts/stamp5_bench.py 8 192 192 512 512 3 0 0
} > gpurun_out/pers_stamps.txt 2>&1
for rep in 1 2; do for pe in 1 0; do
  for shp in "8 768 768 128 128" "8 384 384 256 256" "8 192 192 512 512" "8 96 96 512 512"; do
    echo -n "PERSIST=$pe " ; E2EFT_PERSIST=$pe timeout 120 python scripts/conv_bench.py $shp 3 30 fp16 2>&1 | tail -1
  done
  echo -n "PERSIST=$pe res " ; E2EFT_PERSIST=$pe timeout 120 python scripts/conv_bench.py 8 768 768 128 128 3 30 fp16 1 2>&1 | tail -1
done; done > gpurun_out/pers_ab.txt 2>&1
tail -5 gpurun_out/pers_test.txt; grep -v amdgpu.ids gpurun_out/pers_stamps.txt; cat gpurun_out/pers_ab.txt
