mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_persistent_gpu.py -x -q -m gpu 2>&1 | tail -6 > gpurun_out/swap_test.txt
for rep in 1 2; do for sw in 0 1; do
  for shp in "73728 320 320" "73728 2560 320" "18432 5120 640" "18432 640 640" "4608 1280 1280" "73728 512 512"; do
    echo -n "NOSWAP=$sw " ; E2EFT_PERSIST_NOSWAP=$sw timeout 120 python scripts/gemm_bench.py $shp 2>&1 | tail -1
  done
done; done > gpurun_out/swap_ab.txt 2>&1
cat gpurun_out/swap_test.txt gpurun_out/swap_ab.txt
