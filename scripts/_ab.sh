mkdir -p gpurun_out
E2EFT_PERSIST_GRID=8 timeout 1500 python -m pytest tests -q -m gpu 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm version\|^Hostname\|^Librccl" | tail -4 > gpurun_out/r02c_gpu_tests_persist_grid8.log
cat gpurun_out/r02c_gpu_tests_persist_grid8.log
