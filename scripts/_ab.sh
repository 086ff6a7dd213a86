mkdir -p gpurun_out
timeout 1500 python -m pytest tests -q -m gpu 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm version\|^Hostname\|^Librccl" | tail -8 > gpurun_out/r02b_gpu_tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE OK')" 2>&1 | tail -3 >> gpurun_out/r02b_gpu_tests.log
timeout 600 python scripts/soak_determinism.py 2>&1 | tail -4 > gpurun_out/r02b_soak.log
timeout 600 python scripts/stress_conv_stats.py 2>&1 | tail -4 >> gpurun_out/r02b_soak.log
for pe in 1 0; do echo -n "PERSIST=$pe res " ; E2EFT_PERSIST=$pe timeout 120 python scripts/conv_bench.py 8 768 768 128 128 3 30 fp16 1 2>&1 | tail -1; done >> gpurun_out/r02b_soak.log
cat gpurun_out/r02b_gpu_tests.log gpurun_out/r02b_soak.log
