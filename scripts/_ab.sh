mkdir -p gpurun_out
timeout 1500 python -m pytest tests -q -m gpu 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm version\|^Hostname\|^Librccl" | tail -4 > gpurun_out/r02c_gpu_tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE OK')" 2>&1 | tail -3 >> gpurun_out/r02c_gpu_tests.log
cat gpurun_out/r02c_gpu_tests.log
