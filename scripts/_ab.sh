mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_ops_gpu.py tests/test_bwd_gpu.py tests/test_model_gpu.py tests/test_fullsize_parity_gpu.py tests/test_train_gpu.py tests/test_persistent_gpu.py -x -q -m gpu 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm version\|^Hostname\|^Librccl" | tail -6 > gpurun_out/gnsmall_test.txt
for pe in 1 0 1 0; do
  E2EFT_GN_SMALL=$pe timeout 600 python bench.py --steps 15 --warmup 4 --no-train-leg --no-cpu-baseline 2>/dev/null | python -c "import sys, json; j = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('GN_SMALL=$pe', round(j['value'], 2), 'img/s', round(j['ms_per_step'], 2), 'ms; unet', round(j['stages']['ms_per_step']['unet'], 2), 'gn', round(j['roofline']['other_kernels']['groupnorm']['ms_per_step'], 2))"
done > gpurun_out/gnsmall_ab.txt 2>&1
cat gpurun_out/gnsmall_test.txt gpurun_out/gnsmall_ab.txt
