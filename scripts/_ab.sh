mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_persistent_gpu.py tests/test_ops_gpu.py tests/test_bwd_gpu.py tests/test_fullsize_parity_gpu.py -x -q -m gpu -k "persistent or conv or upsample or dgrad" 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm version\|^Hostname\|^Librccl" | tail -4 > gpurun_out/up2_test.txt
timeout 600 python bench.py --steps 15 --warmup 4 --no-train-leg --no-cpu-baseline --detail gpurun_out/up2_detail.tsv 2>/dev/null | python -c "import sys, json; j = json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(j['value'], 2), 'img/s', round(j['ms_per_step'], 2), 'ms', round(j['roofline']['achieved'], 1))" > gpurun_out/up2_bench.txt
grep "s1u" gpurun_out/up2_detail.tsv >> gpurun_out/up2_bench.txt
cat gpurun_out/up2_test.txt gpurun_out/up2_bench.txt
