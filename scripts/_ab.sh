mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_model_gpu.py tests/test_fullsize_parity_gpu.py tests/test_bwd_gpu.py -x -q -m gpu -k "groupnorm or gn or unet or vae or pipeline or config" 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm version\|^Hostname\|^Librccl" | tail -4 > gpurun_out/gnfin_test.txt
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_inf -o r02 -- python bench.py --no-train-leg --no-cpu-baseline --steps 3 --warmup 1 > gpurun_out/gnfin_bench.json 2>/dev/null
grep -E "gn_finalize|gn_apply|gn_partial" $(find /tmp/prof_inf -name "*kernel_stats.csv" | head -1) | cut -c1-200 > gpurun_out/gnfin_stats.txt
timeout 600 python bench.py --steps 15 --warmup 4 --no-train-leg --no-cpu-baseline 2>/dev/null | python -c "import sys, json; j = json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(j['value'], 2), 'img/s', round(j['ms_per_step'], 2), 'ms', j['roofline']['other_kernels']['groupnorm'])" >> gpurun_out/gnfin_stats.txt
cat gpurun_out/gnfin_test.txt gpurun_out/gnfin_stats.txt
