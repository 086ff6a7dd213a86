mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_fullsize_parity_gpu.py tests/test_model_gpu.py -x -q -m gpu -k "batched or unet or pipeline or geowizard" 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm version\|^Hostname\|^Librccl" | tail -5 > gpurun_out/batched_test.txt
for pe in 1 0 1 0; do
  E2EFT_BATCHED_PROJ=$pe timeout 600 python bench.py --steps 15 --warmup 4 --no-train-leg --no-cpu-baseline > gpurun_out/bp_bench_$pe.json 2> gpurun_out/bp_bench_$pe.err
  python - <<PY
import json
j = json.loads(open("gpurun_out/bp_bench_$pe.json").read().strip().splitlines()[-1])
print("BATCHED_PROJ=$pe", round(j["value"], 2), "img/s", round(j["ms_per_step"], 2), "ms; unet stage", round(j["stages"]["ms_per_step"]["unet"], 2))
PY
done > gpurun_out/bp_ab.txt 2>&1
cat gpurun_out/batched_test.txt gpurun_out/bp_ab.txt
