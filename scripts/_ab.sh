mkdir -p gpurun_out
for pe in 1 0; do
  E2EFT_PERSIST=$pe timeout 600 python bench.py --steps 10 --warmup 3 --no-train-leg --no-cpu-baseline --detail gpurun_out/pers_detail_$pe.tsv > gpurun_out/pers_bench_$pe.json 2> gpurun_out/pers_bench_$pe.err
done
python - <<'PY'
import json
for pe in (1, 0):
    try:
        j = json.loads(open("gpurun_out/pers_bench_%d.json" % pe).read().strip().splitlines()[-1])
        print(pe, j["value"], j["ms_per_step"], j["roofline"])
    except Exception as e:
        print(pe, "ERR", e, open("gpurun_out/pers_bench_%d.err" % pe).read()[-1500:])
PY
