cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out
python bench.py --steps 20 --warmup 5 --detail $O/r02c_bench_per_shape.tsv > $O/r02c_bench_default.json 2> $O/r02c_bench_default.err
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_inf -o r02 -- python bench.py --no-train-leg --no-cpu-baseline --steps 3 --warmup 1 > /dev/null 2>&1
cp $(find /tmp/prof_inf -name "*kernel_stats.csv" | head -1) $O/r02c_rocprofv3_kernel_stats.csv
python bench.py --geowizard --steps 10 --warmup 3 > $O/r02c_bench_geowizard_n1.json 2>/dev/null
python - <<'PY'
import json
j = json.loads(open("gpurun_out/r02c_bench_default.json").read().strip().splitlines()[-1])
r = j["roofline"]
print(j["value"], j["ms_per_step"], r["achieved"], r["frac"], r["kernel_ms_per_step"], {k: round(v["ms_per_step"], 2) for k, v in r["other_kernels"].items()})
print(j["stages"]["ms_per_step"], j["cpu_baseline"]["value"], j["cpu_baseline"]["sample"][:120], j["cpu_baseline"]["leg_seconds"])
print(j["train_step"]["value"], j["train_step_fp32"]["value"])
g = json.loads(open("gpurun_out/r02c_bench_geowizard_n1.json").read().strip().splitlines()[-1]); print("geo", g["value"], g["ms_per_step"])
PY
head -6 $O/r02c_rocprofv3_kernel_stats.csv | cut -c1-150
