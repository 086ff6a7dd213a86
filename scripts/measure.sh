#!/bin/bash
# The round's measurement batch (one gpurun call, on the final commit): whole GPU suite, smoke, PMC HBM traffic of THIS build (stamped with its build id), the default
# bench line, rocprofv3 kernel stats of the inference leg, the bf16 training leg AND the strict-fp32 training leg (the reference's precision), the configs[3] per-GPU
# legs, the loader with 8 ranks.  Everything lands in gpurun_out/ (merged back); the PMC profile is ALSO copied into profiles/ on the box so that the bench of this
# call finds it — the committed copy is made from gpurun_out/ afterwards.   usage: bash scripts/measure.sh <tag> [notests]
TAG=${1:-r06x}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out
mkdir -p $O
python -c "from diffusion_e2e_ft_amd import build as b; print('library build id', b.built_id(), 'source id', b.source_id())" > $O/${TAG}_build_id.txt 2>&1
if [ "$2" != "notests" ]; then
  timeout 2700 python -m pytest tests -q -m gpu 2>&1 | tail -12 > $O/${TAG}_gpu_tests.log
fi
python -c "import __graft_entry__ as g; g.smoke()" > $O/${TAG}_smoke.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pmc_$c -o p -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-train-leg --no-latency-leg --no-geowizard-leg > /dev/null 2>&1
done
python scripts/pmc_traffic.py /tmp/pmc_FETCH_SIZE /tmp/pmc_WRITE_SIZE $O/${TAG}_pmc_hbm_traffic.json 6 > $O/${TAG}_pmc_traffic.log 2>&1
cp $O/${TAG}_pmc_hbm_traffic.json profiles/${TAG}_pmc_hbm_traffic.json
timeout 1500 python bench.py --steps 20 --warmup 5 --detail $O/${TAG}_bench_per_shape.tsv > $O/${TAG}_bench_default.json 2> $O/${TAG}_bench_default.err
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_inf -o p -- python bench.py --no-train-leg --no-cpu-baseline --no-latency-leg --no-geowizard-leg --steps 3 --warmup 1 > /dev/null 2>&1
cp $(find /tmp/prof_inf -name "*kernel_stats.csv" | head -1) $O/${TAG}_rocprofv3_kernel_stats.csv
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_train -o p -- python bench.py --train --steps 2 --warmup 1 > /dev/null 2>&1
cp $(find /tmp/prof_train -name "*kernel_stats.csv" | head -1) $O/${TAG}_train_rocprofv3_kernel_stats.csv
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_train32 -o p -- python bench.py --train --dtype fp32 --steps 2 --warmup 1 > /dev/null 2>&1
cp $(find /tmp/prof_train32 -name "*kernel_stats.csv" | head -1) $O/${TAG}_train_fp32_rocprofv3_kernel_stats.csv
timeout 600 python bench.py --train --dtype fp32 --steps 4 --warmup 1 --detail $O/${TAG}_bench_train_fp32_per_shape.tsv > $O/${TAG}_bench_train_fp32.json 2>/dev/null
timeout 300 python bench.py --c4 --steps 10 --warmup 2 > $O/${TAG}_bench_c4_bf16_n1.json 2>/dev/null
timeout 300 python bench.py --c4 --dtype fp32 --steps 5 --warmup 1 > $O/${TAG}_bench_c4_fp32_n1.json 2>/dev/null
timeout 900 python scripts/loader_bench.py --samples 64 --workers 16 --batch 4 --epochs 3 --ranks 8 > $O/${TAG}_loader_bench.json 2> $O/${TAG}_loader_bench.err
cat $O/${TAG}_build_id.txt; tail -4 $O/${TAG}_gpu_tests.log; tail -2 $O/${TAG}_smoke.log; tail -3 $O/${TAG}_pmc_traffic.log
python - <<PY
import json
j=json.load(open("gpurun_out/${TAG}_bench_default.json"))
r=j["roofline"]
print("inference", j["value"], j["ms_per_step"], "frac", r["frac"], "traffic", r["traffic"], r["traffic_over_algorithmic"], (r["traffic_source"] or r["traffic_note"])[:160])
print("other", json.dumps({k:(round(v["ms_per_step"],2), round(v.get("tflops",0) or v.get("gbs",0))) for k,v in r["other_kernels"].items()}), "stages", j.get("stages",{}).get("ms_per_step"))
for k in ("train_step","train_step_fp32","train_step_fp32_ckpt"):
    t=j.get(k,{}); print(k, t.get("value"), t.get("median_ms_per_step"), t.get("peak_mem_gib"), (t.get("roofline") or {}).get("frac"), t.get("error"))
print("latency", j.get("latency_b1_576x768",{}).get("value"), "geo", j.get("geowizard",{}).get("value"), "cpu", j["cpu_baseline"]["value"], "build", j.get("build_id"))
try:
    l=json.load(open("gpurun_out/${TAG}_loader_bench.json")); print("loader", {k: (round(v.get("images_per_s", v.get("images_per_s_per_core", 0)),1) if isinstance(v, dict) else v) for k,v in l.items() if k in ("reference_cpu","decode_only","device_loader","device_loader_ranks","device_prepare","host_cores","workers")})
except Exception as e: print("loader failed", e)
for n in ("c4_bf16_n1","c4_fp32_n1","train_fp32"):
    try:
        c=json.load(open("gpurun_out/${TAG}_bench_%s.json"%n)); print(n, c["value"], c.get("median_ms_per_step"), c["peak_mem_gib"])
    except Exception as e: print(n, "failed", e)
PY
