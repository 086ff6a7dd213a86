#!/bin/bash
# round-4 measurement batch (one gpurun call): whole GPU suite, default bench line, rocprofv3 kernel stats (inference / training), PMC HBM traffic of the
# SAME build (launch count recorded with it), training legs with per-shape tables, GeoWizard.  Everything lands in gpurun_out/; summaries are copied to profiles/.
TAG=${1:-r04f}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out
timeout 1800 python -m pytest tests -q -m gpu 2>&1 | tail -8 > $O/${TAG}_gpu_tests.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/${TAG}_smoke.log 2>&1
# PMC passes first: the default bench below prints roofline.traffic only from a committed profile of the SAME kernel population
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pmc_$c -o p -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-train-leg --no-latency-leg --no-geowizard-leg > /dev/null 2>&1
done
python scripts/pmc_traffic.py /tmp/pmc_FETCH_SIZE /tmp/pmc_WRITE_SIZE $O/${TAG}_pmc_hbm_traffic.json 6 > $O/${TAG}_pmc_traffic.log 2>&1
cp $O/${TAG}_pmc_hbm_traffic.json profiles/${TAG}_pmc_hbm_traffic.json
timeout 900 python bench.py --steps 20 --warmup 5 --detail $O/${TAG}_bench_per_shape.tsv > $O/${TAG}_bench_default.json 2> $O/${TAG}_bench_default.err
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_inf -o p -- python bench.py --no-train-leg --no-cpu-baseline --no-latency-leg --no-geowizard-leg --steps 3 --warmup 1 > /dev/null 2>&1
cp $(find /tmp/prof_inf -name "*kernel_stats.csv" | head -1) $O/${TAG}_rocprofv3_kernel_stats.csv
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_train -o p -- python bench.py --train --steps 2 --warmup 1 > /dev/null 2>&1
cp $(find /tmp/prof_train -name "*kernel_stats.csv" | head -1) $O/${TAG}_train_rocprofv3_kernel_stats.csv
python bench.py --train --steps 3 --warmup 1 --detail $O/${TAG}_bench_train_per_shape.tsv > $O/${TAG}_bench_train_bf16.json 2>/dev/null
python bench.py --train --dtype bf16 --grad-ckpt --steps 3 --warmup 1 > $O/${TAG}_bench_train_bf16_ckpt.json 2>/dev/null
python bench.py --geowizard --steps 10 --warmup 3 > $O/${TAG}_bench_geowizard_n1.json 2>/dev/null
[ -n "$GRID8" ] && E2EFT_TEST_PERSISTENT_GRID=8 timeout 1800 python -m pytest tests -q -m gpu 2>&1 | tail -8 > $O/${TAG}_gpu_tests_persist_grid8.log
tail -3 $O/${TAG}_gpu_tests.log; [ -n "$GRID8" ] && tail -2 $O/${TAG}_gpu_tests_persist_grid8.log; tail -2 $O/${TAG}_smoke.log; cat $O/${TAG}_pmc_traffic.log | tail -3
python - <<PY
import json
j=json.load(open("gpurun_out/${TAG}_bench_default.json"))
print("inference", j["value"], j["ms_per_step"], j["roofline"]["frac"], j["roofline"]["launches_per_step"], json.dumps({k:(round(v["ms_per_step"],2), round(v.get("tflops",0) or v.get("gbs",0))) for k,v in j["roofline"]["other_kernels"].items()}))
print("train", j.get("train_step",{}).get("value"), j.get("train_step_fp32",{}).get("value"), "latency", j.get("latency_b1_576x768",{}).get("value"), "cpu", j["cpu_baseline"]["value"])
print("geowizard", json.load(open("gpurun_out/${TAG}_bench_geowizard_n1.json"))["value"])
print("pmc", json.dumps({k: (round(v["hbm_bytes_per_launch"]/1e6,1), v.get("launches_per_step")) for k,v in json.load(open("gpurun_out/${TAG}_pmc_hbm_traffic.json"))["kernels"].items()}))
PY
