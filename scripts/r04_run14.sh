#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out
timeout 1500 python -m pytest tests -q -m gpu -x 2>&1 | tail -15 > $O/r04a_gpu_tests.log
tail -6 $O/r04a_gpu_tests.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/r04a_smoke.log 2>&1; tail -2 $O/r04a_smoke.log
timeout 900 python bench.py --steps 10 --warmup 3 --detail $O/r04a_bench_per_shape.tsv > $O/r04a_bench_default.json 2> $O/r04a_bench_default.err
python - <<PY
import json
j=json.load(open("gpurun_out/r04a_bench_default.json"))
print("inference", round(j["value"],2), round(j["ms_per_step"],2), round(j["roofline"]["frac"],4), j["roofline"]["launches_per_step"], json.dumps({k:(round(v["ms_per_step"],2), round(v.get("tflops",0) or v.get("gbs",0))) for k,v in j["roofline"]["other_kernels"].items()}))
print("train", j.get("train_step",{}).get("value"), j.get("train_step_fp32",{}).get("value"), "latency", j.get("latency_b1_576x768",{}).get("value"), "cpu", j["cpu_baseline"]["value"], "geowizard", j.get("geowizard",{}).get("value"), j.get("geowizard",{}).get("error"))
print(json.dumps(j["roofline"].get("by_symbol"), indent=0)[:1500])
PY
