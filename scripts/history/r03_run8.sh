#!/bin/bash
# igemm6 with the fused upsample: its tests, then the whole GPU suite, then the bench legs
O=gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests/test_patch_conv_gpu.py -x -q -s > $O/r03i_patch_tests.log 2>&1; echo "tests rc=$?" >> $O/r03i_patch_tests.log
tail -3 $O/r03i_patch_tests.log
grep -q "2 passed" $O/r03i_patch_tests.log || exit 1
timeout 120 python scripts/conv_bench.py 8 192 192 512 512 3 30 fp16 0 2>&1 | tail -1
timeout 1500 python -m pytest tests -m gpu -x -q > $O/r03i_gpu_tests.log 2>&1; tail -3 $O/r03i_gpu_tests.log
timeout 900 python bench.py --steps 20 --warmup 5 --detail $O/r03i_bench_per_shape.tsv > $O/r03i_bench_default.json 2> $O/r03i_bench_default.err
python - <<'PY'
import json
j=json.loads(open("gpurun_out/r03i_bench_default.json").read().strip().splitlines()[-1])
print("inference", j["value"], j["ms_per_step"], j["roofline"]["frac"], j["roofline"]["launches_per_step"], j["roofline"].get("traffic"))
print({k:(round(v["ms_per_step"],2), round(v["tflops"])) for k,v in j["roofline"]["by_symbol"].items()})
print(j.get("latency_b1_576x768",{}).get("value"), [ (l.get("mode"), l.get("value")) for l in j.get("train_legs",[])] if "train_legs" in j else "")
PY
