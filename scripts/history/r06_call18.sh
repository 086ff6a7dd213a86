#!/bin/bash
# round 6, call 18: igemm6 NORM variant with the lane's sixteen coefficients loaded once per chunk (four of the five ds_read_b128 of a norm k-tile were coefficient reads)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out; mkdir -p $O
timeout 1200 python -m pytest tests/test_fused_norm_conv_gpu.py tests/test_patch_conv_gpu.py tests/test_persistent_gpu.py -x -q -m gpu 2>&1 | tail -6 > $O/r06r_norm_tests.log
{
for rep in 1 2; do
timeout 200 python scripts/norm_conv_bench.py 8 768 768 128 128 20
timeout 200 python scripts/norm_conv_bench.py 8 768 768 256 128 20
done
} > $O/r06r_norm_conv_bench.txt 2>&1
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-train-leg --no-latency-leg --no-geowizard-leg > $O/r06r_bench_quick.json 2> /dev/null
cat $O/r06r_norm_tests.log; grep -v amdgpu.ids $O/r06r_norm_conv_bench.txt
python -c "
import json; j=json.load(open('gpurun_out/r06r_bench_quick.json')); print(j['value'], j['ms_per_step'], j['roofline']['frac']); print({k:(round(v['ms_per_step'],2), round(v['tflops'])) for k,v in j['roofline'].get('by_symbol',{}).items() if 'igemm6' in k})"
