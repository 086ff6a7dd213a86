#!/bin/bash
# round 6, call 20: narrow.hip (conv_out 128 -> 3) with all of a thread's halo loads in flight before the first store (two-pass staging), vector coefficient loads
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out; mkdir -p $O
timeout 1200 python -m pytest tests/test_fused_norm_conv_gpu.py tests/test_ops_gpu.py tests/test_model_gpu.py tests/test_bwd_gpu.py -x -q -m gpu 2>&1 | tail -6 > $O/r06t_narrow_tests.log
{
for rep in 1 2; do
timeout 200 python scripts/norm_conv_bench.py 8 768 768 128 3 20
timeout 200 python scripts/conv_bench.py 8 768 768 128 3 3 20 fp16
timeout 200 python scripts/conv_bench.py 16 576 576 128 3 3 20 fp32
done
} > $O/r06t_narrow_bench.txt 2>&1
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-train-leg --no-latency-leg --no-geowizard-leg --detail $O/r06t_bench_per_shape.tsv > $O/r06t_bench_quick.json 2> /dev/null
cat $O/r06t_narrow_tests.log; grep -v amdgpu.ids $O/r06t_narrow_bench.txt
python -c "
import json; j=json.load(open('gpurun_out/r06t_bench_quick.json')); print(j['value'], j['ms_per_step'], j['roofline']['frac']); print({k:(round(v['ms_per_step'],2), round(v['tflops'])) for k,v in j['roofline'].get('by_symbol',{}).items() if 'narrow' in k or 'igemm6_kernel<_Float16, false, false>' in k})"
