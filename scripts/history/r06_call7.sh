#!/bin/bash
# round 6, call 7: GroupNorm(+SiLU) apply in the exp2 domain (gn_apply / igemm6 NORM / narrow): tests, fused-norm conv timing, quick bench; bf16 training A/B of the
# half-round persistent rule; PMC kernel counters with the two-pass collector; loader host side with 8 ranks
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out; mkdir -p $O
timeout 1500 python -m pytest tests/test_fused_norm_conv_gpu.py tests/test_ops_gpu.py tests/test_patch_conv_gpu.py tests/test_model_gpu.py tests/test_fullsize_gpu.py tests/test_benchmarked_configs_gpu.py tests/test_train_gpu.py -x -q -m gpu 2>&1 | tail -6 > $O/r06g_gn_exp2_tests.log
{
for rep in 1 2; do
timeout 200 python scripts/norm_conv_bench.py 8 768 768 128 128 20
timeout 200 python scripts/norm_conv_bench.py 8 768 768 256 128 20
done
} > $O/r06g_norm_conv_bench.txt 2>&1
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-train-leg --no-latency-leg --no-geowizard-leg --detail $O/r06g_bench_per_shape.tsv > $O/r06g_bench_quick.json 2> $O/r06g_bench_quick.err
for q in 8 2 8 2; do
  timeout 300 python bench.py --train --steps 6 --warmup 2 --set-option persistent_min_qrounds=$q 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print('bf16 train min_qrounds $q', j['value'], j.get('median_ms_per_step'))" >> $O/r06g_train_minq_ab.txt
done
bash scripts/pmc_kernels_r06.sh > /dev/null 2>&1
timeout 900 python scripts/loader_bench.py --samples 64 --workers 16 --batch 4 --epochs 3 --ranks 8 > $O/r06g_loader_bench.json 2> $O/r06g_loader_bench.err
cat $O/r06g_gn_exp2_tests.log $O/r06g_norm_conv_bench.txt $O/r06g_train_minq_ab.txt
python -c "
import json; j=json.load(open('gpurun_out/r06g_bench_quick.json')); print(j['value'], j['ms_per_step'], j['roofline']['frac']); print({k:(round(v['ms_per_step'],2), round(v['tflops'])) for k,v in j['roofline'].get('by_symbol',{}).items() if 'igemm6' in k})
l=json.load(open('gpurun_out/r06g_loader_bench.json')); print(l.get('device_loader_ranks'))"
tail -30 $O/r06_pmc_kernels.txt
