#!/bin/bash
# round 6, call 1: the new parity tests (configs[3] share at 2 x 768^2 with recompute; fp32 768^2 latent), the benchmarked-batch test with -s (fp16 latent
# values recorded), the ADVICE-fix tests; a short bench for the session's A/B baseline
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out; mkdir -p $O
nproc > $O/r06a_host.txt; free -g >> $O/r06a_host.txt
timeout 1500 python -m pytest tests/test_config3_c4_gpu.py "tests/test_fullsize_parity_gpu.py::test_config1_768_fp32_latent_and_depth" "tests/test_benchmarked_configs_gpu.py::test_config1_batch8_768_fp16_every_image_against_oracle" -x -q -s -m gpu --durations=10 2>&1 | grep -v amdgpu.ids > $O/r06a_parity_768_tests.log
timeout 600 python -m pytest tests/test_cross_attn_fold_gpu.py tests/test_train_gpu.py -x -q -m gpu -k "fold or ema" 2>&1 | tail -5 > $O/r06a_advice_tests.log
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-train-leg --no-latency-leg --no-geowizard-leg --detail $O/r06a_bench_per_shape.tsv > $O/r06a_bench_quick.json 2> $O/r06a_bench_quick.err
tail -30 $O/r06a_parity_768_tests.log; cat $O/r06a_advice_tests.log; cat $O/r06a_host.txt
python -c "
import json; j=json.load(open('gpurun_out/r06a_bench_quick.json')); print(j['value'], j['ms_per_step'], j['roofline']['frac'])"
