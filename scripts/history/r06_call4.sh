#!/bin/bash
# round 6, call 4: whole GPU suite on the build with the half-round persistent rule + GroupNorm apply split by size + 2x2 patch kernel; inference glue profile;
# quick bench; loader with 8 ranks
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out; mkdir -p $O
timeout 2400 python -m pytest tests -q -m gpu -x 2>&1 | tail -15 > $O/r06d_gpu_tests.log
timeout 600 python scripts/infer_glue_profile.py 8 768 > $O/r06d_infer_glue_profile.txt 2>&1
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-train-leg --no-latency-leg --no-geowizard-leg --detail $O/r06d_bench_per_shape.tsv > $O/r06d_bench_quick.json 2> $O/r06d_bench_quick.err
timeout 900 python scripts/loader_bench.py --samples 64 --workers 16 --batch 4 --epochs 3 --ranks 8 > $O/r06d_loader_bench_ranks8.json 2> $O/r06d_loader_bench.err
tail -6 $O/r06d_gpu_tests.log; head -40 $O/r06d_infer_glue_profile.txt
python -c "
import json; j=json.load(open('gpurun_out/r06d_bench_quick.json')); print(j['value'], j['ms_per_step'], j['roofline']['frac'], j.get('stages'))
l=json.load(open('gpurun_out/r06d_loader_bench_ranks8.json')); print({k:v for k,v in l.items() if k in ('decode_only','device_loader','device_loader_ranks')})"
