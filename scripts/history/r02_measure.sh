#!/bin/bash
# round-2 measurement batch (one gpurun call): default bench line, rocprofv3 kernel stats (inference / training), PMC HBM traffic, training with
# activation recompute, GeoWizard, GPU test log.  Everything lands in gpurun_out/; the summaries are copied to profiles/ afterwards.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out
python -m pytest tests -q -m gpu 2>&1 | tail -5 > $O/r02b_gpu_tests.log
python bench.py --steps 20 --warmup 5 --detail $O/r02b_bench_per_shape.tsv > $O/r02b_bench_default.json 2> $O/r02b_bench_default.err
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_inf -o r02 -- python bench.py --no-train-leg --no-cpu-baseline --steps 3 --warmup 1 > /dev/null 2>&1
cp $(find /tmp/prof_inf -name "*kernel_stats.csv" | head -1) $O/r02b_rocprofv3_kernel_stats.csv
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_train -o r02 -- python bench.py --train --steps 2 --warmup 1 > /dev/null 2>&1
cp $(find /tmp/prof_train -name "*kernel_stats.csv" | head -1) $O/r02b_train_rocprofv3_kernel_stats.csv
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pmc_$c -o p -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-train-leg > /dev/null 2>&1
done
python scripts/pmc_traffic.py /tmp/pmc_FETCH_SIZE /tmp/pmc_WRITE_SIZE $O/r02b_pmc_hbm_traffic.json > $O/r02b_pmc_traffic.log 2>&1
python bench.py --train --dtype fp32 --micro-batch 32 --grad-ckpt --steps 2 --warmup 1 > $O/r02b_bench_train_fp32_ckpt_1x32.json 2>/dev/null
python bench.py --train --dtype bf16 --grad-ckpt --steps 3 --warmup 1 > $O/r02b_bench_train_bf16_ckpt.json 2>/dev/null
python bench.py --train --steps 3 --warmup 1 --detail $O/r02b_bench_train_per_shape.tsv > $O/r02b_bench_train_bf16.json 2>/dev/null
python bench.py --geowizard --steps 10 --warmup 3 > $O/r02b_bench_geowizard_n1.json 2>/dev/null
du -sh $O; ls $O | head -40
