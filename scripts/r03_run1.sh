#!/bin/bash
# round-3 first GPU call: new kernels / tests first (under their own timeouts), then the whole suite, then the default bench line
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out
timeout 600 python -m pytest tests/test_attn512_gpu.py -q -x -s 2>&1 | tail -40 > $O/r03a_attn512_tests.log
echo "attn512 rc=$?" >> $O/r03a_attn512_tests.log
timeout 900 python -m pytest tests/test_train_gpu.py tests/test_persistent_gpu.py tests/test_ops_gpu.py -q 2>&1 | tail -30 > $O/r03a_changed_tests.log
timeout 1500 python -m pytest tests/test_benchmarked_configs_gpu.py tests/test_fullsize_parity_gpu.py -q -s 2>&1 | grep -v "^$" | tail -80 > $O/r03a_fullsize_tests.log
timeout 1200 python -m pytest tests -q -m gpu --deselect tests/test_benchmarked_configs_gpu.py --deselect tests/test_fullsize_parity_gpu.py 2>&1 | tail -15 > $O/r03a_gpu_tests.log
timeout 900 python bench.py --steps 20 --warmup 5 --detail $O/r03a_bench_per_shape.tsv > $O/r03a_bench_default.json 2> $O/r03a_bench_default.err
tail -c 1500 $O/r03a_attn512_tests.log; tail -5 $O/r03a_changed_tests.log; tail -30 $O/r03a_fullsize_tests.log; tail -5 $O/r03a_gpu_tests.log; head -c 1500 $O/r03a_bench_default.json; tail -3 $O/r03a_bench_default.err
