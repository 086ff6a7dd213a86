#!/bin/bash
# round 4, GPU call 2: est / dL/dest dumps of the HIP bf16 draws (for the loss-gradient analysis), slot dependence of the per-sample arithmetic, fixed J4 tests
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out
BF16_DUMP=$O/r04_est_dump_hip.pt timeout 900 python scripts/bf16_localise.py hip 9 576 $O/r04_bf16_localise_b default > $O/r04_bf16_localise_hip_b.log 2>&1
tail -3 $O/r04_bf16_localise_hip_b.log
timeout 600 python scripts/slot_dependence.py 576 2 bf16 > $O/r04_slot_dependence_bf16.txt 2>&1
grep -n "first\|loss" $O/r04_slot_dependence_bf16.txt | head; awk '$3 != "0" && NR > 2' $O/r04_slot_dependence_bf16.txt | head -12
timeout 1500 python -m pytest tests/test_config2_batch32_gpu.py -q -m gpu -s 2>&1 | grep -v "^$" > $O/r04_run2_tests.log
grep -E "rel err|passed|failed|configs\[2\]|colsum" $O/r04_run2_tests.log | cut -c1-330
