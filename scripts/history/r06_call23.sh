#!/bin/bash
# round 6, call 23 (test-only change): the one-GPU multi-rank tests under the suite's grid-8 variant (rank workers now apply the variant's options) and under the default
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
{ E2EFT_TEST_PERSISTENT_GRID=8 timeout 900 python -m pytest tests/test_two_ranks_one_gpu.py -q -m gpu 2>&1 | tail -4; timeout 900 python -m pytest tests/test_two_ranks_one_gpu.py -q -m gpu 2>&1 | tail -4; } > $O/r06w_two_ranks_both_variants.log
cat $O/r06w_two_ranks_both_variants.log
