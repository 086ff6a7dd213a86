#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out
for sh in "8 5 9216" "8 10 2304" "8 20 576" "2 5 9216" "1 5 6912"; do for v in v1 v2; do echo "== $v $sh"; scripts/bin/attn_$v $sh 20; done; done > $O/r04_attn_v1_v2_b.txt 2>&1
cat $O/r04_attn_v1_v2_b.txt
