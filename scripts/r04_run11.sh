#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out
for sh in "8 5 9216" "8 10 2304" "8 20 576" "2 5 9216"; do for v in v1 v2 v2_nw8; do echo "== $v $sh"; scripts/bin/attn_$v $sh 20; done; done > $O/r04_attn_v1_v2_c.txt 2>&1
scripts/bin/attn_v2_p1 8 5 9216 20 | tail -1 >> $O/r04_attn_v1_v2_c.txt
cat $O/r04_attn_v1_v2_c.txt
