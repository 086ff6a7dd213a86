#!/bin/bash
# round 6, call 28 (no library change): which tensor-library kernels a strict-fp32 training micro-step still launches (host-side cost of the step)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
timeout 900 python scripts/train_glue_profile.py 16 576 stacks fp32 2>&1 | grep -v "amdgpu.ids\|Warning\|warn" | tail -50 > $O/r06B_train_glue_fp32_stacks.txt
timeout 900 python scripts/train_glue_profile.py 16 576 methods fp32 2>&1 | grep -v "amdgpu.ids\|Warning\|warn" | tail -45 > $O/r06B_train_glue_fp32_methods.txt
cat $O/r06B_train_glue_fp32_stacks.txt | cut -c1-220; cat $O/r06B_train_glue_fp32_methods.txt | cut -c1-220
