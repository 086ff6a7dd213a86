#!/bin/bash
# round 4, GPU call 5: HIP side with the rounded-weights reference and the systematic component
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out
timeout 1200 python scripts/bf16_localise.py hip 13 576 $O/r04b_bf16_localise default > $O/r04b_bf16_localise_hip.log 2>&1
grep "draw\|Error" $O/r04b_bf16_localise_hip.log | tail -4
