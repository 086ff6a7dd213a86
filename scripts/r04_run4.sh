#!/bin/bash
# round 4, GPU call 4: HIP side of the bf16 localisation on the oracle's network + CPU-generated sample (identical instance to the cpu side)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out
BF16_DUMP=$O/r04_est_dump_hip.pt timeout 1200 python scripts/bf16_localise.py hip 13 576 $O/r04_bf16_localise_d default > $O/r04_bf16_localise_hip_d.log 2>&1
grep "draw\|loss\|Error" $O/r04_bf16_localise_hip_d.log | tail -30
