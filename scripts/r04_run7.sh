#!/bin/bash
# round 4, GPU call 7: is the wide error distribution of round 3's test instance (device-seeded network + sample) a property of the INSTANCE?  torch bf16 (CPU oracle on the
# box's host cores) and HIP bf16 on that very instance
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out
BF16_INSTANCE=gpuseeded timeout 900 python scripts/bf16_localise.py hip 13 576 $O/r04c_gpuseeded_bf16_localise default > $O/r04c_hip.log 2>&1
grep "draw" $O/r04c_hip.log | tail -13 | awk '{printf "%s ", $NF} END {print ""}'
BF16_INSTANCE=gpuseeded timeout 1500 python scripts/bf16_localise.py cpu 7 576 $O/r04c_gpuseeded_bf16_localise > $O/r04c_cpu.log 2>&1
grep "draw\|run" $O/r04c_cpu.log | tail -16
