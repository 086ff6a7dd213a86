#!/bin/bash
# round 4, GPU call 3: slot dependence at the benchmarked batch (32, bf16) and at 8; est dumps on the CPU-generated sample (same instance as the cpu side)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out
timeout 900 python scripts/slot_dependence.py 576 32 bf16 > $O/r04_slot_dependence_bf16_b32.txt 2>&1
grep -n "first\|loss\|Error" $O/r04_slot_dependence_bf16_b32.txt | head; awk '$3 != "0" && NR > 2' $O/r04_slot_dependence_bf16_b32.txt | head -8
timeout 900 python scripts/slot_dependence.py 576 8 bf16 > $O/r04_slot_dependence_bf16_b8.txt 2>&1
grep -n "first\|loss\|Error" $O/r04_slot_dependence_bf16_b8.txt | head; awk '$3 != "0" && NR > 2' $O/r04_slot_dependence_bf16_b8.txt | head -4
BF16_DUMP=$O/r04_est_dump_hip.pt timeout 900 python scripts/bf16_localise.py hip 13 576 $O/r04_bf16_localise_c default > $O/r04_bf16_localise_hip_c.log 2>&1
grep "draw" $O/r04_bf16_localise_hip_c.log | tail -14
