#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out
timeout 900 python scripts/slot_dependence.py 576 32 bf16 nosplit > $O/r04_slot_dependence_bf16_b32_nosplit.txt 2>&1
grep -n "first\|loss\|Error" $O/r04_slot_dependence_bf16_b32_nosplit.txt | head; awk '$3 != "0" && NR > 2' $O/r04_slot_dependence_bf16_b32_nosplit.txt | head -8
timeout 900 python -m pytest tests/test_config2_batch32_gpu.py -q -m gpu -s -k mirrored 2>&1 | grep -E "configs|passed|failed|Error" | cut -c1-400
