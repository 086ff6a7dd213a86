#!/bin/bash
# round 4, GPU call 1: where the bf16 gradient noise enters (HIP side of scripts/bf16_localise.py, default + diagnostic variants), then the new tests
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out
timeout 900 python scripts/bf16_localise.py hip 13 576 $O/r04_bf16_localise default > $O/r04_bf16_localise_hip.log 2>&1
timeout 900 python scripts/bf16_localise.py hip 7 576 $O/r04_bf16_localise vaeattn_fp32,nostats,nothin,nopatch,noflashbwd > $O/r04_bf16_localise_hip_variants.log 2>&1
grep -h "draw\|Error\|error" $O/r04_bf16_localise_hip.log $O/r04_bf16_localise_hip_variants.log | tail -70
timeout 1500 python -m pytest tests/test_config2_batch32_gpu.py tests/test_abi_gpu_client.py tests/test_two_ranks_one_gpu.py tests/test_abi.py -q -m gpu -s 2>&1 | grep -v "^$" | tail -60 > $O/r04_run1_tests.log
tail -45 $O/r04_run1_tests.log
