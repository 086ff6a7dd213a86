#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out
timeout 600 python -m pytest tests/test_two_ranks_one_gpu.py -q -x --tb=short 2>&1 | grep -v "^$" | tail -40 > $O/r03d_two_ranks.log
E2EFT_LIB=$GRAFT_REPO_ROOT/diffusion-e2e-ft_amd/lib/libe2eft_stamps.so timeout 600 python scripts/noa_probe.py > $O/r03d_noa_probe.txt 2>&1
cat $O/r03d_two_ranks.log | tail -30; cat $O/r03d_noa_probe.txt
