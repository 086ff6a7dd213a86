#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 600 python scripts/train_glue_profile.py 32 576 > gpurun_out/r04g_train_glue_profile.txt 2>&1
tail -100 gpurun_out/r04g_train_glue_profile.txt
