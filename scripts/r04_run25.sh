#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_train_gpu.py -q -m gpu -x -k "born or checkpointing or flat_cast or optimizer" 2>&1 | tail -25 | cut -c1-300
