#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python scripts/grad_sink_probe.py 2 256 2>&1 | tail -45
python scripts/grad_sink_probe.py 4 576 2>&1 | tail -8
