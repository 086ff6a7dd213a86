#!/bin/bash
# last call of round 4: the final tree's library (rebuilt after two dead-variable removals) loads and computes: smoke + the op / wgrad / igemm2-heavy test files
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 60 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
timeout 100 python -m pytest tests/test_ops_gpu.py tests/test_wgrad_gpu.py -q -m gpu -x 2>&1 | tail -3
