#!/bin/bash
# round 6, call 8: whole GPU suite after the narrow.hip rounding fix (the fused conv_out route rounded u * r once, gn_apply twice)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out; mkdir -p $O
python -c "from diffusion_e2e_ft_amd import build as b; print('library build id', b.built_id(), 'source id', b.source_id())" > $O/r06h_build_id.txt 2>&1
timeout 2400 python -m pytest tests -q -m gpu 2>&1 | tail -15 > $O/r06h_gpu_tests.log
cat $O/r06h_build_id.txt $O/r06h_gpu_tests.log
