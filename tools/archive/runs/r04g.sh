#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r04g; rm -rf $O; mkdir -p $O; cd $R; export TMPDIR=/tmp
timeout -k 10 300 python tools/k0var_step_ab.py $O/k0var_step_ab.json 2>&1 | grep -v amdgpu.ids
