#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r04j; rm -rf $O; mkdir -p $O; cd $R; export TMPDIR=/tmp
timeout -k 10 500 python tools/k2_ab.py $O/k2_ab.json nopf,,ns4 2>&1 | grep -v amdgpu.ids
