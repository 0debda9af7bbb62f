#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r04c; rm -rf $O; mkdir -p $O; cd $R; export TMPDIR=/tmp
timeout -k 10 400 python tools/k0k1_state.py $O/k0k1_state.json > $O/k0k1_state.log 2>&1; rc=$?; grep -v amdgpu.ids $O/k0k1_state.log | cut -c1-300; echo "rc=$rc"
