#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r03e; rm -rf $O; mkdir -p $O; cd $R; export TMPDIR=/tmp
timeout -k 10 500 python3 tools/placement_vs_size.py $O/placement_vs_size.json 8 > $O/placement_vs_size.txt 2>&1; echo "rc=$?"
grep -v amdgpu.ids $O/placement_vs_size.txt
