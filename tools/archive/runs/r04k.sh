#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r04k; rm -rf $O; mkdir -p $O; cd $R; export TMPDIR=/tmp
timeout -k 10 600 python tools/ab_libs_f64.py tools/libmrphy_hip_dev.so tools/libmrphy_hip_dev_eo.so tools/libmrphy_hip_dev.so tools/libmrphy_hip_dev_eo.so 2>&1 | grep -v amdgpu.ids | tee $O/f64_ab.txt
