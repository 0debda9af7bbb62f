#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r04d; rm -rf $O; mkdir -p $O; cd $R; export TMPDIR=/tmp
timeout -k 10 120 tools/mall_wr_rd 3 > $O/mall_wr_rd_3g.txt 2>&1; echo "rc=$?"; cat $O/mall_wr_rd_3g.txt
timeout -k 10 120 tools/mall_wr_rd 12 > $O/mall_wr_rd_12g.txt 2>&1; echo "rc=$?"; cat $O/mall_wr_rd_12g.txt
