#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r04b; rm -rf $O; mkdir -p $O; cd $R; export TMPDIR=/tmp
timeout -k 10 300 python tools/block_probe.py $O/block_probe.json 6 7 > $O/block_probe.log 2>&1; rc=$?; cat $O/block_probe.log | cut -c1-400; echo "rc=$rc"
timeout -k 10 300 python bench.py --cube 64 --nT 1024 --no-cpu > $O/bench_cfg1.json 2> $O/bench_cfg1.log; echo "bench rc=$?"; cat $O/bench_cfg1.json | cut -c1-1500
