#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r03d; rm -rf $O; mkdir -p $O; cd $R; export TMPDIR=/tmp
timeout -k 10 500 python3 bench.py --steps 10 --warmup 2 --shard-of 8 --no-cpu > $O/bench_shard8.json 2> $O/bench_shard8.log; echo "bench shard rc=$?"
cat $O/bench_shard8.json
