#!/bin/bash
# The gradient-route part of r05_profiles.sh alone (after a change of the Python layer that does not touch the kernels).
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r05p; mkdir -p $O; cd $R; export TMPDIR=/tmp
rm -rf $O/prof_cfg4 $O/prof_cfg4_both $O/prof_f64
P="rocprofv3 --kernel-trace --output-format csv"
timeout -k 10 300 python3 bench.py --config 4 --steps 10 --warmup 2 > $O/bench_cfg4.json 2> $O/bench_cfg4.log; echo "bench cfg4 rc=$?"
timeout -k 10 300 $P --stats -d $O/prof_cfg4 -- python3 bench.py --config 4 --steps 30 --warmup 2 --no-cpu --grad-route workspace > /dev/null 2> $O/prof_cfg4.log; echo "prof cfg4 rc=$?"
timeout -k 10 300 $P --stats -d $O/prof_cfg4_both -- python3 bench.py --config 4 --steps 10 --warmup 2 --no-cpu > /dev/null 2> $O/prof_cfg4_both.log; echo "prof cfg4 both rc=$?"
timeout -k 10 300 $P --stats -d $O/prof_f64 -- python3 tools/run_kernels.py gradws64 64 1024 10 > /dev/null 2> $O/prof_f64.log; echo "prof f64 rc=$?"
find $O -name '*.db' -delete; find $O -name '*agent_info*' -delete
