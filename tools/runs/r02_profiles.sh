#!/bin/bash
# Round-2 evidence: bench lines + rocprofv3 kernel stats + PMC (FETCH_SIZE / WRITE_SIZE, separate passes)
# for the forward headline (128^3 x 4096) and the gradient path (128^3 x 1024, and configs[4]).
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r02p; rm -rf $O; mkdir -p $O; cd $R; export TMPDIR=/tmp
F="python3 bench.py --steps 5 --warmup 2 --no-cpu"
G="python3 bench.py --mode grad --cube 128 --nT 1024 --no-interp --steps 3 --warmup 1"
G4="python3 bench.py --mode grad --steps 5 --warmup 2"
timeout -k 10 400 python3 bench.py --steps 10 --warmup 2 > $O/bench_fwd.json 2> $O/bench_fwd.log
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_fwd -- $F > $O/bench_fwd_prof.json 2> $O/prof_fwd.log
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch_fwd -- $F --no-fused > /dev/null 2> $O/pmc_fetch_fwd.log
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write_fwd -- $F --no-fused > /dev/null 2> $O/pmc_write_fwd.log
timeout -k 10 300 $G > $O/bench_grad128.json 2> $O/bench_grad128.log
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_grad128 -- $G > /dev/null 2> $O/prof_grad128.log
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch_grad128 -- $G > /dev/null 2> $O/pmc_fetch_grad128.log
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write_grad128 -- $G > /dev/null 2> $O/pmc_write_grad128.log
timeout -k 10 300 $G4 > $O/bench_grad_cfg4.json 2> $O/bench_grad_cfg4.log
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_grad_cfg4 -- $G4 > /dev/null 2> $O/prof_grad_cfg4.log
find $O -name '*.db' -delete; find $O -name '*agent_info*' -delete
for d in prof_fwd prof_grad128 prof_grad_cfg4; do python3 tools/kstats.py $O/$d $d >> $O/kstats.txt 2>&1; done
cat $O/kstats.txt; du -sh $O
