#!/bin/bash
# round-2 probe A: write/read ceilings, K0 placement modes, profiler-attached K0, PMC traffic of the
# gradient-path kernels (baseline before this round's kernel work).  Run from the repo root on the GPU box.
set -x
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r02a
mkdir -p $O
export TMPDIR=/tmp
cd $R
timeout -k 10 120 tools/hbm_ceiling 96 > $O/hbm_ceiling.txt 2>&1
timeout -k 10 420 python3 tools/k0_modes.py > $O/k0_modes.txt 2>&1
timeout -k 10 200 python3 bench.py --steps 5 --warmup 2 --no-cpu > $O/bench_plain.json 2> $O/bench_plain.log
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_fwd -- python3 bench.py --steps 5 --warmup 2 --no-cpu --no-fused > $O/bench_prof.json 2> $O/bench_prof.log
python3 tools/kstats.py $O/prof_fwd fwd > $O/kstats_fwd.txt 2>&1
G="python3 bench.py --mode grad --cube 128 --nT 1024 --no-interp --steps 3 --warmup 1"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_grad -- $G > $O/grad_prof.json 2> $O/grad_prof.log
python3 tools/kstats.py $O/prof_grad grad > $O/kstats_grad.txt 2>&1
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch_grad -- $G > /dev/null 2> $O/pmc_fetch.log
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write_grad -- $G > /dev/null 2> $O/pmc_write.log
python3 tools/pmc_bytes.py $O/pmc_fetch_grad $O/pmc_write_grad > $O/pmc_grad.txt 2>&1
# keep the merged-back output small: drop the big per-dispatch traces except csv summaries
find $O -name '*.db' -delete
du -sh $O
