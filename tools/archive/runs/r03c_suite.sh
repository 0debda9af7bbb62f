#!/bin/bash
# round-3: full GPU suite (writes the parity ledger), smoke, gradient-path timings with the precise adjoint
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r03c; rm -rf $O; mkdir -p $O; cd $R; export TMPDIR=/tmp
export MRPHY_PARITY_LEDGER=$O/parity_ledger.json
timeout -k 10 1100 python3 -m pytest tests -m gpu -x -q -s > $O/pytest_gpu.txt 2>&1
echo "pytest rc=$?" | tee -a $O/pytest_gpu.txt
tail -n 8 $O/pytest_gpu.txt
timeout -k 10 200 python3 -c 'import __graft_entry__ as g; g.smoke()' > $O/smoke.txt 2>&1; echo "smoke rc=$?"; tail -n 2 $O/smoke.txt
timeout -k 10 300 python3 bench.py --mode grad --steps 5 --warmup 2 > $O/bench_grad_cfg4.json 2> $O/bench_grad_cfg4.log; echo "grad cfg4 rc=$?"
timeout -k 10 300 python3 bench.py --mode grad --cube 128 --nT 1024 --no-interp --steps 3 --warmup 1 > $O/bench_grad128.json 2> $O/bench_grad128.log; echo "grad128 rc=$?"
cat $O/bench_grad_cfg4.json $O/bench_grad128.json
