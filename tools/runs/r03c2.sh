#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r03c; mkdir -p $O; cd $R; export TMPDIR=/tmp
export MRPHY_PARITY_LEDGER=$O/parity_ledger.json
timeout -k 10 1100 python3 -m pytest tests -m gpu -x -q -s > $O/pytest_gpu.txt 2>&1
echo "pytest rc=$?" | tee -a $O/pytest_gpu.txt
tail -n 12 $O/pytest_gpu.txt
