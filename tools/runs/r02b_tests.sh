#!/bin/bash
# full GPU suite + the new round-2 tests, output kept under gpurun_out/r02b
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r02b; mkdir -p $O; cd $R
timeout -k 10 1000 python3 -m pytest tests -m gpu -x -q -s > $O/pytest_gpu.txt 2>&1
echo "pytest rc=$?" >> $O/pytest_gpu.txt
tail -5 $O/pytest_gpu.txt
