#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r03h; rm -rf $O; mkdir -p $O; cd $R; export TMPDIR=/tmp
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q -k "round3 or onestep or abi or interp" > $O/pytest_sel.txt 2>&1; echo "pytest rc=$?"; tail -n 30 $O/pytest_sel.txt
