#!/bin/bash
# round 4, first GPU call: the GPU suite on the multi-unit build + the K1 pin A/B
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r04a; rm -rf $O; mkdir -p $O; cd $R; export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; rc=$?; tail -5 $O/pytest.log; echo "pytest rc=$rc"
[ $rc -eq 0 ] || exit $rc
timeout -k 10 600 python tools/k1_pin_ab.py $O/k1_pin_ab.json 9 > $O/k1_pin_ab.log 2>&1; rc=$?; tail -12 $O/k1_pin_ab.log | cut -c1-600; echo "ab rc=$rc"
exit $rc
