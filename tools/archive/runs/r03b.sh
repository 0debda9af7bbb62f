#!/bin/bash
# round-3 probe B: gradient parity with the precise (t-state, compensated) adjoint; one-generation
# sweep (occupancy variants x priority rotation) at 64^3 x 2048 and 128^3 x 1024
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r03b; rm -rf $O; mkdir -p $O; cd $R; export TMPDIR=/tmp
timeout -k 10 600 python3 tools/grad_parity.py $O/grad_parity.json > $O/grad_parity.txt 2>&1; echo "grad_parity rc=$?"
timeout -k 10 400 python3 tools/onegen_sweep.py 64 2048 $O/onegen_64_2048.json > $O/onegen_64_2048.txt 2>&1; echo "sweep64 rc=$?"
timeout -k 10 400 python3 tools/onegen_sweep.py 128 1024 $O/onegen_128_1024.json > $O/onegen_128_1024.txt 2>&1; echo "sweep128 rc=$?"
grep -v amdgpu.ids $O/onegen_64_2048.txt | cut -c1-400
