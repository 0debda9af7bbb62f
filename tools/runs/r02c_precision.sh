#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r02c; mkdir -p $O; cd $R; rm -f $O/sweep.txt
for v in 0 331 341 321 431 541 241; do
  echo "== MRPHY_FWD_VARIANT=$v" >> $O/sweep.txt
  MRPHY_FWD_VARIANT=$v timeout -k 10 300 python3 tools/precision_sweep.py 128 4096 2>&1 | grep -v amdgpu.ids >> $O/sweep.txt
done
echo "== 64^3 x 4096 (one wave generation)" >> $O/sweep.txt
for v in 0 331; do
  MRPHY_FWD_VARIANT=$v timeout -k 10 300 python3 tools/precision_sweep.py 64 4096 2>&1 | grep -v amdgpu.ids >> $O/sweep.txt
done
cat $O/sweep.txt
