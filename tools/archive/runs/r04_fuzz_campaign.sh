#!/bin/bash
# One-off fuzz campaign on the final tree: 3 seeds x 300 forward problems + 3 seeds x 150 gradient problems (fp64, 1e-9 against
# oracle/bloch_c.c and the torch oracle's autograd), campaign coverage (coil counts to 66, pulse lengths on the fp64 line grid).
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r04fz; mkdir -p $O; cd $R
for seed in ${SEEDS:-101 202 303}; do
  MRPHY_FUZZ_SEED=$seed MRPHY_FUZZ_CASES=${FWD_CASES:-300} timeout -k 10 500 python3 -m pytest tests/test_k1_k3.py -m gpu -q -k "test_fuzz_forward_vs_c_restatement" > $O/fwd_$seed.txt 2>&1; rc=$?
  echo "forward seed $seed cases ${FWD_CASES:-300} rc=$rc: $(tail -n 1 $O/fwd_$seed.txt)" | tee -a $O/summary.txt
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit 1; fi
  MRPHY_FUZZ_SEED=$seed MRPHY_FUZZ_CASES=${GRAD_CASES:-150} timeout -k 10 500 python3 -m pytest tests/test_k1_k3.py -m gpu -q -k "test_fuzz_gradients_vs_oracle" > $O/grad_$seed.txt 2>&1; rc=$?
  echo "gradients seed $seed cases ${GRAD_CASES:-150} rc=$rc: $(tail -n 1 $O/grad_$seed.txt)" | tee -a $O/summary.txt
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit 1; fi
done
