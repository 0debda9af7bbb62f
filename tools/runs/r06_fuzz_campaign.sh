#!/bin/bash
# One-off fuzz campaign on the final round-6 tree: 3 seeds x 300 forward problems + 3 seeds x 150 gradient problems (fp64, 1e-9 against
# oracle/bloch_c.c and the torch oracle's autograd; campaign coverage: coil counts to 66, pulse lengths on the fp64 line grid), then
# fp32 forward (1e-5) and gradients (3e-5).  Every gradient case also draws how the history is cut (1 / 2 / 3 / 4 / 8 parts, blocked or
# interleaved, whatever the size: round 6, ABI 5).
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r06fz; mkdir -p $O; cd $R
for seed in ${SEEDS:-111 222 333}; do
  MRPHY_FUZZ_SEED=$seed MRPHY_FUZZ_CASES=${FWD_CASES:-300} timeout -k 10 500 python3 -m pytest tests/test_k1_k3.py -m gpu -q -k "test_fuzz_forward_vs_c_restatement" > $O/fwd_$seed.txt 2>&1; rc=$?
  echo "forward seed $seed cases ${FWD_CASES:-300} rc=$rc: $(tail -n 1 $O/fwd_$seed.txt)" | tee -a $O/summary.txt
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit 1; fi
  MRPHY_FUZZ_SEED=$seed MRPHY_FUZZ_CASES=${GRAD_CASES:-150} timeout -k 10 500 python3 -m pytest tests/test_k1_k3.py -m gpu -q -k "test_fuzz_gradients_vs_oracle" > $O/grad_$seed.txt 2>&1; rc=$?
  echo "gradients seed $seed cases ${GRAD_CASES:-150} rc=$rc: $(tail -n 1 $O/grad_$seed.txt)" | tee -a $O/summary.txt
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit 1; fi
done
MRPHY_FUZZ_SEED=${F32_SEED:-87} MRPHY_FUZZ_CASES=${F32_CASES:-2000} timeout -k 10 700 python3 -m pytest tests/test_k1_k3.py -m gpu -q -s -k "test_fuzz_forward_fp32_vs_fp64_oracle" > $O/f32_fwd.txt 2>&1; rc=$?
echo "fp32 forward seed ${F32_SEED:-87} cases ${F32_CASES:-2000} rc=$rc: $(grep -i "worst" $O/f32_fwd.txt | tail -n 1) $(tail -n 1 $O/f32_fwd.txt)" | tee -a $O/summary.txt
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit 1; fi
MRPHY_FUZZ_SEED=${F32_SEED:-87} MRPHY_FUZZ_CASES=${F32_GRAD_CASES:-400} timeout -k 10 700 python3 -m pytest tests/test_k1_k3.py -m gpu -q -s -k "test_fuzz_gradients_fp32_vs_fp64_oracle" > $O/f32_grad.txt 2>&1; rc=$?
echo "fp32 gradients seed ${F32_SEED:-87} cases ${F32_GRAD_CASES:-400} rc=$rc: $(grep -i "worst" $O/f32_grad.txt | tail -n 1) $(tail -n 1 $O/f32_grad.txt)" | tee -a $O/summary.txt
