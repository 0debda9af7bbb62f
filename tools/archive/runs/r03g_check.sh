#!/bin/bash
# full GPU suite + headline bench (no CPU leg) + gradient benches: the check after a kernel change
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r03g; rm -rf $O; mkdir -p $O; cd $R; export TMPDIR=/tmp
export MRPHY_PARITY_LEDGER=$O/parity_ledger.json
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?"; tail -n 3 $O/pytest_gpu.txt
unset MRPHY_PARITY_LEDGER
timeout -k 10 300 python3 bench.py --steps 10 --warmup 2 --no-cpu > $O/bench.json 2> $O/bench.log; echo "bench rc=$?"
python3 -c "
import json; d=json.load(open('$O/bench.json')); k=d['kernels']
print('step', round(d['ms_per_step'],3), 'K0', round(k['K0_rfgr2beff']['ms'],3), 'K1', round(k['K1_blochsim_fwd']['ms'],3), 'frac', round(d['roofline']['frac'],4))
K2=k['K2_fused_rfgr_fwd']; print('K2', round(K2['ms'],3), K2['valu_slot_frac'], 'bitwise', K2['equals_K0_K1_bitwise'], 'fast', round(K2['fast_step']['ms'],3))"
timeout -k 10 300 python3 bench.py --mode grad --steps 10 --warmup 2 > $O/grad_cfg4.json 2> $O/grad_cfg4.log; echo "grad cfg4 rc=$?"
timeout -k 10 300 python3 bench.py --mode grad --cube 128 --nT 1024 --no-interp --steps 5 --warmup 2 > $O/grad128.json 2> $O/grad128.log; echo "grad128 rc=$?"
python3 -c "
import json
for f in ('$O/grad_cfg4.json','$O/grad128.json'):
    d=json.load(open(f)); print(d['config']['spins'], d['config']['nT'], 'mat', round(d['materialised']['ms_total'],3), {k:round(v,3) for k,v in d['materialised']['stages_ms'].items()}, 'fused fwd', round(d['fused']['ms_fwd_with_checkpoints'],3), 'bwd', round(d['fused']['ms_bwd'],3), d['grad_fused_vs_materialised_rel_l2'])"
