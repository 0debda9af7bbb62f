#!/bin/bash
# SQ counters of the fused forward kernel K2 at a one-round grid (64^3) and at 128^3, same pulse length
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r04i; rm -rf $O; mkdir -p $O; cd $R; export TMPDIR=/tmp
P="rocprofv3 --kernel-trace --output-format csv"
SQ1="SQ_INSTS_VALU SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_CVT SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE"
SQ2="SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD"
for c in "64 1024" "128 1024"; do set -- $c
timeout -k 10 200 $P --pmc $SQ1 -d $O/sq1_$1 -- python3 tools/run_kernels.py k2 $1 $2 4 > $O/sq1_$1.log 2>&1; echo "sq1 $1 rc=$?"
timeout -k 10 200 $P --pmc $SQ2 -d $O/sq2_$1 -- python3 tools/run_kernels.py k2 $1 $2 4 > $O/sq2_$1.log 2>&1; echo "sq2 $1 rc=$?"
python3 tools/pmc_summary.py $O/k2_small_pmc.json k2_$1_$2 $O/sq1_$1 $O/sq2_$1 > $O/k2_$1.txt 2>&1
done
find $O -name '*.db' -delete; find $O -name '*agent_info*' -delete; find $O -name '*kernel_trace.csv' -delete
python3 - <<'PY'
import json
d = json.load(open('gpurun_out/r04i/k2_small_pmc.json'))
for lab, ks in d.items():
    for k, e in ks.items():
        if 'prec_f32' not in k: continue
        w = e['SQ_WAVES']; wc = e['SQ_WAVE_CYCLES']
        print(lab, k[:70])
        print('   waves', w, 'VALU insts/wave-step', e['SQ_INSTS_VALU'] / w / 1024, 'SALU', e['SQ_INSTS_SALU'] / w / 1024, 'SMEM', e['SQ_INSTS_SMEM'] / w / 1024)
        print('   kernel cycles (GRBM/8)', e['GRBM_GUI_ACTIVE'] / 8, 'SQ_BUSY_CYCLES', e['SQ_BUSY_CYCLES'])
        print('   per wave: cycles', wc / w, ' waiting on waitcnt', e['SQ_WAIT_ANY'] / wc, ' waiting for issue', e['SQ_WAIT_INST_ANY'] / wc, ' issuing', e['SQ_ACTIVE_INST_ANY'] / wc, 'valu', e['SQ_ACTIVE_INST_VALU'] / wc, 'scalar', e['SQ_ACTIVE_INST_SCA'] / wc)
PY
