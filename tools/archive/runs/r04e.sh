#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r04e; rm -rf $O; mkdir -p $O; cd $R; export TMPDIR=/tmp
timeout -k 10 500 python tools/k0k1_step_ab.py $O/k0k1_step_ab.json 9 > $O/k0k1_step_ab.log 2>&1; rc=$?
python3 - <<'PY'
import json
d = json.load(open('gpurun_out/r04e/k0k1_step_ab.json'))
for r in d['runs']:
    print(r['size'], r['mode'], 'bitwise', r['bitwise_equal'])
    for k, v in r['cases'].items():
        print(f"   {k:22s} K0 {v['K0_ms']:8.4f}  K1 {v['K1_ms']:8.4f} ({v['K1_frac']:.3f})  step {v['step_ms']:8.4f}")
PY
echo "rc=$rc"; tail -3 $O/k0k1_step_ab.log | cut -c1-300
