#!/bin/bash
# GPU suite + the three bench lines (configs 1, 2 with the shard rehearsal, 4)
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r04f; rm -rf $O; mkdir -p $O; cd $R; export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; rc=$?; tail -5 $O/pytest.log; echo "pytest rc=$rc"
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python bench.py --config 1 --steps 20 > $O/bench_cfg1.json 2> $O/bench_cfg1.log; echo "cfg1 rc=$?"
timeout -k 10 300 python bench.py --config 4 > $O/bench_cfg4.json 2> $O/bench_cfg4.log; echo "cfg4 rc=$?"; tail -3 $O/bench_cfg4.log
timeout -k 10 500 python bench.py --steps 10 --no-cpu --shard-of 8 > $O/bench_cfg2_shard8.json 2> $O/bench_cfg2_shard8.log; echo "cfg2 rc=$?"
python3 - <<'PY'
import json
for f in ('bench_cfg1', 'bench_cfg4', 'bench_cfg2_shard8'):
    try:
        d = json.load(open(f'gpurun_out/r04f/{f}.json'))
    except Exception as e:
        print(f, 'unreadable', e); continue
    print(f, d['config']['baseline_config'], 'value', f"{d['value']:.4g}", 'ms/step', round(d['ms_per_step'], 4))
    print('   roofline', {k: (round(v, 4) if isinstance(v, float) else v) for k, v in (d.get('roofline') or {}).items() if k in ('frac', 'launch_ms', 'achieved', 'kernel')})
    if 'kernels' in d:
        for k, v in d['kernels'].items():
            print('   ', k, {a: (round(b, 4) if isinstance(b, float) else b) for a, b in v.items() if a in ('ms', 'frac_hbm', 'valu_slot_frac', 'launch_ms', 'GBps')})
    if 'shard_rehearsal' in d:
        print('   shard', {k: (round(v, 4) if isinstance(v, float) else v) for k, v in d['shard_rehearsal'].items() if k != 'note'})
    if 'cpu_baseline' in d:
        print('   cpu', {k: v for k, v in d['cpu_baseline'].items() if k in ('value', 'cores', 'seconds', 'gpu_vs_cpu_rel_l2_on_sample', 'gpu_vs_exact_rel_l2_on_sample')})
    if 'materialised' in d and d['materialised']:
        print('   materialised', d['materialised']['ms_total'], d['materialised']['stages_ms'], 'fused', d['fused']['ms_fwd_with_checkpoints'], d['fused']['ms_bwd'])
PY
