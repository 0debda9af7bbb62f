#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r04h; rm -rf $O; mkdir -p $O; cd $R; export TMPDIR=/tmp
for c in "64 1024" "64 2048" "56 1024" "48 1024"; do set -- $c
TIMELINE_WHAT=k2 timeout -k 10 200 python tools/timeline_waves.py $1 $2 $O/k2_timeline_$1_$2.json 2>&1 | grep -v "amdgpu.ids\|simd_last" | python3 -c "
import sys, json
for l in sys.stdin:
    try: r = json.loads(l)
    except Exception: print(l[:300]); continue
    print('$1 $2', r['kernel'], 'event_ms', r.get('event_ms'), 'span_us', r['span_us'], 'waves/cu', r['waves_per_cu_minmax'], 'cus', r['cus_used'], 'end pct', r['end_us_pct'])
"; done
