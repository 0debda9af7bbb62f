"""Per-kernel means of every counter found under one or more rocprofv3 ``--pmc`` output
directories (separate passes), merged into one JSON.

    python tools/pmc_summary.py OUT.json LABEL DIR [DIR ...]
Kernel names are shortened to the template name + arguments; dispatches of one kernel are averaged
(the first dispatch of each kernel is dropped when there are more than two: cold caches/clocks).
FETCH_SIZE / WRITE_SIZE are reported in bytes with the guide's gfx950 corrections (KiB units;
FETCH_SIZE doubled for 16-B/lane coalesced reads) next to the raw values.
"""
import collections
import csv
import glob
import json
import re
import sys


def short(k):
    k = k.replace('(anonymous namespace)::', '').replace('void ', '')
    k = re.sub(r'\(.*\)$', '', k)
    return k.replace('mrphy::', '')


def main():
    outp, label, dirs = sys.argv[1], sys.argv[2], sys.argv[3:]
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    meta = {}
    for d in dirs:
        for f in glob.glob(f'{d}/**/*counter_collection.csv', recursive=True):
            for r in csv.DictReader(open(f)):
                k = short(r['Kernel_Name'])
                if not k.startswith('k_'):
                    continue
                # dispatch ids restart in every run: key by (file, id), so that stale CSVs of an earlier
                # collection in the same directory can never be summed into this one's dispatches
                acc[k][r['Counter_Name']].append(((f, int(r['Dispatch_Id'])), float(r['Counter_Value'])))
                meta[k] = {x: r.get(x) for x in ('VGPR_Count', 'Accum_VGPR_Count', 'SGPR_Count', 'LDS_Block_Size',
                                                  'Scratch_Size', 'Grid_Size', 'Workgroup_Size') if r.get(x) is not None}
    res = {}
    for k, cs in acc.items():
        e = dict(meta.get(k, {}))
        for c, vs in sorted(cs.items()):
            vs = sorted(vs)
            # one dispatch may give several rows (per-dimension instances): sum per dispatch
            per = collections.OrderedDict()
            for did, v in vs:
                per[did] = per.get(did, 0.0) + v
            vals = list(per.values())
            if len(vals) > 2:
                vals = vals[1:]
            e[c] = sum(vals) / len(vals)
            e[c + '.n'] = len(vals)
        if 'FETCH_SIZE' in e:
            e['fetch_bytes_corrected'] = e['FETCH_SIZE'] * 1024 * 2
        if 'WRITE_SIZE' in e:
            e['write_bytes'] = e['WRITE_SIZE'] * 1024
        res[k] = e
    try:
        allr = json.load(open(outp))
    except Exception:
        allr = {}
    allr[label] = res
    json.dump(allr, open(outp, 'w'), indent=1)
    for k, e in res.items():
        print(k[:110])
        for c, v in e.items():
            if not c.endswith('.n'):
                print(f'    {c:28s} {v if isinstance(v, str) else format(v, ".6g")}')


main()
