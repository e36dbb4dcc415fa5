#!/usr/bin/env python
"""profiles/traffic.json from the rocprofv3 --pmc passes written by tools/rocprof_passes.sh.

HBM-side bytes per launch of each hot kernel = (2 * FETCH_SIZE + WRITE_SIZE) KiB: on gfx950 FETCH_SIZE reports
half the bytes of a wide (16 B/lane) coalesced read (MI355X_MICROARCH.md, HBM section); WRITE_SIZE is exact.
    python tools/make_traffic.py gpurun_out/prof_<tag> <workload> [profiles/traffic.json]
"""
import csv, glob, json, os, sys
from collections import defaultdict
root, workload = sys.argv[1], sys.argv[2]
out = sys.argv[3] if len(sys.argv) > 3 else 'profiles/traffic.json'
name = lambda n: 'forward' if ('grand_fwd' in n or 'wide::fwd' in n) else 'backward_target' if 'bwd_target' in n else 'backward_source' if 'bwd_source' in n else None
def variant(n):
    """dense / compact_g / compact_x from the template arguments in the kernel name (None: not distinguishable by name)."""
    if 'bwd_target' in n:
        a = n[n.index('<') + 1:n.index('>')].replace(' ', '').split(',')
        a += ['false'] * (4 - len(a))
        return 'compact_x' if a[3] == 'true' else 'compact_g' if a[2] == 'true' else 'dense'
    if 'bwd_source' in n:
        a = n[n.index('<') + 1:n.index('>')].replace(' ', '').split(',')
        return 'compact_g' if len(a) > 1 and a[1] == 'true' else 'dense'
    return None
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(root, 'pmc*', '**', '*counter_collection.csv'), recursive=True):
    for row in csv.DictReader(open(f)):
        kn = row.get('Kernel_Name', '')
        k = name(kn)
        if k and row['Counter_Name'] in ('FETCH_SIZE', 'WRITE_SIZE'):
            acc[k][row['Counter_Name']].append(float(row['Counter_Value']))
            v = variant(kn)
            if v:
                acc[k + ':' + v][row['Counter_Name']].append(float(row['Counter_Value']))
res = {}
for k, c in acc.items():
    if 'FETCH_SIZE' in c and 'WRITE_SIZE' in c:
        f, w = sum(c['FETCH_SIZE']) / len(c['FETCH_SIZE']), sum(c['WRITE_SIZE']) / len(c['WRITE_SIZE'])
        res[k] = int((2 * f + w) * 1024)
data = json.load(open(out)) if os.path.exists(out) else {}
data[workload] = res
json.dump(data, open(out, 'w'), indent=1)
print(workload, res)
