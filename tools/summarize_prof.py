#!/usr/bin/env python
"""Condenses rocprofv3 CSV output (kernel stats + PMC passes) into a per-kernel text table."""
import csv
import glob
import os
import sys
from collections import defaultdict

root = sys.argv[1]
short = lambda n: ('fwd' if ('grand_fwd' in n or 'wide::fwd' in n) else 'bwd_target' if 'bwd_target' in n else 'bwd_source' if 'bwd_source' in n else n[:40])

print("== kernel stats (rocprofv3 --kernel-trace --stats) ==")
for f in glob.glob(os.path.join(root, 'stats', '**', '*kernel_stats.csv'), recursive=True):
    for row in csv.DictReader(open(f)):
        print(f"{short(row['Name']):<42} calls={row['Calls']:>6} avg_ns={float(row['AverageNs']):>12.0f} total_ns={row['TotalDurationNs']:>12} pct={row['Percentage']}")
print("\n== PMC (mean per dispatch, hot kernels) ==")
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(root, 'pmc*', '**', '*counter_collection.csv'), recursive=True):
    for row in csv.DictReader(open(f)):
        k = row.get('Kernel_Name', '')
        if 'grand_' not in k and 'wide::' not in k:
            continue
        acc[short(k)][row['Counter_Name']].append(float(row['Counter_Value']))
for k in sorted(acc):
    print(k)
    for c in sorted(acc[k]):
        v = acc[k][c]
        print(f"    {c:<34} {sum(v) / len(v):>18.1f}   (n={len(v)})")
