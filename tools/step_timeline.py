#!/usr/bin/env python3
"""Ordered kernel timeline of the last bench step from a rocprofv3 --kernel-trace CSV.

    python tools/step_timeline.py <dir with *_kernel_trace.csv>
"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'adam_step' in r['Kernel_Name']]
a, b = idx[-2], idx[-1]
prev, busy = None, 0.0
for r in rows[a + 1:b + 1]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    gap = (s - prev) / 1e3 if prev else 0.0
    busy += (e - s) / 1e3
    print(f"{gap:7.2f} gap  {(e - s) / 1e3:7.2f} us  {r['Kernel_Name'][:100]}")
    prev = e
print(f"launches {b - a}  busy {busy:.1f} us  span {(int(rows[b]['End_Timestamp']) - int(rows[a]['End_Timestamp'])) / 1e3:.1f} us")
