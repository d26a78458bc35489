#!/usr/bin/env python3
"""Per-kernel, per-grid-size average durations from a rocprofv3 kernel_trace.csv.  usage: kt_breakdown.py <dir> [substr]"""
import collections, csv, glob, sys
f = glob.glob(sys.argv[1] + '/*/*kernel_trace.csv')[0]
sub = sys.argv[2] if len(sys.argv) > 2 else ''
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if sub in r['Kernel_Name']:
        acc[(r['Kernel_Name'].split('(')[0][-40:], r['Grid_Size_X'])].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
for k, v in sorted(acc.items()):
    print(f'{k[0]:42s} grid {k[1]:>9s}  n={len(v):4d}  avg {sum(v) / len(v):8.1f} us  min {min(v):8.1f}')
