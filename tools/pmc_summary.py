#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection.csv files per kernel (mean per dispatch).
usage: tools/pmc_summary.py gpurun_out/pmc_*/**/*_counter_collection.csv [--kernel SUBSTR]"""
import csv, sys, collections, glob
args = [a for a in sys.argv[1:] if not a.startswith('--')]
sub = None
if '--kernel' in sys.argv: sub = sys.argv[sys.argv.index('--kernel') + 1]; args = [a for a in args if a != sub]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for pat in args:
    for path in glob.glob(pat, recursive=True):
        with open(path) as f:
            for row in csv.DictReader(f):
                k = row['Kernel_Name']
                if sub and sub not in k: continue
                acc[k][row['Counter_Name']].append(float(row['Counter_Value']))
for k, cs in acc.items():
    print(k[:110])
    for c, v in sorted(cs.items()):
        print(f'   {c:28s} n={len(v):3d} mean={sum(v)/len(v):16.1f}')
