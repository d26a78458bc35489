#!/usr/bin/env python3
"""Condense the rocprofv3 outputs of tools/collect_profiles.sh into the files that get committed under profiles/:
   <round>_kernel_stats.csv      rocprofv3 --kernel-trace --stats summary (per-kernel calls / avg / min / max ns)
   <round>_pmc_summary.txt       per-kernel mean of every collected counter
   traffic.json                  per loss type: HBM bytes per launch of the fused kernel from FETCH_SIZE/WRITE_SIZE
HBM bytes follow MI355X_MICROARCH.md §HBM: FETCH_SIZE and WRITE_SIZE are in KiB-like units of 1024 B; on gfx950
FETCH_SIZE reports exactly half of the bytes of a wide (16 B/lane) coalesced streaming read -> doubled; WRITE_SIZE is
exact for 16-B-per-lane streaming stores."""
import collections, csv, glob, json, os, sys
src, rnd = sys.argv[1], sys.argv[2]
stats = glob.glob(os.path.join(src, 'kt', '*', '*kernel_stats.csv'))
if stats:
    with open(stats[0]) as f, open(os.path.join(src, f'{rnd}_kernel_stats.csv'), 'w') as g:
        g.write(f.read())
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for path in glob.glob(os.path.join(src, 'pmc_*', '*', '*_counter_collection.csv')):
    with open(path) as f:
        for row in csv.DictReader(f):
            acc[row['Kernel_Name']][row['Counter_Name']].append(float(row['Counter_Value']))
names = {0: 'gwd3d', 1: 'kld3d', 2: 'bd3d'}
traffic = {}
with open(os.path.join(src, f'{rnd}_pmc_summary.txt'), 'w') as g:
    g.write('# mean counter value per dispatch (rocprofv3 --pmc, separate passes; bench.py --steps 3 --warmup 2, 10 M pairs)\n')
    for k, cs in sorted(acc.items()):
        if 'gd3d::' not in k:
            continue
        g.write(k + '\n')
        for c, v in sorted(cs.items()):
            g.write(f'    {c:24s} n={len(v):3d} mean={sum(v) / len(v):18.1f}\n')
        if 'fused_kernel<' in k and 'FETCH_SIZE' in cs and 'WRITE_SIZE' in cs:
            lt = names.get(int(k.split('fused_kernel<')[1].split(',')[0]))
            fetch = sum(cs['FETCH_SIZE']) / len(cs['FETCH_SIZE'])
            write = sum(cs['WRITE_SIZE']) / len(cs['WRITE_SIZE'])
            hbm = 2 * fetch * 1024 + write * 1024
            g.write(f'    -> HBM bytes/launch = 2*FETCH_SIZE*1024 + WRITE_SIZE*1024 = {hbm:.0f}\n')
            if lt:
                traffic[lt] = {'hbm_bytes_per_launch': round(hbm), 'fetch_size_raw': fetch, 'write_size_raw': write,
                               'correction': 'FETCH_SIZE x2 (gfx950 wide coalesced reads), units 1024 B'}
with open(os.path.join(src, 'traffic.json'), 'w') as g:
    json.dump(traffic, g, indent=1)
print(json.dumps(traffic))
