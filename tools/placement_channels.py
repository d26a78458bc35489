"""Per-channel / per-XCD slices of the TCC counters tools/placement_channels.sh collected (derived counters CHk_* / XCCk_*), FAST vs
SLOW placement of the first read stream: mean per launch over 10 launches of each phase."""
import collections, csv, glob, sys
acc = collections.defaultdict(lambda: {'fast': [], 'slow': []})
for path in glob.glob(sys.argv[1] + '/**/*_counter_collection.csv', recursive=True):
    rows = [r for r in csv.DictReader(open(path)) if 'probe_add_kernel' in r['Kernel_Name']]
    by = collections.defaultdict(list)
    for r in rows:
        by[r['Counter_Name']].append((int(r['Dispatch_Id']), float(r['Counter_Value'])))
    for name, vals in by.items():
        vals.sort()
        if len(vals) < 24:
            print('#', name, 'unexpected dispatch count', len(vals)); continue
        vals = vals[-24:]                                    # the driver's last two phases: 12 fast, 12 slow launches
        acc[name]['fast'] += [v for _, v in vals[2:12]]
        acc[name]['slow'] += [v for _, v in vals[14:24]]
def mean(v): return sum(v) / max(len(v), 1)
for short, what in (('RD', 'TCC_EA0_RDREQ (read requests to the fabric)'), ('RDLVL', 'TCC_EA0_RDREQ_LEVEL (summed occupancy: requests in flight x cycles)'), ('TAGST', 'TCC_TAG_STALL')):
    for dim, n, label in (('CH', 16, 'L2 channel (summed over the 8 XCDs)'), ('XCC', 8, 'XCD (summed over its 16 channels)')):
        f = [mean(acc[f'{dim}{k}_{short}']['fast']) for k in range(n) if f'{dim}{k}_{short}' in acc]
        s = [mean(acc[f'{dim}{k}_{short}']['slow']) for k in range(n) if f'{dim}{k}_{short}' in acc]
        if not f: continue
        print(f'== {what}, per {label}')
        print('   fast: ' + ' '.join(f'{x / 1e3:9.1f}' for x in f) + f'   (k; spread max/min {max(f) / max(min(f), 1):.3f})')
        print('   slow: ' + ' '.join(f'{x / 1e3:9.1f}' for x in s) + f'   (k; spread max/min {max(s) / max(min(s), 1):.3f})')
        print('   s/f : ' + ' '.join(f'{(b / a if a else float("nan")):9.3f}' for a, b in zip(f, s)))
