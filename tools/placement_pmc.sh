#!/bin/bash
# Run ON THE GPU BOX: what differs between a fast and a slow placement of a read stream — address translation or
# the memory side?  rocprofv3 --pmc passes over tools/placement_pmc_driver.py (12 fast then 12 slow launches of the probe kernel).
#   tools/placement_pmc.sh <out-file>
set -u
export TMPDIR=/tmp
OUT=${1:-gpurun_out/placement_pmc.txt}
: > $OUT
D=gpurun_out/_pmc_place
i=0
for cnt in "GRBM_GUI_ACTIVE" \
           "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum" \
           "TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum TCP_UTCL1_TRANSLATION_MISS_UNDER_MISS_sum TCP_UTCL1_STALL_INFLIGHT_MAX_sum" \
           "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum" \
           "TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_STALL_sum" \
           "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum" \
           "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_LEVEL_sum" \
           "TCC_TAG_STALL_sum TCC_BUBBLE_sum TCC_BUSY_sum"; do
  i=$((i + 1))
  d=$D/p$i
  mkdir -p $d
  rocprofv3 --pmc $cnt --output-format csv -d $d -- python3 tools/placement_pmc_driver.py > $d.log 2>&1 || tail -3 $d.log
  grep -h "x at" $d.log >> $OUT
done
python3 - "$D" >> $OUT <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: {'fast': [], 'slow': []})
for path in glob.glob(sys.argv[1] + '/**/*_counter_collection.csv', recursive=True):
    rows = [r for r in csv.DictReader(open(path)) if 'probe_add_kernel' in r['Kernel_Name']]
    by = collections.defaultdict(list)
    for r in rows:
        by[r['Counter_Name']].append((int(r['Dispatch_Id']), float(r['Counter_Value'])))
    for name, vals in by.items():
        vals.sort()
        if len(vals) != 24:
            print('#', name, 'unexpected dispatch count', len(vals)); continue
        acc[name]['fast'] += [v for _, v in vals[2:12]]      # the first two launches of each phase are warm-up
        acc[name]['slow'] += [v for _, v in vals[14:24]]
print('# counter: mean per launch over 10 launches, FAST placement | SLOW placement | ratio')
for name in sorted(acc):
    f = sum(acc[name]['fast']) / max(len(acc[name]['fast']), 1); s = sum(acc[name]['slow']) / max(len(acc[name]['slow']), 1)
    print(f'{name:48s} {f:18.1f} {s:18.1f}   {s / f if f else float("nan"):6.3f}')
PY
rm -rf $D
cat $OUT
