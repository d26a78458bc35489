#!/bin/bash
# Run ON THE GPU BOX: per-CHANNEL view of a fast and a slow placement of the fused kernel's first read stream:
# is the 7 % a channel-aliasing effect (requests piling up on few of the 16 L2 channels per XCD) that the kernel could avoid by
# issuing its two read streams in a different interleave?   rocprofv3 --pmc <per-instance TCC counters> with JSON output (the csv
# sums the instances), over tools/placement_channels_driver.py (it searches this process's separately allocated buffers for the fastest and the slowest
# pair of read streams, then 12 launches on each).
#   tools/placement_channels.sh <out-file>
set -u
export TMPDIR=/tmp
OUT=${1:-gpurun_out/placement_channels.txt}
: > $OUT
D=gpurun_out/_pmc_chan
rm -rf $D; mkdir -p $D
i=0
# per-instance slices are derived counters (tools/placement_channels.yaml: reduce(select(TCC_EA0_RDREQ,[DIMENSION_INSTANCE=[k]]),sum) ...)
for short in RD RDLVL TAGST; do
  for dim in CH XCC; do
    if [ $dim = CH ]; then names=$(for k in $(seq 0 15); do echo -n "CH${k}_$short "; done); else names=$(for k in $(seq 0 7); do echo -n "XCC${k}_$short "; done); fi
    i=$((i + 1))
    rocprofv3 -E tools/placement_channels.yaml --pmc $names --output-format csv -d $D/p$i -- python3 tools/placement_channels_driver.py > $D/p$i.log 2>&1 || tail -5 $D/p$i.log >> $OUT
    grep -h "x at\|search over" $D/p$i.log >> $OUT
  done
done
python3 tools/placement_channels.py $D >> $OUT 2>&1
rm -rf $D
cat $OUT
