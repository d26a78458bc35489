#!/bin/bash
# Run ON THE GPU BOX: L2 / fabric counters of the scatter-reduce kernels per launch, one channel count at a time
# (rocprofv3 --pmc, separate passes; tools/scatter_pmc_driver.py).   tools/scatter_pmc.sh <out-file>
set -u
export TMPDIR=/tmp
OUT=${1:-gpurun_out/scatter_pmc.txt}
: > $OUT
for c in 64 10 3; do
  D=gpurun_out/_pmc_sc_$c
  for cnt in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCP_TCC_READ_REQ_sum TCC_REQ_sum"; do
    d=$D/$(echo $cnt | tr ' ' '_')
    mkdir -p $d
    rocprofv3 --pmc $cnt --output-format csv -d $d -- python3 tools/scatter_pmc_driver.py $c > $d.log 2>&1 || tail -3 $d.log
  done
  echo "## c = $c: n = 2 000 000 uniformly random points -> 214 251 voxels, reduce = sum; mean per launch over 3 launches" >> $OUT
  echo "## (FETCH_SIZE / WRITE_SIZE in KiB; FETCH_SIZE = TCC_EA0_RDREQ x 64 B: a 128-B line request is tallied at 64 B, MI355X_MICROARCH.md)" >> $OUT
  python3 tools/pmc_summary.py "$D/**/*_counter_collection.csv" --kernel vox:: >> $OUT 2>&1
  rm -rf $D
done
cat $OUT
