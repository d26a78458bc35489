#!/bin/bash
# Run ON THE GPU BOX, with profiles/r06_acc_experiment.patch APPLIED (the experiment was removed from the product; record:
# profiles/r06_acc_ab.txt): A/B of the loss sum's second stage on one box, processes alternated —
#   all    GD3D_ACC=all   (round 6): exact integer accumulation in the fused kernel (512 shards) + reduce_slots_kernel
#   small  GD3D_ACC=small (round 5): per-tile partial stores + the 1024-thread reduce_partials_kernel
#   sNNN   the accumulator form built with NNN shards (tools/build_variants.py sNNN="-DGD_ACC_SHARDS=NNN")
# every form under rocprofv3 --kernel-trace --stats (durations of the fused kernels and of the reduce kernels without event
# overhead), then the launch-floor probe.
set -u
export TMPDIR=/tmp
OUT=gpurun_out/${1:-acc_ab}
mkdir -p $OUT
B="python3 bench.py --gpus 1 --steps 20 --warmup 5 --cpu-sample 0 --no-traffic --no-standins"
FORMS="all small"
for v in tools/variants/libgd3d_s*.so; do [ -f $v ] && FORMS="$FORMS $(basename $v .so | sed s/libgd3d_//)"; done
run() {   # $1 form, rest: command
  local f=$1; shift
  case $f in
    all|small) GD3D_ACC=$f "$@" ;;
    *) GD3D_LIB=tools/variants/libgd3d_$f.so GD3D_HOST=python GD3D_ACC=all "$@" ;;
  esac
}
for i in 1 2; do
  for f in $FORMS; do
    run $f $B > $OUT/bench_${f}_$i.json 2>> $OUT/bench.err
    python3 - $OUT/bench_${f}_$i.json $f $i <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
r = d['roofline']
print(sys.argv[2], sys.argv[3], 'value', d['value'], 'ms/step', d['ms_per_step'], 'unit', d['value_unit_grad'], 'one_alloc', d.get('value_one_allocation'),
      'kernel_ms', r['kernel_ms'], 'frac', r['frac'], 'frac_step', r['frac_step'], flush=True)
PY
  done
done | tee $OUT/summary.txt
for f in $FORMS; do
  run $f rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_$f -- $B > $OUT/kt_$f.log 2>&1
  cp $OUT/kt_$f/*/*kernel_stats.csv $OUT/kernel_stats_$f.csv 2>/dev/null
  echo "== $f" >> $OUT/summary.txt
  grep -E "fused_kernel|reduce_|grad_finish" $OUT/kernel_stats_$f.csv | cut -c1-220 >> $OUT/summary.txt
  rm -rf $OUT/kt_$f
done
echo "== launch floor (tools/launch_floor.hip)" >> $OUT/summary.txt
[ -x tools/launch_floor ] && ./tools/launch_floor >> $OUT/summary.txt 2>&1
cat $OUT/summary.txt
