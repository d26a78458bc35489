#!/bin/bash
# Run ON THE GPU BOX: fused-kernel times of library variants (tools/build_variants.py) against the product library, processes
# alternated on one box, both input allocation forms in every process (bench.py's main region = one allocation per array, third
# region = row ranges of one allocation); golden tests first for every variant.
#   tools/variant_kernel_ab.sh out-dir variant ...
set -u
export TMPDIR=/tmp
OUT=gpurun_out/$1; shift
mkdir -p $OUT
B="python3 bench.py --gpus 1 --steps 20 --warmup 5 --cpu-sample 0 --no-traffic --no-standins"
for v in "$@"; do
  GD3D_LIB=tools/variants/libgd3d_$v.so GD3D_HOST=python python3 -m pytest tests/test_gpu_gd_loss.py -m gpu -x -q -k "golden or ragged or determinism or full_size or unaligned" > $OUT/pytest_$v.log 2>&1
  echo "$v: $(tail -1 $OUT/pytest_$v.log)"
done | tee $OUT/summary.txt
for i in 1 2 3; do
  for v in product "$@"; do
    if [ $v = product ]; then GD3D_HOST=python $B > $OUT/bench_${v}_$i.json 2>> $OUT/bench.err
    else GD3D_LIB=tools/variants/libgd3d_$v.so GD3D_HOST=python $B > $OUT/bench_${v}_$i.json 2>> $OUT/bench.err; fi
    python3 - $OUT/bench_${v}_$i.json $v $i <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); r = d['roofline']
print(f"{sys.argv[2]:12s} {sys.argv[3]} value {d['value']:9.1f} one_alloc {d['value_one_allocation']:9.1f} kernel_ms {r['kernel_ms']} ceiling_ms {r['copy_ceiling_ms']}", flush=True)
PY
  done
done | tee -a $OUT/summary.txt
