#!/bin/bash
# same-box A/B of per-loss occupancy caps (workgroups per CU for gwd/kld/bd) inside bench.py, two rounds, interleaved
out=gpurun_out/r02i; mkdir -p $out
for round in 1 2; do for v in c888 c666 c667 c567 c668; do
  GD3D_LIB=tools/variants/libgd3d_$v.so python bench.py --steps 30 --warmup 5 --cpu-sample 0 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); r=d['roofline']
print('$v', d['value'], d['ms_per_step'], r['kernel_ms'], 'probe', r['copy_ceiling_ms'])" | tee -a $out/lds_ab2_$1.txt
done; done
