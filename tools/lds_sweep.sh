#!/bin/bash
# occupancy-cap sweep of the fused kernel inside bench.py (run on the GPU box): per loss type, dynamic LDS bytes requested
# (160 KiB / bytes = workgroups per CU, at most 8)
out=gpurun_out/r02g; mkdir -p $out
for cfg in "0,0,0" "20480,20480,20480" "23400,23400,23400" "27300,27300,27300" "32768,27300,27300" "32768,27300,23400" "40960,27300,23400" "32768,23400,23400" "32768,32768,32768" "0,0,0"; do
  GD3D_MIN_LDS=$cfg python bench.py --steps 30 --warmup 5 --cpu-sample 0 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); r=d['roofline']
print('$cfg', d['value'], d['ms_per_step'], r['kernel_ms'], 'probe', r['copy_ceiling_ms'])" | tee -a $out/lds_sweep.txt
done
