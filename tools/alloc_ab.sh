#!/bin/bash
# does torch's allocator configuration change the placement lottery?  one bench.py process per row
out=gpurun_out/r02j; mkdir -p $out
for round in 1 2; do for conf in default "expandable_segments:True"; do
  if [ "$conf" = default ]; then unset PYTORCH_HIP_ALLOC_CONF PYTORCH_CUDA_ALLOC_CONF; else export PYTORCH_HIP_ALLOC_CONF=$conf PYTORCH_CUDA_ALLOC_CONF=$conf; fi
  python bench.py --steps 20 --warmup 5 --cpu-sample 0 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); r=d['roofline']
print('$conf', d['value'], d['ms_per_step'], r['kernel_ms'], 'probe', r['copy_ceiling_ms'], 'frac', r['frac'])" | tee -a $out/alloc_ab.txt
done; done
