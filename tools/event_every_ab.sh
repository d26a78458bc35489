#!/bin/bash
# Run ON THE GPU BOX: what binding HIP event pairs to the fused dispatches costs the timed step (bench.py --event-every N).
OUT=${1:-gpurun_out/event_every.txt}
: > $OUT
for e in 1 5 0 5 1 0; do
  python3 bench.py --gpus 1 --steps 20 --warmup 5 --cpu-sample 0 --no-traffic --event-every $e 2>/dev/null > /tmp/_ee.json
  python3 - $e >> $OUT <<'PY'
import json, sys
d = json.load(open('/tmp/_ee.json')); r = d['roofline']; k = r['kernel_ms']
print('event-every', sys.argv[1], 'value', d['value'], 'ms/step', d['ms_per_step'], 'sum of the 3 fused kernels', round(sum(k.values()), 4),
      'step minus kernels (us)', round((d['ms_per_step'] - sum(k.values())) * 1e3, 1), 'frac', r['frac'], 'frac_step', r['frac_step'],
      'host enqueue ms/step', d['config']['host_enqueue_ms_per_step'])
PY
done
cat $OUT
