#!/bin/bash
# A/B experimental builds INSIDE bench.py under rocprofv3 (authoritative per-kernel averages).
#   tools/ab_rocprof.sh t256 t512 ...   (variants built by tools/build_variants.py; two interleaved passes)
export TMPDIR=/tmp
for pass in 1 2; do for v in "$@"; do
  GD3D_LIB=$PWD/tools/variants/libgd3d_$v.so rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ab_$v -- python3 bench.py --steps 30 --warmup 5 --cpu-sample 0 > gpurun_out/ab_$v.log 2>&1
  python3 - "$v" <<'PY'
import csv, glob, json, sys
v = sys.argv[1]
rows = list(csv.DictReader(open(glob.glob(f'gpurun_out/ab_{v}/*/*kernel_stats.csv')[0])))
k = {r['Name'].split('fused_kernel<')[1][0]: float(r['AverageNs']) / 1e3 for r in rows if 'fused_kernel<' in r['Name']}
small = {r['Name'].split('(')[0].split('::')[-1][:24]: round(float(r['AverageNs']) / 1e3, 1) for r in rows if 'reduce_partials' in r['Name'] or 'scale_rows' in r['Name']}
line = [l for l in open(f'gpurun_out/ab_{v}.log') if l.startswith('{"metric"')]
d = json.loads(line[-1]) if line else {}
print(f"{v:8s} rocprof fused us gwd {k.get('0', 0):6.1f} kld {k.get('1', 0):6.1f} bd {k.get('2', 0):6.1f} | {small} | bench value {d.get('value')} events {d.get('roofline', {}).get('kernel_ms')}")
PY
  rm -rf gpurun_out/ab_$v
done; done
