#!/bin/bash
# Run ON THE GPU BOX: issue mix of the fused kernels per WAVE (rocprofv3 --pmc, separate passes; bench.py --steps 3 eager).
#   tools/pmc_issue_mix.sh <out-file> [GD3D_LIB path]
set -u
export TMPDIR=/tmp
OUT=${1:-gpurun_out/pmc_issue_mix.txt}
[ -n "${2:-}" ] && export GD3D_LIB=$2
D=gpurun_out/_pmc_issue
rm -rf $D; mkdir -p $D
SHORT="python3 bench.py --steps 3 --warmup 2 --cpu-sample 0 --no-graph --prewarm 0 --no-traffic"
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VALU_TRANS_F32 --output-format csv -d $D/a -- $SHORT > $D/a.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_ANY --output-format csv -d $D/b -- $SHORT > $D/b.log 2>&1
python3 - $D $OUT <<'PY'
import collections, csv, glob, sys
d, out = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for path in glob.glob(f'{d}/*/*/*_counter_collection.csv'):
    for row in csv.DictReader(open(path)):
        if 'fused_kernel<' in row['Kernel_Name']:
            acc[row['Kernel_Name'].split('fused_kernel<')[1].split('>')[0]][row['Counter_Name']].append(float(row['Counter_Value']))
names = {'0': 'gwd3d', '1': 'kld3d', '2': 'bd3d'}
with open(out, 'w') as f:
    f.write('# per WAVE of the fused kernel (64 pairs), rocprofv3 --pmc over bench.py --steps 3 (10 M pairs); ACTIVE_* are in units of 4 cycles\n')
    for k, cs in sorted(acc.items()):
        m = {c: sum(v) / len(v) for c, v in cs.items()}
        w = m.get('SQ_WAVES', 1.0)
        f.write(f'{names.get(k.split(",")[0].strip(), k)} <{k}>: waves {w:.0f}\n   ' + '  '.join(f'{c[3:]} {m[c] / w:.1f}' for c in sorted(m) if c != 'SQ_WAVES') + '\n')
print(open(out).read())
PY
rm -rf $D
