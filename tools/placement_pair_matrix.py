"""tools/placement_pair_matrix.py — z = x + y (the fused kernel's access mix, 70 M floats per stream) over EVERY ordered pair of NB
separately allocated 280 MB buffers of one process: is the slow placement a property of single buffers or of pairs?  (round 6:
of PAIRS — the buffers fall into two groups, pairs inside a group are fast, pairs across are slow: profiles/r06_placement_channels.txt)
  PAIR_BUFFERS=n  PAIR_BALLAST_GB=g (a g-GiB allocation made first and kept)"""
import os, sys, statistics
sys.path.insert(0, '/root/repo')
import torch
import mmdet3d_gaussian_amd as amd
from mmdet3d_gaussian_amd.gd_loss import DispatchTimer
lib = amd.load_library()
N = 70_000_000; NB = int(os.environ.get('PAIR_BUFFERS', '10'))
ballast_gb = float(os.environ.get('PAIR_BALLAST_GB', '0'))   # allocated FIRST and kept: do the process's first allocations draw a class of their own?
ballast = torch.empty(int(ballast_gb * (1 << 30)), dtype=torch.uint8, device='cuda').zero_() if ballast_gb > 0 else None
bufs = [torch.empty(N, dtype=torch.float32, device='cuda').zero_() for _ in range(NB)]
z = torch.empty(N, dtype=torch.float32, device='cuda').zero_()
torch.cuda.synchronize()
stream = torch.cuda.current_stream().cuda_stream
tm = DispatchTimer()
def run(x, y, zz, reps=5):
    ts = []
    for _ in range(reps):
        assert lib.gd3d_probe_stream(x.data_ptr(), y.data_ptr(), zz.data_ptr(), N, stream, tm.start, tm.stop) == 0
        torch.cuda.synchronize(); ts.append(tm.elapsed_ms() * 1e3)
    return statistics.median(ts)
run(bufs[0], bufs[1], z, 20)
print('addresses', [hex(b.data_ptr()) for b in bufs], hex(z.data_ptr()))
print('       ' + ' '.join(f'y={j:<5d}' for j in range(NB)))
M = [[0.0] * NB for _ in range(NB)]
for i in range(NB):
    for j in range(NB):
        M[i][j] = run(bufs[i], bufs[j], z) if i != j else float('nan')
    print(f'x={i:<3d} ' + ' '.join(f'{M[i][j]:7.1f}' for j in range(NB)))
row = [statistics.mean(v for v in M[i] if v == v) for i in range(NB)]
col = [statistics.mean(M[i][j] for i in range(NB) if i != j) for j in range(NB)]
print('row means', [round(v, 1) for v in row]); print('col means', [round(v, 1) for v in col])
# additive model residual
g = statistics.mean(row)
res = [M[i][j] - (row[i] + col[j] - g) for i in range(NB) for j in range(NB) if i != j]
print('additive-model residual: max |r|', round(max(abs(r) for r in res), 2), 'rms', round((sum(r * r for r in res) / len(res)) ** 0.5, 2))
# single-stream reads: copy x -> z only (y = x)
print('x alone (y = x):', [round(run(b, b, z), 1) for b in bufs])
# does a SKEW between the two read streams help a slow pair?  thread i reads x[i] and y[i + S]: the same two buffers, the same
# bytes per stream (n - S floats), another relation between the addresses read at the same time
flat = [(M[i][j], i, j) for i in range(NB) for j in range(NB) if i != j]
slow, fast = max(flat), min(flat)
for name, (t0, i, j) in (('slow pair', slow), ('fast pair', fast)):
    line = []
    for S_mb in (0, 1, 7, 64, 128):
        S = S_mb * (1 << 20) // 4
        m = (N - S) & ~3
        ts = []
        for _ in range(5):
            assert lib.gd3d_probe_stream(bufs[i].data_ptr(), bufs[j].data_ptr() + 4 * S, z.data_ptr(), m, stream, tm.start, tm.stop) == 0
            torch.cuda.synchronize(); ts.append(tm.elapsed_ms() * 1e3)
        line.append(f'S={S_mb:3d} MiB: {statistics.median(ts) * N / m:6.1f}')
    print(f'{name} ({i},{j}) scaled to the full size, y skewed by S: ' + '  '.join(line))
