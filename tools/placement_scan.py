#!/usr/bin/env python3
"""Does the RELATIVE placement of the three streams of the fused kernel's access mix (two read, one written, 280 MB each) inside
ONE allocation decide the 124-136 us spread of DESIGN.md §5.3?  x at offset 0 of a 4 GiB arena, y and z at swept offsets;
gd3d_probe_stream (z = x + y, nontemporal 16-byte accesses) timed by dispatch-bound events, median of 15 launches per point.
Usage (GPU box): python tools/placement_scan.py > gpurun_out/placement_scan.txt"""
import ctypes, os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mmdet3d_gaussian_amd as amd
from mmdet3d_gaussian_amd.gd_loss import DispatchTimer

lib = amd.load_library()
N = 70_000_000                       # floats per stream: the (10 M, 7) fp32 arrays of the bench
BYTES = N * 4
MB = 1 << 20
arena = torch.empty(4096 * MB, dtype=torch.uint8, device='cuda')
base = arena.data_ptr()
assert base % (2 * MB) == 0, hex(base)
torch.cuda.synchronize()
stream = torch.cuda.current_stream().cuda_stream
tm = DispatchTimer()


def run(ox, oy, oz, reps=15):
    ts = []
    for _ in range(reps):
        rc = lib.gd3d_probe_stream(base + ox, base + oy, base + oz, N, stream, tm.start, tm.stop)
        assert rc == 0, rc
        torch.cuda.synchronize()
        ts.append(tm.elapsed_ms() * 1e3)
    return statistics.median(ts), min(ts)


arena.zero_()
for _ in range(30):
    run(0, 512 * MB, 1024 * MB, 1)
print(f'arena base {base:#x}; stream bytes {BYTES} ({BYTES / MB:.1f} MiB)')
print('# sweep z: x at 0, y at 512 MiB, z at 1024 MiB + k * 2 MiB')
row = []
for k in range(0, 128):
    med, mn = run(0, 512 * MB, (1024 + 2 * k) * MB)
    row.append(med)
    print(f'k={k:3d} z_off={(1024 + 2 * k):5d} MiB  median {med:6.1f}  min {mn:6.1f}', flush=True)
print(f'# z sweep: min {min(row):.1f} max {max(row):.1f}')
print('# sweep y: x at 0, y at 288 MiB + k * 8 MiB, z at 2048 MiB')
row = []
for k in range(0, 64):
    med, mn = run(0, (288 + 8 * k) * MB, 2048 * MB)
    row.append(med)
    print(f'k={k:3d} y_off={(288 + 8 * k):5d} MiB  median {med:6.1f}  min {mn:6.1f}', flush=True)
print(f'# y sweep: min {min(row):.1f} max {max(row):.1f}')
print('# coarse grid: y_off x z_off in steps of 96 MiB (non-overlapping placements only)')
for oy in range(288, 2000, 192):
    line = []
    for oz in range(oy + 288, 3800, 192):
        med, _ = run(0, oy * MB, oz * MB, 7)
        line.append(f'{med:6.1f}')
    print(f'y={oy:5d}: ' + ' '.join(line), flush=True)
# fresh separate allocations for comparison (what torch gives a caller)
bufs = [torch.empty(N, dtype=torch.float32, device='cuda') for _ in range(9)]
for b in bufs:
    b.zero_()
print('# separate torch allocations:', [hex(b.data_ptr()) for b in bufs])
import itertools
for tri in [(0, 1, 2), (3, 4, 5), (6, 7, 8), (0, 4, 8), (1, 5, 6), (2, 3, 7)]:
    ts = []
    for _ in range(15):
        lib.gd3d_probe_stream(bufs[tri[0]].data_ptr(), bufs[tri[1]].data_ptr(), bufs[tri[2]].data_ptr(), N, stream, tm.start, tm.stop)
        torch.cuda.synchronize()
        ts.append(tm.elapsed_ms() * 1e3)
    print(f'triple {tri}: median {statistics.median(ts):6.1f}')

# ---- part 2: is it the SIZE / ALIGNMENT of the individual allocations?  Nine buffers per allocation size, eight triples each;
# then the same triples as 272-MiB slices of one arena
del arena, bufs
torch.cuda.empty_cache()
TRIPLES = [(0, 1, 2), (3, 4, 5), (6, 7, 8), (0, 4, 8), (1, 5, 6), (2, 3, 7), (8, 0, 3), (5, 2, 6)]


def probe(px, py, pz, reps=15):  # noqa: E302
    ts = []
    for _ in range(reps):
        assert lib.gd3d_probe_stream(px, py, pz, N, stream, tm.start, tm.stop) == 0
        torch.cuda.synchronize()
        ts.append(tm.elapsed_ms() * 1e3)
    return statistics.median(ts)


def align_of(p):
    a = 0
    while p % (1 << (a + 1)) == 0 and a < 40:
        a += 1
    return a


warm = torch.empty(3 * N, dtype=torch.float32, device='cuda').zero_()
for _ in range(30):
    probe(warm.data_ptr(), warm.data_ptr() + 4 * N, warm.data_ptr() + 8 * N, 1)
del warm
for size_mb in (0, 272, 320, 512, 1024, 2048):
    nbytes = N * 4 if size_mb == 0 else size_mb * MB
    bufs = [torch.empty(nbytes, dtype=torch.uint8, device='cuda') for _ in range(9)]
    for b in bufs:
        b.zero_()
    ptrs = [b.data_ptr() for b in bufs]
    res = [probe(ptrs[a], ptrs[b], ptrs[c]) for a, b, c in TRIPLES]
    print(f'separate allocations of {nbytes / MB:7.1f} MiB: VA alignment 2^{[align_of(p) for p in ptrs]}  spacing {[(ptrs[i] - ptrs[i + 1]) / MB for i in range(3)]} MiB')
    print('   triples us: ' + ' '.join(f'{r:6.1f}' for r in res) + f'   mean {statistics.mean(res):6.1f}  max {max(res):6.1f}', flush=True)
    del bufs
    torch.cuda.empty_cache()
arena = torch.empty(9 * 272 * MB, dtype=torch.uint8, device='cuda').zero_()
ptrs = [arena.data_ptr() + i * 272 * MB for i in range(9)]
res = [probe(ptrs[a], ptrs[b], ptrs[c]) for a, b, c in TRIPLES]
print(f'one arena of {9 * 272} MiB, nine slices of 272 MiB: VA alignment 2^{align_of(ptrs[0])}')
print('   triples us: ' + ' '.join(f'{r:6.1f}' for r in res) + f'   mean {statistics.mean(res):6.1f}  max {max(res):6.1f}')
