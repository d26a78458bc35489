#!/usr/bin/env python3
"""One allocation or several?  The probe kernel (gd3d_probe_stream: z = x + y over 70 M floats, the fused kernel's access mix) on ten
triples of nine 280 MB buffers that are  A: nine torch allocations;  B: slices of a 4 GiB allocation made first;  C: the same after
6.2 GiB of other allocations;  D: after those were freed again;  E: slices of an exact-size allocation after 6.2 GiB of others.
Run every mode in a FRESH process (placement is drawn per process):  for m in A B C D E; do python tools/placement_recipe.py $m; done
-> profiles/r04_placement_scan.txt part 5."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mmdet3d_gaussian_amd as amd
from mmdet3d_gaussian_amd.gd_loss import DispatchTimer
mode = sys.argv[1]
lib = amd.load_library()
N = 70_000_000
MB = 1 << 20
stream = torch.cuda.current_stream().cuda_stream
tm = DispatchTimer()
TR = [(0, 1, 2), (3, 4, 5), (6, 7, 8), (0, 4, 8), (1, 5, 6), (2, 3, 7), (8, 0, 3), (5, 2, 6), (7, 6, 1), (4, 8, 2)]
def probe(px, py, pz, reps=11):
    ts = []
    for _ in range(reps):
        assert lib.gd3d_probe_stream(px, py, pz, N, stream, tm.start, tm.stop) == 0
        torch.cuda.synchronize(); ts.append(tm.elapsed_ms() * 1e3)
    return statistics.median(ts)
junk = None
if mode in ('C', 'E'):
    junk = [torch.empty(700 * MB, dtype=torch.uint8, device='cuda').zero_() for _ in range(9)]      # 6.2 GiB kept alive
if mode == 'D':
    junk = [torch.empty(700 * MB, dtype=torch.uint8, device='cuda').zero_() for _ in range(9)]
    del junk; torch.cuda.empty_cache()
if mode == 'A':
    bufs = [torch.empty(N, dtype=torch.float32, device='cuda').zero_() for _ in range(9)]
    ptrs = [b.data_ptr() for b in bufs]
elif mode == 'E':   # arena of the exact size needed (not a power of two), after junk
    arena = torch.empty(9 * 268 * MB, dtype=torch.uint8, device='cuda').zero_()
    ptrs = [arena.data_ptr() + i * 268 * MB for i in range(9)]
else:
    arena = torch.empty(4096 * MB, dtype=torch.uint8, device='cuda').zero_()
    ptrs = [arena.data_ptr() + i * 268 * MB for i in range(9)]
for _ in range(30): probe(ptrs[0], ptrs[1], ptrs[2], 1)
res = [probe(ptrs[a], ptrs[b], ptrs[c]) for a, b, c in TR]
print(mode, ' '.join(f'{r:6.1f}' for r in res), f'  mean {statistics.mean(res):6.1f} max {max(res):6.1f}', flush=True)
