#!/usr/bin/env python3
"""Driver for tools/placement_channels.sh: finds, among separately allocated 280 MB buffers of THIS process, the fastest and the
slowest (x, y) pair of read streams for the probe kernel z = x + y (the fused kernel's access mix; DESIGN.md 5.3: 122-135 us by
triple), then launches the probe 12 times on the fast pair and 12 times on the slow pair.  Under rocprofv3 --pmc the LAST 24
dispatches of gd3d::probe_add_kernel are those two phases (the parser takes the last 24)."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mmdet3d_gaussian_amd as amd
from mmdet3d_gaussian_amd.gd_loss import DispatchTimer
lib = amd.load_library()
N = 70_000_000
NB = int(os.environ.get('PLACEMENT_BUFFERS', '10'))
bufs = [torch.empty(N, dtype=torch.float32, device='cuda').zero_() for _ in range(NB)]
z = torch.empty(N, dtype=torch.float32, device='cuda').zero_()
torch.cuda.synchronize()
stream = torch.cuda.current_stream().cuda_stream
tm = DispatchTimer()
def run(x, y, reps):
    ts = []
    for _ in range(reps):
        assert lib.gd3d_probe_stream(x.data_ptr(), y.data_ptr(), z.data_ptr(), N, stream, tm.start, tm.stop) == 0
        torch.cuda.synchronize()
        ts.append(tm.elapsed_ms() * 1e3)
    return statistics.median(ts)
run(bufs[0], bufs[1], 10)   # clocks
res = {}
for i in range(NB):
    for j in range(NB):
        if i != j:
            res[(i, j)] = run(bufs[i], bufs[j], 3)
best = min(res, key=res.get); worst = max(res, key=res.get)
print(f'search over {len(res)} ordered pairs of {NB} separately allocated buffers: fastest {best} {res[best]:.1f} us, slowest {worst} {res[worst]:.1f} us, '
      f'median {statistics.median(res.values()):.1f} us', flush=True)
for name, (i, j) in (('fast', best), ('slow', worst)):
    print(name, 'x at', hex(bufs[i].data_ptr()), 'y at', hex(bufs[j].data_ptr()), ': median', round(run(bufs[i], bufs[j], 12), 1), 'us', flush=True)
