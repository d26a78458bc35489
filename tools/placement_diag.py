#!/usr/bin/env python3
"""Is the 'placement' effect a property of single buffers or of buffer combinations?  10 x 280 MB torch allocations;
(a) z = x + x in place on ONE buffer (1 read + 1 write stream on the same pages), (b) z = x + y over triples."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import mmdet3d_gaussian_amd as amd
from mmdet3d_gaussian_amd import gd_loss as gdl
lib = amd.load_library()
dev = torch.device('cuda:0')
nf = 70_000_000
bufs = [torch.zeros(nf, device=dev) for _ in range(10)]
stream = torch.cuda.current_stream().cuda_stream
def t(x, y, z, it=14):
    tms = []
    for _ in range(it):
        tm = gdl.DispatchTimer()
        assert lib.gd3d_probe_stream(x.data_ptr(), y.data_ptr(), z.data_ptr(), nf, stream, tm.start, tm.stop) == 0
        tms.append(tm)
    torch.cuda.synchronize()
    d = sorted(v.elapsed_ms() for v in tms)[2:]
    return sum(d) / len(d) * 1e3
print('addresses:', [hex(b.data_ptr()) for b in bufs])
self_t = [t(b, b, b) for b in bufs]
print('in-place z=x+x per buffer (us):', [round(v, 1) for v in self_t])
for (i, j, k) in ((0, 1, 2), (3, 4, 5), (6, 7, 8), (0, 4, 8), (9, 5, 1), (2, 3, 7)):
    print(f'triple ({i},{j},{k}): {t(bufs[i], bufs[j], bufs[k]):.1f} us   [self: {self_t[i]:.1f} {self_t[j]:.1f} {self_t[k]:.1f}]')
# one 840 MB allocation carved into three
big = torch.zeros(3 * nf, device=dev)
print('carved from one allocation:', round(t(big[:nf], big[nf:2 * nf], big[2 * nf:]), 1), 'us')
