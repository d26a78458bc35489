#!/usr/bin/env python3
"""Can a sub-2-MiB shift of one stream repair a 'bad' buffer combination?  Separate 2-MiB-aligned torch allocations with
4 MiB of slack each; the worst triple of a few is re-timed with the second / third stream shifted by s bytes."""
import os, sys, itertools
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import mmdet3d_gaussian_amd as amd
from mmdet3d_gaussian_amd import gd_loss as gdl
lib = amd.load_library()
dev = torch.device('cuda:0')
nf = 70_000_000
slack = 2 * 1024 * 1024          # floats = 8 MiB
bufs = [torch.zeros(nf + slack, device=dev) for _ in range(9)]
stream = torch.cuda.current_stream().cuda_stream
def t(x, y, z, it=12):
    tms = []
    for _ in range(it):
        tm = gdl.DispatchTimer()
        assert lib.gd3d_probe_stream(x.data_ptr(), y.data_ptr(), z.data_ptr(), nf, stream, tm.start, tm.stop) == 0
        tms.append(tm)
    torch.cuda.synchronize()
    d = sorted(v.elapsed_ms() for v in tms)[2:]
    return sum(d) / len(d) * 1e3
v = lambda b, s: b[s // 4: s // 4 + nf]
trip = [(0, 1, 2), (3, 4, 5), (6, 7, 8), (0, 4, 8), (1, 5, 6), (2, 3, 7), (8, 0, 3), (5, 2, 6)]
base = {tr: t(v(bufs[tr[0]], 0), v(bufs[tr[1]], 0), v(bufs[tr[2]], 0)) for tr in trip}
print('unshifted triples:', {k: round(x, 1) for k, x in base.items()})
worst = max(base, key=base.get); best = min(base, key=base.get)
for name, tr in (('worst', worst), ('best', best)):
    a, b, c = (bufs[i] for i in tr)
    print(f'--- {name} triple {tr}: {base[tr]:.1f} us unshifted; rows = shift of stream y, columns = shift of stream z (bytes)')
    shifts = [0, 4096, 65536, 262144, 1048576, 2097152 + 4096, 4194304]
    print('        ' + ' '.join(f'{s:>9d}' for s in shifts))
    for sy in shifts:
        print(f'{sy:>8d}' + ' '.join(f'{t(v(a, 0), v(b, sy), v(c, sz), 8):9.1f}' for sz in shifts))
