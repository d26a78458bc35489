#!/usr/bin/env python3
"""Phase times inside the selection kernel (select_kernel only: for maps above 131072 cells the threshold sample and the filtering
pass are launches of their own and "filter pass" below is the thinning of their candidate list)
Phase times inside the selection kernel of the CenterPoint inference slice (center_infer_debug_clocks): per map shape the
us spent in threshold sample / filtering pass / exact radix select (when it ran) / ordering / gather + decode, and the number
of candidates the pass left.  usage: tools/center_infer_phases.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import mmdet3d_gaussian_amd as amd  # noqa: E402

dev = torch.device('cuda:0')
lib = amd.load_library()
names = ('sample', 'filter pass', 'radix select', 'order', 'gather/decode')
# a fresh box idles at a fraction of its clocks: keep the GPU busy for a second first, and between the measured launches
import time
_w = torch.randn(4096, 4096, device=dev)
_t = time.perf_counter()
while time.perf_counter() - _t < 1.5:
    for _ in range(20):
        _w @ _w
    torch.cuda.synchronize()
def probe(blocks, reps=5):
    out = torch.zeros(2, dtype=torch.int64, device=dev)
    vals = []
    for _ in range(reps):
        lib.center_infer_debug_clock_probe(out.data_ptr(), blocks, 100000, torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        vals.append(int(out[0]))
    return min(vals) * 10.0 / 100000      # ns per dependent FMA


print(f'clock probe, ns per dependent FMA: one workgroup after the busy loop {probe(1):.2f}, 4096 workgroups {probe(4096):.2f}, '
      f'one workgroup again {probe(1):.2f}', flush=True)
time.sleep(0.5)
print(f'  after 0.5 s of idling: one workgroup {probe(1, 1):.2f}, again {probe(1, 1):.2f}', flush=True)
for shape, K, kind in (((4, 2, 128, 128), 500, 'rand'), ((1, 1, 128, 128), 500, 'rand'), ((4, 2, 128, 128), 500, 'peaky'),
                       ((2, 2, 180, 180), 500, 'rand'), ((1, 3, 468, 468), 4096, 'rand'), ((1, 2, 128, 128), 500, 'const')):
    g = torch.Generator().manual_seed(1)
    if kind == 'rand':
        heat = torch.rand(shape, generator=g)
    elif kind == 'const':
        heat = torch.zeros(shape)
    else:   # a trained head: almost everything far below, a few hundred blobs
        heat = torch.randn(shape, generator=g) * 0.3 - 6.0
        flat = heat.view(shape[0], -1)
        idx = torch.randint(0, flat.shape[1], (shape[0], 300), generator=g)
        flat.scatter_(1, idx, torch.randn(shape[0], 300, generator=g) * 2.0)
        heat = heat.sigmoid()
    heat = heat.to(dev)
    pred = torch.randn(shape[0], 11, shape[2], shape[3], generator=g).to(dev)
    clocks = torch.zeros(shape[0], 8, dtype=torch.int64, device=dev)
    lib.center_infer_debug_clocks(clocks.data_ptr())
    for _ in range(200):
        amd.select_best(heat, pred, K)
    _w @ _w
    amd.select_best(heat, pred, K)       # the stamps of the last launch stay in the buffer
    torch.cuda.synchronize()
    lib.center_infer_debug_clocks(None)
    c = clocks.cpu().double()
    d = (c[:, 1:6] - c[:, 0:5]) / 100.0          # 100 MHz -> us
    row = '  '.join(f'{n} {float(d[:, i].mean()):7.1f}' for i, n in enumerate(names))
    row += f'  (sample load {float((c[:, 7] - c[:, 0]).mean() / 100.0):5.1f})'
    print(f'{str(shape):20s} K={K:<5d} {kind:6s} {row}   total {float((c[:, 5] - c[:, 0]).mean() / 100.0):7.1f} us   candidates {c[:, 6].long().tolist()}')
