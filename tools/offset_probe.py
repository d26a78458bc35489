#!/usr/bin/env python3
"""Does the fused kernel's time depend on the RELATIVE placement of its three streams (pred read, target read, grad
write)?  10 M pairs, buffers carved at chosen byte offsets out of 2-MiB-aligned slabs; kernel-only timing (20 launches
between HIP events), three losses.  usage: offset_probe.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mmdet3d_gaussian_amd as amd
from mmdet3d_gaussian_amd import gd_loss as gdl
lib = amd.load_library()
dev = torch.device('cuda:0')
n = 10_000_000
nb = n * 7 * 4
SLAB = nb + (64 << 20)
g = torch.Generator(device=dev).manual_seed(0)
src_t = torch.rand(n, 7, generator=g, device=dev) * 2 + 0.5
src_p = src_t + torch.randn(n, 7, generator=g, device=dev) * 0.1
slabs = [torch.empty(SLAB, dtype=torch.uint8, device=dev) for _ in range(3)]
ws = torch.empty(lib.gd3d_loss_workspace_bytes(n), dtype=torch.uint8, device=dev)


def carve(slab, off):
    al = (-slab.data_ptr()) % (2 << 20)            # start from a 2 MiB boundary
    return slab[al + off: al + off + nb].view(torch.float32).view(n, 7)


def time_cfg(op, ot, og, lt):
    p, t, gbuf = carve(slabs[0], op), carve(slabs[1], ot), carve(slabs[2], og)
    p.copy_(src_p); t.copy_(src_t)
    prm = gdl.make_params(lt, 'log1p', 1.0, 1.0, (0, 0, 0.5), {})
    call = lambda: lib.gd3d_loss_fused(prm, p.data_ptr(), t.data_ptr(), None, n, 5.0 / n, None, None, gbuf.data_ptr(), None,
                                       ws.data_ptr(), None)
    for _ in range(5): call()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): assert call() == 0
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 20 * 1e3


K = 1 << 10
cfgs = [(0, 0, 0), (0, 256, 512), (0, K, 2 * K), (0, 4 * K, 8 * K), (0, 16 * K, 32 * K), (0, 64 * K, 128 * K), (0, 256 * K, 512 * K),
        (0, 1 << 20, 0), (0, (1 << 20) + 4 * K, 8 * K), (0, 7168, 14336), (0, 3 * 7168, 5 * 7168), (0, 4 * K, 0), (0, 0, 4 * K)]
print('slab bases mod 2 MiB:', [s.data_ptr() % (2 << 20) for s in slabs], 'physical placement is the driver\'s', flush=True)
for rep in range(2):
    for op, ot, og in cfgs:
        r = [time_cfg(op, ot, og, lt) for lt in ('gwd3d', 'kld3d', 'bd3d')]
        print(f'offsets pred {op:8d} target {ot:8d} grad {og:8d}: gwd {r[0]:6.1f} kld {r[1]:6.1f} bd {r[2]:6.1f} us', flush=True)
