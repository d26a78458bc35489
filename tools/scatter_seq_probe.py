#!/usr/bin/env python3
"""Is the scatter-reduce forward limited by the random row gather or by the kernel?  Same segments, rows physically
pre-sorted (order = identity) vs the real random order."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import mmdet3d_gaussian_amd as amd
from mmdet3d_gaussian_amd.scatter import Scatter, group_points
lib = amd.load_library()
dev = torch.device('cuda:0')
def t(fn, it=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it * 1e3
n, c, grid = 2_000_000, 64, (432, 496, 1)
g = torch.Generator(device=dev).manual_seed(0)
coors = torch.stack([torch.randint(0, s, (n,), generator=g, device=dev) for s in grid], -1).int()
feats = torch.randn(n, c, generator=g, device=dev)
sc = Scatter(coors); v = sc.voxel_coors.shape[0]
order, seg = group_points(sc.pts_voxel_maps, sc.voxel_pts_counts)
out = torch.empty(v, c, device=dev); arg = torch.empty(v, c, dtype=torch.int32, device=dev)
sorted_feats = feats[order.long()].contiguous()
ident = torch.arange(n, device=dev, dtype=order.dtype)
for red, name in ((2, 'max'), (0, 'sum')):
    a = t(lambda: lib.vox_scatter_reduce(feats.data_ptr(), order.data_ptr(), seg.data_ptr(), n, c, v, red, out.data_ptr(), arg.data_ptr() if red == 2 else None, None))
    b = t(lambda: lib.vox_scatter_reduce(sorted_feats.data_ptr(), ident.data_ptr(), seg.data_ptr(), n, c, v, red, out.data_ptr(), arg.data_ptr() if red == 2 else None, None))
    print(f'{name}: random order {a:.1f} us, physically sorted rows {b:.1f} us', flush=True)
