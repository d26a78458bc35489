#!/usr/bin/env python3
"""A short driver for rocprofv3 --pmc passes over the scatter-reduce kernels: `scatter_pmc_driver.py C [C ...]` launches the
forward (sum) and both backward forms three times each at n = 2 M random points -> 214 K voxels for every channel count given,
nothing else.  tools/collect_profiles.sh summarises the counters into <round>_scatter_pmc.txt."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import mmdet3d_gaussian_amd as amd
from mmdet3d_gaussian_amd.scatter import Scatter, group_points

lib = amd.load_library()
dev = torch.device('cuda:0')
n = 2_000_000
for c in [int(a) for a in sys.argv[1:]] or [64, 10, 3]:
    g = torch.Generator(device=dev).manual_seed(0)
    coors = torch.stack([torch.randint(0, s, (n,), generator=g, device=dev) for s in (432, 496, 1)], -1).int()
    feats = torch.randn(n, c, generator=g, device=dev)
    sc = Scatter(coors); v = sc.voxel_coors.shape[0]
    order, seg = group_points(sc.pts_voxel_maps, sc.voxel_pts_counts)
    out = torch.empty(v, c, device=dev); gv = torch.randn(v, c, device=dev); gf = torch.empty(n, c, device=dev)
    for _ in range(3):
        lib.vox_scatter_reduce(feats.data_ptr(), order.data_ptr(), seg.data_ptr(), n, c, v, 0, out.data_ptr(), None, None)
        lib.vox_scatter_backward(gv.data_ptr(), sc.pts_voxel_maps.data_ptr(), sc.voxel_pts_counts.data_ptr(), None, n, c, v, 0, gf.data_ptr(), None)
        lib.vox_scatter_backward_grouped(gv.data_ptr(), order.data_ptr(), seg.data_ptr(), None, n, c, v, 0, gf.data_ptr(), None)
    torch.cuda.synchronize()
