#!/usr/bin/env python3
"""Dynamic scatter-reduce timing: N points x C channels -> V voxels, forward and backward, vs the
torch restatement (index_add / scatter_reduce(amax) + autograd) on the same GPU.  HBM-bound: N*C*4 B read forward."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from mmdet3d_gaussian_amd.scatter import Scatter
dev = torch.device('cuda:0')
def timeit(fn, it):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(it): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / it * 1e6
for n, c, grid in ((120_000, 64, (432, 496, 1)), (2_000_000, 64, (432, 496, 1)), (2_000_000, 10, (432, 496, 1))):
    g = torch.Generator(device=dev).manual_seed(0)
    coors = torch.stack([torch.randint(0, s, (n,), generator=g, device=dev) for s in grid], -1).int()
    feats = torch.randn(n, c, generator=g, device=dev).requires_grad_(True)
    sc = Scatter(coors)
    v = sc.voxel_coors.shape[0]
    idx = sc.pts_voxel_maps.long()
    for red in ('max', 'mean', 'sum'):
        def ours():
            feats.grad = None
            out, _ = sc.reduce(feats, red); out.sum().backward()
        def ref():
            feats.grad = None
            if red == 'max':
                out = torch.full((v, c), float('-inf'), device=dev).scatter_reduce(0, idx[:, None].expand(-1, c), feats, 'amax')
            else:
                out = torch.zeros(v, c, device=dev).index_add(0, idx, feats)
                if red == 'mean': out = out / sc.voxel_pts_counts[:, None]
            out.sum().backward()
        a, b = timeit(ours, 30), timeit(ref, 10)
        fwd_bytes = n * c * 4 + n * 4 + v * c * 4
        print(json.dumps(dict(n=n, c=c, v=v, reduce=red, fused_fwd_bwd_us=round(a, 1), torch_fwd_bwd_us=round(b, 1), speedup=round(b / a, 2))), flush=True)

# the index build itself: one vox_index_build call (round 3) vs the ~20 ATen launches it replaced, batched (b, z, y, x) rows
from mmdet3d_gaussian_amd.scatter import _index_and_grouping, _index_and_grouping_torch
for n, grid in ((120_000, (4, 1, 496, 432)), (960_000, (8, 1, 496, 432)), (2_000_000, (8, 40, 200, 176))):
    g = torch.Generator(device=dev).manual_seed(1)
    coors = torch.stack([torch.randint(0, s, (n,), generator=g, device=dev) for s in grid], -1).int()
    a, b = timeit(lambda: _index_and_grouping(coors), 30), timeit(lambda: _index_and_grouping_torch(coors), 10)
    x, y = _index_and_grouping(coors), _index_and_grouping_torch(coors)
    same = all(torch.equal(p, q) for p, q in zip(x[:3], y[:3])) and torch.equal(x[3][0], y[3][0]) and torch.equal(x[3][1], y[3][1])
    assert same, 'index build differs from its ATen statement'
    print(json.dumps(dict(what='scatter index build (incl. the one host read)', n=n, grid=list(grid), voxels=int(x[0].shape[0]),
                          vox_index_build_us=round(a, 1), aten_statement_us=round(b, 1), speedup=round(b / a, 2), identical=same)), flush=True)
