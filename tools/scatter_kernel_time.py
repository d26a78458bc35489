#!/usr/bin/env python3
"""Kernel-only timing of vox_scatter_reduce / vox_scatter_backward through the C ABI (no autograd)."""
import ctypes, os, sys
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
# cloud 'random': every point in a uniformly random pillar (no relation between memory order and space: the worst case for a
# gather).  cloud 'sweep': the same pillars, but the points arrive as a sensor delivers them — sorted by pillar, then shuffled
# inside windows of 2048 points (a LiDAR sweep stores neighbouring returns next to each other; deskewing and multi-sweep
# concatenation shuffle them locally).
CASES = [(2_000_000, c, (432, 496, 1), 'random') for c in (64, 32, 16, 10, 9, 3)] + \
        [(2_000_000, c, (432, 496, 1), 'sweep') for c in (64, 10, 9, 3)] + \
        [(120_000, 64, (432, 496, 1), 'random'), (120_000, 10, (432, 496, 1), 'random')]
if len(sys.argv) > 1:
    CASES = [(2_000_000, int(a), (432, 496, 1), 'random') for a in sys.argv[1:]]
for n, c, grid, cloud in CASES:
    g = torch.Generator(device=dev).manual_seed(0)
    coors = torch.stack([torch.randint(0, s, (n,), generator=g, device=dev) for s in grid], -1).int()
    if cloud == 'sweep':
        key = (coors[:, 0].long() * grid[1] + coors[:, 1].long()) * grid[2] + coors[:, 2].long()
        coors = coors[torch.sort(key)[1]]
        win = 2048
        m = n // win * win
        perm = torch.argsort(torch.rand(m // win, win, generator=g, device=dev), dim=1) + torch.arange(0, m, win, device=dev)[:, None]
        coors[:m] = coors[perm.reshape(-1)]
    feats = torch.randn(n, c, generator=g, device=dev)
    sc = Scatter(coors); v = sc.voxel_coors.shape[0]
    order, seg = group_points(sc.pts_voxel_maps, sc.voxel_pts_counts)
    out = torch.empty(v, c, device=dev); arg = torch.empty(v, c, dtype=torch.int32, device=dev)
    gv = torch.randn(v, c, device=dev); gf = torch.empty(n, c, device=dev)
    for red, name in ((2, 'max'), (1, 'mean'), (0, 'sum')):
        f = t(lambda: lib.vox_scatter_reduce(feats.data_ptr(), order.data_ptr(), seg.data_ptr(), n, c, v, red, out.data_ptr(), arg.data_ptr() if red == 2 else None, None))
        b = t(lambda: lib.vox_scatter_backward(gv.data_ptr(), sc.pts_voxel_maps.data_ptr(), sc.voxel_pts_counts.data_ptr(), arg.data_ptr() if red == 2 else None, n, c, v, red, gf.data_ptr(), None))
        b2 = t(lambda: lib.vox_scatter_backward_grouped(gv.data_ptr(), order.data_ptr(), seg.data_ptr(), arg.data_ptr() if red == 2 else None, n, c, v, red, gf.data_ptr(), None)) if (c <= 128 or c % 4 == 0) else float('nan')
        fb = n * c * 4 + n * 4 + v * c * 4 * (2 if red == 2 else 1)
        bb = (n * c * 4 + v * c * 4 * (2 if red == 2 else 1)) if red == 2 else (n * c * 4 + n * 4 + v * c * 4)
        print(f'n={n} c={c} v={v} {cloud} {name:4s}: fwd {f:7.1f} us ({fb / f / 1e3:6.0f} GB/s)  bwd(map order) {b:7.1f} us ({bb / b / 1e3:6.0f} GB/s)  bwd(voxel order) {b2:7.1f} us ({bb / b2 / 1e3:6.0f} GB/s)', flush=True)
    # ceiling probes on the same data: a plain random row gather (feats[order], read N*C*4 + write N*C*4) and a
    # streaming copy of the same bytes — what the memory system gives a segmented reduce over randomly placed rows
    gi = t(lambda: torch.index_select(feats, 0, order.long()))
    cp = t(lambda: gf.copy_(feats))
    print(f'n={n} c={c} {cloud}: torch index_select(feats, order) {gi:7.1f} us ({2 * n * c * 4 / gi / 1e3:6.0f} GB/s r+w)   copy {cp:7.1f} us ({2 * n * c * 4 / cp / 1e3:6.0f} GB/s r+w)', flush=True)
