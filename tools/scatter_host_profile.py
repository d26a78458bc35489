#!/usr/bin/env python3
"""Where the host time of a small scatter-reduce fwd+bwd goes (cProfile, n = 120 K points)."""
import cProfile, pstats, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import mmdet3d_gaussian_amd as amd
from mmdet3d_gaussian_amd.scatter import Scatter
dev = torch.device('cuda:0')
g = torch.Generator(device=dev).manual_seed(0)
n, c = 120_000, 64
coors = torch.stack([torch.randint(0, s, (n,), generator=g, device=dev) for s in (432, 496, 1)], -1).int()
f = torch.randn(n, c, generator=g, device=dev).requires_grad_(True)
sc = Scatter(coors)
out, _ = sc.reduce(f, 'sum'); gv = torch.randn_like(out)
def step():
    f.grad = None
    o, _ = sc.reduce(f, 'sum'); o.backward(gv)
for _ in range(50): step()
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(2000): step()
torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats('cumulative').print_stats(18)
