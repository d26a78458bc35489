"""Error vs the reference's fp64 result on the golden families: the reference's own fp32 (ref32) vs this kernel (hip)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch
import mmdet3d_gaussian_amd as amd
from gd_golden import index, pairs
g = pairs()
print('case            family | loss max relerr: ref32 / hip | grad_pred max row-scaled err: ref32 / hip')
for case in ('gwd3d.0', 'gwd3d.1', 'kld3d.0', 'bd3d.0', 'jd3d.0', 'kfiou3d.0'):
    c = index()['pairs']['cases'][case]
    kw = {k: (tuple(v) if isinstance(v, list) else v) for k, v in c['kwargs'].items()}
    for fam in ('kitti', 'near', 'delta', 'large'):
        p = torch.from_numpy(g[f'in.{fam}.pred']).cuda().requires_grad_(True)
        t = torch.from_numpy(g[f'in.{fam}.target']).cuda()
        out = amd.GDLoss(c['loss_type'], reduction='none', **kw)(p, t); out.sum().backward()
        key = f'{case}.{fam}'
        l64, l32, g64, g32 = g[key + '.loss64'], g[key + '.loss32'], g[key + '.gp64'], g[key + '.gp32']
        sc = 1 + np.abs(g64).max(-1, keepdims=True)
        with np.errstate(all='ignore'):
            e = lambda a: np.nanmax(np.abs(a - l64) / (1 + np.abs(l64)))
            eg = lambda a: np.nanmax(np.abs(a - g64) / sc)
            print(f'{case:15s} {fam:6s} | {e(l32):.1e} / {e(out.detach().cpu().numpy()):.1e} | {eg(g32):.1e} / {eg(p.grad.cpu().numpy()):.1e}')
