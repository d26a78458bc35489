#!/usr/bin/env python3
"""Golden vectors for NON-FINITE and degenerate rows, FROM THE REAL REFERENCE (build container only):

    python3 -B tests/golden/make_golden_nonfinite.py     ->  tests/golden/gd_nonfinite.npz

24 KITTI-range pairs; rows 0-6: NaN in column k of pred; rows 7-13: +inf in column k; then zero / negative / huge /
tiny dims, a yaw of 1e6, a centre at 1e20, -inf in a dim, and one identical pair.  For the 7 loss types (log1p, tau 1;
kfiou3d: fun none) the reference's per-pair loss in fp32 and fp64 is stored: values where finite, and the NaN / inf
pattern; and, per pair, whether its gradient row (wrt pred / wrt target) contains a NaN.  Only data is written."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from _ref_loader import load_reference_loss  # noqa: E402

CASES = (('gwd3d', dict(fun='log1p', tau=1.0)), ('kld3d', dict(fun='log1p', tau=1.0)), ('bd3d', dict(fun='log1p', tau=1.0)),
         ('jd3d', dict(fun='log1p', tau=1.0)), ('kld3d_symmax', dict(fun='log1p', tau=1.0)),
         ('kld3d_symmin', dict(fun='log1p', tau=1.0)), ('kfiou3d', dict(fun='none')))


def inputs():
    g = torch.Generator().manual_seed(7)
    n = 24
    r = lambda: torch.rand(n, generator=g)
    t = torch.stack([r() * 70, r() * 80 - 40, r() * 4 - 3, r() * 2 + 0.5, r() * 4 + 0.5, r() * 1.5 + 0.5, r() * 6.28 - 3.14], -1)
    p = t + torch.randn(n, 7, generator=g) * torch.tensor([0.3, 0.3, 0.1, 0.1, 0.1, 0.1, 0.1])
    nan, inf = float('nan'), float('inf')
    for k in range(7):
        p[k, k] = nan
        p[7 + k, k] = inf
    p[14, 3] = 0.0
    p[15, 3:6] = 0.0
    p[16, 4] = -1.0
    t[17, 3] = 0.0
    p[18, 6] = 1e6
    p[19, 0] = 1e20
    p[20, 3] = 1e20
    t[21, 5] = 1e-30
    p[22, 3] = -inf
    p[23] = t[23]
    return p.float().numpy(), t.float().numpy()


def main():
    ref = load_reference_loss()
    p, t = inputs()
    out = {'pred': p, 'target': t}
    with np.errstate(all='ignore'):
        for lt, kw in CASES:
            m = ref.GDLoss(lt, reduction='none', loss_weight=1.0, **kw)
            for dt, tag in ((torch.float32, '32'), (torch.float64, '64')):
                pp = torch.from_numpy(p).to(dt).requires_grad_(True)
                tt = torch.from_numpy(t).to(dt).requires_grad_(True)
                loss = m(pp, tt)
                loss.sum().backward()
                out[f'{lt}.loss{tag}'] = loss.detach().numpy()
                # which gradient ROWS contain a NaN (the element pattern inside a row is an artefact of the autograd graph:
                # clamp masks turn the upstream NaN into 0, later multiplications by NaN operands turn some 0s back)
                out[f'{lt}.gp_nanrow{tag}'] = np.isnan(pp.grad.numpy()).any(1)
                out[f'{lt}.gt_nanrow{tag}'] = np.isnan(tt.grad.numpy()).any(1)
    np.savez_compressed(os.path.join(HERE, 'gd_nonfinite.npz'), **out)
    for lt, _ in CASES:
        print(lt, ''.join('N' if np.isnan(x) else ('I' if np.isinf(x) else '.') for x in out[f'{lt}.loss32']))


if __name__ == '__main__':
    main()
