#!/usr/bin/env python3
"""Golden vectors for the GDAnchor3DHead decoded branch on EXTREME encoded rows (build container only):
    python3 -B tests/golden/make_golden_anchor_extreme.py  ->  tests/golden/anchor_extreme.npz

decode(anchors, pred_enc) / decode(anchors, target_enc) with the DeltaXYZWLHR formulas (mmdet3d's coder is third party and
absent: oracle/head_torch.py restates its published decode, in torch) followed by the REAL reference GDLoss, reduction
'none' (gd_anchor3d_head.py:133-141).  48 positives whose encodings leave the trained regime: size deltas of +-20, +89
(exp overflows), -104 (underflows to the 1e-7 clamp), NaN / inf entries, centre deltas of 1e4 and 1e30, yaw deltas of
1e4 — in pred, in target, or in both.  Stored: per-positive loss in fp32 / fp64 and whether the positive's gradient row wrt
the encoded prediction contains a NaN.  Only data is written."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
from _ref_loader import load_reference_loss  # noqa: E402
from oracle import head_torch  # noqa: E402

CASES = (('gwd3d', dict(fun='log1p', tau=1.0)), ('kld3d', dict(fun='log1p', tau=0.0)), ('bd3d', dict(fun='log1p', tau=1.0)),
         ('kld3d_symmin', dict(fun='none', tau=0.0)))


def inputs():
    rng = np.random.default_rng(31)
    P = 48
    anchors = np.stack([rng.uniform(0, 70, P), rng.uniform(-40, 40, P), rng.uniform(-2, 0, P), rng.uniform(.6, 2, P),
                        rng.uniform(.8, 4, P), rng.uniform(1.4, 1.8, P), rng.choice([0, np.pi / 2], P)], -1).astype(np.float32)
    t = rng.normal(0, 0.3, (P, 7)).astype(np.float32)
    p = (t + rng.normal(0, 0.1, (P, 7))).astype(np.float32)
    nan, inf = np.float32('nan'), np.float32('inf')
    vals = (20.0, -20.0, 89.0, -104.0, nan, inf, -inf, 44.0)
    for k, v in enumerate(vals):                 # rows 0..7: a size delta of pred
        p[k, 3 + k % 3] = v
    for k, v in enumerate(vals):                 # rows 8..15: the same in target
        t[8 + k, 3 + k % 3] = v
    p[16, 3:6] = 89.0; t[16, 3:6] = 89.0         # both overflow
    p[17, 3:6] = -104.0; t[17, 3:6] = -104.0     # both underflow
    p[18, 0] = 1e4; p[19, 1] = 1e30; p[20, 2] = 1e30; p[21, 0] = nan; p[22, 2] = inf
    t[23, 0] = 1e4; t[24, 1] = 1e30; t[25, 2] = nan
    p[26, 6] = 1e4; p[27, 6] = nan; p[28, 6] = inf; t[29, 6] = 1e4
    return anchors, p, t


def main():
    torch.set_num_threads(1)
    ref = load_reference_loss()
    anchors, p_np, t_np = inputs()
    out = {'anchors': anchors, 'pred': p_np, 'target': t_np}
    with np.errstate(all='ignore'):
        for lt, kw in CASES:
            for dtype, tag in ((torch.float32, '32'), (torch.float64, '64')):
                an = torch.from_numpy(anchors).to(dtype)
                p = torch.from_numpy(p_np).to(dtype).requires_grad_(True)
                t = torch.from_numpy(t_np).to(dtype)
                loss = ref.GDLoss(lt, loss_weight=1.0, reduction='none', **kw)(head_torch.delta_decode(an, p),
                                                                                head_torch.delta_decode(an, t))
                loss.sum().backward()
                out[f'{lt}.loss{tag}'] = loss.detach().numpy()
                out[f'{lt}.gp_nanrow{tag}'] = np.isnan(p.grad.numpy()).any(1)
            print(lt, ''.join('N' if np.isnan(x) else ('I' if np.isinf(x) else '.') for x in out[f'{lt}.loss32']),
                  ''.join('N' if x else '.' for x in out[f'{lt}.gp_nanrow32']))
    np.savez_compressed(os.path.join(HERE, 'anchor_extreme.npz'), **out)


if __name__ == '__main__':
    main()
