#!/usr/bin/env python3
"""Golden vectors for the CenterGDHead regression slice on EXTREME raw head outputs, FROM THE REAL REFERENCE (build
container only):  python3 -B tests/golden/make_golden_coder_extreme.py  ->  tests/golden/coder_center_extreme.npz

Same pipeline as make_golden_coder.py (reference coder.decode -> reference GDLoss, reduction 'none'), 40 objects whose
raw outputs leave the trained regime: log-dims of +-20, +89 (exp overflows to inf), -104 (exp underflows to 0 -> the
1e-7 clamp), NaN and inf entries, cell offsets of 1e4 and 1e30, yaws of 1e4.  Stored: per-object loss in fp32 / fp64 and
whether the object's gradient row wrt the raw outputs contains a NaN.  Only data is written."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from _ref_loader import load_reference_loss  # noqa: E402
from make_golden_coder import load_reference_coders  # noqa: E402

CASES = (('gwd3d', dict(fun='log1p', tau=0.0)), ('bd3d', dict(fun='log1p', tau=1.0)), ('kld3d', dict(fun='none', tau=0.0)),
         ('kld3d_symmax', dict(fun='log1p', tau=1.0)))


def inputs():
    g = torch.Generator().manual_seed(23)
    B, K = 1, 40
    locs = torch.stack([torch.randint(0, 128, (B, K), generator=g), torch.randint(0, 128, (B, K), generator=g)], -1)
    frac = torch.rand(B, K, 2, generator=g)
    xy = (locs.float() + frac) * 4 * 0.2 - 51.2
    anno = torch.cat([xy, torch.rand(B, K, 1, generator=g) * 4 - 3,
                      torch.rand(B, K, 3, generator=g) * torch.tensor([2.0, 4.0, 1.5]) + 0.5,
                      (torch.rand(B, K, 1, generator=g) * 2 - 1) * 3.14159, torch.randn(B, K, 2, generator=g)], -1)
    pred = torch.cat([frac, anno[..., 2:3], anno[..., 3:6].log(), anno[..., 6:7], anno[..., 6:7].sin(),
                      anno[..., 6:7].cos(), anno[..., 7:9]], -1)
    pred = pred + torch.randn(B, K, 11, generator=g) * torch.tensor([.15, .15, .1, .08, .08, .08, .1, .05, .05, .1, .1])
    nan, inf = float('nan'), float('inf')
    p = pred[0]
    for k, v in enumerate((20.0, -20.0, 89.0, -104.0, nan, inf, -inf)):     # rows 0..6: one log-dim each
        p[k, 3 + k % 3] = v
    p[7, 3:6] = 89.0
    p[8, 3:6] = -104.0
    p[9, 0] = 1e4
    p[10, 1] = 1e30
    p[11, 0] = nan
    p[12, 1] = inf
    p[13, 2] = 1e30
    p[14, 2] = nan
    p[15, 6] = 1e4
    p[16, 6] = nan
    p[17, 6] = inf
    p[18, 3] = 44.0      # exp = 1.3e19: (1e19)^2 overflows fp32
    p[19, 3:5] = 30.0
    return locs, anno, pred


def main():
    torch.set_num_threads(1)
    ref = load_reference_loss()
    coders = load_reference_coders()
    cfg = dict(pc_range=[-51.2, -51.2], out_size_factor=4, voxel_size=[0.2, 0.2], code_size=9, norm_bbox=True)
    coder = coders.CenterPointBBoxYawCoder(**cfg)
    locs, anno, pred = inputs()
    out = {'locs': locs.numpy(), 'anno': anno.numpy(), 'pred': pred.numpy(),
           'cfg_pc_range': np.array(cfg['pc_range'], np.float64), 'cfg_voxel_size': np.array(cfg['voxel_size'], np.float64),
           'cfg_out_size_factor': np.array(cfg['out_size_factor'])}
    with np.errstate(all='ignore'):
        # the coder on its own (inference path, gd_centerpoint_head.py:244): decode with and without correct_yaw, fp32
        out['decode_noyaw32'] = coder.decode(locs, pred, correct_yaw=False).numpy()
        out['decode_yaw32'] = coder.decode(locs, pred, correct_yaw=True).numpy()
        for lt, kw in CASES:
            for dtype, tag in ((torch.float32, '32'), (torch.float64, '64')):
                p = pred.to(dtype).clone().requires_grad_(True)
                pred_gd = coder.decode(locs, p, correct_yaw=False)[..., :7]
                target_gd = coder.encode(anno.to(dtype))[..., :7]
                loss = ref.GDLoss(lt, loss_weight=1.0, reduction='none', **kw)(pred_gd.reshape(-1, 7), target_gd.reshape(-1, 7))
                loss.sum().backward()
                out[f'{lt}.loss{tag}'] = loss.detach().numpy()
                out[f'{lt}.gp_nanrow{tag}'] = np.isnan(p.grad.numpy().reshape(-1, 11)).any(1)
            print(lt, ''.join('N' if np.isnan(x) else ('I' if np.isinf(x) else '.') for x in out[f'{lt}.loss32']),
                  ''.join('N' if x else '.' for x in out[f'{lt}.gp_nanrow32']))
    np.savez_compressed(os.path.join(HERE, 'coder_center_extreme.npz'), **out)


if __name__ == '__main__':
    main()
