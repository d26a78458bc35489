#!/usr/bin/env python3
"""Golden vectors for SURVEY.md §8f-2: the CenterPoint yaw coder + the CenterGDHead regression slice, FROM THE REAL
REFERENCE (build container only):  python3 -B tests/golden/make_golden_coder.py

Imports /root/reference/mmdet3d_gaussian/core/bbox/coders/{centerpoint_bbox_coders,centerpoint_bbox_yaw_coders}.py
(stubbing only mmdet's BaseBBoxCoder / BBOX_CODERS, which are absent) and the reference GDLoss, and runs
    pred_gd   = coder.decode(locs, pred, correct_yaw=False)[..., :7]        (gd_centerpoint_head.py:422-423)
    target_gd = coder.encode(anno_boxes)[..., :7]                            (:413-415)
    loss      = GDLoss(pred_gd, target_gd, avg_factor=max(num_pos, 1))       (:433-434)
with autograd gradients back to the raw head outputs `pred`.  Writes tests/golden/coder_center.npz (data only)."""
import importlib.util
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from _ref_loader import REF_ROOT, load_reference_loss  # noqa: E402


def load_reference_coders():
    sys.dont_write_bytecode = True
    for name in ('mmdet', 'mmdet.core', 'mmdet.core.bbox', 'mmdet.core.bbox.builder'):
        sys.modules.setdefault(name, types.ModuleType(name))

    class _Reg:
        def register_module(self, *a, **k):
            return lambda c: c
    sys.modules['mmdet.core.bbox'].BaseBBoxCoder = type('BaseBBoxCoder', (), {})
    sys.modules['mmdet.core.bbox.builder'].BBOX_CODERS = _Reg()
    pkg_dir = os.path.join(REF_ROOT, 'mmdet3d_gaussian', 'core', 'bbox', 'coders')
    spec = importlib.util.spec_from_file_location('_ref_coders', os.path.join(pkg_dir, '__init__.py'),
                                                  submodule_search_locations=[pkg_dir])
    mod = importlib.util.module_from_spec(spec)
    sys.modules['_ref_coders'] = mod
    spec.loader.exec_module(mod)
    return mod


def main():
    torch.set_num_threads(1)
    ref = load_reference_loss()
    coders = load_reference_coders()
    cfg = dict(pc_range=[-51.2, -51.2], out_size_factor=4, voxel_size=[0.2, 0.2], code_size=9, norm_bbox=True)
    coder = coders.CenterPointBBoxYawCoder(**cfg)
    g = torch.Generator().manual_seed(11)
    B, K = 2, 96
    locs = torch.stack([torch.randint(0, 128, (B, K), generator=g), torch.randint(0, 128, (B, K), generator=g)], -1)
    # annotated boxes inside the voxel of their centre cell; 9 = 7 + velocity(2)
    frac = torch.rand(B, K, 2, generator=g)
    xy = (locs.float() + frac) * 4 * 0.2 - 51.2
    anno = torch.cat([xy, torch.rand(B, K, 1, generator=g) * 4 - 3,
                      torch.rand(B, K, 3, generator=g) * torch.tensor([2.0, 4.0, 1.5]) + 0.5,
                      (torch.rand(B, K, 1, generator=g) * 2 - 1) * 3.14159, torch.randn(B, K, 2, generator=g)], -1)
    enc = coder.encode(anno)
    # raw head outputs near the encoded target: reg(2), height, log-dim(3), yaw, dir(2), vel(2)
    pred = torch.cat([frac, anno[..., 2:3], anno[..., 3:6].log(), anno[..., 6:7], anno[..., 6:7].sin(),
                      anno[..., 6:7].cos(), anno[..., 7:9]], -1)
    pred = pred + torch.randn(B, K, 11, generator=g) * torch.tensor([.15, .15, .1, .08, .08, .08, .1, .05, .05, .1, .1])
    out = {'locs': locs.numpy(), 'anno': anno.numpy(), 'pred': pred.numpy(), 'enc7': enc[..., :7].numpy(),
           'cfg_pc_range': np.array(cfg['pc_range'], np.float64), 'cfg_voxel_size': np.array(cfg['voxel_size'], np.float64),
           'cfg_out_size_factor': np.array(cfg['out_size_factor']), 'avg_factor': np.array(float(B * K - 7))}
    out['decode_noyaw32'] = coder.decode(locs, pred, correct_yaw=False).numpy()
    out['decode_yaw32'] = coder.decode(locs, pred, correct_yaw=True).numpy()
    for lt, kw in (('gwd3d', dict(fun='log1p', tau=0.0)), ('bd3d', dict(fun='log1p', tau=1.0)),
                   ('kld3d', dict(fun='none', tau=0.0))):
        for dtype, tag in ((torch.float32, '32'), (torch.float64, '64')):
            p = pred.to(dtype).clone().requires_grad_(True)
            pred_gd = coder.decode(locs, p, correct_yaw=False)[..., :7]
            target_gd = coder.encode(anno.to(dtype))[..., :7]
            loss = ref.GDLoss(lt, loss_weight=5.0, **kw)(pred_gd, target_gd, avg_factor=float(out['avg_factor']))
            loss.backward()
            out[f'{lt}.loss{tag}'] = loss.detach().numpy()
            out[f'{lt}.gpred{tag}'] = p.grad.numpy()
    np.savez_compressed(os.path.join(HERE, 'coder_center.npz'), **out)
    print('coder_center.npz', os.path.getsize(os.path.join(HERE, 'coder_center.npz')), 'bytes;',
          {k: float(out[k]) for k in out if k.endswith('loss64')})


if __name__ == '__main__':
    main()
