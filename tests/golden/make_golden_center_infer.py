#!/usr/bin/env python3
"""Golden vectors for the CenterPoint INFERENCE slice that feeds rotated NMS (SURVEY.md §3.3, row aN's call site), FROM THE
REAL REFERENCE coders (build container only):  python3 -B tests/golden/make_golden_center_infer.py

Imports /root/reference/mmdet3d_gaussian/core/bbox/coders/{centerpoint_bbox_coders,centerpoint_bbox_yaw_coders}.py (stubbing
only mmdet's BaseBBoxCoder / BBOX_CODERS, absent here) and runs what CenterHeadRev.get_bboxes runs per task
(gd_centerpoint_head.py:236-244):
    batch_heatmap = heatmap.sigmoid()
    scores, clses, locs, preds = coder.select_best(batch_heatmap, cat(head maps), max_per_img)
    boxes = coder.decode(locs, preds)             CenterPointBBoxCoderRev (rot = atan2(sin, cos)) and
                                                  CenterPointBBoxYawCoder (correct_yaw=True: quarter-turn snap + w/l swap)
The head class itself needs mmdet3d (absent): the mask / NMS / merge steps after this are restated in
oracle/center_infer_torch.py.  Cases: tie-free heat maps (the generator checks that the K+1 best sigmoid scores are distinct,
so torch.topk has one answer), one `ties` case with quantised logits (only order-free properties are compared there).
Writes tests/golden/center_infer.npz (data only)."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden_coder import load_reference_coders  # noqa: E402

CFG = dict(pc_range=[-51.2, -51.2], out_size_factor=4, voxel_size=[0.2, 0.2], code_size=9, norm_bbox=True)
# name: (coder, B, C, H, W, K, quantise, reg present)
CASES = {
    'rev_c1': ('rev', 2, 1, 32, 40, 50, 0, True),
    'rev_c3': ('rev', 1, 3, 24, 24, 80, 0, True),
    'yaw_c2': ('yaw', 2, 2, 40, 32, 64, 0, True),
    'yaw_c1_k500': ('yaw', 1, 1, 64, 64, 500, 0, True),
    'yaw_ties': ('yaw', 2, 2, 32, 32, 48, 8, True),
}


def maps_for(kind, B, H, W, g):
    """raw head maps around plausible values; channel order of _reconstruct_bbox (gd_centerpoint_head.py:202-216, :372-387)"""
    reg = torch.rand(B, 2, H, W, generator=g)
    height = torch.rand(B, 1, H, W, generator=g) * 4 - 3
    dim = torch.randn(B, 3, H, W, generator=g) * 0.4 + 0.6
    yaw = (torch.rand(B, 1, H, W, generator=g) * 2 - 1) * 3.3
    vel = torch.randn(B, 2, H, W, generator=g)
    if kind == 'rev':
        rot = torch.cat([yaw.sin(), yaw.cos()], 1) + torch.randn(B, 2, H, W, generator=g) * 0.1
        return [('reg', reg), ('height', height), ('dim', dim), ('rot', rot), ('vel', vel)]
    turn = torch.randint(-2, 3, (B, 1, H, W), generator=g).float() * (np.pi / 2)
    dirs = torch.cat([(yaw + turn).sin(), (yaw + turn).cos()], 1) + torch.randn(B, 2, H, W, generator=g) * 0.1
    return [('reg', reg), ('height', height), ('dim', dim), ('yaw', yaw), ('dir', dirs), ('vel', vel)]


def main():
    torch.set_num_threads(1)
    coders = load_reference_coders()
    out = {'cfg_pc_range': np.array(CFG['pc_range'], np.float64), 'cfg_voxel_size': np.array(CFG['voxel_size'], np.float64),
           'cfg_out_size_factor': np.array(CFG['out_size_factor'])}
    for name, (kind, B, C, H, W, K, quant, _) in CASES.items():
        coder = (coders.CenterPointBBoxCoderRev if kind == 'rev' else coders.CenterPointBBoxYawCoder)(**CFG)
        seed = 100
        while True:
            g = torch.Generator().manual_seed(seed)
            heat = torch.randn(B, C, H, W, generator=g) * 1.5 - 2.0
            if quant:
                heat = (heat * quant).round() / quant
            sig = heat.sigmoid()
            if quant:
                break
            top = sig.view(B, -1).topk(K + 1)[0]
            distinct = bool((top[:, :-1] > top[:, 1:]).all())
            far = bool(((top - 0.1).abs() > 1e-5).all())     # score_threshold 0.1 of the nuScenes test_cfg: no borderline score
            if distinct and far:
                break
            seed += 1
        maps = maps_for(kind, B, H, W, g)
        batch_pred = torch.cat([m for _, m in maps], dim=1)
        scores, clses, locs, preds = coder.select_best(sig, batch_pred, K)
        boxes = coder.decode(locs, preds)
        out[f'{name}.heat'] = heat.numpy()
        for k, m in maps:
            out[f'{name}.{k}'] = m.numpy()
        out[f'{name}.K'] = np.array(K)
        out[f'{name}.scores'] = scores.numpy()
        out[f'{name}.clses'] = clses.numpy()
        out[f'{name}.locs'] = locs.numpy()
        out[f'{name}.preds'] = preds.numpy()
        out[f'{name}.boxes'] = boxes.numpy()
        print(name, 'seed', seed, 'scores', float(scores.max()), float(scores.min()), 'boxes', tuple(boxes.shape))
    path = os.path.join(HERE, 'center_infer.npz')
    np.savez_compressed(path, **out)
    print('center_infer.npz', os.path.getsize(path), 'bytes')


if __name__ == '__main__':
    main()
