#!/usr/bin/env python3
"""Anchor-head inference slice at the reference's PointPillars geometries, us per call (host + device, synchronised):
  KITTI  (configs/_base_/models/hv_pointpillars_secfpn_kitti.py:89-96): 248 x 216 x 6 anchors, 3 classes, nms_pre 4096, nms_thr 0.01,
         score_thr 0.05, max_num 100; dir_offset / dir_limit_offset (0, 1) = mmdet3d 0.x's head defaults (1.0 has (-pi/2, 0); the config
         sets neither, and the function takes both as arguments)
  Waymo  (BASELINE configs[4], hv_pointpillars_secfpn_waymo.py:59-60, :101-109): 468 x 468 x 6 anchors, 3 classes, nms_pre 4096,
         nms_thr 0.25, score_thr 0.1, max_num 500, dir_offset 0.7854, dir_limit_offset 0
ours  = anchor_head_get_bboxes (score kernel, selection, gather + decode, batched class NMS, collect; one read-back)
eager = mmdet3d's op sequence (oracle/anchor_infer_torch.py's statement) on device tensors with THIS package's nms_gpu per class
Asserts equal numbers of detections and equal boxes."""
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch  # noqa: E402

import mmdet3d_gaussian_amd as amd  # noqa: E402
from oracle import anchor_infer_torch as ait  # noqa: E402
from test_gpu_anchor_infer import head_outputs  # noqa: E402

dev = torch.device('cuda:0')


def timeit(fn, it, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(it):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / it * 1e6


def eager_single(cls, bbox, dirs, anchors, cfg, C, dir_offset, dir_limit_offset):
    dir_cls_score = torch.max(dirs.permute(1, 2, 0).reshape(-1, 2), dim=-1)[1]
    scores = cls.permute(1, 2, 0).reshape(-1, C).sigmoid()
    bbox = bbox.permute(1, 2, 0).reshape(-1, 7)
    if cfg['nms_pre'] > 0 and scores.shape[0] > cfg['nms_pre']:
        max_scores, _ = scores.max(dim=1)
        _, topk = max_scores.topk(cfg['nms_pre'])
        anchors, bbox, scores, dir_cls_score = anchors[topk], bbox[topk], scores[topk], dir_cls_score[topk]
    boxes = ait.delta_decode(anchors, bbox)
    for_nms = ait.bev_xyxyr(boxes)
    bb, ss, ll, dd = [], [], [], []
    for i in range(C):
        m = scores[:, i] > cfg['score_thr']
        if not m.any():
            continue
        sel = amd.nms_gpu(for_nms[m], scores[m, i], cfg['nms_thr'])
        bb.append(boxes[m][sel]); ss.append(scores[m, i][sel]); ll.append(torch.full((len(sel),), i, dtype=torch.long, device=dev)); dd.append(dir_cls_score[m][sel])
    bb, ss, ll, dd = torch.cat(bb), torch.cat(ss), torch.cat(ll), torch.cat(dd)
    if bb.shape[0] > cfg['max_num']:
        inds = ss.sort(descending=True, stable=True)[1][:cfg['max_num']]
        bb, ss, ll, dd = bb[inds], ss[inds], ll[inds], dd[inds]
    rot = ait.limit_period(bb[:, 6] - dir_offset, dir_limit_offset, math.pi)
    bb[:, 6] = rot + dir_offset + math.pi * dd.to(bb.dtype)
    return bb, ss, ll


def main():
    g = torch.Generator().manual_seed(5)
    for name, (B, H, W), cfg, doff, dlim in (
            ('KITTI 248x216x6, nms_pre 4096, max_num 100', (4, 248, 216), dict(use_rotate_nms=True, nms_pre=4096, nms_thr=0.01, score_thr=0.05, max_num=100), 0.0, 1.0),
            ('Waymo 468x468x6, nms_pre 4096, max_num 500', (1, 468, 468), dict(use_rotate_nms=True, nms_pre=4096, nms_thr=0.25, score_thr=0.1, max_num=500), 0.7854, 0.0)):
        cls, bbox, dirs, anchors = [t.to(dev) for t in head_outputs(g, B, 6, 3, H, W, scene=150.0)]
        ours = amd.extras.anchor_head_get_bboxes([cls], [bbox], [dirs], [anchors], cfg, 3, doff, dlim)
        ref = [eager_single(cls[b], bbox[b], dirs[b], anchors, cfg, 3, doff, dlim) for b in range(B)]
        for o, r in zip(ours, ref):
            assert o[0].shape == r[0].shape, (o[0].shape, r[0].shape)
            torch.testing.assert_close(o[0], r[0], rtol=1e-5, atol=1e-5)
            assert torch.equal(o[2], r[2])
        us_a = timeit(lambda: amd.extras.anchor_head_get_bboxes([cls], [bbox], [dirs], [anchors], cfg, 3, doff, dlim), 50)
        us_p = timeit(lambda: amd.extras.anchor_head_get_bboxes([cls], [bbox], [dirs], [anchors], cfg, 3, doff, dlim, padded=True), 100)
        us_b = timeit(lambda: [eager_single(cls[b], bbox[b], dirs[b], anchors, cfg, 3, doff, dlim) for b in range(B)], 5, warm=1)
        print(json.dumps(dict(what=f'anchor head get_bboxes, {name}, batch {B}', detections=[int(o[0].shape[0]) for o in ours],
                              ours_us=round(us_a, 1), ours_padded_no_readback_us=round(us_p, 1), eager_us=round(us_b, 1))), flush=True)


if __name__ == '__main__':
    main()
