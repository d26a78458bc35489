"""oracle/center_infer_torch.py — TEST INFRASTRUCTURE, NOT PRODUCT CODE.

CPU torch restatement of the CenterPoint inference slice that ends in rotated NMS:
  select_best / _topk   /root/reference/mmdet3d_gaussian/core/bbox/coders/centerpoint_bbox_coders.py:23-58
  decode (Rev coder)    centerpoint_bbox_coders.py:87-112          decode (yaw coder): oracle/coder_torch.py
  get_bboxes            /root/reference/mmdet3d_gaussian/models/dense_heads/gd_centerpoint_head.py:218-303
  get_task_detections   gd_centerpoint_head.py:305-361
  _reconstruct_bbox     gd_centerpoint_head.py:202-216 (CenterHeadRev: rot), :372-387 (CenterGDHead: yaw + dir)
select_best and both decodes are PINNED by tests/golden/center_infer.npz, generated from the real coder classes
(tests/golden/make_golden_center_infer.py; tests/test_center_infer_cpu.py).  get_bboxes / get_task_detections need mmdet3d
(absent) to import: restated here from the text, with the NMS of oracle/rbox_oracle.c part 1 (parity unpinned)
and mmdet3d's `LiDARInstance3DBoxes(...).bev` = columns [0, 1, 3, 4, 6] + `xywhr2xyxyr` — PARITY UNPINNED past the decode.
Never imported by the product package."""
import numpy as np
import torch

from . import coder_torch


def topk(scores, K):
    """_topk (:23-49): the K best cells of every class, then the K best of those candidates over all classes.
    -> scores (B,K), class (B,K), y (B,K), x (B,K)"""
    B, C, H, W = scores.shape
    per_class_val, per_class_cell = scores.reshape(B, C, H * W).topk(K)            # (B, C, K)
    per_class_cell = per_class_cell % (H * W)
    rows = (per_class_cell.float() // W).long()                                   # float floor-division, as the reference does
    cols = (per_class_cell % W).long()
    best_val, pick = per_class_val.reshape(B, C * K).topk(K)                      # pick indexes the (class, rank) candidates
    return best_val, (pick // K).long(), rows.reshape(B, -1).gather(1, pick), cols.reshape(B, -1).gather(1, pick)


def select_best(scores, preds, K):
    """select_best (:51-58): scores (B,C,H,W) after the sigmoid, preds (B,N,H,W) -> scores, classes, (x, y) cells, (B,K,N)."""
    val, cls, ys, xs = topk(scores, K)
    channels_last = preds.permute(0, 2, 3, 1)
    picked = torch.stack([channels_last[b, ys[b], xs[b]] for b in range(scores.shape[0])], dim=0)
    return val, cls, torch.stack((xs, ys), dim=-1), picked


def decode_rev(locs, preds, pc_range, out_size_factor, voxel_size, norm_bbox=True):
    """CenterPointBBoxCoderRev.decode (:87-112): cell + offset -> metres, exp of the log dims, rot = atan2(sin, cos)."""
    centre = [((preds[..., k] + locs[..., k]) * out_size_factor * voxel_size[k] + pc_range[k]).unsqueeze(-1) for k in (0, 1)]
    dims = preds[..., 3:6].exp() if norm_bbox else preds[..., 3:6]
    rot = torch.atan2(preds[..., 6], preds[..., 7]).unsqueeze(-1)
    return torch.cat(centre + [preds[..., 2:3], dims, rot, preds[..., 8:]], dim=-1)


def reconstruct(preds_dict, kind):
    pred = []
    if 'reg' in preds_dict:
        pred.append(preds_dict['reg'])
    else:
        b, _, h, w = preds_dict['height'].shape
        pred.append(preds_dict['height'].new_full((b, 2, h, w), 0.5))
    pred += [preds_dict['height'], preds_dict['dim']]
    pred += [preds_dict['rot']] if kind == 'rev' else [preds_dict['yaw'], preds_dict['dir']]
    if 'vel' in preds_dict:
        pred.append(preds_dict['vel'])
    return torch.cat(pred, dim=1)


def center_mask(scores, boxes, score_threshold, post_center_limit_range):
    """:246-250.  `preds[..., i].ge(lo).le(hi)` compares the BOOLEAN result of the first test with the upper limit, as written
    in the reference: kept bit for bit."""
    mask = scores.ge(score_threshold)
    if post_center_limit_range is not None:
        for i in range(3):
            mask = mask * boxes[..., i].ge(post_center_limit_range[i]).le(post_center_limit_range[i + 3])
    return mask


def bev_xyxyr(boxes):
    """xywhr2xyxyr(LiDARInstance3DBoxes(boxes).bev) (:336-337): bev = [x, y, dx, dy, yaw]."""
    b = boxes[:, [0, 1, 3, 4, 6]]
    out = torch.zeros_like(b)
    hw, hh = b[:, 2] / 2, b[:, 3] / 2
    out[:, 0], out[:, 1], out[:, 2], out[:, 3], out[:, 4] = b[:, 0] - hw, b[:, 1] - hh, b[:, 0] + hw, b[:, 1] + hh, b[:, 4]
    return out


def get_bboxes(preds_dicts, kind, coder_cfg, test_cfg, num_classes, sigmoid=torch.sigmoid, stage=None):
    """get_bboxes (:218-303) on CPU tensors.  preds_dicts: list over tasks of dicts of (B,c,H,W) maps incl. 'heatmap'.
    Returns one [bboxes (n,9) with z moved to the box bottom, scores (n,), labels (n,) int32] per sample.
    stage: optional dict that receives, per task, what the decode and the mask produced (for stage-wise checks)."""
    from . import circle_nms, nms_gpu_oracle
    K = test_cfg.get('max_per_img', 128)
    thr = test_cfg.get('score_threshold', 0.1)
    rng = test_cfg.get('post_center_limit_range', None)
    rets = []
    for task_id, pd in enumerate(preds_dicts):
        B = pd['heatmap'].shape[0]
        heat = sigmoid(pd['heatmap'])
        scores, clses, locs, preds = select_best(heat, reconstruct(pd, kind), K)
        if kind == 'rev':
            boxes = decode_rev(locs, preds, coder_cfg['pc_range'], coder_cfg['out_size_factor'], coder_cfg['voxel_size'],
                               coder_cfg.get('norm_bbox', True))
        else:
            boxes = coder_torch.center_decode(locs, preds, coder_cfg['pc_range'], coder_cfg['out_size_factor'],
                                              coder_cfg['voxel_size'], coder_cfg.get('norm_bbox', True), correct_yaw=True)
        mask = center_mask(scores, boxes, thr, rng)
        if stage is not None:
            stage[task_id] = dict(scores=scores, clses=clses, locs=locs, boxes=boxes, mask=mask)
        ret_task = []
        for i in range(B):
            bx, sc, lb = boxes[i][mask[i]], scores[i][mask[i]], clses[i][mask[i]]
            if test_cfg['nms_type'] == 'circle':
                dets = torch.cat([bx[:, [0, 1]], sc.view(-1, 1)], dim=1).numpy()
                keep = torch.as_tensor(np.asarray(circle_nms(dets, test_cfg['min_radius'][task_id],
                                                             post_max_size=test_cfg['post_max_size']), dtype=np.int64))
            elif sc.numel() > 0:
                keep = torch.as_tensor(np.asarray(nms_gpu_oracle(bev_xyxyr(bx).numpy(), sc.numpy(), test_cfg['nms_thr'],
                                                                 pre_max_size=test_cfg['pre_max_size'],
                                                                 post_max_size=test_cfg['post_max_size']), dtype=np.int64))
            else:
                keep = torch.zeros(0, dtype=torch.int64)
            ret_task.append(dict(bboxes=bx[keep], scores=sc[keep], labels=lb[keep]))
        rets.append(ret_task)
    out = []
    for i in range(len(rets[0])):
        bboxes = torch.cat([r[i]['bboxes'] for r in rets]).clone()
        bboxes[:, 2] = bboxes[:, 2] - bboxes[:, 5] * 0.5
        flag, labels = 0, []
        for j, nc in enumerate(num_classes):
            labels.append((rets[j][i]['labels'] + flag).int())
            flag += nc
        out.append([bboxes, torch.cat([r[i]['scores'] for r in rets]), torch.cat(labels)])
    return out

