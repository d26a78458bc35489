"""oracle/anchor_infer_torch.py — TEST INFRASTRUCTURE, NOT PRODUCT CODE.

CPU torch restatement of the anchor-head inference path that the reference's GDAnchor3DHead INHERITS unchanged from mmdet3d
(/root/reference/mmdet3d_gaussian/models/dense_heads/gd_anchor3d_head.py:10: `class GDAnchor3DHead(Anchor3DHead)`, no get_bboxes of
its own) — third party, absent here, version not pinned by the reference; restated from the published mmdet3d 0.x text,
PARITY UNPINNED:
  Anchor3DHead.get_bboxes_single       mmdet3d/models/dense_heads/anchor3d_head.py   (per level: sigmoid scores, nms_pre best
                                       anchors by their best class, delta decode; all levels concatenated; multi-class NMS;
                                       direction-bin correction of the yaw)
  box3d_multiclass_nms                 mmdet3d/core/post_processing/box3d_nms.py
  DeltaXYZWLHRBBoxCoder.decode         mmdet3d/core/bbox/coders/delta_xyzwhlr_bbox_coder.py
  limit_period                         mmdet3d/core/bbox/structures/utils.py
with the NMS of oracle/rbox_oracle.c part 1.  BASELINE configs[4] (Waymo PointPillars: nms_pre 4096, 3 classes, nms_thr 0.25,
score_thr 0.1, max_num 500) and configs[1] (KITTI) run through this path at inference.  Never imported by the product package."""
import math

import numpy as np
import torch


def limit_period(val, offset=0.5, period=math.pi):
    return val - torch.floor(val / period + offset) * period


def delta_decode(anchors, deltas):
    xa, ya, za, wa, la, ha, ra = torch.split(anchors, 1, dim=-1)
    xt, yt, zt, wt, lt, ht, rt = torch.split(deltas, 1, dim=-1)
    za = za + ha / 2
    diagonal = torch.sqrt(la ** 2 + wa ** 2)
    xg = xt * diagonal + xa
    yg = yt * diagonal + ya
    zg = zt * ha + za
    lg = torch.exp(lt) * la
    wg = torch.exp(wt) * wa
    hg = torch.exp(ht) * ha
    rg = rt + ra
    zg = zg - hg / 2
    return torch.cat([xg, yg, zg, wg, lg, hg, rg], dim=-1)


def bev_xyxyr(boxes):
    b = boxes[:, [0, 1, 3, 4, 6]]
    hw, hh = b[:, 2] / 2, b[:, 3] / 2
    return torch.stack((b[:, 0] - hw, b[:, 1] - hh, b[:, 0] + hw, b[:, 1] + hh, b[:, 4]), dim=-1)


def box3d_multiclass_nms(mlvl_bboxes, mlvl_bboxes_for_nms, mlvl_scores, score_thr, max_num, nms_thr, use_rotate_nms, mlvl_dir_scores):
    from . import nms_gpu_oracle
    num_classes = mlvl_scores.shape[1] - 1
    bboxes, scores, labels, dir_scores = [], [], [], []
    for i in range(num_classes):
        cls_inds = mlvl_scores[:, i] > score_thr
        if not cls_inds.any():
            continue
        _scores = mlvl_scores[cls_inds, i]
        _for_nms = mlvl_bboxes_for_nms[cls_inds, :]
        selected = torch.as_tensor(np.asarray(nms_gpu_oracle(_for_nms.numpy(), _scores.numpy(), nms_thr, normal=not use_rotate_nms), dtype=np.int64))
        bboxes.append(mlvl_bboxes[cls_inds, :][selected])
        scores.append(_scores[selected])
        labels.append(torch.full((len(selected),), i, dtype=torch.long))
        dir_scores.append(mlvl_dir_scores[cls_inds][selected])
    if bboxes:
        bboxes, scores, labels, dir_scores = torch.cat(bboxes), torch.cat(scores), torch.cat(labels), torch.cat(dir_scores)
        if bboxes.shape[0] > max_num:
            inds = scores.sort(descending=True, stable=True)[1][:max_num]
            bboxes, scores, labels, dir_scores = bboxes[inds], scores[inds], labels[inds], dir_scores[inds]
    else:
        bboxes, scores = mlvl_scores.new_zeros((0, mlvl_bboxes.size(-1))), mlvl_scores.new_zeros((0,))
        labels, dir_scores = mlvl_scores.new_zeros((0,), dtype=torch.long), mlvl_scores.new_zeros((0,), dtype=torch.long)
    return bboxes, scores, labels, dir_scores


def get_bboxes_single(cls_scores, bbox_preds, dir_cls_preds, mlvl_anchors, cfg, num_classes, box_code_size=7, dir_offset=0.0,
                      dir_limit_offset=1.0, stage=None):
    """one sample: per level cls_score (A*C, H, W), bbox_pred (A*code, H, W), dir_cls_pred (A*2, H, W), anchors (H*W*A, code)"""
    mlvl_bboxes, mlvl_scores, mlvl_dir_scores = [], [], []
    for cls_score, bbox_pred, dir_cls_pred, anchors in zip(cls_scores, bbox_preds, dir_cls_preds, mlvl_anchors):
        dir_cls_score = torch.max(dir_cls_pred.permute(1, 2, 0).reshape(-1, 2), dim=-1)[1]
        scores = cls_score.permute(1, 2, 0).reshape(-1, num_classes).sigmoid()
        bbox_pred = bbox_pred.permute(1, 2, 0).reshape(-1, box_code_size)
        nms_pre = cfg.get('nms_pre', -1)
        if nms_pre > 0 and scores.shape[0] > nms_pre:
            max_scores, _ = scores.max(dim=1)
            _, topk_inds = max_scores.topk(nms_pre)
            anchors, bbox_pred, scores, dir_cls_score = anchors[topk_inds, :], bbox_pred[topk_inds, :], scores[topk_inds, :], dir_cls_score[topk_inds]
        mlvl_bboxes.append(delta_decode(anchors, bbox_pred))
        mlvl_scores.append(scores)
        mlvl_dir_scores.append(dir_cls_score)
    mlvl_bboxes, mlvl_scores, mlvl_dir_scores = torch.cat(mlvl_bboxes), torch.cat(mlvl_scores), torch.cat(mlvl_dir_scores)
    if stage is not None:
        stage.update(boxes=mlvl_bboxes, scores=mlvl_scores, dirs=mlvl_dir_scores)
    padded = torch.cat([mlvl_scores, mlvl_scores.new_zeros(mlvl_scores.shape[0], 1)], dim=1)
    bboxes, scores, labels, dir_scores = box3d_multiclass_nms(mlvl_bboxes, bev_xyxyr(mlvl_bboxes), padded, cfg.get('score_thr', 0), cfg['max_num'],
                                                              cfg['nms_thr'], cfg.get('use_rotate_nms', True), mlvl_dir_scores)
    if bboxes.shape[0] > 0:
        dir_rot = limit_period(bboxes[..., 6] - dir_offset, dir_limit_offset, math.pi)
        bboxes[..., 6] = dir_rot + dir_offset + math.pi * dir_scores.to(bboxes.dtype)
    return bboxes, scores, labels
