"""oracle/anchor_cls_torch.py — TEST INFRASTRUCTURE, NOT PRODUCT CODE.

The classification and direction terms of GDAnchor3DHead.loss_single as plain torch ops with autograd:
  /root/reference/mmdet3d_gaussian/models/dense_heads/gd_anchor3d_head.py:84-92   (loss_cls on the permuted class maps)
  gd_anchor3d_head.py:143-149 (loss_dir on the positives' direction logits; the positives are `labels in [0, num_classes)`, :101-103)
with the two loss modules they call, both mmdet's (third party, absent: restated from the published 2.x text — PARITY UNPINNED):
  FocalLoss(use_sigmoid=True)            py_sigmoid_focal_loss + weight_reduce_loss(mean, avg_factor)
  CrossEntropyLoss(use_sigmoid=False)    F.cross_entropy(reduction='none') * weight, then the same reduction
(On a GPU mmdet routes the same loss through mmcv's fused sigmoid_focal_loss op, which evaluates log(max(p, FLT_MIN)): the same
values up to |logit| ~ 87, a clamp beyond; this restatement and the kernel follow the unclamped formula.)
Never imported by the product package."""
import torch
import torch.nn.functional as F


def sigmoid_focal_loss(pred, target, weight, gamma, alpha, avg_factor, loss_weight):
    """pred (N, C) logits, target (N,) labels in [0, C] (C = background), weight (N,)"""
    C = pred.shape[1]
    t = F.one_hot(target, num_classes=C + 1)[:, :C].type_as(pred)
    p = pred.sigmoid()
    pt = (1 - p) * t + p * (1 - t)
    focal_weight = (alpha * t + (1 - alpha) * (1 - t)) * pt.pow(gamma)
    loss = F.binary_cross_entropy_with_logits(pred, t, reduction='none') * focal_weight
    loss = loss * weight.reshape(-1, 1)
    return loss_weight * loss.sum() / avg_factor


def cross_entropy_loss(pred, label, weight, avg_factor, loss_weight):
    loss = F.cross_entropy(pred, label, reduction='none') * weight
    return loss_weight * loss.sum() / avg_factor


def cls_dir_losses(cls_score, dir_cls_preds, labels, label_weights, dir_targets, dir_weights, num_classes, num_total_samples,
                   gamma=2.0, alpha=0.25, cls_weight=1.0, dir_weight=0.2):
    """cls_score (B, A*C, H, W), dir_cls_preds (B, A*2, H, W); labels / label_weights / dir_targets / dir_weights (B, H*W*A).
    Returns (loss_cls, loss_dir); with no positive anchor loss_dir is `pos_dir_cls_preds.sum()` = 0 (:157-158)."""
    labels, label_weights = labels.reshape(-1), label_weights.reshape(-1)
    cls = cls_score.permute(0, 2, 3, 1).reshape(-1, num_classes)
    loss_cls = sigmoid_focal_loss(cls, labels, label_weights, gamma, alpha, num_total_samples, cls_weight)
    dirs = dir_cls_preds.permute(0, 2, 3, 1).reshape(-1, 2)
    pos = ((labels >= 0) & (labels < num_classes)).nonzero(as_tuple=False).reshape(-1)
    if len(pos) > 0:
        loss_dir = cross_entropy_loss(dirs[pos], dir_targets.reshape(-1)[pos], dir_weights.reshape(-1)[pos], num_total_samples, dir_weight)
    else:
        loss_dir = dirs[pos].sum()
    return loss_cls, loss_dir
