"""`GDAnchor3DHead.loss_single` end to end on the device: classification, regression and direction terms of one level.

The reference's method (/root/reference/mmdet3d_gaussian/models/dense_heads/gd_anchor3d_head.py:62-161) composed from this
package's two fused slices — `anchor_head_cls_dir_loss` (:84-92, :143-149) and `anchor_head_bbox_loss` (:95-141, :150-161) — in
four launches forward, with no permuted copies of the head's maps, no `nonzero` and no host read-back: the reference syncs
twice (`labels.max().item()` at :90, `nonzero` at :101-103).  Static shapes: `GraphedStep` can capture forward and backward.
"""
from .anchor_cls import anchor_head_cls_dir_loss
from .head_loss import anchor_head_bbox_loss


def gd_anchor_head_loss_single(loss_cls, loss_bbox, loss_dir, loss_decoded_bbox, train_cfg, num_classes, cls_score, bbox_pred,
                               dir_cls_preds, labels, label_weights, bbox_targets, bbox_weights, dir_targets, dir_weights, anchor_list,
                               num_total_samples, diff_rad_by_sin=True, use_direction_classifier=True):
    """loss_cls / loss_bbox / loss_dir : the head's modules or config dicts (FocalLoss(use_sigmoid=True), SmoothL1Loss / L1Loss,
                    CrossEntropyLoss);  loss_decoded_bbox : this package's GDLoss;
    train_cfg     : the head's train_cfg, read for 'code_weight' and 'decode_weight' (:124-131);
    cls_score (B, A*C, H, W), bbox_pred (B, A*7, H, W), dir_cls_preds (B, A*2, H, W) : the level's raw outputs;
    labels, label_weights, dir_targets, dir_weights (B, H*W*A), bbox_targets, bbox_weights (B, H*W*A, 7), anchor_list
                    (H*W*A, 7) or (H, W, ..., 7) : what mmdet3d's `anchor_target_3d` hands to loss_single.
    The reference's `assert labels.max().item() <= self.num_classes` (:90) is a host sync and is not repeated: a label above
    num_classes is background for every term here, as it is for the reference's one-hot and its positive mask.
    Returns (loss_cls, loss_bbox, loss_dir) as the reference does (loss_dir None without direction classifier)."""
    get = (lambda k: train_cfg.get(k, None)) if train_cfg is not None else (lambda k: None)
    if not use_direction_classifier:
        dir_cls_preds = None
    l_cls, l_dir = anchor_head_cls_dir_loss(loss_cls, loss_dir, cls_score, dir_cls_preds, labels, label_weights, dir_targets, dir_weights,
                                            num_classes, num_total_samples)
    l_bbox = anchor_head_bbox_loss(loss_decoded_bbox, loss_bbox, bbox_pred, bbox_targets, bbox_weights, labels, anchor_list, num_classes,
                                   num_total_samples, code_weight=get('code_weight'), decode_weight=get('decode_weight'),
                                   diff_rad_by_sin=diff_rad_by_sin, dense=True)
    return l_cls, l_bbox, l_dir
