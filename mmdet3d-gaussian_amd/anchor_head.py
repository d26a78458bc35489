"""`GDAnchor3DHead.loss` and `loss_single` end to end on the device: target assignment, then the classification, regression and
direction terms of the level.

The reference's method (/root/reference/mmdet3d_gaussian/models/dense_heads/gd_anchor3d_head.py:62-161) composed from this
package's two fused slices — `anchor_head_cls_dir_loss` (:84-92, :143-149) and `anchor_head_bbox_loss` (:95-141, :150-161) — in
four launches forward, with no permuted copies of the head's maps, no `nonzero` and no host read-back: the reference syncs
twice (`labels.max().item()` at :90, `nonzero` at :101-103).  Static shapes: `GraphedStep` can capture forward and backward.
"""
import torch

from .anchor_cls import anchor_head_cls_dir_loss
from .anchor_targets import anchor_head_get_targets
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
    num_total_samples : a number, or a one-element fp32 device tensor (the kernels divide by it: no read-back).
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


def _one(x, name):
    if isinstance(x, (list, tuple)):
        if len(x) != 1:
            raise RuntimeError(f'gd_anchor_head_loss: {len(x)} feature levels in {name}; the reference heads have one')
        return x[0]
    return x


def gd_anchor_head_loss(loss_cls, loss_bbox, loss_dir, loss_decoded_bbox, train_cfg, num_classes, anchors, cls_scores, bbox_preds,
                        dir_cls_preds, gt_bboxes, gt_labels, assign_per_class=True, dir_offset=0.0, diff_rad_by_sin=True,
                        use_direction_classifier=True, sampling=False, static=False):
    """`GDAnchor3DHead.loss` (/root/reference/mmdet3d_gaussian/models/dense_heads/gd_anchor3d_head.py:167-240) for the one feature
    level the reference's anchor heads have: `anchor_target_3d` (:206-214, mmdet3d's mixin: `anchor_head_get_targets`), the
    normaliser `num_total_samples = num_total_pos` (:221-222, sampling=False) and `loss_single` (:227-238).
    anchors       : the level's grid (H, W, S, R, 7) (what `get_anchors` / `grid_anchors` return with reshape_out=False: one grid for
                    every sample);  cls_scores / bbox_preds / dir_cls_preds: the head's outputs (tensors or one-element lists);
    gt_bboxes, gt_labels : the batch's ground truth (see anchor_head_get_targets);  train_cfg: 'assigner', 'pos_weight',
                    'code_weight', 'decode_weight';  the loss modules as in gd_anchor_head_loss_single.
    ONE read-back for the batch: the per-sample (positives, negatives) counts that make the normaliser.
    static=True   : none at all — the normaliser stays on the device and the loss kernels divide by it themselves (the same bits as the
                    eager form), so with ground truth padded to a fixed number of rows (label -1 for the padding) the whole method
                    is one stream-ordered sequence that `GraphedStep` captures; its backward launches nothing.
    Returns the reference's dict: loss_cls, loss_bbox, loss_dir, each a one-element list."""
    cls_score, bbox_pred = _one(cls_scores, 'cls_scores'), _one(bbox_preds, 'bbox_preds')
    dir_pred = _one(dir_cls_preds, 'dir_cls_preds') if use_direction_classifier else None
    get = (lambda k, d=None: train_cfg.get(k, d)) if isinstance(train_cfg, dict) else (lambda k, d=None: getattr(train_cfg, k, d))
    tg = anchor_head_get_targets(anchors, gt_bboxes, gt_labels, get('assigner'), num_classes, assign_per_class=assign_per_class,
                                 pos_weight=get('pos_weight', -1), dir_offset=dir_offset, sampling=sampling, padded=static)
    labels, label_weights, bbox_targets, bbox_weights, dir_targets, dir_weights = tg[:6]
    flat = anchors.reshape(-1, 7)
    if static:
        # sum_b max(positives_b, 1) stays on the device: the loss kernels divide by it themselves, as the host form divides by the number
        norm = tg[6][:, 0].clamp(min=1).sum().to(torch.float32)
        l_cls, l_bbox, l_dir = gd_anchor_head_loss_single(loss_cls, loss_bbox, loss_dir, loss_decoded_bbox, train_cfg, num_classes,
                                                          cls_score, bbox_pred, dir_pred, labels, label_weights, bbox_targets, bbox_weights,
                                                          dir_targets, dir_weights, flat, norm, diff_rad_by_sin, use_direction_classifier)
    else:
        l_cls, l_bbox, l_dir = gd_anchor_head_loss_single(loss_cls, loss_bbox, loss_dir, loss_decoded_bbox, train_cfg, num_classes,
                                                          cls_score, bbox_pred, dir_pred, labels, label_weights, bbox_targets, bbox_weights,
                                                          dir_targets, dir_weights, flat, float(tg[6]), diff_rad_by_sin,
                                                          use_direction_classifier)
    return dict(loss_cls=[l_cls], loss_bbox=[l_bbox], loss_dir=[l_dir])
