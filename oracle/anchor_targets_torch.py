"""oracle/anchor_targets_torch.py — TEST INFRASTRUCTURE, NOT PRODUCT CODE.

CPU torch restatement of the anchor heads' target assignment, which the reference's GDAnchor3DHead calls at
/root/reference/mmdet3d_gaussian/models/dense_heads/gd_anchor3d_head.py:206-214 (`self.anchor_target_3d(...)`) and INHERITS
from mmdet3d / mmdet (third party, absent here, versions not pinned by the reference; restated from the published mmdet3d 0.x /
mmdet 2.x text — PARITY UNPINNED):
  AnchorTrainMixin.anchor_target_3d / anchor_target_3d_single / anchor_target_single_assigner   mmdet3d/models/dense_heads/train_mixins.py
  get_direction_target                                                                          (same file)
  MaxIoUAssigner.assign / assign_wrt_overlaps                                                   mmdet/core/bbox/assigners/max_iou_assigner.py
  PseudoSampler.sample                                                                          mmdet/core/bbox/samplers/pseudo_sampler.py
  BboxOverlapsNearest3D, LiDARInstance3DBoxes.nearest_bev, limit_period                         mmdet3d/core/bbox/iou_calculators, structures
  bbox_overlaps (mode 'iou', eps 1e-6)                                                          mmdet/core/bbox/iou_calculators/iou2d_calculator.py
  DeltaXYZWLHRBBoxCoder.encode                                                                  mmdet3d/core/bbox/coders
for the configuration the reference ships (configs/_base_/models/hv_pointpillars_secfpn_kitti.py:39-92, ..._waymo.py): one
feature level, anchors (H, W, S sizes, R rotations, 7) with reshape_out=False, a list of S MaxIoUAssigners, assign_per_class,
no ignore boxes, sampling=False (PseudoSampler).  Divisions by Python scalars are true divisions (the CPU kernels'; the CUDA
kernels multiply by the reciprocal).  Never imported by the product package."""
import math

import torch


def limit_period(val, offset=0.5, period=math.pi):
    return val - torch.floor(val / period + offset) * period


def nearest_bev(boxes):
    bev = boxes[:, [0, 1, 3, 4, 6]]
    normed = torch.abs(limit_period(bev[:, -1], 0.5, math.pi))
    turned = (normed > math.pi / 4)[..., None]
    xywh = torch.where(turned, bev[:, [0, 1, 3, 2]], bev[:, :4])
    centers, dims = xywh[:, :2], xywh[:, 2:]
    return torch.cat([centers - dims / 2, centers + dims / 2], dim=-1)


def bbox_overlaps(b1, b2, eps=1e-6):
    area1 = (b1[:, 2] - b1[:, 0]) * (b1[:, 3] - b1[:, 1])
    area2 = (b2[:, 2] - b2[:, 0]) * (b2[:, 3] - b2[:, 1])
    lt = torch.max(b1[:, None, :2], b2[None, :, :2])
    rb = torch.min(b1[:, None, 2:], b2[None, :, 2:])
    wh = (rb - lt).clamp(min=0)
    overlap = wh[..., 0] * wh[..., 1]
    union = area1[:, None] + area2[None, :] - overlap
    union = torch.max(union, union.new_tensor([eps]))
    return overlap / union


def max_iou_assign(overlaps, gt_labels, pos_iou_thr, neg_iou_thr, min_pos_iou, match_low_quality=True, gt_max_assign_all=True):
    """assign_wrt_overlaps for num_gts > 0: overlaps (G, N) -> assigned_gt_inds (N,): -1 ignore, 0 negative, g + 1 positive"""
    num_gts, num_bboxes = overlaps.shape
    assigned = overlaps.new_full((num_bboxes,), -1, dtype=torch.long)
    max_overlaps, argmax_overlaps = overlaps.max(dim=0)
    gt_max_overlaps, gt_argmax_overlaps = overlaps.max(dim=1)
    assigned[(max_overlaps >= 0) & (max_overlaps < neg_iou_thr)] = 0
    pos = max_overlaps >= pos_iou_thr
    assigned[pos] = argmax_overlaps[pos] + 1
    if match_low_quality:
        for i in range(num_gts):
            if gt_max_overlaps[i] >= min_pos_iou:
                if gt_max_assign_all:
                    assigned[overlaps[i, :] == gt_max_overlaps[i]] = i + 1
                else:
                    assigned[gt_argmax_overlaps[i]] = i + 1
    return assigned


def delta_encode(src, dst):
    xa, ya, za, wa, la, ha, ra = torch.split(src, 1, dim=-1)
    xg, yg, zg, wg, lg, hg, rg = torch.split(dst, 1, dim=-1)
    za = za + ha / 2
    zg = zg + hg / 2
    diagonal = torch.sqrt(la ** 2 + wa ** 2)
    xt = (xg - xa) / diagonal
    yt = (yg - ya) / diagonal
    zt = (zg - za) / ha
    lt = torch.log(lg / la)
    wt = torch.log(wg / wa)
    ht = torch.log(hg / ha)
    rt = rg - ra
    return torch.cat([xt, yt, zt, wt, lt, ht, rt], dim=-1)


def get_direction_target(anchors, reg_targets, dir_offset=0.0, num_bins=2):
    rot_gt = reg_targets[..., 6] + anchors[..., 6]
    offset_rot = limit_period(rot_gt - dir_offset, 0, 2 * math.pi)
    dir_cls_targets = torch.floor(offset_rot / (2 * math.pi / num_bins)).long()
    return torch.clamp(dir_cls_targets, min=0, max=num_bins - 1)


def single_assigner(cfg, anchors, gt_bboxes, gt_labels, num_classes, pos_weight, dir_offset):
    """anchor_target_single_assigner with a PseudoSampler: anchors (n, 7), gt_bboxes (g, 7), gt_labels (g,)"""
    n = anchors.shape[0]
    bbox_targets, bbox_weights = torch.zeros_like(anchors), torch.zeros_like(anchors)
    dir_targets = anchors.new_zeros(n, dtype=torch.long)
    dir_weights = anchors.new_zeros(n, dtype=torch.float)
    labels = anchors.new_zeros(n, dtype=torch.long)
    label_weights = anchors.new_zeros(n, dtype=torch.float)
    if len(gt_bboxes) > 0:
        overlaps = bbox_overlaps(nearest_bev(gt_bboxes), nearest_bev(anchors))
        assigned = max_iou_assign(overlaps, gt_labels, cfg['pos_iou_thr'], cfg['neg_iou_thr'], cfg['min_pos_iou'],
                                  cfg.get('match_low_quality', True), cfg.get('gt_max_assign_all', True))
        pos_inds = torch.nonzero(assigned > 0, as_tuple=False).squeeze(-1).unique()
        neg_inds = torch.nonzero(assigned == 0, as_tuple=False).squeeze(-1).unique()
    else:
        pos_inds = torch.zeros(0, dtype=torch.long, device=anchors.device)
        neg_inds = torch.arange(n, device=anchors.device)
    labels += num_classes
    if len(pos_inds) > 0:
        gt_of = assigned[pos_inds] - 1
        pos_bbox_targets = delta_encode(anchors[pos_inds], gt_bboxes[gt_of])
        bbox_targets[pos_inds, :] = pos_bbox_targets
        bbox_weights[pos_inds, :] = 1.0
        dir_targets[pos_inds] = get_direction_target(anchors[pos_inds], pos_bbox_targets, dir_offset)
        dir_weights[pos_inds] = 1.0
        labels[pos_inds] = gt_labels[gt_of]
        label_weights[pos_inds] = 1.0 if pos_weight <= 0 else pos_weight
    if len(neg_inds) > 0:
        label_weights[neg_inds] = 1.0
    return labels, label_weights, bbox_targets, bbox_weights, dir_targets, dir_weights, pos_inds, neg_inds


def anchor_target_3d_single(anchors, gt_bboxes, gt_labels, assigners, num_classes, assign_per_class=True, pos_weight=-1, dir_offset=0.0):
    """anchors (H, W, S, R, 7) (a leading 1 allowed); one assigner config per size class, or a single dict for all anchors"""
    if isinstance(assigners, dict):
        return single_assigner(assigners, anchors.reshape(-1, 7), gt_bboxes, gt_labels, num_classes, pos_weight, dir_offset)
    S, R = anchors.shape[-3], anchors.shape[-2]
    assert len(assigners) == S
    feat = anchors.numel() // (S * R * 7)
    parts = [[] for _ in range(8)]
    for i, cfg in enumerate(assigners):
        cur = anchors[..., i, :, :].reshape(-1, 7)
        if assign_per_class:
            m = gt_labels == i
            res = single_assigner(cfg, cur, gt_bboxes[m, :], gt_labels[m], num_classes, pos_weight, dir_offset)
        else:
            res = single_assigner(cfg, cur, gt_bboxes, gt_labels, num_classes, pos_weight, dir_offset)
        for k in (0, 1, 4, 5):
            parts[k].append(res[k].reshape(feat, 1, R))
        for k in (2, 3):
            parts[k].append(res[k].reshape(feat, 1, R, 7))
        parts[6].append(res[6])
        parts[7].append(res[7])
    out = [torch.cat(parts[k], dim=-2).reshape(-1) for k in (0, 1)]
    out += [torch.cat(parts[k], dim=-3).reshape(-1, 7) for k in (2, 3)]
    out += [torch.cat(parts[k], dim=-2).reshape(-1) for k in (4, 5)]
    out += [torch.cat(parts[6]).reshape(-1), torch.cat(parts[7]).reshape(-1)]
    return tuple(out)


def anchor_target_3d(anchors, gt_bboxes_list, gt_labels_list, assigners, num_classes, assign_per_class=True, pos_weight=-1, dir_offset=0.0):
    """one level: returns the stacked (B, N[, 7]) targets and (num_total_pos, num_total_neg) as the mixin counts them"""
    res = [anchor_target_3d_single(anchors, b, l, assigners, num_classes, assign_per_class, pos_weight, dir_offset)
           for b, l in zip(gt_bboxes_list, gt_labels_list)]
    num_total_pos = sum(max(r[6].numel(), 1) for r in res)
    num_total_neg = sum(max(r[7].numel(), 1) for r in res)
    return tuple(torch.stack([r[k] for r in res], 0) for k in range(6)) + (num_total_pos, num_total_neg)


def range_anchors(feature_size, ranges, sizes, rotations):
    """Anchor3DRangeGenerator.grid_anchors for one level with reshape_out=False: (1, H, W, S, R, 7) [x, y, z, size..., rot]"""
    per = []
    for rng, size in zip(ranges, sizes):
        z = torch.linspace(rng[2], rng[5], 1)
        y = torch.linspace(rng[1], rng[4], feature_size[0])
        x = torch.linspace(rng[0], rng[3], feature_size[1])
        rot = torch.tensor(rotations, dtype=torch.float32)
        rets = list(torch.meshgrid(x, y, z, rot, indexing='ij'))
        rets = [r.unsqueeze(-2).unsqueeze(-1) for r in rets]                     # (X, Y, Z, 1, R, 1)
        sz = torch.tensor(size, dtype=torch.float32).reshape(1, 1, 1, 1, 1, 3).repeat(*rets[0].shape[:3], 1, rets[0].shape[4], 1)
        rets.insert(3, sz)
        per.append(torch.cat(rets, dim=-1).permute(2, 1, 0, 3, 4, 5))            # (Z, Y, X, 1, R, 7)
    return torch.cat(per, dim=-3)
