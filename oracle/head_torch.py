"""oracle/head_torch.py — TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Op-for-op PyTorch restatement (autograd does the backward) of the regression part of
GDAnchor3DHead.loss_single, /root/reference/mmdet3d_gaussian/models/dense_heads/gd_anchor3d_head.py:95-161:

    permute/reshape (:95-99) -> pos_inds = nonzero(0 <= label < C) (:101-105) -> gathers (:107-112)
    -> code_weight / decode_weight = pos_bbox_weights * new_tensor(cfg) (:124-131)
    -> loss_decoded_bbox(decode(anchors, pred), decode(anchors, target), decode_weight, avg_factor) (:133-141)
    -> add_sin_difference (:150-152) -> += loss_bbox(pred', target', code_weight, avg_factor) (:153-157)
    -> no positives: pos_bbox_pred.sum() (:160)

Third-party pieces, absent from /root/reference and unpinned (no requirements file), restated from their published
formulas — PARITY UNPINNED for these three, the Gaussian term is pinned through oracle/gd_torch.py's own golden tests:
  * mmdet3d DeltaXYZWLHRBBoxCoder.decode            (mmdet3d/core/bbox/coders/delta_xyzwhlr_bbox_coder.py)
  * mmdet3d Anchor3DHead.add_sin_difference         (mmdet3d/models/dense_heads/anchor3d_head.py)
  * mmdet  smooth_l1_loss / l1_loss + weighted_loss (mmdet/models/losses/smooth_l1_loss.py)
Never imported by the product package.
"""
import torch

from . import gd_torch


def delta_decode(anchors, deltas):
    xa, ya, za, wa, la, ha, ra = anchors.unbind(-1)
    xt, yt, zt, wt, lt, ht, rt = deltas.unbind(-1)
    za = za + ha / 2
    diagonal = torch.sqrt(la ** 2 + wa ** 2)
    xg, yg, zg = xt * diagonal + xa, yt * diagonal + ya, zt * ha + za
    lg, wg, hg = torch.exp(lt) * la, torch.exp(wt) * wa, torch.exp(ht) * ha
    return torch.stack([xg, yg, zg - hg / 2, wg, lg, hg, rt + ra], -1)


def add_sin_difference(boxes1, boxes2):
    rad_pred = torch.sin(boxes1[..., 6:7]) * torch.cos(boxes2[..., 6:7])
    rad_tg = torch.cos(boxes1[..., 6:7]) * torch.sin(boxes2[..., 6:7])
    return (torch.cat([boxes1[..., :6], rad_pred, boxes1[..., 7:]], -1),
            torch.cat([boxes2[..., :6], rad_tg, boxes2[..., 7:]], -1))


def smooth_l1(pred, target, weight, avg_factor, beta, loss_weight):
    """mmdet SmoothL1Loss (beta > 0) / L1Loss (beta == 0), reduction='mean' with avg_factor -> sum / avg_factor."""
    diff = torch.abs(pred - target)
    loss = torch.where(diff < beta, 0.5 * diff * diff / beta, diff - 0.5 * beta) if beta > 0 else diff
    if weight is not None:
        loss = loss * weight
    return loss_weight * loss.sum() / avg_factor


def loss_single_bbox(bbox_pred, bbox_targets, bbox_weights, labels, anchor_list, num_classes, num_total_samples,
                     gd=None, sl1=None, code_weight=None, decode_weight=None, diff_rad_by_sin=True):
    """gd: dict(loss_type, fun, tau, alpha, loss_weight, **kw) or None; sl1: dict(beta, loss_weight) or None.
    Returns loss_bbox (0-dim, attached to bbox_pred)."""
    B = bbox_pred.shape[0]
    bp = bbox_pred.permute(0, 2, 3, 1).reshape(-1, 7)
    bt = bbox_targets.reshape(-1, 7)
    bw = bbox_weights.reshape(-1, 7)
    labels = labels.reshape(-1)
    pos = ((labels >= 0) & (labels < num_classes)).nonzero(as_tuple=False).reshape(-1)
    pp, pt, pw = bp[pos], bt[pos], bw[pos]
    if len(pos) == 0:
        return pp.sum()
    anchors = anchor_list.reshape(-1, 7).repeat(B, 1)[pos]
    cw = pw * pw.new_tensor(code_weight) if code_weight else None
    dw = pw * pw.new_tensor(decode_weight) if decode_weight else None
    loss = bp.new_zeros(())
    if gd is not None:
        gd = dict(gd)
        loss = loss + gd_torch.gd_loss(delta_decode(anchors, pp), delta_decode(anchors, pt), gd.pop('loss_type'), weight=dw,
                                       avg_factor=num_total_samples, **gd)
    if sl1 is not None:
        if diff_rad_by_sin:
            pp, pt = add_sin_difference(pp, pt)
        loss = loss + smooth_l1(pp, pt, cw, num_total_samples, sl1['beta'], sl1['loss_weight'])
    return loss


def center_head_task_losses(preds, pos_ind, anno, num_pos, coder_cfg, gd, l1_loss_weight, code_weights):
    """One task of CenterGDHead.loss (gd_centerpoint_head.py:409-434) op for op, autograd for the backward.
    preds: dict of (B,c,H,W) maps ('reg' optional, 'vel' optional); pos_ind (n,3) long [b,x,y]; anno (n, 7|9);
    coder_cfg: dict(pc_range, out_size_factor, voxel_size, norm_bbox) — CenterPointBBoxYawCoder.encode/decode restated
    (centerpoint_bbox_yaw_coders.py:11-31, correct_yaw=False); gd: dict(loss_type, loss_weight, **kw).
    Returns (loss_l1, loss_gd)."""
    parts = []
    if 'reg' in preds:
        parts.append(preds['reg'])
    else:
        b, _, h, w = preds['height'].shape
        parts.append(preds['height'].new_full((b, 2, h, w), 0.5))                       # :377-378
    parts += [preds['height'], preds['dim'], preds['yaw'], preds['dir']]
    if 'vel' in preds:
        parts.append(preds['vel'])
    pred = torch.cat(parts, dim=1)                                                     # _reconstruct_bbox :372-387
    pred = pred[pos_ind[:, 0], :, pos_ind[:, 2], pos_ind[:, 1]]                        # _gather_feat :59-63
    yaw = anno[..., 6]
    target_box = torch.cat((anno[..., :7], torch.stack((yaw.sin(), yaw.cos()), -1), anno[..., 7:]), -1)   # encode
    target_l1, target_gd = target_box[..., 7:], target_box[..., :7]
    locs = pos_ind[:, 1:].to(pred.dtype)
    osf, vs, pc = coder_cfg['out_size_factor'], coder_cfg['voxel_size'], coder_cfg['pc_range']
    x = (pred[..., 0] + locs[..., 0]) * osf * vs[0] + pc[0]
    y = (pred[..., 1] + locs[..., 1]) * osf * vs[1] + pc[1]
    dim = pred[..., 3:6].exp() if coder_cfg['norm_bbox'] else pred[..., 3:6]
    pred_gd = torch.cat((x[:, None], y[:, None], pred[..., 2:3], dim, pred[..., 6:7]), -1)
    pred_l1 = pred[..., 7:]
    w = target_l1.new_tensor(code_weights).unsqueeze(0).expand_as(target_l1)
    avg = max(num_pos, 1)
    if target_box.numel() == 0:
        z = pred.new_zeros((1,))
        return z, z
    loss_l1 = l1_loss_weight * ((pred_l1 - target_l1).abs() * w).sum() / avg           # mmdet L1Loss, mean + avg_factor
    gd = dict(gd)
    loss_gd = gd_torch.gd_loss(pred_gd, target_gd, gd.pop('loss_type'), avg_factor=avg, **gd)
    return loss_l1, loss_gd
