"""oracle/heat_focal_torch.py — TEST INFRASTRUCTURE, NOT PRODUCT CODE.

The heat-map loss term of CenterGDHead.loss (/root/reference/mmdet3d_gaussian/models/dense_heads/gd_centerpoint_head.py:403-411)
as plain torch ops with autograd.  The two functions it calls are third party and absent here (mmdet3d `clip_sigmoid`, mmdet 2.x
`GaussianFocalLoss` / `gaussian_focal_loss` with `weighted_loss`): restated from their published text — PARITY UNPINNED.
Never imported by the product package."""
import torch


def clip_sigmoid(x, eps=1e-4):
    return torch.clamp(x.sigmoid(), min=eps, max=1 - eps)


def gaussian_focal_loss(pred, gaussian_target, alpha=2.0, gamma=4.0):
    eps = 1e-12
    pos_weights = gaussian_target.eq(1)
    neg_weights = (1 - gaussian_target).pow(gamma)
    pos_loss = -(pred + eps).log() * (1 - pred).pow(alpha) * pos_weights
    neg_loss = -(1 - pred + eps).log() * pred.pow(alpha) * neg_weights
    return pos_loss + neg_loss


def heatmap_loss(logits, target, alpha=2.0, gamma=4.0, loss_weight=1.0):
    """:405-411 for one task: returns (loss, num_pos)"""
    pred = clip_sigmoid(logits)
    num_pos = target.eq(1).float().sum().item()
    loss = loss_weight * gaussian_focal_loss(pred, target, alpha, gamma).sum() / max(num_pos, 1)
    return loss, num_pos
