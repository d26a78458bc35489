"""oracle/coder_torch.py — TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Elementwise torch restatement of CenterPointBBoxYawCoder
(/root/reference/mmdet3d_gaussian/core/bbox/coders/centerpoint_bbox_yaw_coders.py:11-16 encode, :18-56 decode; base class
centerpoint_bbox_coders.py:7-21), PINNED bit for bit by tests/golden/coder_center.npz, which
tests/golden/make_golden_coder.py generated from the real reference classes (tests/test_coders_cpu.py).
It is the checker of the device coder (mmdet3d-gaussian_amd/csrc/coders.hip); autograd provides the reference backward.
Never imported by the product package.
"""
import math

import torch


def center_encode(boxes):
    yaw = boxes[..., 6]
    return torch.cat((boxes[..., :7], yaw.sin().unsqueeze(-1), yaw.cos().unsqueeze(-1), boxes[..., 7:]), dim=-1)


def center_decode(locs, preds, pc_range, out_size_factor, voxel_size, norm_bbox=True, correct_yaw=True):
    xy = [(preds[..., k] + locs[..., k]) * out_size_factor * voxel_size[k] + pc_range[k] for k in (0, 1)]
    dim = preds[..., 3:6].exp() if norm_bbox else preds[..., 3:6]
    yaw = preds[..., 6]
    if correct_yaw:
        with torch.no_grad():
            quarter_turns = torch.floor((torch.atan2(preds[..., 7], preds[..., 8]) - yaw) / (math.pi / 2) + 0.5)
            odd = quarter_turns.long() % 2 != 0
        yaw = yaw + quarter_turns * (math.pi / 2)
        dim = torch.where(odd.unsqueeze(-1), dim[..., [1, 0, 2]], dim)
    cols = [xy[0].unsqueeze(-1), xy[1].unsqueeze(-1), preds[..., 2:3], dim, yaw.unsqueeze(-1), preds[..., 9:]]
    return torch.cat(cols, dim=-1)


def point_decode(priors, preds, correct_yaw=True):
    """PointBBoxYawCoder.decode (/root/reference/mmdet3d_gaussian/core/bbox/coders/point_bbox_yaw_coders.py:19-52), PINNED bit for
    bit by tests/golden/coder_point.npz (tests/golden/make_golden_coder_point.py, the real class); encode (:12-16) has the
    statements of center_encode."""
    scale = priors[..., 2]
    x = preds[..., 0] * scale + priors[..., 0]
    y = preds[..., 1] * scale + priors[..., 1]
    e = preds[..., 3:6].exp()
    dim = torch.stack((e[..., 0] * scale, e[..., 1] * scale, e[..., 2]), dim=-1)
    yaw = preds[..., 6]
    if correct_yaw:
        with torch.no_grad():
            quarter_turns = torch.floor((torch.atan2(preds[..., 7], preds[..., 8]) - yaw) / (math.pi / 2) + 0.5)
            odd = quarter_turns.long() % 2 != 0
        yaw = yaw + quarter_turns * (math.pi / 2)
        dim = torch.where(odd.unsqueeze(-1), dim[..., [1, 0, 2]], dim)
    return torch.cat((x.unsqueeze(-1), y.unsqueeze(-1), preds[..., 2:3], dim, yaw.unsqueeze(-1), preds[..., 9:]), dim=-1)
