"""Host mirrors of the bbox coders on either side of the loss.

* ``CenterPointBBoxYawCoder`` — /root/reference/mmdet3d_gaussian/core/bbox/coders/centerpoint_bbox_yaw_coders.py:8-56
  (encode :11-16, decode :18-56) on top of ``CenterPointBBoxCoderRev`` (centerpoint_bbox_coders.py:7-21).
  Pinned by tests/golden/coder_center.npz (generated from the real reference classes).
* ``DeltaXYZWLHRBBoxCoder`` — mmdet3d (third party, absent, unpinned); restated from its published formulas; the
  reference calls it at models/dense_heads/gd_anchor3d_head.py:133-136.

These are elementwise glue in torch.  In TRAINING the decode that feeds GDLoss is not executed from here but inside
the fused kernel (head_loss.py -> gd3d_loss_fused_decoded); the torch versions serve inference-time decoding, target
encoding, and as the readable statement of what the kernel prologue computes.
"""
import math

import torch


class CenterPointBBoxYawCoder:
    def __init__(self, pc_range, out_size_factor, voxel_size, code_size=9, norm_bbox=True):
        self.pc_range = pc_range
        self.out_size_factor = out_size_factor
        self.voxel_size = voxel_size
        self.code_size = code_size
        self.norm_bbox = norm_bbox

    def encode(self, target_boxes):
        yaw = target_boxes[..., 6]
        direction = torch.stack((yaw.sin(), yaw.cos()), dim=-1)
        return torch.cat((target_boxes[..., :7], direction, target_boxes[..., 7:]), dim=-1)

    def decode(self, locs, preds, correct_yaw=True):
        x = (preds[..., 0] + locs[..., 0]) * self.out_size_factor * self.voxel_size[0] + self.pc_range[0]
        y = (preds[..., 1] + locs[..., 1]) * self.out_size_factor * self.voxel_size[1] + self.pc_range[1]
        z = preds[..., 2]
        dim = preds[..., 3:6]
        if self.norm_bbox:
            dim = dim.exp()
        yaw = preds[..., 6]
        if correct_yaw:
            with torch.no_grad():
                direction = torch.atan2(preds[..., 7], preds[..., 8])
                num_rot90 = torch.floor((direction - yaw) / (math.pi / 2) + 0.5)
                no_swap_wh = (num_rot90.long() % 2 == 0)
            yaw = yaw + num_rot90 * (math.pi / 2)
            dim = dim.where(no_swap_wh.unsqueeze(-1), dim[..., [1, 0, 2]])
        return torch.cat((x.unsqueeze(-1), y.unsqueeze(-1), z.unsqueeze(-1), dim, yaw.unsqueeze(-1), preds[..., 9:]),
                         dim=-1)


class DeltaXYZWLHRBBoxCoder:
    """mmdet3d anchor-delta coder for (x, y, z, w, l, h, r) boxes (code_size 7)."""

    def __init__(self, code_size=7):
        self.box_dim = code_size

    @staticmethod
    def encode(src_boxes, dst_boxes):
        xa, ya, za, wa, la, ha, ra = src_boxes[..., :7].unbind(-1)
        xg, yg, zg, wg, lg, hg, rg = dst_boxes[..., :7].unbind(-1)
        za = za + ha / 2
        zg = zg + hg / 2
        diagonal = torch.sqrt(la ** 2 + wa ** 2)
        return torch.stack(((xg - xa) / diagonal, (yg - ya) / diagonal, (zg - za) / ha, torch.log(wg / wa),
                            torch.log(lg / la), torch.log(hg / ha), rg - ra), dim=-1)

    @staticmethod
    def decode(anchors, deltas):
        xa, ya, za, wa, la, ha, ra = anchors[..., :7].unbind(-1)
        xt, yt, zt, wt, lt, ht, rt = deltas[..., :7].unbind(-1)
        za = za + ha / 2
        diagonal = torch.sqrt(la ** 2 + wa ** 2)
        hg = torch.exp(ht) * ha
        return torch.stack((xt * diagonal + xa, yt * diagonal + ya, zt * ha + za - hg / 2, torch.exp(wt) * wa,
                            torch.exp(lt) * la, hg, rt + ra), dim=-1)
