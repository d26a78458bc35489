"""oracle/pvrcnn_torch.py — TEST INFRASTRUCTURE, NOT PRODUCT CODE.

CPU torch restatement of the inference slice of the reference's PVRCNNBboxHead:
  get_bboxes        /root/reference/mmdet3d_gaussian/models/roi_heads/bbox_heads/pvrcnn_bbox_head.py:352-409
  multi_class_nms   pvrcnn_bbox_head.py:411-480
restated statement by statement from the reference's text (the head class needs mmdet3d / mmcv to import), with the third-party helpers
it calls (mmdet3d, absent, version not pinned — PARITY UNPINNED):
  DeltaXYZWLHRBBoxCoder.decode, rotation_3d_in_axis (axis 2: counter-clockwise in mmdet3d 1.0, the transpose in 0.x — `clockwise`),
  LiDARInstance3DBoxes.bev = columns [0, 1, 3, 4, 6], xywhr2xyxyr, and nms_gpu / nms_normal_gpu = oracle/rbox_oracle.c part 1.
Never imported by the product package."""
import numpy as np
import torch


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


def rotation_3d_in_axis_z(points, angles, clockwise=False):
    """points (N, M, 3), angles (N,): rot_mat_T rows as mmdet3d 1.0 stacks them for axis 2, transposed for `clockwise`"""
    rot_sin, rot_cos = torch.sin(angles), torch.cos(angles)
    ones, zeros = torch.ones_like(rot_cos), torch.zeros_like(rot_cos)
    rot_mat_T = torch.stack([torch.stack([rot_cos, rot_sin, zeros]), torch.stack([-rot_sin, rot_cos, zeros]), torch.stack([zeros, zeros, ones])])
    if clockwise:
        rot_mat_T = rot_mat_T.transpose(0, 1)
    return torch.einsum('aij,jka->aik', points, rot_mat_T)


def xywhr2xyxyr(b):
    hw, hh = b[:, 2] / 2, b[:, 3] / 2
    return torch.stack((b[:, 0] - hw, b[:, 1] - hh, b[:, 0] + hw, b[:, 1] + hh, b[:, 4]), dim=-1)


def decode_rois(rois, bbox_pred, clockwise=False):
    """:376-386"""
    roi_boxes = rois[..., 1:]
    roi_ry = roi_boxes[..., 6].view(-1)
    roi_xyz = roi_boxes[..., 0:3].view(-1, 3)
    local = roi_boxes.clone().detach()
    local[..., 0:3] = 0
    boxes = delta_decode(local, bbox_pred)
    boxes[..., 0:3] = rotation_3d_in_axis_z(boxes[..., 0:3].unsqueeze(1), roi_ry, clockwise).squeeze(1)
    boxes[:, 0:3] += roi_xyz
    return boxes


def multi_class_nms(box_probs, box_preds, score_thr, nms_thr, use_rotate_nms=True):
    """:438-480 -> selected indices (LongTensor) or []"""
    from . import nms_gpu_oracle
    C = box_probs.shape[1]
    boxes_for_nms = xywhr2xyxyr(box_preds[:, [0, 1, 3, 4, 6]])
    st = score_thr if isinstance(score_thr, list) else [score_thr] * C
    nt = nms_thr if isinstance(nms_thr, list) else [nms_thr] * C
    selected_list = []
    for k in range(C):
        keep = box_probs[:, k] >= st[k]
        if keep.int().sum() > 0:
            original = keep.nonzero(as_tuple=False).view(-1)
            sel = nms_gpu_oracle(boxes_for_nms[keep].numpy(), box_probs[keep, k].numpy(), nt[k], normal=not use_rotate_nms)
            sel = torch.as_tensor(np.asarray(sel, dtype=np.int64))
            if sel.shape[0] == 0:
                continue
            selected_list.append(original[sel])
    return torch.cat(selected_list, dim=0) if len(selected_list) > 0 else []


def get_bboxes(rois, cls_score, bbox_pred, class_labels, class_pred, cfg, clockwise=False, decoded=None):
    """:352-409; `decoded` replaces the decode's output (stage-wise tests: the NMS then sees the boxes it is given, bit for bit)"""
    roi_batch_id = rois[..., 0]
    batch_size = int(roi_batch_id.max().item() + 1)
    rcnn = decode_rois(rois, bbox_pred, clockwise) if decoded is None else decoded
    out = []
    for b in range(batch_size):
        m = roi_batch_id == b
        cur_boxes = rcnn[m]
        cur_score = cls_score[m].view(-1)
        sel = multi_class_nms(class_pred[b], cur_boxes, cfg['score_thr'], cfg['nms_thr'], cfg['use_rotate_nms'])
        out.append((cur_boxes[sel], cur_score[sel], class_labels[b][sel]))
    return out
