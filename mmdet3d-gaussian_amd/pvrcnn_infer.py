"""The inference slice of `PVRCNNBboxHead` around its NMS (the second call site of `nms_gpu` in the reference):
/root/reference/mmdet3d_gaussian/models/roi_heads/bbox_heads/pvrcnn_bbox_head.py:352-409 (`get_bboxes`) with `multi_class_nms` (:411-480).

The reference decodes the residuals with ~15 elementwise ops, reads the batch size back (`roi_batch_id.max().item()`), then loops over
the samples, and inside `multi_class_nms` over the classes with two host syncs each around `nms_gpu`.  Here: one decode launch
(csrc/coders.hip `roi_decode_kernel`), the class problems of all samples as ONE batched NMS (`nms_gpu_batched` over (sample, class) groups), one read-back.
mmdet3d's `DeltaXYZWLHRBBoxCoder.decode` and `rotation_3d_in_axis` are third party and absent: restated; the rotation's sense changed
between mmdet3d 0.x and 1.0 — `clockwise` selects (default: 1.0's counter-clockwise, which the reference's call without the 0.x
`+ pi / 2` matches).  GPU tensors only.
"""
import torch

from . import _lib
from .iou3d import _thresh_tensor, nms_gpu_batched


def pvrcnn_head_get_bboxes(rois, cls_score, bbox_pred, class_labels, class_pred, cfg, batch_size=None, clockwise=False, return_decoded=False):
    """rois         : (R, 8) [batch_id, x, y, z, dx, dy, dz, yaw] of all samples (sample ids ascending or not);
    cls_score    : (R, 1) or (R,) the head's scores;  bbox_pred: (R, 7) its residuals;
    class_labels : per sample (R_b,) the rois' labels;  class_pred: per sample (R_b, C) the probabilities the NMS ranks by — both in the
                   order of the sample's rois inside `rois` (what `roi_batch_id == b` selects);
    cfg          : test_cfg with `score_thr`, `nms_thr` (numbers or per-class lists) and `use_rotate_nms`;
    batch_size   : number of samples; default len(class_pred) (the reference reads `roi_batch_id.max()` back).
    Returns per sample (boxes (n, 7), scores (n,), labels (n,)): the kept rois class after class, as the reference returns them (its
    `box_type_3d(...)` wrapper aside).  return_decoded=True adds the (R, 7) decoded boxes and (R, 5) NMS rectangles as a fourth item."""
    if not rois.is_cuda:
        raise RuntimeError('pvrcnn_head_get_bboxes: the MI355X implementation has no CPU path')
    if rois.dim() != 2 or rois.shape[1] != 8 or bbox_pred.shape != (rois.shape[0], 7):
        raise RuntimeError(f'pvrcnn_head_get_bboxes: rois {tuple(rois.shape)} / bbox_pred {tuple(bbox_pred.shape)} are not (R, 8) / (R, 7)')
    B = len(class_pred) if batch_size is None else int(batch_size)
    if len(class_pred) != B or len(class_labels) != B:
        raise RuntimeError(f'pvrcnn_head_get_bboxes: {B} samples but {len(class_pred)} class_pred / {len(class_labels)} class_labels entries')
    get = (lambda k: cfg[k]) if isinstance(cfg, dict) else (lambda k: getattr(cfg, k))
    lib = _lib.load()
    dev = rois.device
    R = rois.shape[0]
    r32 = rois.detach()
    r32 = r32 if (r32.dtype == torch.float32 and r32.is_contiguous()) else r32.float().contiguous()
    p32 = bbox_pred.detach()
    p32 = p32 if (p32.dtype == torch.float32 and p32.is_contiguous()) else p32.float().contiguous()
    boxes = torch.empty((R, 7), dtype=torch.float32, device=dev)
    bev = torch.empty((R, 5), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        _lib.check(lib.coder_roi_decode(r32.data_ptr(), 8, 1, p32.data_ptr(), R, int(bool(clockwise)), boxes.data_ptr(), bev.data_ptr(),
                                        torch.cuda.current_stream().cuda_stream), 'coder_roi_decode')
    bid = rois[:, 0].to(torch.int64)
    # the sample's rois in their order inside `rois`: a stable sort by sample id gives, per sample, the global rows in that order
    order = torch.sort(bid, stable=True)[1]
    sizes = [int(p.shape[0]) for p in class_pred]
    if sum(sizes) != R:
        raise RuntimeError(f'pvrcnn_head_get_bboxes: class_pred holds {sum(sizes)} rows for {R} rois')
    # the b-th run of the stable sort must have exactly len(class_pred[b]) rows: equal SUMS with different per-sample counts (or batch
    # ids outside 0..B-1) would pair probabilities with the wrong rois silently, where the reference fails on a shape mismatch
    per_sample = torch.bincount(bid.clamp(min=0), minlength=len(sizes)).tolist() if R else [0] * len(sizes)
    if (R and int(bid.min()) < 0) or per_sample != sizes:
        raise RuntimeError(f'pvrcnn_head_get_bboxes: rois per sample {per_sample} do not match the class_pred sizes {sizes}')
    probs = torch.cat([p.to(dev) for p in class_pred], dim=0).float()           # sample-major = the order of `order`
    labels = torch.cat([l.to(dev) for l in class_labels], dim=0)
    scores = cls_score.reshape(-1)[order]
    boxes_s, bev_s = boxes[order], bev[order]
    # group (sample b, class k) = the rois of sample b (a contiguous range of the sample-major order) with probability >= score_thr[k]:
    # the class problems of every sample as ONE batched NMS (the per-sample, per-class loops of :393-405 and :455-475), one read-back
    C = probs.shape[1]
    st = get('score_thr') if isinstance(get('score_thr'), (list, tuple)) else [get('score_thr')] * C
    nt = list(get('nms_thr')) if isinstance(get('nms_thr'), (list, tuple)) else [get('nms_thr')] * C
    starts = [0]
    for n in sizes:
        starts.append(starts[-1] + n)
    above = probs.t() >= _thresh_tensor(st, C, dev).unsqueeze(1)                       # (C, R)
    pos = torch.arange(R, device=dev)
    lo = torch.tensor(starts[:-1], device=dev).unsqueeze(1)
    hi = torch.tensor(starts[1:], device=dev).unsqueeze(1)
    member = (pos.unsqueeze(0) >= lo) & (pos.unsqueeze(0) < hi)                       # (B, R)
    valid = (member.unsqueeze(1) & above.unsqueeze(0)).reshape(B * C, R)
    rank = probs.t().unsqueeze(0).expand(B, C, R).reshape(B * C, R).contiguous()
    kept = nms_gpu_batched(bev_s, rank, nt * B, valid, normal=not get('use_rotate_nms'))
    out = []
    for b in range(B):
        sel = [k for k in kept[b * C:(b + 1) * C] if k.shape[0] > 0]
        if not sel:                        # nothing kept: the reference indexes with `[]`
            out.append((boxes_s[:0], scores[:0], labels[:0]))
            continue
        g = torch.cat(sel, dim=0) if len(sel) > 1 else sel[0]
        out.append((boxes_s[g], scores[g], labels[g]))
    if return_decoded:
        return out, (boxes, bev)
    return out
