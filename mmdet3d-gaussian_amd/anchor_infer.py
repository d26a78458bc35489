"""Anchor-head inference slice on the device: head maps of a batch -> detections per sample (csrc/anchor_infer.hip + the selection
kernel of csrc/center_infer.hip + the batched NMS of csrc/rbox.hip), one host read-back.

The reference's GDAnchor3DHead inherits its inference from mmdet3d (`class GDAnchor3DHead(Anchor3DHead)`,
/root/reference/mmdet3d_gaussian/models/dense_heads/gd_anchor3d_head.py:10): this is the call surface of mmdet3d's
`Anchor3DHead.get_bboxes` for the PointPillars / SECOND heads of the reference's configs (sigmoid classification, box code size 7,
DeltaXYZWLHRBBoxCoder, direction classifier), with `box3d_multiclass_nms` inside.  Third-party semantics, restated (unpinned).
GPU tensors only: there is no CPU path.
"""
import ctypes

import torch

from . import _lib


def _f32c(t):
    if t.dtype != torch.float32:
        t = t.float()
    return t if t.is_contiguous() else t.contiguous()


def anchor_head_get_bboxes(cls_scores, bbox_preds, dir_cls_preds, mlvl_anchors, cfg, num_classes, dir_offset=0.0, dir_limit_offset=1.0,
                           input_metas=None, box_code_size=7, padded=False, return_candidates=False):
    """cls_scores / bbox_preds / dir_cls_preds : per level the head's outputs for the WHOLE batch, (B, A*C, H, W), (B, A*7, H, W),
                     (B, A*2, H, W) — what `Anchor3DHead.forward` returns;
    mlvl_anchors   : per level the anchors (H*W*A, 7) (or any shape that reshapes to it), in the head's order (h, w, a);
    cfg            : the head's test_cfg: nms_pre, score_thr, nms_thr, max_num, use_rotate_nms;
    dir_offset, dir_limit_offset : the head's attributes (KITTI configs: 0.7854 / 0).
    Returns per sample (bboxes (n,7), scores (n,), labels (n,) int64) — wrapped by input_metas[i]['box_type_3d'](bboxes, box_dim=7)
    when metas are given.  padded=True: no read-back, dict(bboxes (B,max_num,7), scores, labels, counts) on the device
    (counts[b] = -1: a device-side NMS scan gave up, the sample's rows are void)."""
    if box_code_size != 7:
        raise RuntimeError('anchor_head_get_bboxes: box code size 7 only')
    L = len(cls_scores)
    if not (len(bbox_preds) == len(dir_cls_preds) == len(mlvl_anchors) == L) or L == 0 or L > 4:
        raise RuntimeError('anchor_head_get_bboxes: 1..4 levels, one entry per level in every list')
    if not cls_scores[0].is_cuda:
        raise RuntimeError('anchor_head_get_bboxes: the MI355X implementation has no CPU path')
    lib = _lib.load_extras()
    dev = cls_scores[0].device
    C = int(num_classes)
    B = cls_scores[0].shape[0]
    if cls_scores[0].shape[1] % C:
        raise RuntimeError(f'anchor_head_get_bboxes: {cls_scores[0].shape[1]} class channels are no multiple of {C} classes')
    A = cls_scores[0].shape[1] // C
    levels = (_lib.AnchorInferLevel * L)()
    keep = []
    for l in range(L):
        cs, bp, dp = _f32c(cls_scores[l]), _f32c(bbox_preds[l]), _f32c(dir_cls_preds[l])
        H, W = cs.shape[2], cs.shape[3]
        an = _f32c(mlvl_anchors[l].to(dev).reshape(-1, 7))
        if tuple(cs.shape) != (B, A * C, H, W) or tuple(bp.shape) != (B, A * 7, H, W) or tuple(dp.shape) != (B, A * 2, H, W) or an.shape[0] != H * W * A:
            raise RuntimeError(f'level {l}: cls {tuple(cs.shape)}, bbox {tuple(bp.shape)}, dir {tuple(dp.shape)}, anchors {tuple(an.shape)} '
                               f'do not describe B={B}, A={A}, C={C}, H={H}, W={W}')
        keep += [cs, bp, dp, an]
        levels[l].cls_score, levels[l].bbox_pred, levels[l].dir_cls_pred, levels[l].anchors = cs.data_ptr(), bp.data_ptr(), dp.data_ptr(), an.data_ptr()
        levels[l].height, levels[l].width = H, W
    get = (lambda k, d=None: cfg.get(k, d)) if hasattr(cfg, 'get') else (lambda k, d=None: getattr(cfg, k, d))
    d = _lib.AnchorInferDesc()
    d.num_levels, d.batch, d.num_anchors, d.num_classes = L, B, A, C
    d.nms_pre = int(get('nms_pre', -1))
    d.max_num = int(get('max_num'))
    d.use_rotate_nms = int(bool(get('use_rotate_nms', True)))
    d.score_thr, d.nms_thr = float(get('score_thr', 0)), float(get('nms_thr'))
    d.dir_offset, d.dir_limit_offset = float(dir_offset), float(dir_limit_offset)
    d.levels = levels
    offs = (ctypes.c_int64 * 3)()
    K = int(lib.anchor_infer_candidates(ctypes.byref(d), offs))
    if K < 0:
        raise RuntimeError('anchor_head_get_bboxes: unsupported configuration (nms_pre above 4096 per level, more than 16384 candidates '
                           'in total, more than 16 classes, or max_num < 1)')
    M = d.max_num
    with torch.cuda.device(dev):
        ws = torch.empty(lib.anchor_infer_workspace_bytes(ctypes.byref(d)), dtype=torch.uint8, device=dev)
        boxes = torch.empty((B, M, 7), dtype=torch.float32, device=dev)
        scores = torch.empty((B, M), dtype=torch.float32, device=dev)
        labels = torch.empty((B, M), dtype=torch.int64, device=dev)
        count = torch.empty(B, dtype=torch.int64, device=dev)
        _lib.check(lib.anchor_infer_bboxes(ctypes.byref(d), ws.data_ptr(), boxes.data_ptr(), scores.data_ptr(), labels.data_ptr(),
                                           count.data_ptr(), torch.cuda.current_stream().cuda_stream), 'anchor_infer_bboxes')
    cands = None
    if return_candidates:
        cands = dict(boxes=ws[offs[0]:offs[0] + 4 * B * K * 7].view(torch.float32).view(B, K, 7),
                     scores=ws[offs[1]:offs[1] + 4 * B * C * K].view(torch.float32).view(B, C, K),
                     dirs=ws[offs[2]:offs[2] + 4 * B * K].view(torch.int32).view(B, K))
    if padded:
        out = dict(bboxes=boxes, scores=scores, labels=labels, counts=count)
        return (out, cands) if return_candidates else out
    ns = count.tolist()          # the one sync: B data-dependent detection counts
    if min(ns, default=0) < 0:   # a device-side NMS scan gave up (include/gd3d.h: num_keep = -1): the result is void
        raise RuntimeError('anchor_infer: a device-side NMS scan gave up (count -1); the result is void')
    out = []
    for i in range(B):
        bx = boxes[i, :ns[i]]
        if input_metas is not None:
            bx = input_metas[i]['box_type_3d'](bx, box_dim=7)
        out.append((bx, scores[i, :ns[i]], labels[i, :ns[i]]))
    return (out, cands) if return_candidates else out
