"""Rotated BEV NMS / IoU — host-side mirror of the mmdet3d ``iou3d`` surface the reference calls.

``nms_gpu(boxes, scores, thresh, pre_max_size=None, post_max_size=None)`` is what the reference
imports from third-party mmdet3d and calls at
/root/reference/mmdet3d_gaussian/models/dense_heads/gd_centerpoint_head.py:9,340-345 and
models/roi_heads/bbox_heads/pvrcnn_bbox_head.py:12,463-464 (and, through upstream
``box3d_multiclass_nms``, for the PointPillars heads).  Same argument names and meaning, same
return (LongTensor of kept indices into the INPUT order, by descending score).  Sorting stays in
torch; the suppression mask AND the greedy scan run on the device (csrc/rbox.hip), so only the
final count crosses to the host (the returned tensor has a data-dependent length).

``iou_bev`` / ``iou_3d`` are GPU counterparts of the reference's CPU eval helpers
(/root/reference/mmdet3d_gaussian/ops/eval/affinity.cpp:8-81), ``(D,7) x (G,7) -> (D,G)``.
"""
import ctypes

import torch

from . import _lib


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr())


def _stream(dev):
    return ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)


def _check_boxes(boxes, cols, name):
    if not boxes.is_cuda:
        raise RuntimeError(f'{name}: the MI355X implementation has no CPU path; tensors must be on the GPU')
    if boxes.dim() != 2 or boxes.shape[1] != cols:
        raise RuntimeError(f'{name}: expected (N,{cols}) boxes, got {tuple(boxes.shape)}')
    return boxes.to(torch.float32).contiguous()


def _nms(boxes, scores, thresh, pre_max_size, post_max_size, normal):
    name = 'nms_normal_gpu' if normal else 'nms_gpu'
    boxes = _check_boxes(boxes, 5, name)
    if scores.shape[0] != boxes.shape[0]:
        raise RuntimeError(f'{name}: {boxes.shape[0]} boxes but {scores.shape[0]} scores')
    lib = _lib.load()
    order = scores.sort(0, descending=True)[1]
    if pre_max_size is not None:
        order = order[:pre_max_size]
    order = order.contiguous()
    n = order.shape[0]
    dev = boxes.device
    if n == 0:
        return order.new_zeros((0,))
    with torch.cuda.device(dev):
        keep = torch.empty(n, dtype=torch.int64, device=dev)
        num = torch.empty(1, dtype=torch.int64, device=dev)
        ws = torch.empty(lib.rnms_workspace_bytes(n), dtype=torch.uint8, device=dev)
        # the kernels read boxes[order[i]] themselves and emit kept indices in the caller's numbering
        fn = lib.rnms_normal_bev_ordered if normal else lib.rnms_bev_ordered
        _lib.check(fn(boxes.data_ptr(), order.data_ptr(), n, float(thresh), keep.data_ptr(), num.data_ptr(),
                      ws.data_ptr(), torch.cuda.current_stream().cuda_stream), name)
    k = int(num.item())  # the one unavoidable sync: the result length is data dependent
    keep = keep[:k]
    if post_max_size is not None:
        keep = keep[:post_max_size]
    return keep


def nms_gpu(boxes, scores, thresh, pre_max_size=None, post_max_size=None, pre_maxsize=None):
    """Rotated BEV NMS.  boxes (N,5) [x1,y1,x2,y2,ry]; returns kept indices (LongTensor).
    `pre_maxsize` is the spelling of older mmdet3d releases and is accepted as an alias."""
    if pre_max_size is None:
        pre_max_size = pre_maxsize
    return _nms(boxes, scores, thresh, pre_max_size, post_max_size, normal=False)


def nms_normal_gpu(boxes, scores, thresh):
    """Axis-aligned BEV NMS (angle ignored), mmdet3d `nms_normal_gpu`."""
    return _nms(boxes, scores, thresh, None, None, normal=True)


def boxes_iou_bev(boxes_a, boxes_b):
    """Pairwise rotated BEV IoU of [x1,y1,x2,y2,ry] boxes: (M,5),(N,5) -> (M,N) (mmdet3d `boxes_iou_bev`)."""
    a = _check_boxes(boxes_a, 5, 'boxes_iou_bev')
    b = _check_boxes(boxes_b, 5, 'boxes_iou_bev')
    lib = _lib.load()
    out = torch.empty((a.shape[0], b.shape[0]), dtype=torch.float32, device=a.device)
    with torch.cuda.device(a.device):
        _lib.check(lib.riou_bev_xyxyr(_ptr(a), a.shape[0], _ptr(b), b.shape[0], _ptr(out), _stream(a.device)),
                   'riou_bev_xyxyr')
    return out


def iou_bev(det, gt):
    """(D,7),(G,7) [x,y,z,w,h,l,yaw] -> (D,G) BEV IoU; GPU counterpart of ops/eval `iou_bev`."""
    d = _check_boxes(det, 7, 'iou_bev')
    g = _check_boxes(gt, 7, 'iou_bev')
    lib = _lib.load()
    out = torch.empty((d.shape[0], g.shape[0]), dtype=torch.float32, device=d.device)
    with torch.cuda.device(d.device):
        _lib.check(lib.riou_eval_bev(_ptr(d), d.shape[0], _ptr(g), g.shape[0], _ptr(out), _stream(d.device)),
                   'riou_eval_bev')
    return out


def iou_3d(det, gt, z_offset=0.5):
    """(D,7),(G,7) -> (D,G) 3D IoU with the reference's `z_offset` convention (affinity.cpp:26-29)."""
    d = _check_boxes(det, 7, 'iou_3d')
    g = _check_boxes(gt, 7, 'iou_3d')
    lib = _lib.load()
    out = torch.empty((d.shape[0], g.shape[0]), dtype=torch.float32, device=d.device)
    with torch.cuda.device(d.device):
        _lib.check(lib.riou_eval_3d(_ptr(d), d.shape[0], _ptr(g), g.shape[0], float(z_offset), _ptr(out),
                                    _stream(d.device)), 'riou_eval_3d')
    return out


def xywhr2xyxyr(boxes_xywhr):
    """(N,5) [cx,cy,w,h,r] -> [x1,y1,x2,y2,r] (mmdet3d helper used at gd_centerpoint_head.py:336-337)."""
    out = torch.zeros_like(boxes_xywhr)
    half_w = boxes_xywhr[:, 2] / 2
    half_h = boxes_xywhr[:, 3] / 2
    out[:, 0] = boxes_xywhr[:, 0] - half_w
    out[:, 1] = boxes_xywhr[:, 1] - half_h
    out[:, 2] = boxes_xywhr[:, 0] + half_w
    out[:, 3] = boxes_xywhr[:, 1] + half_h
    out[:, 4] = boxes_xywhr[:, 4]
    return out
