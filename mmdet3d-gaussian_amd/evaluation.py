"""Device-side counterparts of the reference's compiled evaluation helpers, and nothing above them:

* ``trans_bev`` (and ``iou_3d`` / ``iou_bev`` in iou3d.py) — /root/reference/mmdet3d_gaussian/ops/eval/affinity.cpp:8-105
  as bound in ops/eval/eval_utils.cpp:26-36;
* ``match_coco`` — ops/eval/matcher.cpp:8-74.

The callables the reference wraps around them (core/evaluation/affinity.py, matcher.py: a few forwarding lines each) and the mAP
accumulation above those (mean_ap_flexible.py, breakdown.py) are out of this build's scope: the reference's own
classes call these two functions unchanged.

The reference computes all of this on the CPU from numpy arrays.  Here the affinity matrix is produced and consumed in
HBM; numpy inputs are accepted and moved to the current device, results are returned as tensors on that device.
CPU tensors (and numpy inputs on a machine without a GPU) take the library's `_cpu` twins — the reference's own helpers are CPU code.
"""
import numpy as np
import torch

from . import _lib
from .iou3d import iou_3d, iou_bev


def _dev_tensor(x, dtype, name, cpu_ok=False):
    """numpy arrays (what the reference's evaluation passes) go to the current GPU when there is one; torch tensors stay on
    their device.  What ends up on the CPU is legal only where a `_cpu` twin exists (`cpu_ok`)."""
    from_numpy = not isinstance(x, torch.Tensor)
    if isinstance(x, np.ndarray):
        x = torch.from_numpy(np.ascontiguousarray(x))
    if not isinstance(x, torch.Tensor):
        x = torch.as_tensor(x)
    if not x.is_cuda:
        if from_numpy and torch.cuda.is_available():
            x = x.cuda()
        elif not cpu_ok:
            raise RuntimeError(f'{name}: this function has no CPU path' + ('' if torch.cuda.is_available() else ' and no GPU is visible'))
    return x.to(dtype).contiguous()


def trans_bev(det_bboxes, gt_bboxes):
    """(D,C>=2),(G,C'>=2) -> (D,G) distance between BEV centres (columns 0,1), affinity.cpp:83-105."""
    d = _dev_tensor(det_bboxes, torch.float32, 'trans_bev', cpu_ok=True)
    g = _dev_tensor(gt_bboxes, torch.float32, 'trans_bev', cpu_ok=True).to(d.device)
    if d.dim() != 2 or g.dim() != 2 or d.shape[1] < 2 or g.shape[1] < 2:
        raise RuntimeError(f'trans_bev: expected (D,>=2) and (G,>=2), got {tuple(d.shape)} and {tuple(g.shape)}')
    out = torch.empty((d.shape[0], g.shape[0]), dtype=torch.float32, device=d.device)
    if not d.is_cuda:   # the reference's own helper is CPU code (affinity.cpp:83-105): the `_cpu` twin
        _lib.check(_lib.load().riou_eval_trans_bev_cpu(d.data_ptr(), d.shape[0], d.shape[1], g.data_ptr(), g.shape[0], g.shape[1],
                                                       out.data_ptr(), torch.get_num_threads()), 'riou_eval_trans_bev_cpu')
        return out
    with torch.cuda.device(d.device):
        _lib.check(_lib.load().riou_eval_trans_bev(d.data_ptr(), d.shape[0], d.shape[1], g.data_ptr(), g.shape[0],
                                                   g.shape[1], out.data_ptr(),
                                                   torch.cuda.current_stream().cuda_stream), 'riou_eval_trans_bev')
    return out


def match_coco(cost_mat, cost_thrs, is_ignore, is_crowd):
    """(D,G) costs, (T) thresholds, (G) bool flags -> (T,D) int32 tensor: matched gt index or -1 (matcher.cpp:8-74)."""
    cost = _dev_tensor(cost_mat, torch.float32, 'match_coco', cpu_ok=True)
    if cost.dim() != 2:
        raise RuntimeError(f'match_coco: cost matrix must be 2-D, got {tuple(cost.shape)}')
    dev = cost.device
    thrs = _dev_tensor(cost_thrs, torch.float32, 'match_coco', cpu_ok=True).to(dev).reshape(-1)
    ign = _dev_tensor(is_ignore, torch.uint8, 'match_coco', cpu_ok=True).to(dev).reshape(-1)
    crowd = _dev_tensor(is_crowd, torch.uint8, 'match_coco', cpu_ok=True).to(dev).reshape(-1)
    D, G = cost.shape
    if ign.numel() != G or crowd.numel() != G:
        raise RuntimeError(f'match_coco: {G} gts but {ign.numel()} ignore / {crowd.numel()} crowd flags')
    T = thrs.numel()
    out = torch.empty((T, D), dtype=torch.int32, device=dev)
    if not cost.is_cuda:   # the reference's matcher is CPU code (matcher.cpp:8-74): the `_cpu` twin
        _lib.check(_lib.load().eval_match_coco_cpu(cost.data_ptr(), thrs.data_ptr(), ign.data_ptr(), crowd.data_ptr(), D, G, T,
                                                   out.data_ptr(), torch.get_num_threads()), 'eval_match_coco_cpu')
        return out
    with torch.cuda.device(dev):
        _lib.check(_lib.load().eval_match_coco(cost.data_ptr(), thrs.data_ptr(), ign.data_ptr(), crowd.data_ptr(), D, G, T,
                                               out.data_ptr(), torch.cuda.current_stream().cuda_stream),
                   'eval_match_coco')
    return out
