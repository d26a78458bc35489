"""Classification and direction terms of the anchor heads' loss in one pass over the head's maps (csrc/anchor_cls.hip).

What `GDAnchor3DHead.loss_single` does at /root/reference/mmdet3d_gaussian/models/dense_heads/gd_anchor3d_head.py:84-92 (a permuted
copy of the class maps into mmdet's sigmoid FocalLoss) and :143-149 (the positives' direction logits gathered into mmdet's
CrossEntropyLoss) — both loss modules third party, absent here, restated from the published text — as two launches forward
(with the gradient), none backward when the upstream gradient is the unit one, and no host sync.  GPU tensors only.
"""
import torch

from . import _lib
from .gd_loss import _is_unit_grad, guard_double_backward


def _get(cfg, key, default):
    return cfg.get(key, default) if isinstance(cfg, dict) else getattr(cfg, key, default)


def _focal_cfg(loss_cls):
    kind = _get(loss_cls, 'type', type(loss_cls).__name__)
    if kind != 'FocalLoss' or not _get(loss_cls, 'use_sigmoid', True):
        raise RuntimeError(f'anchor_head_cls_dir_loss: loss_cls is {kind!r}; the reference heads configure FocalLoss(use_sigmoid=True)')
    if _get(loss_cls, 'reduction', 'mean') != 'mean':
        raise RuntimeError("anchor_head_cls_dir_loss: loss_cls reduction must be 'mean' (sum / avg_factor)")
    if _get(loss_cls, 'activated', False):
        raise RuntimeError('anchor_head_cls_dir_loss: FocalLoss(activated=True) takes probabilities; the head passes logits')
    return float(_get(loss_cls, 'gamma', 2.0)), float(_get(loss_cls, 'alpha', 0.25)), float(_get(loss_cls, 'loss_weight', 1.0))


def _ce_cfg(loss_dir):
    kind = _get(loss_dir, 'type', type(loss_dir).__name__)
    if kind != 'CrossEntropyLoss' or _get(loss_dir, 'use_sigmoid', False) or _get(loss_dir, 'use_mask', False):
        raise RuntimeError(f'anchor_head_cls_dir_loss: loss_dir is {kind!r}; the reference heads configure CrossEntropyLoss(use_sigmoid=False)')
    if _get(loss_dir, 'reduction', 'mean') != 'mean' or _get(loss_dir, 'class_weight', None) is not None:
        raise RuntimeError("anchor_head_cls_dir_loss: loss_dir must have reduction 'mean' and no class_weight")
    return float(_get(loss_dir, 'loss_weight', 1.0))


def _f32c(x):
    return x if (x.dtype == torch.float32 and x.is_contiguous()) else x.float().contiguous()


def _i64c(x):
    return x if (x.dtype == torch.int64 and x.is_contiguous()) else x.long().contiguous()


class _ClsDir(torch.autograd.Function):
    @staticmethod
    def forward(ctx, cfg, targets, cls_score, dir_cls_preds):
        lib = _lib.load_extras()
        gamma, alpha, cls_scale, dir_scale, C, avg_dev = cfg
        labels, label_weights, dir_targets, dir_weights = targets
        B, AC, H, W = cls_score.shape
        A = AC // C
        xc = _f32c(cls_score)
        xd = None if dir_cls_preds is None else _f32c(dir_cls_preds)
        dev = xc.device
        with torch.cuda.device(dev):
            gc = torch.empty_like(xc) if ctx.needs_input_grad[2] else None
            gd = torch.empty_like(xd) if (xd is not None and ctx.needs_input_grad[3]) else None
            out = torch.empty(2, dtype=torch.float32, device=dev)
            ws = torch.empty(lib.gd3d_anchor_cls_dir_workspace_bytes(B, H, W), dtype=torch.uint8, device=dev)
            ptr = lambda t: None if t is None else t.data_ptr()
            if avg_dev is not None:       # cls_scale / dir_scale are the loss weights; the kernels divide by *avg_dev
                rc = lib.gd3d_anchor_cls_dir_loss_dyn(xc.data_ptr(), ptr(xd), labels.data_ptr(), label_weights.data_ptr(), ptr(dir_targets),
                                                      ptr(dir_weights), B, A, C, H, W, gamma, alpha, cls_scale, dir_scale, avg_dev.data_ptr(),
                                                      ptr(gc), ptr(gd), out.data_ptr(), ws.data_ptr(), torch.cuda.current_stream().cuda_stream)
            else:
                rc = lib.gd3d_anchor_cls_dir_loss(xc.data_ptr(), ptr(xd), labels.data_ptr(), label_weights.data_ptr(), ptr(dir_targets),
                                                  ptr(dir_weights), B, A, C, H, W, gamma, alpha, cls_scale, dir_scale, ptr(gc), ptr(gd),
                                                  out.data_ptr(), ws.data_ptr(), torch.cuda.current_stream().cuda_stream)
            _lib.check(rc, 'gd3d_anchor_cls_dir_loss')
        ctx.state = (gc, gd, cls_score.dtype, None if dir_cls_preds is None else dir_cls_preds.dtype)
        return out[0], out[1]          # two outputs of the node (no select nodes in the caller's graph)

    @staticmethod
    @guard_double_backward
    def backward(ctx, grad_cls, grad_dir):
        gc, gd, dtc, dtd = ctx.state
        # the library's own constant 1.0 (gd_loss.unit_grad) is known by address: the stored maps are final, nothing is launched
        rc = None if gc is None else (gc if _is_unit_grad(grad_cls) else gc * grad_cls).to(dtc)
        rd = None if gd is None else (gd if _is_unit_grad(grad_dir) else gd * grad_dir).to(dtd)
        return None, None, rc, rd


def anchor_head_cls_dir_loss(loss_cls, loss_dir, cls_score, dir_cls_preds, labels, label_weights, dir_targets, dir_weights, num_classes,
                             num_total_samples=None):
    """loss_cls / loss_dir : the head's modules (mmdet FocalLoss(use_sigmoid=True) / CrossEntropyLoss(use_sigmoid=False)) or their
                          config dicts (gamma, alpha, loss_weight, reduction='mean'); loss_dir None with dir_cls_preds None = a head
                          without direction classifier;
    cls_score           : (B, A*C, H, W) the head's raw class maps, NOT permuted;  dir_cls_preds (B, A*2, H, W) or None;
    labels, dir_targets : (B, H*W*A) integer, label == num_classes = background, negative = ignored for the direction term;
    label_weights, dir_weights : (B, H*W*A);  num_total_samples : the avg_factor of both terms — a number, or a one-element fp32 device
                          tensor that the kernels divide by (no read-back; e.g. sum_b max(positives_b, 1)) — (None = the batch size B, the
                          reference's fallback `int(cls_score.shape[0])` at :85-86, taken before its permute).
    Returns (loss_cls, loss_dir) 0-dim tensors on the device, differentiable wrt the maps; loss_dir is 0 (still attached to the
    graph) when there is no positive anchor, as `pos_dir_cls_preds.sum()` at :157-158; None without direction classifier."""
    if not cls_score.is_cuda:
        raise RuntimeError('anchor_head_cls_dir_loss: the MI355X implementation has no CPU path')
    if cls_score.dim() != 4 or cls_score.shape[1] % num_classes:
        raise RuntimeError(f'anchor_head_cls_dir_loss: cls_score {tuple(cls_score.shape)} is not (B, A*{num_classes}, H, W)')
    B, AC, H, W = cls_score.shape
    A = AC // num_classes
    N = H * W * A
    gamma, alpha, wc = _focal_cfg(loss_cls)
    if dir_cls_preds is not None:
        if tuple(dir_cls_preds.shape) != (B, A * 2, H, W) or dir_cls_preds.device != cls_score.device:
            raise RuntimeError(f'anchor_head_cls_dir_loss: dir_cls_preds {tuple(dir_cls_preds.shape)} is not ({B}, {A * 2}, {H}, {W})')
        wd = _ce_cfg(loss_dir)
        if dir_targets is None or dir_weights is None:
            raise RuntimeError('anchor_head_cls_dir_loss: dir_cls_preds without dir_targets / dir_weights')
    else:
        wd = 0.0
    per = [labels, label_weights] + ([dir_targets, dir_weights] if dir_cls_preds is not None else [])
    for t in per:
        if t.numel() != B * N or t.device != cls_score.device:
            raise RuntimeError(f'anchor_head_cls_dir_loss: a per-anchor tensor has {t.numel()} entries on {t.device}, expected {B * N} on {cls_score.device}')
    if num_total_samples is None:
        num_total_samples = B
    avg_dev = None
    if isinstance(num_total_samples, torch.Tensor):       # a 0-dim fp32 device tensor: the kernels divide by it (no read-back)
        from .head_loss import _avg_tensor
        avg_dev = _avg_tensor(num_total_samples, cls_score.device, 'anchor_head_cls_dir_loss')
        avg = 1.0
    else:
        avg = float(num_total_samples)
        if not avg > 0:
            raise RuntimeError(f'anchor_head_cls_dir_loss: num_total_samples = {num_total_samples}')
    targets = (_i64c(labels.detach()), _f32c(label_weights.detach()),
               None if dir_cls_preds is None else _i64c(dir_targets.detach()), None if dir_cls_preds is None else _f32c(dir_weights.detach()))
    l_cls, l_dir = _ClsDir.apply((gamma, alpha, wc / avg, wd / avg, int(num_classes), avg_dev), targets, cls_score, dir_cls_preds)
    return l_cls, (l_dir if dir_cls_preds is not None else None)
