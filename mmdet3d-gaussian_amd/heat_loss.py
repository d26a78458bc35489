"""Heat-map classification loss of the CenterPoint heads for all tasks in one pass (csrc/heat_focal.hip).

What `CenterGDHead.loss` does per task at /root/reference/mmdet3d_gaussian/models/dense_heads/gd_centerpoint_head.py:403-411
— mmdet3d's `clip_sigmoid` on the head's heat-map logits, `num_pos = target.eq(1).sum().item()` (a host sync per task) and
mmdet's `GaussianFocalLoss(..., avg_factor=max(num_pos, 1))` (both third party, absent here: restated from the published
text) — as two launches forward for every task together, one launch backward and no host sync.  GPU tensors only.
"""
import ctypes

import torch

from . import _lib
from .gd_loss import guard_double_backward

CLIP_EPS = 1e-4      # mmdet3d clip_sigmoid(x, eps=1e-4)
LOG_EPS = 1e-12      # mmdet gaussian_focal_loss: eps = 1e-12


def _cfg(loss_cls):
    get = (lambda k, d: loss_cls.get(k, d)) if isinstance(loss_cls, dict) else (lambda k, d: getattr(loss_cls, k, d))
    kind = get('type', type(loss_cls).__name__)
    if kind != 'GaussianFocalLoss':
        raise RuntimeError(f'center_head_heatmap_loss: loss_cls is {kind!r}; the reference heads configure GaussianFocalLoss')
    if get('reduction', 'mean') != 'mean':
        raise RuntimeError("center_head_heatmap_loss: reduction must be 'mean' (sum / avg_factor), as the reference configures it")
    return float(get('alpha', 2.0)), float(get('gamma', 4.0)), float(get('loss_weight', 1.0))


def _tasks(logits, targets, grads):
    arr = (_lib.HeatFocalTask * len(logits))()
    for t, (x, y) in enumerate(zip(logits, targets)):
        arr[t].logits, arr[t].target, arr[t].n = x.data_ptr(), y.data_ptr(), x.numel()
        arr[t].grad = grads[t].data_ptr() if grads[t] is not None else None
    return arr


class _HeatFocal(torch.autograd.Function):
    @staticmethod
    def _launch(cfg, xs, targets, need):
        """One gd3d_heat_focal_loss launch set -> (per-task gradient maps | None, out (3, T): losses, factors, num_pos)."""
        lib = _lib.load_extras()
        alpha, gamma, weight = cfg
        T = len(xs)
        dev = xs[0].device
        with torch.cuda.device(dev):
            grads = [torch.empty_like(x) if nd else None for x, nd in zip(xs, need)]
            arr = _tasks(xs, targets, grads)
            out = torch.empty((3, T), dtype=torch.float32, device=dev)        # losses, factors, num_pos
            ws = torch.empty(lib.gd3d_heat_focal_workspace_bytes(arr, T), dtype=torch.uint8, device=dev)
            _lib.check(lib.gd3d_heat_focal_loss(arr, T, alpha, gamma, CLIP_EPS, LOG_EPS, weight, out[0].data_ptr(), out[1].data_ptr(),
                                                out[2].data_ptr(), ws.data_ptr(), torch.cuda.current_stream().cuda_stream),
                       'gd3d_heat_focal_loss')
        return grads, out

    @staticmethod
    def forward(ctx, cfg, targets, *logits):
        need = [bool(ctx.needs_input_grad[2 + t]) for t in range(len(logits))]
        xs = [x if (x.dtype == torch.float32 and x.is_contiguous()) else x.float().contiguous() for x in logits]
        grads, out = _HeatFocal._launch(cfg, xs, targets, need)
        # the fp32 logits and targets a retain_graph replay re-launches from go through save_for_backward: version-checked (an
        # in-place edit between forward and that backward raises instead of returning another input's gradients) and released
        # with the graph; ctx.state holds no input tensor
        ctx.save_for_backward(*xs, *targets)
        ctx.state = (cfg, grads, out, need, [x.dtype for x in logits], [x.shape for x in logits])
        ctx.used = False
        losses, num_pos = out[0], out[2]
        ctx.mark_non_differentiable(num_pos)
        return losses, num_pos

    @staticmethod
    @guard_double_backward
    def backward(ctx, grad_losses, _grad_num_pos):
        lib = _lib.load_extras()
        cfg, grads, out, need, dtypes, shapes = ctx.state
        saved = ctx.saved_tensors
        T = len(saved) // 2
        xs, targets = list(saved[:T]), list(saved[T:])
        if ctx.used:   # retain_graph replay: the maps of the first backward were scaled in place and handed over: launch again,
            grads, out = _HeatFocal._launch(cfg, xs, targets, need)   # as the loss / anchor-head / centre-head nodes do
        else:
            ctx.state = (cfg, None, None, need, dtypes, shapes)
        ctx.used = True
        with torch.cuda.device(xs[0].device):
            up = grad_losses.to(torch.float32).contiguous()
            _lib.check(lib.gd3d_heat_focal_scale(_tasks(xs, targets, grads), T, out[1].data_ptr(), up.data_ptr(),
                                                 torch.cuda.current_stream().cuda_stream), 'gd3d_heat_focal_scale')
        res = [None if g is None else (g.view(shp) if dt == torch.float32 else g.view(shp).to(dt)) for g, dt, shp in zip(grads, dtypes, shapes)]
        return (None, None) + tuple(res)


def center_head_heatmap_loss(loss_cls, heatmap_logits, heatmap_targets):
    """loss_cls        : the head's `loss_cls` (an mmdet GaussianFocalLoss module, or its config dict: alpha, gamma, loss_weight,
                      reduction='mean');
    heatmap_logits  : per task the head's RAW heat-map output (B, C_t, H, W) — before any sigmoid (the reference replaces
                      preds_dict['heatmap'] by its clipped sigmoid in place at :405; this function leaves it alone);
    heatmap_targets : per task the Gaussian target maps of `get_targets` (same shapes).
    Returns (losses (T,), num_pos (T,)) on the device: losses[t] = `task{t}.loss_heatmap`, differentiable wrt the logits;
    num_pos[t] = number of target cells equal to 1 (what the reference passes on as avg_factor of the regression losses)."""
    if len(heatmap_logits) != len(heatmap_targets) or not heatmap_logits:
        raise RuntimeError(f'center_head_heatmap_loss: {len(heatmap_logits)} logit maps and {len(heatmap_targets)} target maps')
    if len(heatmap_logits) > 16:
        raise RuntimeError('center_head_heatmap_loss: at most 16 tasks per call')
    if not heatmap_logits[0].is_cuda:
        raise RuntimeError('center_head_heatmap_loss: the MI355X implementation has no CPU path')
    targets = []
    for t, (x, y) in enumerate(zip(heatmap_logits, heatmap_targets)):
        if x.shape != y.shape or x.device != y.device:
            raise RuntimeError(f'task {t}: logits {tuple(x.shape)} on {x.device} vs targets {tuple(y.shape)} on {y.device}')
        y = y.detach()
        targets.append(y if (y.dtype == torch.float32 and y.is_contiguous()) else y.float().contiguous())
    return _HeatFocal.apply(_cfg(loss_cls), targets, *heatmap_logits)
