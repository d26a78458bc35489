"""Head-level slices of the two dense heads that call GDLoss, with the bbox-coder decode fused into the kernel
(SURVEY.md §8f-1/f-2).  Each function restates the reference lines it replaces and keeps their argument meaning.

* ``anchor_head_decoded_loss``  — GDAnchor3DHead.loss_single, the decoded-box branch
  (/root/reference/mmdet3d_gaussian/models/dense_heads/gd_anchor3d_head.py:95-141).
* ``center_head_gd_loss``       — CenterGDHead.loss, the `loss_gd` branch per task
  (/root/reference/mmdet3d_gaussian/models/dense_heads/gd_centerpoint_head.py:413-434).
Gather of the positives stays in torch (index kernels); decode x2 + loss + all their backward nodes are one launch.
"""
import ctypes

import torch

from . import _lib
from .gd_loss import _is_unit_grad, guard_double_backward


def _prologue(kind, aux, norm_bbox=False, out_size_factor=1.0, voxel_size=(1.0, 1.0), pc_range=(0.0, 0.0)):
    p = _lib.Prologue()
    p.kind = kind
    p.norm_bbox = int(bool(norm_bbox))
    p.aux = aux.data_ptr()
    p.out_size_factor = float(out_size_factor)
    p.voxel_size = (ctypes.c_float * 2)(float(voxel_size[0]), float(voxel_size[1]))
    p.pc_range = (ctypes.c_float * 2)(float(pc_range[0]), float(pc_range[1]))
    p._keepalive = aux  # the struct only holds a raw pointer
    return p


def anchor_decoded_gd_loss(loss_module, anchors, pos_bbox_pred, pos_bbox_targets, weight=None, avg_factor=None):
    """loss_module(coder.decode(anchors, pos_bbox_pred), coder.decode(anchors, pos_bbox_targets), weight,
    avg_factor=avg_factor) with mmdet3d's DeltaXYZWLHRBBoxCoder, in one launch (gd_anchor3d_head.py:133-141)."""
    anchors = anchors.reshape(-1, 7).to(torch.float32).contiguous()
    n = pos_bbox_pred.reshape(-1, 7).shape[0]
    if anchors.shape[0] != n or pos_bbox_targets.reshape(-1, 7).shape[0] != n:
        raise RuntimeError(f'anchors {tuple(anchors.shape)}, pred {tuple(pos_bbox_pred.shape)} and targets '
                           f'{tuple(pos_bbox_targets.shape)} must have the same number of rows')
    if anchors.device != pos_bbox_pred.device:
        raise RuntimeError('anchors and predictions live on different devices')
    pro = _prologue(1, anchors)
    return loss_module(pos_bbox_pred, pos_bbox_targets, weight, avg_factor=avg_factor, _prologue=pro)


def anchor_head_decoded_loss(loss_module, bbox_pred, bbox_targets, bbox_weights, labels, anchor_list, num_classes,
                             num_total_samples, decode_weight=None, box_code_size=7):
    """The decoded-box part of GDAnchor3DHead.loss_single (gd_anchor3d_head.py:95-141).

    bbox_pred: (B, A*code, H, W) head output; bbox_targets / bbox_weights: (B, H*W*A, code); labels: (B, H*W*A);
    anchor_list: (H*W*A, code) anchors of one sample (repeated over the batch as the reference does, :110-112).
    decode_weight: train_cfg['decode_weight'] (list of `code` floats) or None.
    Returns loss_bbox of the decoded branch (`pos_bbox_pred.sum()` when there are no positives, :160-161)."""
    mini_batch = bbox_pred.shape[0]
    bbox_pred = bbox_pred.permute(0, 2, 3, 1).reshape(-1, box_code_size)
    bbox_targets = bbox_targets.reshape(-1, box_code_size)
    bbox_weights = bbox_weights.reshape(-1, box_code_size)
    labels = labels.reshape(-1)
    pos_inds = ((labels >= 0) & (labels < num_classes)).nonzero(as_tuple=False).reshape(-1)
    pos_bbox_pred = bbox_pred[pos_inds]
    if len(pos_inds) == 0:
        return pos_bbox_pred.sum()
    pos_bbox_targets = bbox_targets[pos_inds]
    anchors = anchor_list.reshape(-1, box_code_size).repeat(mini_batch, 1)[pos_inds]
    weight = None
    if decode_weight:
        weight = bbox_weights[pos_inds] * bbox_weights.new_tensor(decode_weight)
    return anchor_decoded_gd_loss(loss_module, anchors, pos_bbox_pred, pos_bbox_targets, weight,
                                  avg_factor=num_total_samples)


def _anchor_head_fused(bbox_pred, bbox_targets, bbox_weights, anchors, pos_or_labels, params, dw, scale, dense=False,
                       num_classes=0, sl1=None):
    """selection / gather of the positives + decode x2 + loss(es) + gradient scatter into the NCHW head output: one launch,
    behind the host glue's `anchor_head` node (_pynode.GDAnchorHead, or its C++ twin in csrc/torch_node.cpp:
    `_lib.load_node()`): the zero-filled-then-scattered gradient waits in the node; backward hands it over, scaled on the device unless the upstream gradient is gd_loss.unit_grad; a second backward
    under retain_graph launches again; differentiating the gradient raises.  `pos_or_labels` is either the (P,) int64
    positive list (dense=False) or the (M,) int64 label map (dense=True); `scale` a float or, for a device-resident
    normaliser (dense form only), the tuple (avg_dev, gd_weight, sl1_weight)."""
    avg_dev, w_gd, w_sl1 = (scale if isinstance(scale, tuple) else (None, 0.0, 0.0))
    return _lib.load_node().anchor_head(bbox_pred, bbox_targets, bbox_weights, anchors, pos_or_labels, ctypes.addressof(params),
                                        0 if sl1 is None else ctypes.addressof(sl1), None if dw is None else [float(x) for x in dw],
                                        bool(dense), int(num_classes), 0.0 if avg_dev is not None else float(scale), avg_dev,
                                        float(w_gd), float(w_sl1))


def _avg_tensor(t, dev, who):
    """num_total_samples as a device-resident normaliser: one fp32 value on `dev`, detached"""
    if t.numel() != 1 or t.device != dev:
        raise RuntimeError(f'{who}: a tensor num_total_samples must hold one value on {dev}, got {tuple(t.shape)} on {t.device}')
    t = t.detach().reshape(())
    return t if t.dtype == torch.float32 else t.float()


def _seven(w, name):
    """train_cfg['code_weight'] / ['decode_weight']: a list of 7 or a scalar (the shipped configs say `decode_weight=1`,
    which `bbox_weights.new_tensor(1)` broadcasts, gd_anchor3d_head.py:128-131).  Falsy -> None, as `if w:` does."""
    if isinstance(w, torch.Tensor):
        w = w.tolist()
    if not w:
        return None
    if isinstance(w, (int, float)):
        return [float(w)] * 7
    if len(w) != 7:
        raise RuntimeError(f'{name} must be a scalar or have 7 entries')
    return [float(x) for x in w]


def _head_operands(bbox_pred, bbox_targets, bbox_weights, labels, anchor_list, need_weights):
    B, C, H, W = bbox_pred.shape
    if C % 7 != 0:
        raise RuntimeError(f'bbox_pred has {C} channels, expected a multiple of the box code size 7')
    M = B * H * W * (C // 7)
    labels = labels.reshape(-1)
    # the kernel indexes by position: every operand must cover exactly the B*H*W*A anchors (no OOB reads on the GPU)
    if labels.numel() != M or bbox_targets.numel() != M * 7 or anchor_list.numel() != (M // B) * 7 or \
            (need_weights and (bbox_weights is None or bbox_weights.numel() != M * 7)):
        raise RuntimeError(f'shape mismatch: bbox_pred {tuple(bbox_pred.shape)} implies {M} anchors; labels '
                           f'{labels.numel()}, bbox_targets {bbox_targets.numel() // 7}, anchors per sample '
                           f'{anchor_list.numel() // 7}, bbox_weights '
                           f'{None if bbox_weights is None else bbox_weights.numel() // 7}')
    for t in (bbox_targets, labels, anchor_list) + ((bbox_weights,) if need_weights else ()):
        if t.device != bbox_pred.device:
            raise RuntimeError('head-loss operands live on different devices')
    bp = (bbox_pred if bbox_pred.dtype == torch.float32 else bbox_pred.float()).contiguous()
    weights = bbox_weights.reshape(-1, 7).to(torch.float32).contiguous() if need_weights else None
    return (bp, bbox_targets.reshape(-1, 7).to(torch.float32).contiguous(), weights,
            anchor_list.reshape(-1, 7).to(torch.float32).contiguous(), labels)


def _select(labels, num_classes, dense):
    if dense:
        return labels.to(torch.int64).contiguous()
    return ((labels >= 0) & (labels < num_classes)).nonzero(as_tuple=False).reshape(-1).contiguous()


def anchor_head_decoded_loss_fused(loss_module, bbox_pred, bbox_targets, bbox_weights, labels, anchor_list, num_classes,
                                   num_total_samples, decode_weight=None, dense=True):
    """Same contract as anchor_head_decoded_loss, but everything between the raw NCHW head output and the loss runs
    INSIDE one kernel (no permute/reshape copy, no index kernels, no scatter in backward).
      dense=True : one thread per ANCHOR tests its label itself; torch.nonzero() — a host sync — and the compaction go
                   away too, the call is fully asynchronous (no-positive batches simply yield 0 and a zero gradient);
      dense=False: one thread per positive of a torch.nonzero() list (less device work, one sync).
    Requires a reduced loss (mean/sum) without per-call kwargs."""
    from .gd_loss import GDLoss
    assert isinstance(loss_module, GDLoss) and loss_module.reduction != 'none'
    dw = _seven(decode_weight, 'decode_weight')
    bp, bt, weights, anchors, labels = _head_operands(bbox_pred, bbox_targets, bbox_weights, labels, anchor_list,
                                                      dw is not None)
    sel = _select(labels, num_classes, dense)
    if not dense and sel.numel() == 0:
        return bbox_pred.sum() * 0
    if num_total_samples is None:
        num_total_samples = int(bbox_pred.shape[0])      # loss_single: `int(cls_score.shape[0])`, the batch size (:85-86)
    den = num_total_samples if loss_module.reduction == 'mean' else 1.0
    scale = float(loss_module.loss_weight) / float(den)
    return _anchor_head_fused(bp, bt, weights, anchors, sel, loss_module._params({}), dw, scale, bool(dense), int(num_classes), None)


def anchor_head_bbox_loss(loss_decoded_bbox, loss_bbox, bbox_pred, bbox_targets, bbox_weights, labels, anchor_list,
                          num_classes, num_total_samples, code_weight=None, decode_weight=None, diff_rad_by_sin=True,
                          dense=True):
    """`loss_bbox` as GDAnchor3DHead.loss_single returns it (gd_anchor3d_head.py:95-161), in ONE launch:

        loss_decoded_bbox(decode(anchors, pos_pred), decode(anchors, pos_targets), decode_weight, avg_factor)  (:133-141)
      + loss_bbox(pos_pred', pos_targets', code_weight, avg_factor)                                            (:150-159)

    loss_decoded_bbox: this package's GDLoss (reduction 'mean').  loss_bbox: the encoded-box regression loss — mmdet's
    SmoothL1Loss / L1Loss module itself or anything with `.beta` (absent/0 = L1), `.loss_weight`, `.reduction`, or a
    config dict (`dict(type='SmoothL1Loss', beta=1/9, loss_weight=2.0)`).  code_weight / decode_weight:
    train_cfg entries (list of 7, scalar, or falsy = weight None, :124-131).  diff_rad_by_sin: add_sin_difference
    (:150-152).  No positives: 0 with a zero gradient (:160-161).  num_total_samples: a number, or a one-element fp32 device tensor
    that the kernel divides the two loss weights by itself (dense form; no read-back of the positives' count)."""
    from .gd_loss import GDLoss
    assert isinstance(loss_decoded_bbox, GDLoss)
    if loss_decoded_bbox.reduction != 'mean':
        raise ValueError('avg_factor can not be used with reduction="sum"' if loss_decoded_bbox.reduction == 'sum'
                         else 'anchor_head_bbox_loss needs a reduced loss')
    if isinstance(loss_bbox, dict):
        kind = loss_bbox.get('type', 'SmoothL1Loss')
        beta = float(loss_bbox.get('beta', 1.0)) if kind == 'SmoothL1Loss' else 0.0
        lw, red = float(loss_bbox.get('loss_weight', 1.0)), loss_bbox.get('reduction', 'mean')
    else:
        beta = float(getattr(loss_bbox, 'beta', 0.0))
        lw, red = float(loss_bbox.loss_weight), getattr(loss_bbox, 'reduction', 'mean')
        kind = type(loss_bbox).__name__
    if kind not in ('SmoothL1Loss', 'L1Loss'):
        raise RuntimeError(f'encoded-box loss {kind!r} is not fused; supported: SmoothL1Loss, L1Loss')
    if kind == 'SmoothL1Loss' and not beta > 0:
        raise AssertionError('SmoothL1Loss needs beta > 0')          # mmdet smooth_l1_loss: assert beta > 0
    if red != 'mean':
        raise ValueError('avg_factor can not be used with reduction="sum"' if red == 'sum'
                         else 'anchor_head_bbox_loss needs a reduced loss')
    cw, dw = _seven(code_weight, 'code_weight'), _seven(decode_weight, 'decode_weight')
    bp, bt, weights, anchors, labels = _head_operands(bbox_pred, bbox_targets, bbox_weights, labels, anchor_list,
                                                      cw is not None or dw is not None)
    sel = _select(labels, num_classes, dense)
    if not dense and sel.numel() == 0:
        return bbox_pred.sum() * 0
    if num_total_samples is None:
        num_total_samples = int(bbox_pred.shape[0])      # loss_single: `int(cls_score.shape[0])`, the batch size (:85-86)
    dyn = isinstance(num_total_samples, torch.Tensor)    # a 0-dim fp32 device tensor: the kernels divide by it (no read-back)
    if dyn:
        if not dense:
            raise RuntimeError('anchor_head_bbox_loss: a device-resident num_total_samples needs the dense form')
        avg_dev = _avg_tensor(num_total_samples, bbox_pred.device, 'anchor_head_bbox_loss')
    sl1 = _lib.SmoothL1()
    sl1.beta = beta
    sl1.scale = 0.0 if dyn else lw / float(num_total_samples)
    sl1.diff_rad_by_sin = int(bool(diff_rad_by_sin))
    sl1.has_code_weight = int(cw is not None)
    sl1.code_weight = (ctypes.c_float * 7)(*(cw or [1.0] * 7))
    scale = (avg_dev, float(loss_decoded_bbox.loss_weight), lw) if dyn else float(loss_decoded_bbox.loss_weight) / float(num_total_samples)
    return _anchor_head_fused(bp, bt, weights, anchors, sel, loss_decoded_bbox._params({}), dw, scale, bool(dense), int(num_classes), sl1)


def center_head_gd_loss(loss_module, coder, pos_ind, pred, anno_boxes, num_pos):
    """loss_gd of one CenterGDHead task (gd_centerpoint_head.py:413-434):
        target_gd = coder.encode(anno_boxes)[..., :7]
        pred_gd   = coder.decode(pos_ind[..., 1:], pred, correct_yaw=False)[..., :7]
        loss_gd   = loss_module(pred_gd, target_gd, avg_factor=max(num_pos, 1))
    pos_ind: (B, K, 3) long [b, x, y]; pred: (B, K, C>=7) gathered raw head outputs [reg(2), height, dim(3), yaw, ...];
    anno_boxes: (B, K, >=7).  The decode runs inside the kernel; the gradient reaches pred[..., :7]."""
    target_gd = anno_boxes[..., :7].reshape(-1, 7)        # encode() leaves the first 7 entries as they are (:11-16)
    locs = pos_ind[..., 1:].reshape(-1, 2).to(torch.float32).contiguous()
    if locs.shape[0] != pred[..., :7].reshape(-1, 7).shape[0] or locs.shape[0] != target_gd.shape[0]:
        raise RuntimeError(f'pos_ind {tuple(pos_ind.shape)}, pred {tuple(pred.shape)} and anno_boxes '
                           f'{tuple(anno_boxes.shape)} must describe the same number of objects')
    pro = _prologue(2, locs, norm_bbox=coder.norm_bbox, out_size_factor=coder.out_size_factor,
                    voxel_size=coder.voxel_size, pc_range=coder.pc_range)
    pred7 = pred[..., :7].reshape(-1, 7)
    if pred7.numel() == 0:
        # the reference returns new_zeros((1,)) (:436-438); same value and shape here, but attached to `pred` so that the
        # head's parameters still receive a (zero) gradient on a step without objects
        return (pred.sum() * 0).reshape(1)
    return loss_module(pred7, target_gd, avg_factor=max(num_pos, 1), _prologue=pro)


_CENTER_HEADS = ('reg', 'height', 'dim', 'yaw', 'dir', 'vel')
# objects per task above which center_head_losses sorts the cell keys between its two launches (gd3d_center_head_finish);
# below it the scanning form's two launches are cheaper than the sort's (tests/perf/config4_head.py)
CENTER_SORT_MIN_N = 8192
_CENTER_CH = (2, 1, 3, 1, 2, 2)
_CENTER_TASKS = {}


def _center_head_launch(meta, maps, need):
    """gd3d_center_head_loss on all tasks -> (losses (T,2), per-map gradient views | None, task table)."""
    lib = _lib.load()
    params, pro, layout, pos_inds, annos, scales, cw, n_l1 = meta[:8]     # meta[8:] only keeps device operands alive
    T = len(layout)
    dev = maps[0].device
    # ONE zero fill covers every gradient map of every task (36 maps at 6 tasks) and the per-task cell counters
    # (B*H*W int32 each, all-zero bits): views into a single flat buffer
    sizes = [m.numel() if nd else 0 for m, nd in zip(maps, need)]
    wants = [any(need[k] for k in layout[t] if k >= 0) for t in range(T)]
    cells = []
    for t in range(T):
        ref = maps[layout[t][1]]
        cells.append(ref.shape[0] * ref.shape[2] * ref.shape[3] if wants[t] else 0)
    flat = torch.zeros(sum(sizes) + sum(cells), dtype=torch.float32, device=dev) if any(need) else None
    grads, off = [], 0
    for m, nd, sz in zip(maps, need, sizes):
        grads.append(flat[off:off + sz].view_as(m) if nd else None)
        off += sz
    # the task table (~30 ctypes fields per task) is rebuilt only when an address, a size or a scale changed: a training loop
    # whose tensors come back from the caching allocator at the same addresses pays for it once
    key = (dev.index, tuple(m.data_ptr() for m in maps), tuple(m.shape for m in maps), tuple(need), 0 if flat is None else flat.data_ptr(),
           tuple(p.data_ptr() for p in pos_inds), tuple(p.shape[0] for p in pos_inds), tuple(a.data_ptr() for a in annos),
           tuple(a.shape[1] if a.dim() == 2 else 7 for a in annos), tuple(scales), tuple(tuple(r) for r in layout))
    hit = _CENTER_TASKS.get(key)
    if hit is not None:
        tasks, max_n = hit
    else:
        tasks = (_lib.CenterTask * T)()
        max_n = 0
        for t in range(T):
            tk = tasks[t]
            for h in range(6):
                k = layout[t][h]
                tk.maps[h] = maps[k].data_ptr() if k >= 0 else None
                tk.grads[h] = grads[k].data_ptr() if (k >= 0 and grads[k] is not None) else None
            ref = maps[layout[t][1]]
            tk.B, tk.H, tk.W = ref.shape[0], ref.shape[2], ref.shape[3]
            tk.cell_count = flat.data_ptr() + 4 * off if wants[t] else None
            off += cells[t]
            tk.n = pos_inds[t].shape[0]
            tk.pos_ind = pos_inds[t].data_ptr()
            tk.anno = annos[t].data_ptr()
            tk.anno_cols = annos[t].shape[1] if annos[t].dim() == 2 else 7
            if len(scales[t]) == 4:
                tk.gd_weight, tk.l1_weight, tk.rows_dev, tk.avg_dev = scales[t]
            else:
                tk.gd_scale, tk.l1_scale = scales[t]
            max_n = max(max_n, tk.n)
        if len(_CENTER_TASKS) >= 16:
            _CENTER_TASKS.clear()
        _CENTER_TASKS[key] = (tasks, max_n)
    losses = torch.empty((T, 2), dtype=torch.float32, device=dev)
    ws = torch.empty(lib.gd3d_center_head_workspace_bytes(T, max_n) // 4, dtype=torch.float32, device=dev)
    cwp = (ctypes.c_float * max(n_l1, 1))(*[float(x) for x in cw[:n_l1]])
    with torch.cuda.device(dev):
        stream = torch.cuda.current_stream().cuda_stream
        if any(need) and max_n > CENTER_SORT_MIN_N:
            # many objects: the per-cell accumulation walks the keys in SORTED order (one stable batched sort of the
            # tasks' key rows between the two launches) instead of scanning them once per shared-cell object
            _lib.check(lib.gd3d_center_head_stage(params, pro, tasks, T, cwp, n_l1, losses.data_ptr(), ws.data_ptr(), stream),
                       'gd3d_center_head_stage')
            off, stride = ctypes.c_int64(), ctypes.c_int64()
            _lib.check(lib.gd3d_center_head_keys(T, max_n, ctypes.byref(off), ctypes.byref(stride)), 'gd3d_center_head_keys')
            rows = torch.as_strided(ws.view(torch.int32), (T, max_n), (stride.value // 4, 1), off.value // 4)
            order = torch.sort(rows, dim=1, stable=True).indices.contiguous()
            rc = lib.gd3d_center_head_finish(params, pro, tasks, T, cwp, n_l1, losses.data_ptr(), ws.data_ptr(),
                                             order.data_ptr(), stream)
        else:
            rc = lib.gd3d_center_head_loss(params, pro, tasks, T, cwp, n_l1, losses.data_ptr(), ws.data_ptr(), stream)
    _lib.check(rc, 'gd3d_center_head_loss')
    return losses, grads, tasks


class _CenterHeadFused(torch.autograd.Function):
    """All tasks' loss_l1 / loss_gd straight from the NCHW head maps: two launches forward (loss + staged gradients, then
    the deterministic per-cell accumulation + loss sums), one launch backward.  `layout[t][h]` = index into `maps` of
    head h of task t, or -1."""

    @staticmethod
    def forward(ctx, meta, *maps):
        need = [bool(ctx.needs_input_grad[1 + k]) for k in range(len(maps))]
        losses, grads, tasks = _center_head_launch(meta, maps, need)
        ctx.tasks, ctx.grads, ctx.used, ctx.replay = tasks, grads, False, (meta, maps, need)
        return losses

    @staticmethod
    @guard_double_backward
    def backward(ctx, grad_losses):
        lib = _lib.load()
        if ctx.used:  # retain_graph replay: the saved maps were scaled in place: recompute them
            _, grads, tasks = _center_head_launch(*ctx.replay)
        else:  # hand the maps over (no reference left here: a leaf's AccumulateGrad then keeps its map instead of cloning it)
            grads, tasks, ctx.used = ctx.grads, ctx.tasks, True
            ctx.grads = ctx.tasks = None
        go = grad_losses.contiguous().float()
        dev = go.device
        with torch.cuda.device(dev):
            rc = lib.gd3d_center_head_scale(tasks, len(tasks), go.data_ptr(), torch.cuda.current_stream().cuda_stream)
        _lib.check(rc, 'gd3d_center_head_scale')
        return (None,) + tuple(grads)


def center_head_losses(loss_gd, loss_bbox, coder, preds_dicts, pos_inds, anno_boxes, num_pos, code_weights, rows=None):
    """`loss_l1` and `loss_gd` of EVERY CenterGDHead task (gd_centerpoint_head.py:402-441) in one launch, reading the
    head outputs where they lie: no `_reconstruct_bbox` cat (:372-387), no `_gather_feat` (:59-63), no coder calls, no
    per-task loss modules and none of their autograd nodes.

    loss_gd: this package's GDLoss (reduction 'mean');  loss_bbox: mmdet's L1Loss module, anything with `.loss_weight`
    (and `.reduction` == 'mean'), or a config dict;  coder: CenterPointBBoxYawCoder;
    preds_dicts: per task a dict with the raw head outputs 'height', 'dim', 'yaw', 'dir' and optionally 'reg', 'vel'
    (B, c, H, W) — `preds_dict[0]` of the reference;  pos_inds: per task (n,3) long [batch, x, y] (:72-80);
    anno_boxes: per task (n, 7 | 9);  num_pos: per task the positive count (avg_factor = max(num_pos, 1), :405-408);
    code_weights: train_cfg['code_weights'] for [sin, cos(, vx, vy)].
    Device-resident form (no host value about the tasks' sizes: the call can sit inside a captured hipGraph):
    rows = (T+1,) int64 DEVICE tensor, task t = rows [rows[t], rows[t+1]) of ONE shared `pos_inds` (N,3) / `anno_boxes` (N,C)
    pair (what `center_head_get_targets(..., padded=True)` returns), and num_pos a (T,) float DEVICE tensor (what
    `center_head_heatmap_loss` returns): the kernels read both where they lie.
    Returns a list of (loss_l1, loss_gd) pairs, one per task (0-dim tensors; the graph reaches the head maps)."""
    from .gd_loss import GDLoss
    assert isinstance(loss_gd, GDLoss)
    if loss_gd.reduction != 'mean':
        raise ValueError('center_head_losses needs reduction="mean" (avg_factor is given)')
    if isinstance(loss_bbox, dict):
        kind, lw, red = loss_bbox.get('type', 'L1Loss'), float(loss_bbox.get('loss_weight', 1.0)), loss_bbox.get('reduction', 'mean')
    else:
        kind, lw, red = type(loss_bbox).__name__, float(loss_bbox.loss_weight), getattr(loss_bbox, 'reduction', 'mean')
    if kind != 'L1Loss' or red != 'mean':
        raise RuntimeError(f'encoded-box loss {kind!r} (reduction {red!r}) is not fused; supported: L1Loss, mean')
    T = len(preds_dicts)
    dyn = rows is not None
    if dyn:
        if not (isinstance(pos_inds, torch.Tensor) and isinstance(anno_boxes, torch.Tensor) and isinstance(num_pos, torch.Tensor)):
            raise RuntimeError('center_head_losses: with `rows`, pos_inds / anno_boxes are the shared (N,3) / (N,C) tensors and '
                               'num_pos a (T,) device tensor')
        if rows.dtype != torch.int64 or rows.numel() != T + 1 or not rows.is_cuda or num_pos.numel() != T or not num_pos.is_cuda:
            raise RuntimeError('center_head_losses: rows must be (T+1,) int64 and num_pos (T,) float32, both on the device')
        rows, num_pos = rows.contiguous(), num_pos.to(torch.float32).contiguous()
        pos_inds, anno_boxes = [pos_inds] * T, [anno_boxes] * T
    if not (len(pos_inds) == len(anno_boxes) == T) or (not dyn and len(num_pos) != T) or T == 0 or T > 8:
        raise RuntimeError('center_head_losses: 1..8 tasks, one entry per task in every list')
    has_vel = all('vel' in d for d in preds_dicts)
    n_l1 = 4 if has_vel else 2
    cw = [float(x) for x in code_weights]
    if len(cw) != n_l1:
        raise RuntimeError(f'{len(cw)} code_weights for {n_l1} L1 channels (dir{", vel" if has_vel else ""})')
    maps, layout, pis, ans, scales = [], [], [], [], []
    dev = preds_dicts[0]['height'].device
    for t, d in enumerate(preds_dicts):
        row = []
        B, _, H, W = d['height'].shape
        for h, name in enumerate(_CENTER_HEADS):
            if name in d and not (name == 'vel' and not has_vel):
                m = d[name]
                if tuple(m.shape) != (B, _CENTER_CH[h], H, W) or m.device != dev:
                    raise RuntimeError(f"task {t}: head '{name}' has shape {tuple(m.shape)}, expected {(B, _CENTER_CH[h], H, W)}")
                row.append(len(maps))
                maps.append((m if m.dtype == torch.float32 else m.float()).contiguous())
            elif name in ('height', 'dim', 'yaw', 'dir'):
                raise RuntimeError(f"task {t}: head '{name}' is missing")
            else:
                row.append(-1)
        layout.append(row)
        pi = pos_inds[t].reshape(-1, 3).to(device=dev, dtype=torch.int64).contiguous()
        an = anno_boxes[t]
        an = an.reshape(pi.shape[0], an.shape[-1] if an.dim() >= 1 and an.numel() else 7 + (2 if has_vel else 0))
        an = an.to(device=dev, dtype=torch.float32).contiguous()
        if an.shape[0] and an.shape[1] < 7 + (2 if has_vel else 0):
            raise RuntimeError(f'task {t}: anno_boxes has {an.shape[1]} columns')
        # (index VALUES are range-checked inside the kernel: an out-of-map cell touches no memory and turns the task's
        #  losses into NaN — no host-side min/max, i.e. no sync)
        pis.append(pi)
        ans.append(an)
        if dyn:    # (gd weight, l1 weight, address of rows[t], address of num_pos[t]): the division happens on the device
            scales.append((float(loss_gd.loss_weight), lw, rows.data_ptr() + 8 * t, num_pos.data_ptr() + 4 * t))
        else:
            avg = max(float(num_pos[t]), 1.0)
            scales.append((float(loss_gd.loss_weight) / avg, lw / avg))
    pro = _prologue(2, maps[0], norm_bbox=coder.norm_bbox, out_size_factor=coder.out_size_factor,
                    voxel_size=coder.voxel_size, pc_range=coder.pc_range)
    losses = _CenterHeadFused.apply((loss_gd._params({}), pro, layout, pis, ans, scales, cw, n_l1, rows, num_pos if dyn else None), *maps)
    flat = losses.reshape(-1).unbind(0)          # one autograd node for all 2T scalars
    return [(flat[2 * t], flat[2 * t + 1]) for t in range(T)]
