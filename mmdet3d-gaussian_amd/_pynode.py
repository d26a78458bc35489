"""Host glue above the C ABI in PYTHON: `torch.autograd.Function` + ctypes over libgd3d.so.

This is the package's first-class host layer — the shape of the reference's own native ops, a Python `Function` over a
native entry point (/root/reference/mmdet3d_gaussian/ops/voxel/scatter.py:29-72, ops/vsa/group_points.py:7-93) — and needs
nothing but `hipcc`'s output: no torch headers, no pybind11, no libtorch link.  csrc/torch_node.cpp (-> _gd3d_node.so) is an
OPTIONAL accelerator with the same function surface (a C++ node saves 7-30 us of Python per training-size call,
profiles/r04_node_ab.txt); `_lib.load_node()` picks one of the two (`GD3D_HOST=python|cpp`).  Both are glue: the same
`extern "C"` entry points with the same arguments, so values and gradients are bit-identical between them
(tests/test_autograd_node.py, tests/test_gpu_host_glue.py, and the GPU suites parametrised over both).

Surface (argument for argument that of csrc/torch_node.cpp):
  reduced(...)        GDLoss's reduced forms ('mean' / 'sum'): one fused launch writes the loss sum AND the final gradients; they
                      wait in the node; backward hands them over (scaled on the device unless the upstream gradient is
                      gd_loss.unit_grad, known by address); a retain_graph replay recomputes; double backward raises.
  anchor_head(...)    the anchor-head regression slice (head_loss.py, gd_anchor3d_head.py:95-161) as one node.
  scatter_reduce(...) dynamic scatter-reduce (scatter.py, ops/voxel/scatter.py:29-72) as one node.
  nms_scored(...)     nms_gpu's scored path: allocate, launch, read the count back, cut.
  set_unit_grad / finish_calls / bind: bookkeeping the C++ module also exports.
"""
import ctypes
import threading
import time

import torch

from . import _lib

IMPLEMENTATION = 'python'

_raw_stream = torch._C._cuda_getCurrentRawStream
_get_device = torch._C._cuda_getDevice
_set_device = torch._C._cuda_setDevice

_FINISH_CALLS = 0          # gd3d_grad_finish launches made by backward so far (tests: unit_grad must make none)
_UNIT_GRAD = {}            # device index (-1: the CPU) -> address of the library's constant 1.0


def guard_double_backward(impl):
    """The backward functions here return gradients that ctypes kernels wrote: no autograd graph hangs off them.  Under
    `create_graph=True` (the only case in which grad mode is ON inside a backward) the results are put behind torch's
    DelayedError node, so differentiating them again RAISES instead of silently treating them as constants.  That is what
    torch.autograd.function.once_differentiable does — except that it only does so when an incoming GRADIENT requires grad,
    which the ones tensor of a plain `autograd.grad(loss, x, create_graph=True)` does not; the gradients here depend on the
    saved INPUTS, so the guard is unconditional (as in the C++ twin).  A plain backward pays one flag test."""
    def backward(ctx, *grads):
        if not torch.is_grad_enabled():
            return impl(ctx, *grads)
        with torch.no_grad():
            outputs = impl(ctx, *grads)
        single = not isinstance(outputs, tuple)
        if single:
            outputs = (outputs,)
        err = torch._C._functions.DelayedError(
            b'trying to differentiate twice a function that was marked with @once_differentiable', len(outputs))
        alias = []
        for v in outputs:
            if v is not None:
                v = v.detach()
                v.requires_grad = True
            alias.append(v)
        res = err(*alias)
        return res[0] if single else res
    return backward


class _on_device:
    """Minimal device guard (raw accessors: no Python-level bookkeeping on the per-call path); yields the raw current stream."""
    __slots__ = ('idx', 'prev')

    def __init__(self, dev):
        self.idx = dev.index

    def __enter__(self):
        self.prev = _get_device()
        if self.prev != self.idx:
            _set_device(self.idx)
        return _raw_stream(self.idx)

    def __exit__(self, *exc):
        if self.prev != self.idx:
            _set_device(self.prev)
        return False


def bind(path):
    """The C++ module resolves the C ABI from `path` here; the Python glue calls through the ctypes binding `_lib.load()`
    made of the same image.  Returns the ABI version."""
    import os
    if not os.path.isfile(path):
        raise RuntimeError(f'gd3d glue: cannot open {path}')
    return int(_lib.load().gd3d_abi_version(None))


def set_unit_grad(device_index, address):
    _UNIT_GRAD[-1 if device_index < 0 else int(device_index)] = int(address)


def finish_calls():
    return _FINISH_CALLS


def _is_unit_grad(g):
    a = _UNIT_GRAD.get(g.device.index if g.is_cuda else -1)
    return a is not None and g.dim() == 0 and g.dtype == torch.float32 and g.data_ptr() == a


def _ptr(t):
    return None if t is None else t.data_ptr()


def _copy_struct(cls, addr):
    """A private copy of the caller's ctypes struct (the node outlives the call; GDLoss caches and rewrites its params)."""
    return cls.from_buffer_copy(ctypes.string_at(addr, ctypes.sizeof(cls)))


def _weights(w):
    if w is None:
        return None, None
    return (None, w.data_ptr()) if w.dim() == 2 else (w.data_ptr(), None)


# ---- GDLoss, reduced forms ------------------------------------------------------------------------------------------------------

class GDLossReduced(torch.autograd.Function):
    """scale * sum_i w_i L_i with the final gradients produced by the SAME launch (see module docstring)."""

    # (the non-tensor arguments travel as ONE tuple: Function.apply walks its argument list several times per call — functorch
    #  unwrapping, needs_input_grad — and thirteen arguments cost ~3 us more than four)
    @staticmethod
    def forward(ctx, pred, target, weight, call):
        params, prologue, aux, scale, select, ticket, ev0, ev1, ws_floats, flag_box = call
        need_gp, need_gt = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        gp, gt, buf = _reduced_launch(pred, target, weight, params, prologue, scale, select, ticket, ev0, ev1, ws_floats,
                                      need_gp, need_gt, True)
        ctx.save_for_backward(pred, target)      # version-checked: an in-place edit before backward raises, a released graph too
        ctx.weight, ctx.aux, ctx.gp, ctx.gt = weight, aux, gp, gt
        ctx.buf = buf if select else None        # owns the any-positive flag backward reads on the device
        ctx.call = (params, prologue, scale, select)
        ctx.want, ctx.used = (need_gp, need_gt), False
        if flag_box is not None:   # the flag leaves through a side door: a second autograd OUTPUT that is a view of the same
            flag_box.append(buf[1:2].view(torch.int32))   # buffer as the result would make every backward pay view bookkeeping
        return buf[0]

    @staticmethod
    @guard_double_backward
    def backward(ctx, grad_out):
        global _FINISH_CALLS
        pred, target = ctx.saved_tensors
        params, prologue, scale, select = ctx.call
        n = pred.shape[0]
        if ctx.used:   # retain_graph replay: the buffers of the first backward were handed over (and scaled in place)
            gp, gt, _ = _reduced_launch(pred, target, ctx.weight, params, prologue, scale, False, 0, 0, 0, 0, ctx.want[0],
                                        ctx.want[1], False)
        else:          # hand the buffers over: with no reference left here a leaf's AccumulateGrad keeps them instead of cloning
            gp, gt = ctx.gp, ctx.gt
            ctx.gp = ctx.gt = None
            ctx.used = True
        if select or not _is_unit_grad(grad_out):
            lib = _lib.load()
            g = grad_out if grad_out.dtype == torch.float32 else grad_out.float()
            if not pred.is_cuda:
                g = g.reshape(1)
                for arr in (gp, gt):
                    if arr is not None:
                        _lib.check(lib.gd3d_scale_rows_cpu(arr.data_ptr(), g.data_ptr(), 0, n, torch.get_num_threads()),
                                   'gd3d_scale_rows_cpu')
            else:
                # one launch for both arrays: reads g (and the any-positive flag) on the device; leaves without touching memory
                # when g == 1 and the normal branch was taken (no host sync)
                _FINISH_CALLS += 1
                with _on_device(pred.device) as stream:
                    rc = lib.gd3d_grad_finish(_ptr(gp), _ptr(gt), g.data_ptr(), n,
                                              ctx.buf.data_ptr() + 4 if select else None,
                                              ctx.weight.data_ptr() if select else None,
                                              pred.data_ptr() if (select and prologue is not None) else None,
                                              prologue if select else None, stream)
                if rc != 0:
                    _lib.check(rc, 'gd3d_grad_finish')
        return gp, gt, None, None


def _reduced_launch(pred, target, weight, params, prologue, scale, select, ticket, ev0, ev1, ws_floats, need_gp, need_gt,
                    want_sum):
    """One fused launch.  With `want_sum`: a buffer [0] = the fp32 result, [1] = the int32 any-positive flag, [4:] = workspace
    (16-byte aligned); without (the replay): gradients only."""
    lib = _lib.load()
    n = pred.shape[0]
    w1, w7 = _weights(weight)
    if not pred.is_cuda:
        gp = torch.empty_like(pred) if need_gp else None
        gt = torch.empty_like(target) if need_gt else None
        buf = torch.empty(4 + ws_floats, dtype=torch.float32) if want_sum else None
        base = buf.data_ptr() if want_sum else None
        rc = lib.gd3d_loss_fused_cpu(params, pred.data_ptr(), target.data_ptr(), w1, w7, n, scale, None, base, _ptr(gp), _ptr(gt),
                                     base + 16 if want_sum else None, torch.get_num_threads())
        if rc != 0:
            _lib.check(rc, 'gd3d_loss_fused_cpu')
        return gp, gt, buf
    with _on_device(pred.device) as stream:
        gp = torch.empty_like(pred) if need_gp else None
        gt = torch.empty_like(target) if need_gt else None
        if not want_sum:
            buf = None
            rc = lib.gd3d_loss_fused_decoded(params, prologue, pred.data_ptr(), target.data_ptr(), w1, w7, n, scale, None, None,
                                             _ptr(gp), _ptr(gt), None, stream)
        else:
            buf = torch.empty(4 + ws_floats, dtype=torch.float32, device=pred.device)
            base = buf.data_ptr()
            flag = base + 4 if select else None
            if ticket:     # training-size call: ONE launch, the last workgroup finishes the sum
                rc = lib.gd3d_loss_fused_one_launch(params, prologue, pred.data_ptr(), target.data_ptr(), w1, w7, n, scale, base, flag,
                                                    _ptr(gp), _ptr(gt), base + 16, ticket, stream)
            elif select:
                rc = lib.gd3d_loss_fused_select(params, prologue, pred.data_ptr(), target.data_ptr(), w7, n, scale, base, flag,
                                                _ptr(gp), _ptr(gt), base + 16, stream, ev0 or None, ev1 or None)
            else:
                rc = lib.gd3d_loss_fused_timed(params, prologue, pred.data_ptr(), target.data_ptr(), w1, w7, n, scale, None, base,
                                               _ptr(gp), _ptr(gt), base + 16, stream, ev0 or None, ev1 or None)
    if rc != 0:
        _lib.check(rc, 'gd3d_loss_fused')
    return gp, gt, buf


def reduced(pred, target, weight, params, prologue, aux, scale, select, ticket, ev_start, ev_stop, ws_floats, want_flag):
    """Returns (loss sum as a 0-dim fp32 tensor, the int32 any-positive flag (1,) or None).  params / prologue: addresses of
    gd3d_params / gd3d_prologue (copied); aux: the tensor prologue->aux points into; ticket: device int32 of
    gd3d_loss_fused_one_launch or 0 (two-stage form); ev_start / ev_stop: hipEvent_t from gd3d_prof_event_create or 0;
    ws_floats: gd3d_loss_workspace_bytes(n) / 4."""
    if not (pred.dim() == 2 and pred.shape[1] == 7 and pred.dtype == torch.float32 and pred.is_contiguous() and
            target.shape == pred.shape and target.dtype == torch.float32 and target.is_contiguous() and
            target.device == pred.device):
        raise RuntimeError('gd3d node: pred / target must be contiguous fp32 (N, 7) tensors on one device')
    if not (pred.is_cuda or pred.device.type == 'cpu'):
        raise RuntimeError(f'gd3d node: no implementation for device {pred.device}')
    n = pred.shape[0]
    if weight is not None and not (weight.dtype == torch.float32 and weight.is_contiguous() and weight.device == pred.device and
                                   weight.dim() in (1, 2) and weight.shape[0] == n and (weight.dim() == 1 or weight.shape[1] == 7)):
        raise RuntimeError("gd3d node: weight must be a contiguous fp32 (N,) or (N, 7) tensor on pred's device")
    if select and not (weight is not None and weight.dim() == 2 and pred.is_cuda):
        raise RuntimeError('gd3d node: select needs an (N, 7) weight on the GPU')
    if prologue and not pred.is_cuda:
        raise RuntimeError('GDLoss: the head-level fusions (bbox-coder prologue) are GPU-only')
    p = _copy_struct(_lib.Params, params)
    pro = _copy_struct(_lib.Prologue, prologue) if prologue else None
    return reduced_trusted(pred, target, weight, p, pro, aux, float(scale), bool(select), int(ticket), int(ev_start), int(ev_stop),
                           int(ws_floats), want_flag)


def reduced_trusted(pred, target, weight, params, prologue, aux, scale, select, ticket, ev_start, ev_stop, ws_floats, want_flag):
    """`reduced` for a caller that has validated its operands itself and hands over ctypes STRUCTS it never rewrites
    (GDLoss.forward: `_rows`, the weight normalisation and `make_params`, which builds a new struct whenever a hyper-parameter
    changes): no checks, no copies — the Python glue's per-call cost is what separates it from the C++ node."""
    box = [] if want_flag else None
    total = GDLossReduced.apply(pred, target, weight, (params, prologue, aux, scale, select, ticket, ev_start, ev_stop, ws_floats, box))
    return total, (box[0] if want_flag else None)


# ---- the anchor-head regression slice ----------------------------------------------------------------------------------------------

def _anchor_head_launch(bbox_pred, bbox_targets, bbox_weights, anchors, sel, params, sl1, dw, dense, num_classes, scale, avg_dev,
                        w_gd, w_sl1, need_grad):
    """One gd3d_anchor_head_bbox_loss[_dyn] launch -> (loss scalar tensor, zero-filled-then-scattered NCHW gradient | None)."""
    lib = _lib.load()
    B, C, H, W = bbox_pred.shape
    P = sel.numel()
    with _on_device(bbox_pred.device) as stream:
        grad = torch.zeros_like(bbox_pred) if need_grad else None
        buf = torch.empty(4 + lib.gd3d_loss_workspace_bytes(P) // 4, dtype=torch.float32, device=bbox_pred.device)
        base = buf.data_ptr()
        dwp = None if dw is None else (ctypes.c_float * 7)(*dw)
        if avg_dev is not None:   # the normaliser stays on the device (dense form only)
            rc = lib.gd3d_anchor_head_bbox_loss_dyn(params, sl1, bbox_pred.data_ptr(), B, C // 7, H, W, bbox_targets.data_ptr(),
                                                    _ptr(bbox_weights), dwp, anchors.data_ptr(), sel.data_ptr(), num_classes, w_gd,
                                                    w_sl1, avg_dev.data_ptr(), base, _ptr(grad), base + 16, stream)
        else:
            rc = lib.gd3d_anchor_head_bbox_loss(params, sl1, bbox_pred.data_ptr(), B, C // 7, H, W, bbox_targets.data_ptr(),
                                                _ptr(bbox_weights), dwp, anchors.data_ptr(), None if dense else sel.data_ptr(), P,
                                                sel.data_ptr() if dense else None, num_classes, scale, base, _ptr(grad), base + 16,
                                                stream)
    if rc != 0:
        _lib.check(rc, 'gd3d_anchor_head_bbox_loss')
    return buf[0], grad


class GDAnchorHead(torch.autograd.Function):
    @staticmethod
    def forward(ctx, bbox_pred, args):
        loss, grad = _anchor_head_launch(bbox_pred, *args, ctx.needs_input_grad[0])
        ctx.save_for_backward(bbox_pred)
        ctx.grad, ctx.used, ctx.args = grad, False, args
        return loss

    @staticmethod
    @guard_double_backward
    def backward(ctx, grad_out):
        (bbox_pred,) = ctx.saved_tensors   # raises after a released graph; checks in-place edits of the head output
        if ctx.used:   # retain_graph replay: the first gradient was handed over (and scaled in place): launch again
            g = _anchor_head_launch(bbox_pred, *ctx.args, True)[1]
        else:
            g, ctx.grad, ctx.used = ctx.grad, None, True
        if not _is_unit_grad(grad_out):
            go = grad_out if grad_out.dtype == torch.float32 else grad_out.float()
            with _on_device(g.device) as stream:
                rc = _lib.load().gd3d_scale_rows(g.data_ptr(), go.data_ptr(), 0, g.numel() // 7, stream)
            if rc != 0:
                _lib.check(rc, 'gd3d_scale_rows')
        return g, None


def anchor_head(bbox_pred, bbox_targets, bbox_weights, anchors, sel, params, sl1, decode_weight, dense, num_classes, scale,
                avg_dev, w_gd, w_sl1):
    """Selection / gather of the positives + decode x2 + loss(es) + the gradient scattered into the NCHW head output: ONE launch
    (argument meaning: head_loss._anchor_head_fused)."""
    if not (bbox_pred.is_cuda and bbox_pred.dim() == 4 and bbox_pred.shape[1] % 7 == 0 and bbox_pred.dtype == torch.float32 and
            bbox_pred.is_contiguous()):
        raise RuntimeError('gd3d node: bbox_pred must be a contiguous fp32 (B, A*7, H, W) tensor on the GPU')
    for t in (bbox_targets, anchors):
        if not (t.dtype == torch.float32 and t.is_contiguous() and t.device == bbox_pred.device):
            raise RuntimeError("gd3d node: targets / anchors must be contiguous fp32 tensors on bbox_pred's device")
    if not (sel.dtype == torch.int64 and sel.is_contiguous() and sel.device == bbox_pred.device):
        raise RuntimeError("gd3d node: the positive list / label map must be a contiguous int64 tensor on bbox_pred's device")
    if bbox_weights is not None and not (bbox_weights.dtype == torch.float32 and bbox_weights.is_contiguous() and
                                         bbox_weights.device == bbox_pred.device):
        raise RuntimeError("gd3d node: bbox_weights must be a contiguous fp32 tensor on bbox_pred's device")
    if decode_weight is not None and len(decode_weight) != 7:
        raise RuntimeError('gd3d node: decode_weight must hold 7 values')
    if avg_dev is not None and not (dense and avg_dev.dtype == torch.float32 and avg_dev.numel() == 1 and
                                    avg_dev.device == bbox_pred.device):
        raise RuntimeError("gd3d node: a device-resident normaliser needs the dense form and one fp32 value on bbox_pred's device")
    p = _copy_struct(_lib.Params, params)
    s = _copy_struct(_lib.SmoothL1, sl1) if sl1 else None
    dw = None if decode_weight is None else [float(x) for x in decode_weight]
    return GDAnchorHead.apply(bbox_pred, (bbox_targets, bbox_weights, anchors, sel, p, s, dw, bool(dense), int(num_classes),
                                          float(scale), avg_dev, float(w_gd), float(w_sl1)))


# ---- dynamic scatter-reduce -----------------------------------------------------------------------------------------------------------

class GDScatterReduce(torch.autograd.Function):
    """forward: vox_scatter_reduce over the grouped points; backward: the voxel-ordered form for rows of 128 bytes and more
    (c % 4 == 0, 32 <= c <= 256, aligned: every gradient row read once and streamed to its points), the map-ordered gather for
    narrower rows (c = 10: 40 us against 118 us, profiles/r04_scatter_kernel_time.txt)."""

    @staticmethod
    def forward(ctx, feats, pmap, count, red, order, seg):
        lib = _lib.load()
        n, c = feats.shape
        v = count.numel()
        with _on_device(feats.device) as stream:
            f32 = feats.contiguous() if feats.dtype == torch.float32 else feats.float().contiguous()
            out = torch.empty((v, c), dtype=torch.float32, device=feats.device)
            argmax = torch.empty((v, c), dtype=torch.int32, device=feats.device) if red == 2 else None
            rc = lib.vox_scatter_reduce(f32.data_ptr(), order.data_ptr(), seg.data_ptr(), n, c, v, red, out.data_ptr(),
                                        _ptr(argmax), stream)
        if rc != 0:
            _lib.check(rc, 'vox_scatter_reduce')
        ctx.red, ctx.shape, ctx.in_dtype, ctx.has_argmax = red, (n, c, v), feats.dtype, argmax is not None
        ctx.save_for_backward(*((pmap, count, order, seg) + ((argmax,) if argmax is not None else ())))   # version-checked
        return out if feats.dtype == torch.float32 else out.to(feats.dtype)

    @staticmethod
    @guard_double_backward
    def backward(ctx, grad_voxel_feats):
        lib = _lib.load()
        saved = ctx.saved_tensors
        pmap, count, order, seg = saved[:4]
        am = saved[4].data_ptr() if ctx.has_argmax else None
        n, c, v = ctx.shape
        g = grad_voxel_feats.contiguous().float()
        with _on_device(g.device) as stream:
            gf = torch.empty((n, c), dtype=torch.float32, device=g.device)
            if c % 4 == 0 and 32 <= c <= 256 and g.data_ptr() % 16 == 0:
                rc = lib.vox_scatter_backward_grouped(g.data_ptr(), order.data_ptr(), seg.data_ptr(), am, n, c, v, ctx.red,
                                                      gf.data_ptr(), stream)
            else:
                rc = lib.vox_scatter_backward(g.data_ptr(), pmap.data_ptr(), count.data_ptr(), am, n, c, v, ctx.red, gf.data_ptr(),
                                              stream)
        if rc != 0:
            _lib.check(rc, 'vox_scatter_backward')
        return (gf if ctx.in_dtype == torch.float32 else gf.to(ctx.in_dtype)), None, None, None, None, None


def scatter_reduce(feats, point2voxel_map, voxel_points_count, reduce, order, seg):
    if not (feats.is_cuda and feats.dim() == 2):
        raise RuntimeError('scatter_reduce: feats must be an (N, C) tensor on the GPU')
    for t in (point2voxel_map, voxel_points_count, order, seg):
        if not (t.dtype == torch.int32 and t.is_contiguous() and t.device == feats.device):
            raise RuntimeError("scatter_reduce: map / count / order / seg must be contiguous int32 tensors on feats' device")
    n, v = feats.shape[0], voxel_points_count.numel()
    if not (point2voxel_map.numel() == n and order.numel() == n and seg.numel() == v + 1 and 0 <= reduce <= 2):
        raise RuntimeError('scatter_reduce: inconsistent index tensors')
    return GDScatterReduce.apply(feats, point2voxel_map, voxel_points_count, int(reduce), order, seg)


# ---- nms_gpu's scored path --------------------------------------------------------------------------------------------------------------

_NMS_WS = {}
_PENDING = -(1 << 62)
_MAILBOX = threading.local()


def count_mailbox(g):
    """This thread's mailbox: (pinned int64 tensor, its numpy view) of at least g words.  nms_gpu's result length is data
    dependent; instead of a blocking 8-byte device-to-host copy (a copy call + a stream synchronisation, ~10 us on this stack) the
    scan kernel writes its count straight into pinned host memory — every NMS entry point takes `num_keep` as a plain pointer —
    and the host polls the word: 55.0 -> 49.8 us per nms_gpu-sized call (n = 4096; profiles/r06_nms_batched.txt).  The kept
    ids stay on the device, stream-ordered as before.  One mailbox per thread: a call blocks until its words arrive."""
    cur = getattr(_MAILBOX, 'box', None)
    if cur is None or cur[0].numel() < g:
        t = torch.empty(max(64, g), dtype=torch.int64).pin_memory()
        cur = _MAILBOX.box = (t, t.numpy())
    return cur


def wait_counts(words, g, dev):
    """Poll the first g mailbox words until the kernels have written them all; returns them as ints.  After 0.2 s without them
    the device is synchronised and the words are read once more (a word still pending then raises)."""
    spins, deadline = 0, None
    while True:
        vals = [int(words[i]) for i in range(g)]
        if _PENDING not in vals:
            return vals
        spins += 1
        if (spins & 0x3ff) == 0:
            now = time.perf_counter()
            if deadline is None:
                deadline = now + 0.2
            elif now > deadline:
                torch.cuda.synchronize(dev)
                vals = [int(words[i]) for i in range(g)]
                if _PENDING in vals:
                    raise RuntimeError('nms_gpu: the NMS kernels finished without reporting a count')
                return vals


def nms_scored(boxes, scores, thresh, n_keep, normal, padded, post_max):
    """nms_gpu's scored path (iou3d.py: <= rnms_scored_max_n() candidates, fp32 scores): the three allocations, the launch and —
    unless `padded` — the one read-back of the count and the cut to it.  Returns (keep, num): padded: keep (n_keep) int64 whose
    first num[0] entries are valid, num (1) int64 on the device; otherwise keep is cut to the count (and to post_max when >= 0)
    and num is None."""
    if not (boxes.is_cuda and boxes.dim() == 2 and boxes.shape[1] == 5 and boxes.dtype == torch.float32 and boxes.is_contiguous() and
            scores.dim() == 1 and scores.shape[0] == boxes.shape[0] and scores.dtype == torch.float32 and scores.is_contiguous() and
            scores.device == boxes.device and 0 < n_keep <= boxes.shape[0]):
        raise RuntimeError('gd3d node: nms_scored takes contiguous fp32 (N,5) boxes and (N) scores on one GPU and 0 < n_keep <= N')
    lib = _lib.load()
    n_all = boxes.shape[0]
    nbytes = _NMS_WS.get((n_all, n_keep))
    if nbytes is None:
        if len(_NMS_WS) > 4096:
            _NMS_WS.clear()
        nbytes = _NMS_WS[(n_all, n_keep)] = int(lib.rnms_scored_workspace_bytes(n_all, n_keep))
    dev = boxes.device
    with _on_device(dev) as stream:
        keep = torch.empty(n_keep, dtype=torch.int64, device=dev)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        if padded:
            num = torch.empty(1, dtype=torch.int64, device=dev)
            dst = num.data_ptr()
        else:   # the count goes straight into pinned host memory and is polled there (see count_mailbox)
            box, words = count_mailbox(1)
            words[0] = _PENDING
            dst = box.data_ptr()
        rc = lib.rnms_scored(int(normal), boxes.data_ptr(), scores.data_ptr(), n_all, n_keep, float(thresh), keep.data_ptr(),
                             dst, ws.data_ptr(), stream)
    if rc != 0:
        _lib.check(rc, 'nms_normal_gpu' if normal else 'nms_gpu')
    if padded:
        return keep, num
    k = wait_counts(words, 1, dev)[0]   # the one unavoidable wait: the result length is data dependent
    if k < 0:             # the scan kernel's failure mark (a bounded polling loop gave up: never observed)
        raise RuntimeError(f'nms_gpu: the device-side NMS scan gave up (num_keep = {k}); the result is void')
    if post_max >= 0 and k > post_max:
        k = post_max
    return keep[:k], None
