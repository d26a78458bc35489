"""GDLoss — host-side mirror of the reference's loss module on top of the fused HIP kernels.

Mirrors /root/reference/mmdet3d_gaussian/models/losses/gaussian_distance_loss.py:251-310
(``GDLoss``): same constructor arguments, asserts and defaults, same ``forward`` signature and
host logic (reduction override, all-zero-weight early-out, (N,7)->(N,) weight mean, kwargs merge,
``* loss_weight``), and the mmdet ``weighted_loss`` / ``weight_reduce_loss`` reduction rules the
loss functions are wrapped in there (:42,109,144,...).  What differs is below the module: the
reference runs preprocess x2 + ~110-145 ATen ops + autograd; here one fused kernel computes the
loss AND the final gradient(s) in a single pass over the (N,7) rows (csrc/gd3d_loss.hip), reached
through the C ABI of include/gd3d.h.

Like the reference module (whose forward is device-agnostic, :280-310), GDLoss follows the device of its tensors: GPU tensors
take the HIP kernels, CPU tensors the library's `_cpu` twins (gd3d_loss_fused_cpu: the kernel's own per-pair math compiled
for the host, csrc/gd3d_cpu.cpp).  The two never substitute for each other: a GPU tensor whose kernel cannot run raises.
"""
import ctypes
import os
from copy import deepcopy

import torch
from torch import nn
from . import _lib
from ._pynode import guard_double_backward  # noqa: F401  (head_loss / heat_loss / anchor_cls import it from here)
from .registry import LOSSES, register_with_mmdet

LOSS_TYPES = {'gwd3d': 0, 'kld3d': 1, 'bd3d': 2, 'jd3d': 3, 'kld3d_symmax': 4, 'kld3d_symmin': 5, 'kfiou3d': 6}
FUNS = {'none': 0, 'log1p': 1, 'expm1': 2, 'nlog': 3}

# bench.py sets this to a list; every fused launch then appends a DispatchTimer: a pair of HIP events bound to the
# begin / end timestamps of that kernel's own dispatch (gd3d_loss_fused_timed), inside the timed region.
PROFILE_EVENTS = None
# measurement switch (tests/perf/small_p_latency.py): 1 = decide the no-positive-weight early-out on the HOST as the
# reference does (torch.any + a device-to-host wait per call) instead of inside the fused launch
_HOST_WEIGHT_CHECK = os.environ.get('GD3D_HOST_WEIGHT_CHECK', '0') == '1'


class DispatchTimer:
    """Two hipEvent_t owned by libgd3d.so's runtime, handed to gd3d_loss_fused_timed; `elapsed_ms()` after a
    synchronize is the execution time of the one fused dispatch they were bound to."""
    __slots__ = ('start', 'stop', '_abi')

    def __init__(self):
        self.start = self.stop = None
        self._abi = _library()
        for slot in ('start', 'stop'):
            h = ctypes.c_void_p()
            _lib.check(self._abi.gd3d_prof_event_create(ctypes.byref(h)), 'gd3d_prof_event_create')
            setattr(self, slot, h.value)

    def elapsed_ms(self):
        ms = ctypes.c_float()
        _lib.check(self._abi.gd3d_prof_event_elapsed_ms(self.start, self.stop, ctypes.byref(ms)),
                   'gd3d_prof_event_elapsed_ms')
        return ms.value

    def __del__(self):
        for slot in ('start', 'stop'):
            h = getattr(self, slot, None)
            if h:
                self._abi.gd3d_prof_event_destroy(h)
                setattr(self, slot, None)


def make_params(loss_type, fun, tau, alpha, center_offset, kwargs):
    """Loss hyper-parameters -> gd3d_params.  `normalize` (gwd3d) / `sqrt` (others) are the only
    keyword arguments the reference's loss functions accept beyond fun/tau/alpha (ref :42,109,...);
    anything else is a TypeError there as well."""
    kwargs = dict(kwargs)
    if loss_type == 'gwd3d':
        flag = kwargs.pop('normalize', True)
    elif loss_type == 'kfiou3d':
        flag = kwargs.pop('sqrt', False)
    else:
        flag = kwargs.pop('sqrt', True)
    if kwargs:
        raise TypeError(f'{loss_type}_loss() got unexpected keyword argument(s) {sorted(kwargs)}')
    if isinstance(center_offset, torch.Tensor):
        center_offset = center_offset.detach().cpu().tolist()
    p = _lib.Params()
    p.loss_type = LOSS_TYPES[loss_type]
    p.fun = FUNS[fun]
    p.tau = float(tau)
    p.alpha = float(alpha)
    p.center_offset = (ctypes.c_float * 3)(*[float(c) for c in center_offset])
    p.flag = int(bool(flag))
    return p


def _ptr(t):
    return None if t is None else t.data_ptr()


_LIB = None
# raw-handle accessors (no Python-level device bookkeeping on the per-call path)
_raw_stream = torch._C._cuda_getCurrentRawStream
_get_device = torch._C._cuda_getDevice
_set_device = torch._C._cuda_setDevice


def _library():
    global _LIB
    if _LIB is None:
        _LIB = _lib.load()
    return _LIB


def _node():
    """The host glue the reduced forms go through (`_lib.load_node()`): _pynode.py — a torch.autograd.Function over ctypes — or
    the optional C++ node of csrc/torch_node.cpp (a Python Function costs 4 us per forward and 23-41 us per backward before it
    does anything: profiles/r04_small_p_latency.jsonl).  Same C ABI calls either way."""
    return _lib.load_node()


def _rows(t):
    """(…,7) any float dtype -> contiguous fp32 (N,7), on the device it lives on (the hot path is fp32: heads call it
    under @force_fp32, gd_anchor3d_head.py:167)."""
    if t.dim() != 2 or t.shape[1] != 7:
        t = t.reshape(-1, 7)
    if t.dtype != torch.float32:
        t = t.float()
    return t if t.is_contiguous() else t.contiguous()


_WS_FLOATS = {}
_ONE_MAX = None
_TICKETS = {}
_GRAPH_TICKETS = {}      # device index -> [int32 pool, slots handed out]
_GRAPH_SLOTS = 4096
_capturing = torch._C._cuda_isCurrentStreamCapturing


def _one_launch_max():
    global _ONE_MAX
    if _ONE_MAX is None:
        _ONE_MAX = int(_library().gd3d_one_launch_max_n()) if os.environ.get('GD3D_TWO_STAGE', '0') != '1' else -1
    return _ONE_MAX


def _ticket(dev_index, stream):
    """The arrival-ticket word of gd3d_loss_fused_one_launch.  Eager calls: one zeroed int32 per (device, stream),
    allocated once; every call leaves it zero again and calls on one stream are ordered, so they share it.  A call that is
    being CAPTURED into a hipGraph gets a word of its own from a per-device pool allocated outside the capture (a replay
    may run while eager calls use the capture stream's word); None when no such word is available -> two-stage form."""
    if _capturing():
        pool = _GRAPH_TICKETS.get(dev_index)
        if pool is None or pool[1] >= _GRAPH_SLOTS:
            return None
        pool[1] += 1
        return pool[0].data_ptr() + 16 * (pool[1] - 1)
    t = _TICKETS.get((dev_index, stream))
    if t is None:
        if len(_TICKETS) > 256:
            _TICKETS.clear()
        dev = torch.device('cuda', dev_index)
        t = _TICKETS[(dev_index, stream)] = torch.zeros(4, dtype=torch.int32, device=dev)
        if dev_index not in _GRAPH_TICKETS:
            _GRAPH_TICKETS[dev_index] = [torch.zeros(4 * _GRAPH_SLOTS, dtype=torch.int32, device=dev), 0]
    return t.data_ptr()


def _ws_floats(n):
    """gd3d_loss_workspace_bytes(n) / 4, memoised (one ctypes call less per forward)."""
    k = _WS_FLOATS.get(n)
    if k is None:
        if len(_WS_FLOATS) > 4096:
            _WS_FLOATS.clear()
        k = _WS_FLOATS[n] = _library().gd3d_loss_workspace_bytes(n) // 4
    return k


_UNIT_GRAD = {}


def unit_grad(device):
    """The upstream gradient 1.0 as a CONSTANT of this library: one read-only 0-dim fp32 tensor per device.

    `loss.backward()` makes torch fill a fresh ones tensor (one launch) and our backward then has to READ it on the device
    to learn that nothing needs scaling (one more launch, gd3d_grad_finish's early exit).  A training step that passes this
    tensor instead -- `torch.autograd.backward([l0, l1, l2], grad_tensors=[unit_grad(dev)] * 3)` -- is recognised by
    ADDRESS (no read, no sync): the gradients the fused forward launch wrote are already final and backward launches
    nothing.  Never write to the returned tensor."""
    dev = torch.device(device)
    if dev.type == 'cpu':
        idx = 'cpu'
    elif dev.type == 'cuda':
        idx = dev.index if dev.index is not None else _get_device()
    else:
        raise RuntimeError(f'unit_grad: no implementation for device type {dev.type!r}')
    t = _UNIT_GRAD.get(idx)
    if t is None:
        t = _UNIT_GRAD[idx] = torch.ones((), dtype=torch.float32, device=dev if idx == 'cpu' else torch.device('cuda', idx))
        _lib.register_unit_grad(-1 if idx == 'cpu' else idx, t.data_ptr())   # the glue (either one) knows it by address
    return t


# `loss.backward()` without an explicit gradient makes torch FILL a ones tensor (one launch) that each GDLoss node then has to
# read on the device to learn that nothing needs scaling (one early-exit launch per node): four tiny dispatches at a 4.1 us
# floor each per three-loss step (profiles/r06_acc_ab.txt), 3 % of the 10 M-pair step.  The reduced forms therefore return their
# scalar as a LossValue — a torch.Tensor subclass that changes ONE thing: `.backward()` with no gradient on a 0-dim fp32 value
# passes the library's unit constant (`unit_grad`) as the root gradient instead of letting torch fill a fresh one.  Sums and
# products of LossValues with anything are LossValues (torch's subclass propagation), so `(l0 + l1 + l2).backward()`, mmdet's
# `_parse_losses` sum and an AMP `(loss * scale).backward()` all start from the constant; an `Add` node hands the same tensor on
# and the nodes recognise it by address (no launch), a `Mul` node hands on a new tensor and the nodes scale on the device as
# before.  Values and gradients are bit-identical to the plain path (a finish launch at g == 1 touches nothing).
# `torch.autograd.backward(loss)`, `torch.autograd.grad`, an explicit `gradient=` and non-fp32 values take torch's own path.
# GD3D_UNIT_ROOT=0 switches the subclass off (the reduced forms then return plain tensors, as through round 5).
_UNIT_ROOT = os.environ.get('GD3D_UNIT_ROOT', '1') != '0'
_TENSOR_BACKWARD = torch.Tensor.backward


class LossValue(torch.Tensor):
    """A loss scalar of this package: a plain tensor in every respect but `.backward()` (see the note above)."""

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        if func is _TENSOR_BACKWARD and _UNIT_ROOT:
            kw = dict(kwargs) if kwargs else {}
            self = args[0]
            gradient = kw.pop('gradient', args[1] if len(args) > 1 else None)
            if (gradient is None and self.dim() == 0 and self.dtype == torch.float32 and self.requires_grad and
                    (self.is_cuda or self.device.type == 'cpu')):
                names = ('retain_graph', 'create_graph', 'inputs')
                for k, v in zip(names, args[2:]):
                    kw.setdefault(k, v)
                with torch._C.DisableTorchFunctionSubclass():
                    return torch.autograd.backward(self, unit_grad(self.device), kw.get('retain_graph'),
                                                   kw.get('create_graph', False), inputs=kw.get('inputs'))
        return super().__torch_function__(func, types, args, kwargs)


    def __deepcopy__(self, memo):
        # (torch's default deepcopy of a subclass instance runs new_empty() with subclass dispatch off and then rejects the result)
        return self.as_subclass(torch.Tensor).__deepcopy__(memo).as_subclass(LossValue)


def _as_loss_value(t):
    """The freshly produced loss scalar as a LossValue — by class assignment (the object is ours alone; torch itself does this in
    nn.parameter.UninitializedParameter.materialize): `as_subclass` would put an AliasBackward node behind every loss."""
    if _UNIT_ROOT and type(t) is torch.Tensor:
        if (t.device.index if t.is_cuda else 'cpu') not in _UNIT_GRAD:
            # the constant is made HERE, eagerly, at the first reduced call on a device — never inside `.backward()`, which may run
            # under a hipGraph capture (its allocation would then belong to that graph's private pool); a first call that is
            # itself being captured returns a plain tensor (torch's own ones-fill is captured instead)
            if t.is_cuda and _capturing():
                return t
            unit_grad(t.device)
        t.__class__ = LossValue
    return t


def _is_unit_grad(g):
    t = _UNIT_GRAD.get(g.device.index if g.is_cuda else 'cpu')
    return t is not None and g.data_ptr() == t.data_ptr() and g.dim() == 0 and g.dtype == torch.float32


def _weight_ptrs(row_weight):
    if row_weight is None:
        return None, None
    if row_weight.dim() == 2:   # (N,7): the kernel takes the row mean itself
        return None, row_weight.data_ptr()
    return row_weight.data_ptr(), None


def per_pair_call(params, pred, target, row_weight, scale, want_loss, want_gp, want_gt, prologue=None):
    """One launch of the fused kernel without the reduce stage, on the current stream of pred's device (CPU tensors: the
    `_cpu` twin, torch's intra-op thread count as the team size).  `prologue`: None or a _lib.Prologue (bbox-coder decode
    fused into the kernel, head_loss.py).  Returns (loss|None, grad_pred|None, grad_target|None)."""
    lib = _library()
    n = pred.shape[0]
    w1, w7 = _weight_ptrs(row_weight)
    if not pred.is_cuda:
        if prologue is not None:
            raise RuntimeError('GDLoss: the head-level fusions (bbox-coder prologue) are GPU-only')
        loss = torch.empty(n, dtype=torch.float32) if want_loss else None
        gp = torch.empty_like(pred) if want_gp else None
        gt = torch.empty_like(target) if want_gt else None
        rc = lib.gd3d_loss_fused_cpu(params, pred.data_ptr(), target.data_ptr(), w1, w7, n, scale, _ptr(loss), None, _ptr(gp),
                                     _ptr(gt), None, torch.get_num_threads())
        if rc != 0:
            _lib.check(rc, 'gd3d_loss_fused_cpu')
        return loss, gp, gt
    dev = pred.device
    prev = _get_device()
    switch = prev != dev.index
    if switch:
        _set_device(dev.index)
    try:
        loss = torch.empty(n, dtype=torch.float32, device=dev) if want_loss else None
        gp = torch.empty_like(pred) if want_gp else None
        gt = torch.empty_like(target) if want_gt else None
        rc = lib.gd3d_loss_fused_decoded(params, prologue, pred.data_ptr(), target.data_ptr(), w1, w7, n, scale, _ptr(loss), None,
                                         _ptr(gp), _ptr(gt), None, _raw_stream(dev.index))
    finally:
        if switch:
            _set_device(prev)
    if rc != 0:
        _lib.check(rc, 'gd3d_loss_fused')
    return loss, gp, gt


def reduced_call(params, pred, target, row_weight, scale, prologue=None, select=False, want_flag=False):
    """scale * sum_i w_i L_i as a 0-dim tensor whose autograd node (_pynode.GDLossReduced, or its C++ twin in
    csrc/torch_node.cpp) holds the final gradients the SAME launch produced.  With `select` the value and the gradient are those of the reference's early-out
    `(pred * weight).sum()` when no weight entry is > 0; `want_flag` also returns the int32 (1,) any-positive flag.
    Backward: the gradients are handed over, scaled by the upstream gradient on the device (gd3d_grad_finish) unless that is
    `unit_grad`; a second backward under retain_graph recomputes them; differentiating them again raises."""
    node = _node()
    n = pred.shape[0]
    ticket = ev0 = ev1 = 0
    if pred.is_cuda:
        ev = PROFILE_EVENTS
        if ev is not None:  # profiling: the same single call, with a HIP event pair bound to the fused kernel's own dispatch
            tm = DispatchTimer()
            ev.append(tm)
            ev0, ev1 = tm.start, tm.stop
        elif n <= _one_launch_max():
            # training-size call: ONE launch, the last workgroup finishes the sum (per-stream arrival ticket, zeroed once)
            idx = pred.device.index
            ticket = _ticket(idx, _raw_stream(idx)) or 0
    aux = None if prologue is None else getattr(prologue, "_keepalive", None)
    trusted = getattr(node, 'reduced_trusted', None)
    if trusted is not None:   # the Python glue: operands were validated by GDLoss.forward, the structs are never rewritten
        return trusted(pred, target, row_weight, params, prologue, aux, scale, select, ticket, ev0, ev1, _ws_floats(n), want_flag)
    return node.reduced(pred, target, row_weight, ctypes.addressof(params), 0 if prologue is None else ctypes.addressof(prologue),
                        aux, scale, select, ticket, ev0, ev1, _ws_floats(n), want_flag)


class _GDPerPair(torch.autograd.Function):
    """(scale * w_i * L_i)_i ; backward re-runs the fused kernel with the upstream row gradient folded in."""

    @staticmethod
    def forward(ctx, pred, target, row_weight, params, scale, prologue=None):
        loss = per_pair_call(params, pred, target, row_weight, scale, True, False, False, prologue)[0]
        ctx.replay = (pred, target, row_weight, params, scale, prologue)
        return loss

    @staticmethod
    @guard_double_backward
    def backward(ctx, grad_out):
        pred, target, row_weight, params, scale, prologue = ctx.replay
        rw = grad_out.reshape(-1).to(torch.float32)
        if row_weight is not None:
            rw = rw * (row_weight.mean(dim=-1) if row_weight.dim() == 2 else row_weight)
        rw = rw.contiguous()
        _, gp, gt = per_pair_call(params, pred, target, rw, scale, False, ctx.needs_input_grad[0], ctx.needs_input_grad[1],
                                  prologue)
        return gp, gt, None, None, None, None


@LOSSES.register_module()
class GDLoss(nn.Module):
    """Gaussian-distance box regression loss (GWD / KLD / BCD / JD / sym-KLD / KFIoU), reference
    signature (gaussian_distance_loss.py:261-264, :280-286).

    Precision: inputs are EVALUATED IN FP32 regardless of their dtype (cast in, result cast back to pred's dtype); the
    reference follows the input dtype (gaussian_distance_loss.py:8-21), so an fp64 caller of it gets fp64 arithmetic and here
    an fp64 label with fp32 accuracy.  The heads call the loss under @force_fp32."""

    BAG_GD_LOSS = tuple(LOSS_TYPES)

    def __init__(self, loss_type, center_offset=(0, 0, 0.5), fun='log1p', tau=1.0, alpha=1.0, reduction='mean',
                 loss_weight=1.0, **kwargs):
        super().__init__()
        assert reduction in ['none', 'sum', 'mean']
        assert loss_type in self.BAG_GD_LOSS
        if loss_type not in ['kfiou3d']:
            assert fun in ['log1p', 'none']
        else:
            assert fun in ['nlog', 'expm1', 'none']
        self.loss_type = loss_type
        self.center_offset = center_offset
        self.fun = fun
        self.tau = tau
        self.alpha = alpha
        self.reduction = reduction
        self.loss_weight = loss_weight
        self.kwargs = kwargs
        self._params_cache = None  # (key, gd3d_params) for calls without per-call kwargs

    def _params(self, call_kwargs):
        # kfiou3d ignores tau (ref :247) — the kernel is told tau = 0
        tau = 0.0 if self.loss_type == 'kfiou3d' else self.tau
        if call_kwargs:
            _kwargs = deepcopy(self.kwargs)
            _kwargs.update(call_kwargs)  # ref :293-294
            return make_params(self.loss_type, self.fun, tau, self.alpha, self.center_offset, _kwargs)
        # the attributes are plain and may be reassigned by the user: cheap identity / value key, rebuilt when it changes
        co = self.center_offset
        # (values, not identities: `loss.kwargs['sqrt'] = False` or an in-place write to a center_offset tensor must be
        #  seen, as the reference re-reads both on every call, ref :293-299)
        key = (self.loss_type, self.fun, tau, self.alpha,
               (id(co), co._version) if isinstance(co, torch.Tensor) else tuple(co), tuple(self.kwargs.items()))
        cache = self._params_cache
        if cache is None or cache[0] != key:
            cache = self._params_cache = (key, make_params(self.loss_type, self.fun, tau, self.alpha, self.center_offset,
                                                           self.kwargs))
        return cache[1]

    def forward(self, pred, target, weight=None, avg_factor=None, reduction_override=None, **kwargs):
        # `_prologue` is internal (head_loss.py): a bbox-coder decode fused into the kernel; `pred` (and `target`
        # for the anchor coder) are then the ENCODED rows and the returned gradient is wrt them.
        prologue = kwargs.pop('_prologue', None)
        assert reduction_override in (None, 'none', 'mean', 'sum')
        reduction = reduction_override if reduction_override else self.reduction
        # ref :290-292: `if weight is not None and not torch.any(weight > 0) and reduction != 'none':
        #                    return (pred * weight).sum()`
        # For the (N,7) weights the heads pass, that test, both results and both gradients are evaluated by the fused
        # launch itself and selected on the device (`select`): no extra pass over the weights, no host sync.  Any other
        # weight shape keeps the reference's host-side test: its early-out expression only broadcasts (or raises, as
        # there) for particular shapes, which cannot be decided from the device.
        select = False
        if weight is not None and reduction != 'none':
            if weight.shape == pred.shape and weight.is_cuda and pred.is_cuda and not _HOST_WEIGHT_CHECK:
                select = True
            elif not torch.any(weight > 0):
                return (pred * weight).sum()
        params = self._params(kwargs)
        # ref :295-296 `weight.mean(dim=-1)` for an (N,7) weight: done inside the kernel
        weight7 = weight is not None and weight.shape == pred.shape

        out_dtype = pred.dtype
        p = _rows(pred)
        t = _rows(target)
        if p.shape != t.shape:
            raise RuntimeError(f'pred {tuple(pred.shape)} and target {tuple(target.shape)} disagree')
        if p.device != t.device:
            raise RuntimeError(f'pred is on {p.device} and target on {t.device}')
        if not p.is_cuda and p.device.type != 'cpu':
            raise RuntimeError(f'GDLoss: no implementation for device {p.device}')
        if prologue is not None and not p.is_cuda:
            raise RuntimeError('GDLoss: the head-level fusions (bbox-coder prologue) are GPU-only')
        n = p.shape[0]
        w = None
        if weight is not None:
            w = (weight if weight.dim() == 2 else weight.reshape(-1, 7)) if weight7 else weight.reshape(-1)
            if w.dtype != torch.float32 or w.device != p.device or not w.is_contiguous():
                w = w.to(device=p.device, dtype=torch.float32).contiguous()
            if w.shape[0] != n:
                raise RuntimeError(f'weight has {w.shape[0]} rows for {n} boxes')

        # mmdet weight_reduce_loss folded into one scalar for the kernel
        post_div = None
        if reduction == 'none':
            scale = self.loss_weight
        elif avg_factor is None:
            scale = self.loss_weight / n if (reduction == 'mean' and n > 0) else self.loss_weight
        elif reduction == 'mean':
            if isinstance(avg_factor, torch.Tensor):
                scale, post_div = self.loss_weight, avg_factor  # no host sync: divide on the device
            else:
                scale = self.loss_weight / avg_factor
        else:
            raise ValueError('avg_factor can not be used with reduction="sum"')

        if reduction == 'none':
            out = _GDPerPair.apply(p, t, w, params, float(scale), prologue)
        else:
            want_flag = select and post_div is not None
            out, flag = reduced_call(params, p, t, w, float(scale), prologue, select, want_flag)
            if post_div is not None:
                if select:  # the early-out value is not divided by avg_factor (it returns before the loss is called)
                    post_div = torch.where(flag.reshape(()) != 0, post_div.to(torch.float32), 1.0)
                out = out / post_div
            if n == 0 and reduction == 'mean' and avg_factor is None and not select:
                out = out + float('nan')  # torch: mean of an empty tensor is nan
            out = _as_loss_value(out)      # `.backward()` on it (or on sums of it) starts from the unit constant: see LossValue
        return out if out_dtype == torch.float32 else out.to(out_dtype)

    def extra_repr(self):
        return (f'loss_type={self.loss_type!r}, fun={self.fun!r}, tau={self.tau}, alpha={self.alpha}, '
                f'reduction={self.reduction!r}, loss_weight={self.loss_weight}')


register_with_mmdet(GDLoss)
