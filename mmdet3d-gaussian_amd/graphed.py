"""Launch-bound slices of a training step as ONE hipGraph: forward and backward captured once, replayed per step.

Every op of this package is stream-ordered, allocation-free inside the library and free of host syncs (the dense head forms
test labels in the kernel instead of `nonzero()`), so a slice with static shapes — the regression losses of an anchor head over
all anchors, the CenterPoint head losses of all tasks, the three losses of the benchmark step — can be captured with its
backward and replayed with new values in the same buffers: at KITTI geometry (6 x 321 408 anchors) 28 us per step instead of
90 us of eager Python (profiles/r03_config_standins.jsonl).  The reference has no counterpart (its step is eager PyTorch,
~600-850 ATen calls per loss); this is the MI355X-side answer to "the step is launch-bound, not bandwidth-bound".
"""
import torch

from .gd_loss import unit_grad


def _flatten(out):
    """-> (flat list of tensors, function that puts a list of the same length back into the structure of `out`)"""
    if isinstance(out, torch.Tensor):
        return [out], lambda xs: xs[0]
    if isinstance(out, dict):
        keys, items = list(out), [out[k] for k in out]
        wrap = lambda res: dict(zip(keys, res))          # noqa: E731
    elif isinstance(out, (list, tuple)):
        items, kind = list(out), type(out)
        wrap = lambda res: kind(res)                      # noqa: E731
    else:
        raise TypeError(f'GraphedStep: the step function returned {type(out).__name__}; expected tensors in tuples / lists / dicts')
    flat, rebuilds = [], []
    for o in items:
        f, r = _flatten(o)
        rebuilds.append((len(f), r))
        flat += f

    def rebuild(xs):
        res, i = [], 0
        for n, r in rebuilds:
            res.append(r(xs[i:i + n]))
            i += n
        return wrap(res)
    return flat, rebuild


class GraphedStep:
    """step = GraphedStep(fn, example_inputs);  losses, grads = step(*inputs)

    fn(*tensors) -> loss tensor(s) (a tensor, or a tuple / list / dict of them, nested tuples allowed); every tensor the
    step reads must come in through its arguments (anything closed over is frozen at its captured address).  The capture runs
    `fn` and `torch.autograd.backward` over all returned losses on private copies of the example inputs; a call copies the new
    inputs into those buffers (same shapes and dtypes), replays, and returns the losses in the structure `fn` returned them and
    the gradients of the inputs that require grad (None for the others), in argument order.  Both are views of the graph's own
    output buffers: valid until the next call.  Shapes are static: pad to the maximum (zero weights / labels outside the
    classes) where the count of positives varies, as the dense head forms of this package do."""

    def __init__(self, fn, example_inputs, warmup=3):
        if not example_inputs or not all(isinstance(x, torch.Tensor) and x.is_cuda for x in example_inputs):
            raise RuntimeError('GraphedStep: example_inputs must be GPU tensors')
        self._static = [x.detach().clone().requires_grad_(x.requires_grad) for x in example_inputs]
        self._fn = fn
        dev = self._static[0].device
        with torch.cuda.device(dev):
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(max(int(warmup), 1)):     # first calls build the library, set kernel attributes, fill caches
                    self._run()
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize(dev)
            self._graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self._graph):
                self._losses, self._rebuild = self._run()
            self._grads = [x.grad for x in self._static]

    def _run(self):
        for x in self._static:
            x.grad = None
        flat, rebuild = _flatten(self._fn(*self._static))
        need = [l for l in flat if l.requires_grad]
        if need:
            # 0-dim fp32 losses get the library's constant 1.0 (gd_loss.unit_grad): this package's backward functions recognise it by
            # address and launch nothing for it; torch's own nodes just read it
            torch.autograd.backward(need, [unit_grad(l.device) if (l.dim() == 0 and l.dtype == torch.float32) else torch.ones_like(l)
                                           for l in need])
        return [l.detach() for l in flat], rebuild

    def __call__(self, *inputs):
        if len(inputs) != len(self._static):
            raise RuntimeError(f'GraphedStep: {len(self._static)} inputs were captured, {len(inputs)} given')
        with torch.no_grad():
            for s, x in zip(self._static, inputs):
                if x.shape != s.shape or x.dtype != s.dtype:
                    raise RuntimeError(f'GraphedStep: input {tuple(x.shape)} {x.dtype} where {tuple(s.shape)} {s.dtype} was captured')
                if x.data_ptr() != s.data_ptr():
                    s.copy_(x)
        self._graph.replay()
        return self._rebuild(self._losses), list(self._grads)

    def static_inputs(self):
        """The graph's own input buffers: write new values into them directly to skip the copy in __call__."""
        return list(self._static)
