#!/usr/bin/env python3
"""A/B of how the 3-loss benchmark step is ENQUEUED, all variants in one process on the same buffers (buffer placement
moves every number of a process by up to 5 %, DESIGN.md §5.3, so only same-process comparisons mean anything).

  sum_bwd     : (l0 + l1 + l2).backward()                                   -- round 2's step
  unit_bwd    : torch.autograd.backward([l0, l1, l2], grad_tensors = the library's unit_grad constant x 3)
  *_defer     : the reduce stage of every loss on a side stream (a `deferred_sums` context that existed in gd_loss.py for
                this measurement only: 25-60 us per step SLOWER eagerly and as a graph, removed; profiles/r03_step_variants.jsonl)
  eager / graph : launched per step / one hipGraph replay per step
  round 5, what the three early-exit grad_finish launches of the plain step cost, under the PYTHON glue so that
  the launch can be intercepted (MEASUREMENT ONLY: both hacks return wrong gradients whenever the upstream gradient is not 1):
  sum_bwd_py            : the plain step through the Python glue (the like-for-like base of the next two)
  sum_bwd_py_no_finish  : every 0-dim upstream gradient treated as the unit gradient -> no grad_finish launch at all: the UPPER
                          BOUND of what any coalescing of the three launches could save
  sum_bwd_py_one_finish : only the first node of each backward launches grad_finish: the launch count a coalesced form would have

Prints one JSON line per variant: us per step (wall, 60 steps after 10 warm-ups) and the loss values (must agree)."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
import mmdet3d_gaussian_amd as amd  # noqa: E402
from mmdet3d_gaussian_amd import gd_loss as gdl  # noqa: E402

LOSSES = bench.LOSSES


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
    dev = torch.device('cuda:0')
    amd.load_library()
    pred0, tgt = bench.synthetic_pairs(n, 0, dev)
    preds = {lt: pred0.clone().requires_grad_(True) for lt in LOSSES}
    del pred0
    mods = {lt: amd.build_loss(dict(type='GDLoss', loss_type=lt, fun='log1p', tau=1.0, alpha=1.0, reduction='mean',
                                    loss_weight=5.0)) for lt in LOSSES}
    unit = [gdl.unit_grad(dev)] * 3

    def forward(defer):
        for lt in LOSSES:
            preds[lt].grad = None
        if defer and hasattr(gdl, 'deferred_sums'):
            with gdl.deferred_sums():
                return [mods[lt](preds[lt], tgt) for lt in LOSSES]
        return [mods[lt](preds[lt], tgt) for lt in LOSSES]

    def step_sum(defer=False):
        ls = forward(defer)
        (ls[0] + ls[1] + ls[2]).backward()
        return [l.detach() for l in ls]

    def step_unit(defer=False):
        ls = forward(defer)
        torch.autograd.backward(ls, grad_tensors=unit)
        return [l.detach() for l in ls]

    variants = [('sum_bwd', lambda: step_sum()), ('unit_bwd', lambda: step_unit())]
    from mmdet3d_gaussian_amd import _lib, _pynode
    real_is_unit = _pynode._is_unit_grad

    def with_python_glue(fn, hack):
        def run():
            _lib.set_host_glue('python')
            calls = [0]

            def one(g):
                calls[0] += 1
                return real_is_unit(g) or calls[0] % 3 != 1
            _pynode._is_unit_grad = {'none': real_is_unit, 'no_finish': lambda g: True, 'one_finish': one}[hack]
            try:
                return fn()
            finally:
                _pynode._is_unit_grad = real_is_unit
                _lib.set_host_glue(None)
        return run
    variants += [('sum_bwd_py', with_python_glue(step_sum, 'none')), ('sum_bwd_py_no_finish', with_python_glue(step_sum, 'no_finish')),
                 ('sum_bwd_py_one_finish', with_python_glue(step_sum, 'one_finish')), ('unit_bwd_py', with_python_glue(step_unit, 'none'))]
    if hasattr(gdl, 'deferred_sums'):
        variants += [('sum_bwd_defer', lambda: step_sum(True)), ('unit_bwd_defer', lambda: step_unit(True))]

    def timed(fn, steps=60, warm=10):
        for _ in range(warm):
            outs = fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            outs = fn()
        host = time.perf_counter() - t0
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / steps * 1e6, host / steps * 1e6, outs

    # clock ramp
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 1.0:
        for _ in range(20):
            step_unit()
        torch.cuda.synchronize()

    for rep in range(2):
        for name, fn in variants:
            us, host, outs = timed(fn)
            print(json.dumps({'variant': name, 'launch': 'eager', 'rep': rep, 'us_per_step': round(us, 1),
                              'host_us': round(host, 1), 'losses': [round(o.item(), 6) for o in outs]}), flush=True)
        for name, fn in variants[:2]:
            try:
                side = torch.cuda.Stream()
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):
                    for _ in range(3):
                        fn()
                torch.cuda.current_stream().wait_stream(side)
                torch.cuda.synchronize()
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, capture_error_mode='thread_local'):
                    gouts = fn()
                us, host, _ = timed(lambda: (g.replay(), gouts)[1])
                print(json.dumps({'variant': name, 'launch': 'graph', 'rep': rep, 'us_per_step': round(us, 1),
                                  'host_us': round(host, 1), 'losses': [round(o.item(), 6) for o in gouts]}), flush=True)
                del g
            except Exception as e:  # noqa: BLE001
                print(json.dumps({'variant': name, 'launch': 'graph', 'error': repr(e)[:300]}), flush=True)
                torch.cuda.synchronize()

    # the three fused kernels alone (event pair bound to each dispatch), for the overhead the step adds on top of them
    evs = {lt: [] for lt in LOSSES}
    for _ in range(30):
        for lt in LOSSES:
            gdl.PROFILE_EVENTS = evs[lt]
            preds[lt].grad = None
            mods[lt](preds[lt], tgt)
        gdl.PROFILE_EVENTS = None
    torch.cuda.synchronize()
    k = {lt: sum(t.elapsed_ms() for t in evs[lt][5:]) / len(evs[lt][5:]) * 1e3 for lt in LOSSES}
    print(json.dumps({'fused_kernel_us': {a: round(b, 2) for a, b in k.items()}, 'sum_us': round(sum(k.values()), 1)}))


if __name__ == '__main__':
    main()
