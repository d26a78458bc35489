#!/usr/bin/env python3
"""Host-side cost of GDLoss at training sizes (SURVEY.md §8d config 2 stand-in: KLD tau=0 log1p loss_weight=5, P positives,
weight (P,7) = 1, avg_factor = P).  Per P:
  isolated_us   : one forward + its own backward() per step (includes torch's autograd-engine thread hand-off, which any
                  op chain pays once per backward() call);
  amortised_us  : 8 GDLoss calls summed, ONE backward() — the shape of a real training step, where the loss is one of
                  many nodes of a single backward pass; per-call cost = step / 8;
  forward_us    : forward only;
  eager_torch_us: the reference-style eager PyTorch op chain on the same GPU (oracle/gd_torch.py), isolated.
Run twice for the weight-path A/B: GD3D_HOST_WEIGHT_CHECK=1 decides the early-out on the host as the reference does."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import math, torch
import mmdet3d_gaussian_amd as amd
from mmdet3d_gaussian_amd import gd_loss as gdl
from oracle import gd_torch
dev = torch.device('cuda:0')
def pairs(n, seed=0):
    g = torch.Generator(device=dev).manual_seed(seed)
    lo = torch.tensor([0, -40, -3, 0.5, 0.5, 0.5, -math.pi], device=dev); hi = torch.tensor([70, 40, 1, 2.5, 4.5, 2.0, math.pi], device=dev)
    t = torch.rand(n, 7, generator=g, device=dev) * (hi - lo) + lo
    p = t + torch.randn(n, 7, generator=g, device=dev) * torch.tensor([0.3, 0.3, 0.1, 0.1, 0.1, 0.1, 0.1], device=dev)
    return p.contiguous().requires_grad_(True), t.contiguous()
def timeit(fn, iters):
    for _ in range(10): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(iters): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / iters * 1e6
class _Floor(torch.autograd.Function):
    """What ANY Python autograd.Function costs on this host: forward hands back a preallocated scalar, backward a
    preallocated gradient; no kernel is launched, nothing is allocated."""
    @staticmethod
    def forward(ctx, x, out, g):
        ctx.g = g
        return out.detach()
    @staticmethod
    def backward(ctx, go):
        return ctx.g, None, None
def floor_rows():
    sets = [pairs(4096, s) for s in range(8)]
    outs = [torch.zeros((), device=dev) for _ in sets]; gs = [torch.zeros(4096, 7, device=dev) for _ in sets]
    p, _ = sets[0]
    def iso():
        p.grad = None; _Floor.apply(p, outs[0], gs[0]).backward()
    def fwd():
        _Floor.apply(p, outs[0], gs[0])
    def amort():
        tot = 0
        for (pp, _), o, g in zip(sets, outs, gs):
            pp.grad = None
            tot = tot + _Floor.apply(pp, o, g)
        tot.backward()
    return dict(mode='FLOOR: empty Python autograd.Function (no launch, no allocation)', P=4096, isolated_us=round(timeit(iso, 300), 1),
                forward_us=round(timeit(fwd, 300), 1), amortised_us_per_call=round(timeit(amort, 75) / 8, 1))
# warm the host (clock ramp, allocator, autograd engine thread) for two seconds before anything is timed
_t0 = time.perf_counter()
_wp, _wt = pairs(4096)
_wm = amd.GDLoss('kld3d', fun='log1p', tau=0.0, loss_weight=5.0)
while time.perf_counter() - _t0 < 2.0:
    _wp.grad = None; _wm(_wp, _wt).backward()
torch.cuda.synchronize()
print(json.dumps(floor_rows()), flush=True)
mode = 'host torch.any check (reference control flow)' if gdl._HOST_WEIGHT_CHECK else 'early-out resolved in the fused launch'
lt = 'kld3d'
mod = amd.GDLoss(lt, fun='log1p', tau=0.0, loss_weight=5.0)
for weighted, sizes in ((True, (64, 512, 4096, 100_000, 1_000_000, 10_000_000)), (False, (64, 512, 4096, 100_000, 1_000_000, 10_000_000)),
                        (True, (4096,)), (False, (4096,))):     # the last two repeat one size in the other order
    for P in sizes:
        sets = [pairs(P, s) for s in range(8 if P <= 100_000 else 1)]
        w = torch.ones(P, 7, device=dev) if weighted else None
        p, t = sets[0]
        def iso():
            p.grad = None; mod(p, t, w, avg_factor=P).backward()
        def fwd():
            mod(p, t, w, avg_factor=P)
        def amort():
            tot = 0
            for pp, tt in sets:
                pp.grad = None
                tot = tot + mod(pp, tt, w, avg_factor=P)
            tot.backward()
        def ref():
            p.grad = None; gd_torch.gd_loss(p, t, lt, weight=w, avg_factor=P, loss_weight=5.0, fun='log1p', tau=0.0).backward()
        iters = 300 if P <= 100_000 else 20
        # the timed call and the eager op chain are the same loss
        v_ours, v_ref = mod(p, t, w, avg_factor=P).item(), gd_torch.gd_loss(p, t, lt, weight=w, avg_factor=P, loss_weight=5.0, fun='log1p', tau=0.0).item()
        assert abs(v_ours - v_ref) <= 1e-4 * (1 + abs(v_ref)), (P, weighted, v_ours, v_ref)
        row = dict(mode=mode, weight='(P,7)' if weighted else None, P=P, isolated_us=round(timeit(iso, iters), 1),
                   forward_us=round(timeit(fwd, iters), 1))
        if len(sets) == 8:
            row['amortised_us_per_call'] = round(timeit(amort, max(iters // 4, 10)) / 8, 1)
        row['eager_torch_us'] = round(timeit(ref, max(5, iters // 10)), 1)
        print(json.dumps(row), flush=True)
