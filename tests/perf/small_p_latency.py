#!/usr/bin/env python3
"""SURVEY.md §8d config 2 stand-in: loss-level train-step slice of the PointPillars/KITTI KLD config (tau=0, log1p,
loss_weight=5): P positives, weight (P,7)=1, avg_factor=P; forward + backward per step.
Compares the fused HIP path (GDLoss) with the reference-style eager PyTorch op chain on the same MI355X
(oracle/gd_torch.py — written in entry form, fewer kernels than the reference's own bmm chain, so conservative)."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import math, torch
import mmdet3d_gaussian_amd as amd
from oracle import gd_torch
dev = torch.device('cuda:0')
def pairs(n):
    g = torch.Generator(device=dev).manual_seed(0)
    lo = torch.tensor([0, -40, -3, 0.5, 0.5, 0.5, -math.pi], device=dev); hi = torch.tensor([70, 40, 1, 2.5, 4.5, 2.0, math.pi], device=dev)
    t = torch.rand(n, 7, generator=g, device=dev) * (hi - lo) + lo
    p = t + torch.randn(n, 7, generator=g, device=dev) * torch.tensor([0.3, 0.3, 0.1, 0.1, 0.1, 0.1, 0.1], device=dev)
    return p.contiguous().requires_grad_(True), t.contiguous()
def timeit(fn, iters):
    for _ in range(10): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(iters): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / iters * 1e6
rows = []
for lt in ('kld3d', 'gwd3d', 'bd3d'):
    mod = amd.GDLoss(lt, fun='log1p', tau=0.0, loss_weight=5.0)
    for P in (64, 512, 4096, 100_000, 1_000_000, 10_000_000):
        p, t = pairs(P); w = torch.ones(P, 7, device=dev)
        def ours():
            p.grad = None; mod(p, t, w, avg_factor=P).backward()
        def ref():
            p.grad = None; gd_torch.gd_loss(p, t, lt, weight=w, avg_factor=P, loss_weight=5.0, fun='log1p', tau=0.0).backward()
        iters = 200 if P <= 100_000 else 20
        a = timeit(ours, iters); b = timeit(ref, max(5, iters // 4))
        l1 = mod(p, t, w, avg_factor=P).item(); l2 = gd_torch.gd_loss(p, t, lt, weight=w, avg_factor=P, loss_weight=5.0, fun='log1p', tau=0.0).item()
        rows.append(dict(loss=lt, P=P, fused_us=round(a, 1), eager_torch_us=round(b, 1), speedup=round(b / a, 1), loss_fused=l1, loss_eager=l2))
        print(json.dumps(rows[-1]), flush=True)
