#!/usr/bin/env python3
"""Head-level train-step slice latency: decode(anchors, pred) + decode(anchors, target)
+ GDLoss(decode_weight (P,7), avg_factor) forward + backward.
  fused   : amd.anchor_decoded_gd_loss (decode inside the kernel, 1 launch fwd)
  unfused : torch coder mirror + amd.GDLoss (plain fused loss, decode as torch ops + autograd)
  eager   : torch coder mirror + oracle/gd_torch.py (reference-style eager op chain)"""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
import mmdet3d_gaussian_amd as amd
from oracle import gd_torch
dev = torch.device('cuda:0')
coder = amd.DeltaXYZWLHRBBoxCoder()
def timeit(fn, iters):
    """best of 3 timed loops (first-use code-object loads and allocator growth stay out of the number)"""
    for _ in range(10): fn()
    best = float('inf')
    for _ in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(iters): fn()
        torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t0) / iters * 1e6)
    return best
for lt in ('kld3d', 'gwd3d', 'bd3d'):
    mod = amd.GDLoss(lt, fun='log1p', tau=0.0, loss_weight=5.0)
    for P in (64, 512, 4096, 65536):
        g = torch.Generator(device=dev).manual_seed(P)
        an = torch.rand(P, 7, generator=g, device=dev) * torch.tensor([70, 80, 1, 1.5, 3, .5, 1.5], device=dev) + torch.tensor([0, -40, -2, .6, .9, 1.4, 0], device=dev)
        te = torch.randn(P, 7, generator=g, device=dev) * 0.3
        pe = (te + torch.randn(P, 7, generator=g, device=dev) * 0.1).requires_grad_(True)
        w = torch.ones(P, 7, device=dev)
        def fused():
            pe.grad = None; amd.anchor_decoded_gd_loss(mod, an, pe, te, w, avg_factor=P).backward()
        def unfused():
            pe.grad = None; mod(coder.decode(an, pe), coder.decode(an, te), w, avg_factor=P).backward()
        def eager():
            pe.grad = None
            gd_torch.gd_loss(coder.decode(an, pe), coder.decode(an, te), lt, weight=w, avg_factor=P, loss_weight=5.0, fun='log1p', tau=0.0).backward()
        # the three forms are the same loss: fused decode vs torch decode + plain fused loss (1e-6), vs the eager op chain (1e-4)
        lf = amd.anchor_decoded_gd_loss(mod, an, pe, te, w, avg_factor=P).item()
        lu = mod(coder.decode(an, pe), coder.decode(an, te), w, avg_factor=P).item()
        le = gd_torch.gd_loss(coder.decode(an, pe), coder.decode(an, te), lt, weight=w, avg_factor=P, loss_weight=5.0, fun='log1p', tau=0.0).item()
        assert abs(lf - lu) <= 1e-5 * (1 + abs(lu)) and abs(lf - le) <= 1e-4 * (1 + abs(le)), (lt, P, lf, lu, le)
        a, b, c = timeit(fused, 200), timeit(unfused, 100), timeit(eager, 50)
        print(json.dumps(dict(loss=lt, P=P, fused_us=round(a, 1), unfused_us=round(b, 1), eager_torch_us=round(c, 1),
                              speedup_vs_eager=round(c / a, 1))), flush=True)

# gather-fused variant from raw NCHW head output (KITTI geometry: 248 x 216 x 6 anchors), vs the torch-gather path
B, A, H, W, C = 2, 6, 248, 216, 3
n_per = H * W * A
anchors = torch.rand(n_per, 7, device=dev) * torch.tensor([70, 80, 1, 1.5, 3, .5, 1.5], device=dev) + torch.tensor([0, -40, -2, .6, .9, 1.4, 0], device=dev)
bbox_pred = (torch.randn(B, A * 7, H, W, device=dev) * 0.1).requires_grad_(True)
bbox_targets = torch.randn(B, n_per, 7, device=dev) * 0.2
bbox_weights = torch.ones(B, n_per, 7, device=dev)
for npos in (200, 2000):
    labels = torch.full((B, n_per), C, device=dev, dtype=torch.long)
    idx = torch.randperm(B * n_per, device=dev)[:npos]
    labels.view(-1)[idx] = 0
    mod = amd.GDLoss('kld3d', fun='log1p', tau=0.0, loss_weight=5.0)
    dw = [1.0] * 7
    def gather_torch():
        bbox_pred.grad = None
        amd.anchor_head_decoded_loss(mod, bbox_pred, bbox_targets, bbox_weights, labels, anchors, C, float(npos), dw).backward()
    def gather_fused():
        bbox_pred.grad = None
        amd.anchor_head_decoded_loss_fused(mod, bbox_pred, bbox_targets, bbox_weights, labels, anchors, C, float(npos), dw).backward()
    a, b = timeit(gather_fused, 100), timeit(gather_torch, 50)
    print(json.dumps(dict(slice='loss_single decoded branch from NCHW', B=B, anchors_per_sample=n_per, positives=npos,
                          gather_fused_us=round(a, 1), torch_gather_us=round(b, 1), speedup=round(b / a, 2))), flush=True)

# the WHOLE regression loss of loss_single (:95-161): GD on decoded boxes + SmoothL1 on encoded boxes (code_weight,
# add_sin_difference) in one launch, vs the eager torch restatement of the same lines (oracle/head_torch.py)
from oracle import head_torch  # noqa: E402
sl1 = dict(type='SmoothL1Loss', beta=1.0 / 9.0, loss_weight=2.0)
cw = [1., 1., 1., 0., 0., 0., 0.]
for npos in (200, 2000):
    labels = torch.full((B, n_per), C, device=dev, dtype=torch.long)
    labels.view(-1)[torch.randperm(B * n_per, device=dev)[:npos]] = 0
    mod = amd.GDLoss('kfiou3d', fun='nlog', loss_weight=5.0)     # kitti kfiou5 config: the one with live code weights
    def full_fused():
        bbox_pred.grad = None
        amd.anchor_head_bbox_loss(mod, sl1, bbox_pred, bbox_targets, bbox_weights, labels, anchors, C, float(npos),
                                  code_weight=cw, decode_weight=1).backward()
    def full_eager():
        bbox_pred.grad = None
        head_torch.loss_single_bbox(bbox_pred, bbox_targets, bbox_weights, labels, anchors, C, float(npos),
                                    gd=dict(loss_type='kfiou3d', fun='nlog', loss_weight=5.0),
                                    sl1=dict(beta=1.0 / 9.0, loss_weight=2.0), code_weight=cw, decode_weight=1).backward()
    a, b = timeit(full_fused, 100), timeit(full_eager, 30)
    print(json.dumps(dict(slice='loss_single regression loss (GD + SmoothL1) from NCHW', B=B, anchors_per_sample=n_per,
                          positives=npos, fused_us=round(a, 1), eager_torch_us=round(b, 1), speedup=round(b / a, 1))), flush=True)
