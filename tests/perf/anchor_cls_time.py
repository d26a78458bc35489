#!/usr/bin/env python3
"""Anchor heads' classification + direction loss, forward + backward, us per call (synchronised) at the reference's PointPillars
training geometries:
  KITTI  248 x 216 cells x 6 anchors (3 classes x 2 rotations), 3 classes, batch 6
  Waymo  468 x 468 x 6 anchors, 3 classes, batch 2
ours  = anchor_head_cls_dir_loss (one pass over the NCHW maps + a one-workgroup finish; backward returns the stored gradients)
eager = the reference's op sequence (oracle/anchor_cls_torch.py's statement: permuted copies, mmdet FocalLoss / CrossEntropyLoss
        as torch ops with autograd) on device tensors
Then GDAnchor3DHead.loss_single as a whole (classification + regression + direction, forward + backward) at the KITTI geometry:
ours eager, ours as one hipGraph (GraphedStep), and the reference's op sequence (oracle/head_torch.py + oracle/anchor_cls_torch.py
statements) on device tensors.  Prints the kernel's algorithmic bytes next to the times."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch  # noqa: E402

import mmdet3d_gaussian_amd as amd  # noqa: E402
from oracle import anchor_cls_torch as ORA  # noqa: E402
from test_gpu_anchor_cls import CE, FOCAL, make  # noqa: E402

dev = torch.device('cuda:0')


def timeit(fn, it, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(it):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / it * 1e6


def main():
    for name, (B, A, C, H, W) in (('kitti', (6, 6, 3, 248, 216)), ('waymo', (2, 6, 3, 468, 468))):
        cls, dirs, labels, lw, dt, dw = [t.to(dev) for t in make(B, A, C, H, W, seed=1, pos_frac=0.002)]
        avg = float(max(int(((labels >= 0) & (labels < C)).sum()), 1))
        cls.requires_grad_(True)
        dirs.requires_grad_(True)

        def ours():
            cls.grad = dirs.grad = None
            a, b = amd.extras.anchor_head_cls_dir_loss(FOCAL, CE, cls, dirs, labels, lw, dt, dw, C, avg)
            (a + b).backward()
            return a, b

        def eager():
            cls.grad = dirs.grad = None
            a, b = ORA.cls_dir_losses(cls, dirs, labels, lw, dt, dw, C, avg)
            (a + b).backward()
            return a, b

        a, b = ours()
        gc, gd = cls.grad.clone(), dirs.grad.clone()
        ea, eb = eager()
        assert abs(a.item() - ea.item()) <= 1e-4 * abs(ea.item()) and abs(b.item() - eb.item()) <= 1e-4 * abs(eb.item()) + 1e-7
        assert (gc - cls.grad).abs().max().item() <= 1e-4 * cls.grad.abs().max().item()
        assert (gd - dirs.grad).abs().max().item() <= 1e-4 * dirs.grad.abs().max().item() + 1e-9
        t_ours, t_eager = timeit(ours, 50), timeit(eager, 10)
        with torch.no_grad():
            t_fwd = timeit(lambda: amd.extras.anchor_head_cls_dir_loss(FOCAL, CE, cls, dirs, labels, lw, dt, dw, C, avg), 50)
        n = B * H * W * A
        algo = n * C * 8 + n * 2 * 4 + n * 12           # class logits in + gradients out, direction gradients out, label + weight
        print(json.dumps(dict(geometry=name, batch=B, anchors_per_sample=H * W * A, classes=C, ours_fwd_bwd_us=round(t_ours, 1),
                              ours_fwd_only_us=round(t_fwd, 1), eager_fwd_bwd_us=round(t_eager, 1), speedup=round(t_eager / t_ours, 1),
                              algorithmic_mb=round(algo / 1e6, 2))), flush=True)


def whole():
    from oracle import head_torch
    from test_gpu_anchor_cls import SL1, TRAIN_CFG
    B, A, C, H, W = 6, 6, 3, 248, 216
    cls, dirs, labels, lw, dt, dw = [t.to(dev) for t in make(B, A, C, H, W, seed=2, pos_frac=0.002)]
    g = torch.Generator().manual_seed(3)
    n = H * W * A
    anchors = (torch.rand(n, 7, generator=g) * torch.tensor([70, 80, 1, 1.5, 3, 0.5, 1.5]) + torch.tensor([0, -40, -2, .6, .9, 1.4, 0])).to(dev)
    bbox = (torch.randn(B, A * 7, H, W, generator=g) * 0.15).to(dev)
    bt = (torch.randn(B, n, 7, generator=g) * 0.2).to(dev)
    bw = ((labels >= 0) & (labels < C)).float().unsqueeze(-1).expand(B, n, 7).contiguous()
    avg = float(max(int(((labels >= 0) & (labels < C)).sum()), 1))
    mod = amd.GDLoss('kld3d', fun='log1p', tau=1.0, loss_weight=5.0)
    for t in (cls, bbox, dirs):
        t.requires_grad_(True)

    def fn(cls, bbox, dirs, labels, lw, bt, bw, dt, dw, anchors):
        return amd.extras.gd_anchor_head_loss_single(FOCAL, SL1, CE, mod, TRAIN_CFG, C, cls, bbox, dirs, labels, lw, bt, bw, dt, dw, anchors, avg)
    args = (cls, bbox, dirs, labels, lw, bt, bw, dt, dw, anchors)

    def ours():
        cls.grad = bbox.grad = dirs.grad = None
        a, b, c = fn(*args)
        (a + b + c).backward()

    def eager():
        cls.grad = bbox.grad = dirs.grad = None
        a, c = ORA.cls_dir_losses(cls, dirs, labels, lw, dt, dw, C, avg)
        b = head_torch.loss_single_bbox(bbox, bt, bw, labels, anchors, C, avg, gd=dict(loss_type='kld3d', fun='log1p', tau=1.0, loss_weight=5.0),
                                        sl1=dict(beta=SL1['beta'], loss_weight=SL1['loss_weight']), code_weight=TRAIN_CFG['code_weight'],
                                        decode_weight=TRAIN_CFG['decode_weight'], diff_rad_by_sin=True)
        (a + b + c).backward()
    step = amd.GraphedStep(fn, args)
    t_ours, t_graph, t_eager = timeit(ours, 50), timeit(lambda: step(*args), 50), timeit(eager, 5)
    print(json.dumps(dict(method='GDAnchor3DHead.loss_single', geometry='kitti', batch=B, anchors_per_sample=n, positives=int(avg),
                          ours_eager_us=round(t_ours, 1), ours_graph_us=round(t_graph, 1), reference_ops_on_gpu_us=round(t_eager, 1),
                          speedup_eager=round(t_eager / t_ours, 1), speedup_graph=round(t_eager / t_graph, 1))), flush=True)


if __name__ == '__main__':
    main()
    whole()
