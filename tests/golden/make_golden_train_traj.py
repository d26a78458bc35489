#!/usr/bin/env python3
"""A short optimisation run with THE REFERENCE's GDLoss as the objective (build container only).

    python3 -B tests/golden/make_golden_train_traj.py

BASELINE.md: the reference's only evidence that its loss arithmetic is right is at the model level (KITTI AP after training).
The nearest thing that fits in a fixture: 256 boxes are fitted to 256 targets by plain SGD with momentum, the objective being the
reference module itself (gaussian_distance_loss.py:251-310, loaded through tests/golden/_ref_loader.py), once in fp32 (what a
training run does) and once in fp64 (the arbiter).  Stored per configuration: start boxes, targets, the loss after every step and
the boxes after the last one.  tests/test_train_traj.py runs the same loop with this package's GDLoss (CPU twin under
-m "not gpu", HIP path under -m gpu): a drop-in must follow the fp64 curve at least as closely as the reference's own fp32 does,
values AND gradients compounded over 150 steps.  Only data is written (tests/golden/train_traj.npz).
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from _ref_loader import load_reference_loss  # noqa: E402

STEPS, LR, MOM = 150, 0.05, 0.9
#         name                 loss_type  ctor kwargs                                              weighted
CASES = [('gwd3d_log1p_tau1', 'gwd3d', dict(fun='log1p', tau=1.0, loss_weight=5.0), False),       # configs/kitti/*gwd5tau1*
         ('kld3d_log1p_tau1', 'kld3d', dict(fun='log1p', tau=1.0, loss_weight=5.0), False),       # configs/kitti/*kld5tau1*
         ('bd3d_log1p_tau1', 'bd3d', dict(fun='log1p', tau=1.0, loss_weight=5.0), True),          # configs/kitti/*bd5tau1*, (N,7) weights + avg_factor
         ('gwd3d_none_tau0', 'gwd3d', dict(fun='none', tau=0.0, loss_weight=5.0), False),         # README row "GWD, tau=0, f(x)=x"
         ('kfiou3d', 'kfiou3d', dict(fun='none', loss_weight=1.0), False),
         ('jd3d_log1p_tau1', 'jd3d', dict(fun='log1p', tau=1.0, loss_weight=5.0), False),
         ('kld3d_symmax_log1p', 'kld3d_symmax', dict(fun='log1p', tau=1.0, loss_weight=5.0), False),
         ('kld3d_symmin_none', 'kld3d_symmin', dict(fun='none', tau=0.0, loss_weight=1.0), False),
         ('kld3d_none_nosqrt', 'kld3d', dict(fun='none', tau=0.0, loss_weight=1.0, sqrt=False), False)]


def start_and_target(seed):
    g = torch.Generator().manual_seed(seed)
    lo = torch.tensor([0, -40, -3, 0.5, 0.5, 0.5, -3.14159])
    hi = torch.tensor([70, 40, 1, 2.5, 4.5, 2.0, 3.14159])
    tgt = torch.rand(256, 7, generator=g) * (hi - lo) + lo
    start = tgt + torch.randn(256, 7, generator=g) * torch.tensor([0.5, 0.5, 0.2, 0.2, 0.2, 0.2, 0.3])
    start[:, 3:6] = start[:, 3:6].clamp(min=0.3)
    return start.float(), tgt.float()


def run(module, start, tgt, weight, avg_factor, steps=STEPS, lr=LR, mom=MOM):
    """SGD with momentum written out (no optimiser object: the update is p -= lr * v, v = mom * v + g, in the tensors' dtype)."""
    p = start.clone().requires_grad_(True)
    v = torch.zeros_like(p)
    curve = []
    for _ in range(steps):
        loss = module(p, tgt, weight, avg_factor=avg_factor) if weight is not None else module(p, tgt)
        (g,) = torch.autograd.grad(loss, p)
        curve.append(float(loss.detach()))
        with torch.no_grad():
            v.mul_(mom).add_(g)
            p.add_(v, alpha=-lr)
    return np.array(curve, np.float64), p.detach()


def main():
    ref = load_reference_loss()
    out = {'cases': np.array([c[0] for c in CASES]), 'steps': np.int64(STEPS), 'lr': np.float64(LR), 'mom': np.float64(MOM)}
    for k, (name, lt, kw, weighted) in enumerate(CASES):
        start, tgt = start_and_target(100 + k)
        w = None
        avg = None
        if weighted:
            w = torch.ones(256, 7)
            w[::5] = 0.0                     # a fifth of the rows carry no weight (negatives that slipped into the slice)
            avg = 205.0
        mod = ref.GDLoss(lt, **kw)
        c32, p32 = run(mod, start, tgt, w, avg)
        c64, p64 = run(mod, start.double(), tgt.double(), None if w is None else w.double(), avg)
        out[f'{name}.start'], out[f'{name}.target'] = start.numpy(), tgt.numpy()
        if w is not None:
            out[f'{name}.weight'], out[f'{name}.avg_factor'] = w.numpy(), np.float64(avg)
        out[f'{name}.loss_type'] = np.array(lt)
        out[f'{name}.kwargs'] = np.array(repr(kw))
        out[f'{name}.curve32'], out[f'{name}.curve64'] = c32, c64
        out[f'{name}.final32'], out[f'{name}.final64'] = p32.numpy(), p64.numpy()
        print(f'{name}: loss {c64[0]:.6f} -> {c64[-1]:.6f}; |curve32 - curve64| max {np.abs(c32 - c64).max():.3e}; '
              f'|final32 - final64| max {np.abs(p32.double().numpy() - p64.numpy()).max():.3e}')
    path = os.path.join(HERE, 'train_traj.npz')
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path), 'bytes')


if __name__ == '__main__':
    main()
