#!/usr/bin/env python3
"""BASELINE.json configs[0] as a golden fixture FROM THE REAL REFERENCE: "1k synthetic KITTI 7-dof box pairs, GWD loss
fwd only, PyTorch-CPU reference (plumbing, no GPU)", plus the reference's early-out for non-positive weights.

Run in the build container only (needs /root/reference):   python3 -B tests/golden/make_golden_config0.py

Writes tests/golden/gd_config0.npz (data only):
  pred, target (1000,7) fp32: torch.manual_seed(0); pred then target, each an independent uniform draw over the KITTI
      ranges of SURVEY.md §8d (x 0..70, y -40..40, z -3..1, w .5..2.5, h .5..4.5, l .5..2, yaw -pi..pi).  SURVEY.md §4
      quotes N = 1000 seed-0 means of the reference for an unstated generator (gwd 3.1693 / 0.7546, kld 3.4692 / 0.7703,
      bd 2.6529 / 0.7180 at tau 0 / 1); this generator reproduces them to 0.5 % (3.1728 / 0.7537, 3.4909 / 0.7703,
      2.6632 / 0.7173), which is what tests/test_oracle_gd.py asserts next to the exact fixture values.
  <loss>.tau<0|1>.loss32 / .loss64: per-pair forward values of GDLoss(loss, fun='log1p' ('expm1' for kfiou3d), tau,
      reduction='none') in fp32 and fp64 on the CPU.
  early.<kind>.out32/out64/gp32/gp64: GDLoss('kld3d', fun='log1p', tau=1, loss_weight=5)(pred, target, weight (N,7),
      avg_factor=37) and its gradient for weights without any positive entry (gaussian_distance_loss.py:290-292):
      kind = neg (all negative), zero, mixed (ordinary weights, the normal branch, for contrast).
"""
import math
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from _ref_loader import load_reference_loss  # noqa: E402


def inputs():
    lo = torch.tensor([0, -40, -3, 0.5, 0.5, 0.5, -math.pi])
    hi = torch.tensor([70, 40, 1, 2.5, 4.5, 2.0, math.pi])
    torch.manual_seed(0)
    pred = torch.rand(1000, 7) * (hi - lo) + lo
    target = torch.rand(1000, 7) * (hi - lo) + lo
    return pred.float().contiguous(), target.float().contiguous()


def early_weights(n):
    g = torch.Generator().manual_seed(1)
    return {'neg': -torch.rand(n, 7, generator=g) - 0.01, 'zero': torch.zeros(n, 7),
            'mixed': torch.rand(n, 7, generator=g) - 0.3}


def main():
    torch.set_num_threads(1)
    ref = load_reference_loss()
    pred, target = inputs()
    out = {'pred': pred.numpy(), 'target': target.numpy()}
    for lt in ('gwd3d', 'kld3d', 'bd3d', 'jd3d', 'kld3d_symmax', 'kld3d_symmin', 'kfiou3d'):
        fun = 'expm1' if lt == 'kfiou3d' else 'log1p'
        for tau in (0.0, 1.0):
            for dtype, tag in ((torch.float32, '32'), (torch.float64, '64')):
                mod = ref.GDLoss(lt, fun=fun, tau=tau, reduction='none')
                out[f'{lt}.tau{int(tau)}.loss{tag}'] = mod(pred.to(dtype), target.to(dtype)).numpy()
    for kind, w in early_weights(1000).items():
        out[f'early.{kind}.w'] = w.numpy()
        for dtype, tag in ((torch.float32, '32'), (torch.float64, '64')):
            p = pred.detach().to(dtype).clone().requires_grad_(True)
            res = ref.GDLoss('kld3d', fun='log1p', tau=1.0, loss_weight=5.0)(p, target.to(dtype), w.to(dtype), avg_factor=37.0)
            res.backward()
            out[f'early.{kind}.out{tag}'] = res.detach().numpy()
            out[f'early.{kind}.gp{tag}'] = p.grad.numpy()
    np.savez_compressed(os.path.join(HERE, 'gd_config0.npz'), **out)
    print('gd_config0.npz', os.path.getsize(os.path.join(HERE, 'gd_config0.npz')), 'bytes')
    for lt in ('gwd3d', 'kld3d', 'bd3d'):
        print(lt, [round(float(out[f'{lt}.tau{t}.loss32'].mean()), 4) for t in (0, 1)])
    print({k: float(out[f'early.{k}.out32']) for k in ('neg', 'zero', 'mixed')})


if __name__ == '__main__':
    main()
