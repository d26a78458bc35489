"""A short optimisation run with GDLoss as the objective, against the same run with the REFERENCE module (fixture
tests/golden/train_traj.npz, written by tests/golden/make_golden_train_traj.py from the real gaussian_distance_loss.py): 256
boxes fitted to 256 targets by SGD with momentum, 150 steps.  Values and gradients compound here: a drop-in has to follow the
reference's fp64 curve at least as closely as the reference's own fp32 run does (the reference's only evidence for
its loss arithmetic is at the model level).  CPU tensors take the `_cpu` twin, GPU tensors the HIP kernel."""
import ast
import os

import numpy as np
import pytest
import torch

import mmdet3d_gaussian_amd as amd

G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'train_traj.npz'))
CASES = [str(c) for c in G['cases']]


def _run(name, device):
    start = torch.from_numpy(G[f'{name}.start']).to(device)
    tgt = torch.from_numpy(G[f'{name}.target']).to(device)
    w = torch.from_numpy(G[f'{name}.weight']).to(device) if f'{name}.weight' in G.files else None
    avg = float(G[f'{name}.avg_factor']) if w is not None else None
    mod = amd.GDLoss(str(G[f'{name}.loss_type']), **ast.literal_eval(str(G[f'{name}.kwargs'])))
    lr, mom = float(G['lr']), float(G['mom'])
    p = start.clone().requires_grad_(True)
    v = torch.zeros_like(p)
    curve = []
    for _ in range(int(G['steps'])):
        loss = mod(p, tgt, w, avg_factor=avg) if w is not None else mod(p, tgt)
        (g,) = torch.autograd.grad(loss, p)
        curve.append(loss.detach())
        with torch.no_grad():
            v.mul_(mom).add_(g)
            p.add_(v, alpha=-lr)
    return torch.stack(curve).double().cpu().numpy(), p.detach().double().cpu().numpy()


def _check(name, device):
    curve, final = _run(name, device)
    c32, c64 = G[f'{name}.curve32'], G[f'{name}.curve64']
    f32, f64 = G[f'{name}.final32'].astype(np.float64), G[f'{name}.final64']
    ref_curve = np.abs(c32 - c64).max()                 # how far the reference's own fp32 run strays from its fp64 run
    ref_final = np.abs(f32 - f64).max()
    our_curve, our_final = np.abs(curve - c64).max(), np.abs(final - f64).max()
    # the run makes real progress (this is not a comparison of two flat lines) ...
    assert c64[-1] < 0.75 * c64[0]
    # ... and this package's fp32 path stays within the reference's own fp32-vs-fp64 envelope (x 1.5 for two different fp32
    # roundings of a trajectory that amplifies them; 2e-6 / 2e-5 floors where the reference's fp32 happens to land on fp64)
    assert our_curve <= 1.5 * ref_curve + 2e-6, (name, device, our_curve, ref_curve)
    assert our_final <= 1.5 * ref_final + 2e-5, (name, device, our_final, ref_final)
    return our_curve, ref_curve, our_final, ref_final


@pytest.mark.parametrize('name', CASES)
def test_cpu_twin_follows_the_reference_run(name):
    _check(name, 'cpu')


@pytest.mark.gpu
@pytest.mark.parametrize('name', CASES)
def test_hip_path_follows_the_reference_run(name):
    _check(name, 'cuda')


if __name__ == '__main__':
    for n in CASES:
        print(n, ['%.3e' % x for x in _check(n, 'cpu')])
