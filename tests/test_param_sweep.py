"""Randomised hyper-parameter sweeps (seeded): alpha, tau, center_offset, flags, funs, weights.
CPU: hand-derived C oracle (fp64) vs the autograd-based torch restatement (two independent derivations).
GPU: the HIP kernel vs the fp64 oracle under the standard tolerance policy."""
import numpy as np
import pytest
import torch

import oracle
from gd_golden import check_close, oracle32_bounds
from oracle import gd_torch

LT = ['gwd3d', 'kld3d', 'bd3d', 'jd3d', 'kld3d_symmax', 'kld3d_symmin', 'kfiou3d']


def _case(seed):
    rng = np.random.default_rng(seed)
    lt = LT[seed % len(LT)]
    fun = rng.choice(['expm1', 'nlog', 'none']) if lt == 'kfiou3d' else rng.choice(['log1p', 'none'])
    kw = dict(fun=str(fun), tau=float(rng.choice([0.0, 0.5, 1.0, 1.75, 3.0])), alpha=float(np.float32(rng.uniform(0.3, 3.0))),   # fp32-representable: gd3d_params carries floats
              center_offset=tuple(float(x) for x in rng.uniform(-0.5, 0.5, 3)))
    if rng.random() < 0.5:
        kw['normalize' if lt == 'gwd3d' else 'sqrt'] = bool(rng.random() < 0.5)
    n = 96
    t = np.stack([rng.uniform(-50, 50, n), rng.uniform(-50, 50, n), rng.uniform(-3, 1, n), rng.uniform(.3, 6, n),
                  rng.uniform(.3, 12, n), rng.uniform(.3, 4, n), rng.uniform(-6, 6, n)], -1)
    p = t + rng.normal(0, 1, (n, 7)) * np.array([.4, .4, .2, .15, .15, .1, .2])
    p[:, 3:6] = np.abs(p[:, 3:6]) + 0.05
    w = rng.uniform(0, 2, n)
    return lt, kw, p.astype(np.float32), t.astype(np.float32), w.astype(np.float32)


@pytest.mark.parametrize('seed', range(28))
def test_c_oracle_vs_autograd_restatement(seed):
    lt, kw, p, t, w = _case(seed)
    prm = oracle.make_params(lt, **kw)
    r = oracle.gd_loss(p, t, prm, row_weight=w, scale=0.37)
    tp = torch.from_numpy(p).double().requires_grad_(True); tt = torch.from_numpy(t).double().requires_grad_(True)
    loss = gd_torch.pair_loss(tp, tt, lt, **kw) * torch.from_numpy(w).double() * 0.37
    loss.sum().backward()
    np.testing.assert_allclose(r['loss'], loss.detach().numpy(), rtol=1e-8, atol=1e-10)
    for ours, ref in ((r['grad_pred'], tp.grad.numpy()), (r['grad_target'], tt.grad.numpy())):
        scale = np.abs(ref).max(-1, keepdims=True)
        assert (np.abs(ours - ref) <= 1e-7 * (1 + scale)).all()


@pytest.mark.gpu
@pytest.mark.parametrize('seed', range(28))
def test_hip_kernel_param_sweep(seed):
    import mmdet3d_gaussian_amd as amd
    lt, kw, p, t, w = _case(seed)
    prm = oracle.make_params(lt, **kw)
    ref = oracle.gd_loss(p, t, prm, row_weight=w.astype(np.float64), scale=2.0)
    r32 = oracle.gd_loss(p, t, prm, row_weight=w, scale=2.0, dtype=np.float32)
    from gd_golden import grad_bound, loss_bound
    pp = torch.from_numpy(p).cuda().requires_grad_(True); tt = torch.from_numpy(t).cuda().requires_grad_(True)
    mod = amd.GDLoss(lt, reduction='none', loss_weight=2.0, **kw)
    out = mod(pp, tt, torch.from_numpy(w).cuda())
    out.sum().backward()
    check_close(f'{seed}.loss', out.detach().cpu().numpy(), ref['loss'], loss_bound(ref['loss'], r32['loss']))
    check_close(f'{seed}.gp', pp.grad.cpu().numpy(), ref['grad_pred'], grad_bound(ref['grad_pred'], r32['grad_pred']))
    check_close(f'{seed}.gt', tt.grad.cpu().numpy(), ref['grad_target'], grad_bound(ref['grad_target'], r32['grad_target']))
