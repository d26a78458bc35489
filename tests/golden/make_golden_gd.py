#!/usr/bin/env python3
"""Generate golden vectors for the Gaussian-distance losses FROM THE REAL REFERENCE.

Run in the build container only (needs /root/reference):

    python3 -B tests/golden/make_golden_gd.py

Imports /root/reference/mmdet3d_gaussian/models/losses/gaussian_distance_loss.py
(via tests/golden/_ref_loader.py), runs ``GDLoss`` on seeded inputs in fp32 and
fp64 on CPU, and stores inputs + outputs (+ autograd gradients) in

    tests/golden/gd_pairs.npz     per-pair loss / grads, 7 loss types x param grid x 5 input families
    tests/golden/gd_module.npz    GDLoss.forward glue: weights, avg_factor, reductions, shapes
    tests/golden/gd_index.json    the case list (names -> parameters)

Only data is written; no reference source text is stored.
"""
import json
import math
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from _ref_loader import load_reference_loss  # noqa: E402

N_FAM = 128
N_IDENT = 16


def family_inputs():
    """Five input families. Returns {name: (pred, target)} fp32 numpy."""
    g = torch.Generator().manual_seed(20261003)

    def U(n, lo, hi):
        return torch.rand(n, generator=g) * (hi - lo) + lo

    fam = {}
    # 1. KITTI-like (SURVEY.md §8d recipe)
    n = N_FAM
    tgt = torch.stack([U(n, 0, 70), U(n, -40, 40), U(n, -3, 1), U(n, 0.5, 2.5),
                       U(n, 0.5, 4.5), U(n, 0.5, 2.0), U(n, -math.pi, math.pi)], -1)
    sigma = torch.tensor([0.3, 0.3, 0.1, 0.1, 0.1, 0.1, 0.1])
    fam['kitti'] = (tgt + torch.randn(n, 7, generator=g) * sigma, tgt)
    # 2. near-identical (sigma = 1e-3): the fp32 cancellation regime
    tgt = torch.stack([U(n, 0, 70), U(n, -40, 40), U(n, -3, 1), U(n, 0.5, 2.5),
                       U(n, 0.5, 4.5), U(n, 0.5, 2.0), U(n, -math.pi, math.pi)], -1)
    fam['near'] = (tgt + torch.randn(n, 7, generator=g) * 1e-3, tgt)
    # 3. encoded-delta-like: |v| < 1, negative "dims" -> clamp(1e-7) (Waymo config quirk)
    pred = torch.rand(n, 7, generator=g) * 2 - 1
    tgt = torch.rand(n, 7, generator=g) * 2 - 1
    # a few rows with positive small dims so that both clamp branches appear
    pred[::4, 3:6] = pred[::4, 3:6].abs() + 0.05
    tgt[::4, 3:6] = tgt[::4, 3:6].abs() + 0.05
    fam['delta'] = (pred, tgt)
    # 4. large scale: nuScenes ranges, dims up to 30 m, yaw outside [-pi, pi]
    tgt = torch.stack([U(n, -54, 54), U(n, -54, 54), U(n, -5, 3), U(n, 0.3, 30),
                       U(n, 0.3, 30), U(n, 0.3, 10), U(n, -3 * math.pi, 3 * math.pi)], -1)
    sigma = torch.tensor([1.0, 1.0, 0.3, 0.5, 0.5, 0.3, 0.4])
    pred = tgt + torch.randn(n, 7, generator=g) * sigma
    pred[:, 3:6] = pred[:, 3:6].abs() + 0.05
    fam['large'] = (pred, tgt)
    # 5. identical boxes (loss only; reference autograd yields NaN grads here)
    n = N_IDENT
    tgt = torch.stack([U(n, 0, 70), U(n, -40, 40), U(n, -3, 1), U(n, 0.5, 2.5),
                       U(n, 0.5, 4.5), U(n, 0.5, 2.0), U(n, -math.pi, math.pi)], -1)
    fam['ident'] = (tgt.clone(), tgt)
    return {k: (p.float().contiguous().numpy(), t.float().contiguous().numpy())
            for k, (p, t) in fam.items()}


def pair_cases():
    """(name, loss_type, ctor kwargs) grid."""
    cases = []
    for lt in ('gwd3d', 'kld3d', 'bd3d', 'jd3d', 'kld3d_symmax', 'kld3d_symmin'):
        flag = 'normalize' if lt == 'gwd3d' else 'sqrt'
        grid = [
            dict(fun='log1p', tau=1.0, alpha=1.0),
            dict(fun='none', tau=0.0, alpha=1.0),
            dict(fun='log1p', tau=2.0, alpha=2.0),
            dict(fun='none', tau=0.0, alpha=1.0, **{flag: False}),
            dict(fun='log1p', tau=0.5, alpha=0.5, center_offset=(0.1, 0.2, 0.3)),
        ]
        for i, kw in enumerate(grid):
            cases.append((f'{lt}.{i}', lt, kw))
    grid = [
        dict(fun='expm1'),
        dict(fun='nlog'),
        dict(fun='none', tau=1.0),  # tau is ignored by kfiou3d (ref :247)
        dict(fun='expm1', center_offset=(0.5, 0.5, 0.5), alpha=2.0),
    ]
    for i, kw in enumerate(grid):
        cases.append((f'kfiou3d.{i}', 'kfiou3d', kw))
    return cases


def run_pairs(ref, pred_np, tgt_np, lt, kw, dtype):
    pred = torch.from_numpy(pred_np).to(dtype).requires_grad_(True)
    tgt = torch.from_numpy(tgt_np).to(dtype).requires_grad_(True)
    mod = ref.GDLoss(lt, reduction='none', loss_weight=1.0, **kw)
    loss = mod(pred, tgt)
    loss.sum().backward()
    return (loss.detach().numpy(), pred.grad.numpy(), tgt.grad.numpy())


def gen_pairs(ref):
    fams = family_inputs()
    out = {}
    index = {'families': {k: int(v[0].shape[0]) for k, v in fams.items()}, 'cases': {}}
    for fname, (p, t) in fams.items():
        out[f'in.{fname}.pred'] = p
        out[f'in.{fname}.target'] = t
    with np.errstate(all='ignore'):
        for name, lt, kw in pair_cases():
            index['cases'][name] = dict(loss_type=lt, kwargs={k: (list(v) if isinstance(v, tuple) else v)
                                                             for k, v in kw.items()})
            for fname, (p, t) in fams.items():
                l32, gp32, gt32 = run_pairs(ref, p, t, lt, kw, torch.float32)
                l64, gp64, gt64 = run_pairs(ref, p, t, lt, kw, torch.float64)
                key = f'{name}.{fname}'
                out[key + '.loss32'] = l32
                out[key + '.loss64'] = l64
                if fname != 'ident':
                    out[key + '.gp32'] = gp32
                    out[key + '.gp64'] = gp64
                    out[key + '.gt32'] = gt32
                    out[key + '.gt64'] = gt64
    return out, index


def module_cases():
    """GDLoss.forward glue cases: (name, loss_type, ctor kw, call spec)."""
    cases = []
    base = dict(fun='log1p', tau=1.0, alpha=1.0, loss_weight=5.0)
    for lt in ('gwd3d', 'kld3d', 'bd3d'):
        cases += [
            (f'{lt}.mean', lt, dict(base, reduction='mean'), dict()),
            (f'{lt}.sum', lt, dict(base, reduction='sum'), dict()),
            (f'{lt}.none', lt, dict(base, reduction='none'), dict()),
            (f'{lt}.mean.w1', lt, dict(base, reduction='mean'), dict(weight='w1')),
            (f'{lt}.mean.w7', lt, dict(base, reduction='mean'), dict(weight='w7')),
            (f'{lt}.mean.w7.avg', lt, dict(base, reduction='mean'), dict(weight='w7', avg_factor=37.5)),
            (f'{lt}.none.w1.avg', lt, dict(base, reduction='none'), dict(weight='w1', avg_factor=11.0)),
            (f'{lt}.sum.w1', lt, dict(base, reduction='sum'), dict(weight='w1')),
            (f'{lt}.override_sum', lt, dict(base, reduction='mean'), dict(reduction_override='sum')),
            (f'{lt}.override_none.w7', lt, dict(base, reduction='mean'),
             dict(weight='w7', reduction_override='none')),
            (f'{lt}.zero_weight', lt, dict(base, reduction='mean'), dict(weight='w0', avg_factor=4.0)),
            (f'{lt}.zero_weight7.sum', lt, dict(base, reduction='sum'), dict(weight='w07')),
            (f'{lt}.shape244', lt, dict(base, reduction='mean'), dict(reshape=[2, 64, 7])),
            (f'{lt}.shape244.none', lt, dict(base, reduction='none'), dict(reshape=[2, 64, 7])),
        ]
    cases += [
        ('gwd3d.kw_normalize', 'gwd3d', dict(base, reduction='mean', normalize=False), dict()),
        ('gwd3d.call_normalize', 'gwd3d', dict(base, reduction='mean'), dict(call_kwargs=dict(normalize=False))),
        ('kld3d.call_sqrt', 'kld3d', dict(base, reduction='mean', sqrt=True), dict(call_kwargs=dict(sqrt=False))),
        ('kfiou3d.mean.w1', 'kfiou3d', dict(fun='expm1', loss_weight=2.0, reduction='mean'), dict(weight='w1')),
        ('jd3d.mean.avg', 'jd3d', dict(base, reduction='mean'), dict(avg_factor=50.0)),
    ]
    return cases


def gen_module(ref, fams):
    p_np, t_np = fams['kitti']
    n = p_np.shape[0]
    g = torch.Generator().manual_seed(7)
    w1 = torch.rand(n, generator=g)
    w1[::5] = 0.0
    w7 = torch.rand(n, 7, generator=g)
    w7[::7] = 0.0
    up = torch.rand(n, generator=g) + 0.5  # upstream grad for vector outputs
    weights = {'w1': w1, 'w7': w7, 'w0': torch.zeros(n), 'w07': torch.zeros(n, 7)}
    out = {'w1': w1.numpy(), 'w7': w7.numpy(), 'up': up.numpy()}
    index = {}
    for name, lt, ctor, call in module_cases():
        index[name] = dict(loss_type=lt, ctor=ctor, call=call)
        for dtype, tag in ((torch.float32, '32'), (torch.float64, '64')):
            pred = torch.from_numpy(p_np).to(dtype)
            tgt = torch.from_numpy(t_np).to(dtype)
            shape = call.get('reshape')
            if shape:
                pred = pred.reshape(shape)
                tgt = tgt.reshape(shape)
            pred.requires_grad_(True)
            mod = ref.GDLoss(lt, **ctor)
            kwargs = {}
            if 'weight' in call:
                kwargs['weight'] = weights[call['weight']].to(dtype)
            if 'avg_factor' in call:
                kwargs['avg_factor'] = call['avg_factor']
            if 'reduction_override' in call:
                kwargs['reduction_override'] = call['reduction_override']
            kwargs.update(call.get('call_kwargs', {}))
            try:
                res = mod(pred, tgt, **kwargs)
            except RuntimeError:
                # reference behaviour worth pinning: (N,) all-zero weight hits
                # `(pred * weight).sum()` (ref :290-292) which cannot broadcast
                index[name]['raises'] = 'RuntimeError'
                continue
            if res.dim() == 0:
                res.backward()
            else:
                res.backward(up.to(dtype).reshape(res.shape))
            out[f'{name}.out{tag}'] = res.detach().numpy()
            out[f'{name}.gp{tag}'] = pred.grad.numpy().reshape(-1, 7)
    return out, index


def main():
    torch.set_num_threads(1)
    ref = load_reference_loss()
    pairs, pidx = gen_pairs(ref)
    fams = family_inputs()
    module, midx = gen_module(ref, fams)
    np.savez_compressed(os.path.join(HERE, 'gd_pairs.npz'), **pairs)
    np.savez_compressed(os.path.join(HERE, 'gd_module.npz'), **module)
    with open(os.path.join(HERE, 'gd_index.json'), 'w') as f:
        json.dump({'pairs': pidx, 'module': midx,
                   'generator': 'tests/golden/make_golden_gd.py',
                   'torch': torch.__version__}, f, indent=1, sort_keys=True)
    for fn in ('gd_pairs.npz', 'gd_module.npz', 'gd_index.json'):
        print(fn, os.path.getsize(os.path.join(HERE, fn)), 'bytes')
    # headline sanity numbers: N=1000 seed-0 means are printed by the survey; here: kitti-family means
    for name in ('gwd3d.0', 'kld3d.0', 'bd3d.0'):
        print(name, 'kitti mean loss32', float(pairs[name + '.kitti.loss32'].mean()),
              'max|l32-l64|', float(np.abs(pairs[name + '.kitti.loss32'] - pairs[name + '.kitti.loss64']).max()))


if __name__ == '__main__':
    main()
