"""oracle/gd_torch.py (autograd-based torch restatement) against the golden vectors from the real reference. CPU."""
import numpy as np
import pytest
import torch

from gd_golden import families, index, pair_case_names, pairs
from oracle import gd_torch


@pytest.mark.parametrize('case', pair_case_names())
def test_torch_restatement_matches_reference_f64(case):
    c = index()['pairs']['cases'][case]
    g = pairs()
    kw = {k: (tuple(v) if isinstance(v, list) else v) for k, v in c['kwargs'].items()}
    for fam in families(with_ident=False):
        p = torch.from_numpy(g[f'in.{fam}.pred']).double().requires_grad_(True)
        t = torch.from_numpy(g[f'in.{fam}.target']).double().requires_grad_(True)
        loss = gd_torch.pair_loss(p, t, c['loss_type'], **kw)
        loss.sum().backward()
        l64 = g[f'{case}.{fam}.loss64']
        np.testing.assert_allclose(loss.detach().numpy(), l64, rtol=1e-7, atol=1e-9 * (1 + np.abs(l64).max()))
        for ours, key in ((p.grad, 'gp64'), (t.grad, 'gt64')):
            ref = g[f'{case}.{fam}.{key}']
            scale = np.abs(np.where(np.isfinite(ref), ref, 0)).max(-1, keepdims=True)
            err = np.abs(ours.numpy() - ref)
            assert (err[np.isfinite(ref)] <= (1e-6 * (1 + scale) * np.ones_like(ref))[np.isfinite(ref)]).all()


def test_head_torch_restatements_against_independent_formulas():
    """oracle/head_torch.py restates three third-party pieces (mmdet smooth_l1 / l1, mmdet3d add_sin_difference and
    delta decode).  Independent checks: torch's own F.smooth_l1_loss (same published formula), the identity
    sin(a)cos(b) - cos(a)sin(b) = sin(a - b), and decode(encode(x)) == x with the product-side coder mirror."""
    import importlib.util
    import os
    import torch.nn.functional as F
    from oracle import head_torch
    g = torch.Generator().manual_seed(0)
    p = torch.randn(200, 7, generator=g, dtype=torch.float64)
    t = p + torch.randn(200, 7, generator=g, dtype=torch.float64) * 0.2
    w = torch.rand(200, 7, generator=g, dtype=torch.float64)
    for beta in (1.0 / 9.0, 0.5, 1.0):
        ours = head_torch.smooth_l1(p, t, w, 13.0, beta, 2.0)
        ref = 2.0 * (F.smooth_l1_loss(p, t, beta=beta, reduction='none') * w).sum() / 13.0
        assert abs(ours.item() - ref.item()) <= 1e-12 * (1 + abs(ref.item()))
    l1 = head_torch.smooth_l1(p, t, None, 7.0, 0.0, 0.25)
    assert abs(l1.item() - 0.25 * (p - t).abs().sum().item() / 7.0) <= 1e-12
    a, b = head_torch.add_sin_difference(p, t)
    assert torch.allclose(a[:, 6] - b[:, 6], torch.sin(p[:, 6] - t[:, 6]), atol=1e-14)
    assert torch.equal(a[:, :6], p[:, :6]) and torch.equal(b[:, :6], t[:, :6])
    from mmdet3d_gaussian_amd import coders as cm
    anchors = torch.rand(200, 7, generator=g, dtype=torch.float64) + 0.5
    boxes = torch.rand(200, 7, generator=g, dtype=torch.float64) + 0.5
    enc = cm.DeltaXYZWLHRBBoxCoder.encode(anchors, boxes)
    assert torch.allclose(head_torch.delta_decode(anchors, enc), boxes, atol=1e-12)
