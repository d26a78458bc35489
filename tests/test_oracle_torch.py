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
