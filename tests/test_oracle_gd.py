"""Pins oracle/gd_oracle.c against golden vectors produced by the REAL reference
(gaussian_distance_loss.py run in fp32 and fp64; tests/golden/make_golden_gd.py).  CPU only."""
import numpy as np
import pytest

import oracle
from gd_golden import (check_close, families, grad_bound, index, loss_bound, module, pair_case_names, pairs)


@pytest.mark.parametrize('case', pair_case_names())
def test_oracle_f64_matches_reference_f64(case):
    """fp64 oracle == reference run in fp64, to rounding: loss 1e-9, grads 1e-9 of the row scale."""
    c = index()['pairs']['cases'][case]
    prm = oracle.make_params(c['loss_type'], **c['kwargs'])
    g = pairs()
    for fam in families():
        r = oracle.gd_loss(g[f'in.{fam}.pred'], g[f'in.{fam}.target'], prm, dtype=np.float64)
        l64 = g[f'{case}.{fam}.loss64']
        # identical boxes: sqrt of cancellation noise, abs 1e-7 is the resolution of fp64 there
        tol = 1e-7 if fam == 'ident' else 1e-9
        check_close(f'{case}.{fam}.loss', r['loss'], l64, tol * (1 + np.abs(l64)))
        if fam == 'ident':
            continue
        for key, ours in (('gp64', r['grad_pred']), ('gt64', r['grad_target'])):
            ref = g[f'{case}.{fam}.{key}']
            scale = np.nanmax(np.where(np.isfinite(ref), np.abs(ref), 0), axis=-1, keepdims=True)
            check_close(f'{case}.{fam}.{key}', ours, ref, 1e-9 * (1 + scale))
            assert (np.isfinite(ref) == np.isfinite(ours)).all(), 'NaN/inf pattern differs'


@pytest.mark.parametrize('case', pair_case_names())
def test_oracle_f32_within_reference_f32_noise(case):
    """fp32 build (the timed cpu_baseline 'port') obeys the same tolerance policy as the HIP path."""
    c = index()['pairs']['cases'][case]
    prm = oracle.make_params(c['loss_type'], **c['kwargs'])
    g = pairs()
    for fam in families(with_ident=False):
        r = oracle.gd_loss(g[f'in.{fam}.pred'], g[f'in.{fam}.target'], prm, dtype=np.float32)
        key = f'{case}.{fam}'
        check_close(key + '.loss', r['loss'], g[key + '.loss64'], loss_bound(g[key + '.loss64'], g[key + '.loss32']))
        check_close(key + '.gp', r['grad_pred'], g[key + '.gp64'], grad_bound(g[key + '.gp64'], g[key + '.gp32']))
        check_close(key + '.gt', r['grad_target'], g[key + '.gt64'], grad_bound(g[key + '.gt64'], g[key + '.gt32']))


def test_survey_headline_values():
    """N=1000 seed-0 means of the reference (tau 1 / log1p): gwd .7546, kld .7703, bd .7180
    are for another generator; here we pin the kitti-family means stored by the generator instead."""
    g = pairs()
    for case in ('gwd3d.0', 'kld3d.0', 'bd3d.0'):
        c = index()['pairs']['cases'][case]
        prm = oracle.make_params(c['loss_type'], **c['kwargs'])
        r = oracle.gd_loss(g['in.kitti.pred'], g['in.kitti.target'], prm, dtype=np.float64)
        assert abs(r['loss'].mean() - g[f'{case}.kitti.loss64'].mean()) < 1e-12
        assert abs(r['loss_sum'] - g[f'{case}.kitti.loss64'].sum()) < 1e-9


def test_symmetries():
    """Kernel algebra: yaw+pi => distance 0; yaw+pi/2 with w<->h swap => 0 (GWD, tau=0, none)."""
    rng = np.random.default_rng(0)
    n = 64
    t = np.stack([rng.uniform(0, 70, n), rng.uniform(-40, 40, n), rng.uniform(-3, 1, n), rng.uniform(.5, 2.5, n),
                  rng.uniform(.5, 4.5, n), rng.uniform(.5, 2, n), rng.uniform(-3, 3, n)], -1)
    prm = oracle.make_params('gwd3d', fun='none', tau=0.0)
    p = t.copy(); p[:, 6] += np.pi
    assert np.abs(oracle.gd_loss(p, t, prm)['loss']).max() < 1e-6
    p = t.copy(); p[:, 6] += np.pi / 2; p[:, [3, 4]] = p[:, [4, 3]]
    # centre offset (0,0,.5) does not involve w,h so the swap leaves the centre alone
    assert np.abs(oracle.gd_loss(p, t, prm)['loss']).max() < 1e-6


def test_clamped_dim_has_zero_gaussian_grad():
    """Out-of-range dims: clamp(1e-7,1e7) kills the Gaussian route; centre still uses the raw dim (:12-14)."""
    p = np.array([[1., 2., 0.5, -0.3, 1.0, -0.2, 0.3]])
    t = np.array([[1.2, 2.1, 0.4, 1.0, 1.1, 0.9, 0.2]])
    prm = oracle.make_params('kld3d', fun='log1p', tau=1.0, center_offset=(0, 0, 0.5))
    r = oracle.gd_loss(p, t, prm)
    assert r['grad_pred'][0, 3] == 0.0            # w clamped, c0 = 0
    assert r['grad_pred'][0, 5] == 0.5 * r['grad_pred'][0, 2]   # l clamped: only the centre route c2*gZ


def test_weight_scale_contract():
    g = pairs()
    p, t = g['in.kitti.pred'], g['in.kitti.target']
    prm = oracle.make_params('bd3d')
    w = np.linspace(0, 2, p.shape[0])
    a = oracle.gd_loss(p, t, prm)
    b = oracle.gd_loss(p, t, prm, row_weight=w, scale=0.25)
    np.testing.assert_allclose(b['loss'], 0.25 * w * a['loss'], rtol=1e-14)
    np.testing.assert_allclose(b['grad_pred'], 0.25 * w[:, None] * a['grad_pred'], rtol=1e-14)
    assert abs(b['loss_sum'] - b['loss'].sum()) < 1e-12


def test_empty_and_badarg():
    prm = oracle.make_params('gwd3d')
    r = oracle.gd_loss(np.zeros((0, 7)), np.zeros((0, 7)), prm)
    assert r['loss'].shape == (0,) and r['loss_sum'] == 0.0
    prm.loss_type = 99
    with pytest.raises(RuntimeError):
        oracle.gd_loss(np.zeros((1, 7)), np.zeros((1, 7)), prm)


def _module_reduce(out_none, weight, reduction, avg_factor, loss_weight):
    """mmdet weight_reduce_loss semantics on top of per-pair oracle losses."""
    loss = out_none if weight is None else out_none * weight
    if avg_factor is None:
        red = loss if reduction == 'none' else (loss.mean() if reduction == 'mean' else loss.sum())
    elif reduction == 'mean':
        red = loss.sum() / avg_factor
    elif reduction == 'none':
        red = loss
    else:
        raise ValueError
    return red * loss_weight


@pytest.mark.parametrize('case', sorted(index()['module']))
def test_module_glue_fixture_is_reproduced_by_oracle(case):
    """GDLoss.forward glue (weights, avg_factor, reduction, shapes, kwargs merge) restated on top of the
    oracle reproduces the reference module's fp64 outputs and gradients."""
    spec = index()['module'][case]
    m = module()
    if spec.get('raises'):
        pytest.skip('reference raises here; covered by the host-logic test of GDLoss')
    ctor = dict(spec['ctor'])
    call = spec['call']
    reduction = call.get('reduction_override') or ctor.pop('reduction')
    ctor.pop('reduction', None)
    loss_weight = ctor.pop('loss_weight', 1.0)
    ctor.update(call.get('call_kwargs', {}))
    prm = oracle.make_params(spec['loss_type'], **ctor)
    g = pairs()
    p, t = g['in.kitti.pred'], g['in.kitti.target']
    n = p.shape[0]
    w = None
    if 'weight' in call:
        w = {'w1': m['w1'], 'w7': m['w7'], 'w0': np.zeros(n), 'w07': np.zeros((n, 7))}[call['weight']].astype(np.float64)
    out64 = m[case + '.out64']
    gp64 = m[case + '.gp64']
    if w is not None and not (w > 0).any() and reduction != 'none':
        # early-out (pred*weight).sum() (:290-292): value 0, grad = weight
        assert float(out64) == 0.0
        np.testing.assert_array_equal(gp64, np.broadcast_to(w.reshape(n, -1), (n, 7)))
        return
    if w is not None and w.ndim == 2:
        w = w.mean(-1)
    base = oracle.gd_loss(p, t, prm)
    ours = _module_reduce(base['loss'], w, reduction, call.get('avg_factor'), loss_weight)
    np.testing.assert_allclose(np.asarray(ours).reshape(out64.shape), out64, rtol=1e-10, atol=1e-12)
    # gradient: d(out)/d(loss_i) known in closed form
    wi = np.ones(n) if w is None else w
    if reduction == 'none':
        coef = m['up'].astype(np.float64) * wi * loss_weight
    elif reduction == 'sum':
        coef = wi * loss_weight
    else:
        den = call.get('avg_factor') or n
        coef = wi * loss_weight / den
    np.testing.assert_allclose(coef[:, None] * base['grad_pred'], gp64, rtol=1e-9, atol=1e-11)
