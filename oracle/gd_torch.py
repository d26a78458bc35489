"""oracle/gd_torch.py — TEST INFRASTRUCTURE, NOT PRODUCT CODE.

A device-agnostic PyTorch restatement of the reference loss (gaussian_distance_loss.py:8-310) that relies on
torch autograd for the backward, exactly as the reference does.  Two uses:
  * a third, independent implementation for the CPU tests (autograd gradients vs the hand-derived ones of
    gd_oracle.c), checked against the golden vectors;
  * the "reference-style eager PyTorch op chain on the same device" timing baseline of tools/small_p_latency.py
    (SURVEY.md §8d config 2).  It is written in entry form (S11,S12,S22 instead of (N,2,2) bmm chains), which
    needs FEWER kernels than the reference's own code (~60 vs 106-143 top-level ops), so comparisons against it
    are conservative.
Never imported by the product package.
"""
import torch

_SQRT2_TERM = 4.656854249492381


def _gauss(box, c):
    box = box.reshape(-1, 7)
    if not isinstance(c, torch.Tensor):
        c = torch.tensor(c).to(box)          # as the reference (:9-10): fp32-rounded, then cast to the box dtype
    x, y, z, w, h, l, r = box.unbind(-1)
    X, Y, Z = x + c[0] * w, y + c[1] * h, z + c[2] * l                      # :12 (unclamped dims)
    a = 0.5 * w.clamp(min=1e-7, max=1e7)                                    # :13,:19
    b = 0.5 * h.clamp(min=1e-7, max=1e7)
    e = 0.5 * l.clamp(min=1e-7, max=1e7)                                    # :14,:20
    co, si = torch.cos(r), torch.sin(r)
    return dict(X=X, Y=Y, Z=Z, a=a, b=b, e=e, co=co, si=si)


def _rot(dA, dB, co, si):
    return dA * co * co + dB * si * si, (dA - dB) * si * co, dA * si * si + dB * co * co


def _post(d, fun, tau):                                                     # :24-39
    if fun == 'log1p':
        d = torch.log1p(d)
    elif fun == 'expm1':
        d = torch.expm1(d)
    elif fun == 'nlog':
        d = -torch.log(1 - d + 1e-7)
    elif fun != 'none':
        raise ValueError(f'Invalid non-linear function {fun}')
    return 1 - tau / (tau + d) if tau >= 1.0 else d


def _kld(p, t, alpha):                                                      # :109-137, no sqrt / post
    iA, iB, iE = (1 / p['a']) ** 2, (1 / p['b']) ** 2, (1 / p['e']) ** 2
    P11, P12, P22 = _rot(iA, iB, p['co'], p['si'])
    S11, S12, S22 = _rot(t['a'] ** 2, t['b'] ** 2, t['co'], t['si'])
    dX, dY, dZ = p['X'] - t['X'], p['Y'] - t['Y'], p['Z'] - t['Z']
    xyz = 0.5 * (dX * dX * P11 + 2 * dX * dY * P12 + dY * dY * P22) + 0.5 * dZ * dZ * iE
    whlr = 0.5 * (P11 * S11 + 2 * P12 * S12 + P22 * S22) + 0.5 * iE * t['e'] ** 2
    whlr = whlr + (p['a'].log() + p['b'].log() + p['e'].log()) - (t['a'].log() + t['b'].log() + t['e'].log()) - 1.5
    return xyz / (alpha * alpha) + whlr


def pair_loss(pred, target, loss_type, fun='log1p', tau=1.0, alpha=1.0, center_offset=(0, 0, 0.5), **kw):
    """Per-pair loss (N,), differentiable wrt pred and target."""
    p, t = _gauss(pred, center_offset), _gauss(target, center_offset)
    if loss_type == 'gwd3d':                                                # :42-106
        normalize = kw.pop('normalize', True)
        Sp, St = _rot(p['a'] ** 2, p['b'] ** 2, p['co'], p['si']), _rot(t['a'] ** 2, t['b'] ** 2, t['co'], t['si'])
        dxyz = (p['X'] - t['X']) ** 2 + (p['Y'] - t['Y']) ** 2 + (p['Z'] - t['Z']) ** 2
        T = Sp[0] * St[0] + 2 * Sp[1] * St[1] + Sp[2] * St[2]
        D = p['a'] * p['b'] * t['a'] * t['b']
        whlr = p['a'] ** 2 + p['b'] ** 2 + t['a'] ** 2 + t['b'] ** 2 - 2 * (T + 2 * D).clamp(0).sqrt() + (p['e'] - t['e']) ** 2
        d = (dxyz + alpha * alpha * whlr).clamp(0).sqrt()
        if normalize:
            d = d / (2 * ((D.log() + p['e'].log() + t['e'].log()) / 6).exp())
        return _post(d, fun, tau)
    if loss_type == 'kfiou3d':                                              # :227-248
        kw.pop('sqrt', None)
        Sp, St = _rot(p['a'] ** 2, p['b'] ** 2, p['co'], p['si']), _rot(t['a'] ** 2, t['b'] ** 2, t['co'], t['si'])
        S11, S12, S22 = Sp[0] + St[0], Sp[1] + St[1], Sp[2] + St[2]
        det = (S11 * S22 - S12 * S12) * (p['e'] ** 2 + t['e'] ** 2)
        vp, vt = p['a'] * p['b'] * p['e'], t['a'] * t['b'] * t['e']
        inter = vp * vt / det.clamp(min=1e-7).sqrt()
        union = (vp + vt - inter).clamp(min=1e-7)
        return _post(1 - _SQRT2_TERM * inter / union, fun, 0.0)
    sqrt = kw.pop('sqrt', True)
    if loss_type == 'kld3d':
        d = _kld(p, t, alpha)
    elif loss_type == 'jd3d':                                               # :189-198
        d = 0.5 * (_kld(p, t, alpha) + _kld(t, p, alpha))
    elif loss_type in ('kld3d_symmax', 'kld3d_symmin'):                     # :201-224
        d1, d2 = _kld(p, t, alpha), _kld(t, p, alpha)
        if sqrt:
            d1, d2 = d1.clamp(0).sqrt(), d2.clamp(0).sqrt()
        return _post(torch.max(d1, d2) if loss_type.endswith('max') else torch.min(d1, d2), fun, tau)
    elif loss_type == 'bd3d':                                               # :144-186
        Sp, St = _rot(p['a'] ** 2, p['b'] ** 2, p['co'], p['si']), _rot(t['a'] ** 2, t['b'] ** 2, t['co'], t['si'])
        S11, S12, S22 = 0.5 * (Sp[0] + St[0]), 0.5 * (Sp[1] + St[1]), 0.5 * (Sp[2] + St[2])
        Sl = 0.5 * (p['e'] ** 2 + t['e'] ** 2)
        det = (S11 * S22 - S12 * S12).clamp(min=1e-7)
        dX, dY, dZ = p['X'] - t['X'], p['Y'] - t['Y'], p['Z'] - t['Z']
        idet = det.reciprocal()
        xyz = 0.125 * (dX * dX * S22 * idet - 2 * dX * dY * S12 * idet + dY * dY * S11 * idet) + 0.125 * dZ * dZ / Sl
        whlr = 0.5 * (det.log() + Sl.log()) - 0.25 * ((p['a'] ** 2).log() + (p['b'] ** 2).log() + (p['e'] ** 2).log()) \
            - 0.25 * ((t['a'] ** 2).log() + (t['b'] ** 2).log() + (t['e'] ** 2).log())
        d = xyz / (alpha * alpha) + whlr
    else:
        raise KeyError(loss_type)
    if kw:
        raise TypeError(f'unexpected kwargs {sorted(kw)}')
    if sqrt:
        d = d.clamp(0).sqrt()
    return _post(d, fun, tau)


def gd_loss(pred, target, loss_type, weight=None, avg_factor=None, reduction='mean', loss_weight=1.0, **kw):
    """GDLoss.forward semantics (:280-310 + mmdet weight_reduce_loss) on any device."""
    if weight is not None and weight.shape == pred.shape:
        weight = weight.mean(-1)
    loss = pair_loss(pred, target, loss_type, **kw)
    if weight is not None:
        loss = loss * weight.reshape(-1)
    if avg_factor is None:
        loss = loss if reduction == 'none' else (loss.mean() if reduction == 'mean' else loss.sum())
    elif reduction == 'mean':
        loss = loss.sum() / avg_factor
    elif reduction != 'none':
        raise ValueError('avg_factor can not be used with reduction="sum"')
    return loss * loss_weight
