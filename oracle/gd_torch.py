"""oracle/gd_torch.py — TEST INFRASTRUCTURE, NOT PRODUCT CODE.

A device-agnostic PyTorch restatement of the reference loss (gaussian_distance_loss.py:8-310) that relies on
torch autograd for the backward, exactly as the reference does.  Two uses:
  * a third, independent implementation for the CPU tests (autograd gradients vs the hand-derived ones of
    gd_oracle.c), checked against the golden vectors;
  * timing baselines (bench.py `cpu_baseline`, tests/perf/small_p_latency.py), in TWO shapes:
      - `literal_*` (second half of this file): the reference's OWN op chain, op for op — `stack` / `diag_embed` / `bmm` on
        (N,2,2) matrices, the `@weighted_loss` wrapper, GDLoss.forward's host logic — 103 / 110 / 140 top-level ATen ops
        forward for gwd3d / kld3d / bd3d (the reference: 103 / 110 / 140 by the same count here, SURVEY.md §8a quotes 106 / 113 /
        143).  In fp32 it reproduces the reference's fp32 golden values BIT FOR BIT (same ops, same order), so its time is
        the reference's time.  This is `cpu_baseline.torch_chain`.
      - `pair_loss` / `gd_loss` (first half): an entry-form rewrite (S11,S12,S22 as (N,) vectors instead of (N,2,2) matrices:
        no stack / diag_embed / bmm; 131 / 137 / 148 top-level elementwise ops by the same count); measured 2.2-2.6x FASTER
        than the reference's chain (8 threads, 1 M pairs) — NOT the reference's shape.  This is `cpu_baseline.torch_chain_lean`.
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


# =====================================================================================================================
# literal mode: the reference's op chain restated op for op (same ATen calls in the same order on the same shapes).
# Line references are to /root/reference/mmdet3d_gaussian/models/losses/gaussian_distance_loss.py.
# =====================================================================================================================

def _dg(m):
    """main diagonal of a batch of 2x2 matrices, as a view"""
    return m.diagonal(dim1=-2, dim2=-1)


def _sandwich(rot, mid):
    """rot . mid . rot^T with two batched products (:86-87, :115-116, :148-149, :231-232)"""
    return rot.bmm(mid).bmm(rot.permute(0, 2, 1))


def _quad(d, m):
    """d^T m d per pair, d as (N,2,1) columns (:122-123, :170-171)"""
    return d.permute(0, 2, 1).bmm(m).bmm(d).view(-1)


def _det2(m):
    return m[..., 0, 0] * m[..., 1, 1] - m[..., 1, 0] * m[..., 0, 1]


def literal_preprocess(box, center_offset):
    """:8-21 -> (gravity centre (N,3), R (N,2,2), S = diag(w,h)/2 (N,2,2), l/2 (N,))"""
    if not isinstance(center_offset, torch.Tensor):
        center_offset = torch.tensor(center_offset).to(box)
    box = box.reshape(-1, 7)
    centre = box[..., :3] + center_offset[None, :] * box[..., 3:6]
    wh = box[..., 3:5].clamp(min=1e-7, max=1e7)
    ln = box[..., 5].clamp(min=1e-7, max=1e7)
    yaw = box[..., 6]
    cs, sn = torch.cos(yaw), torch.sin(yaw)
    rot = torch.stack((cs, -sn, sn, cs), dim=-1).reshape(-1, 2, 2)
    return centre, rot, 0.5 * torch.diag_embed(wh), 0.5 * ln


def _lit_gwd(P, T, fun, tau, alpha, normalize=True):                        # :42-106
    cp, Rp, Sp, ep = P
    ct, Rt, St, et = T
    d_c = (cp - ct).square().sum(dim=-1)
    d_s = _dg(Sp).square().sum(dim=-1)
    d_s = d_s + _dg(St).square().sum(dim=-1)
    cov_p = _sandwich(Rp, Sp.square())
    cov_t = _sandwich(Rt, St.square())
    prod = cov_p.bmm(cov_t)
    tr = _dg(prod).sum(dim=-1)
    dsq = _dg(Sp).prod(dim=-1)
    dsq = dsq * _dg(St).prod(dim=-1)
    d_s = d_s + (-2) * ((tr + 2 * dsq).clamp(0).sqrt())
    d_s = d_s + (ep - et).square()
    dist = (d_c + alpha * alpha * d_s).clamp(0).sqrt()
    if normalize:
        logsum = dsq.log() + ep.log() + et.log()
        dist = dist / (2 * (logsum / 6).exp())
    return _post(dist, fun, tau)


def _lit_kld(P, T, fun, tau, alpha, sqrt=True):                             # :109-141
    cp, Rp, Sp, ep = P
    ct, Rt, St, et = T
    Sp_inv = _dg(Sp).reciprocal().diag_embed()
    ep_inv = ep.reciprocal()
    icov_p = _sandwich(Rp, Sp_inv.square())
    cov_t = _sandwich(Rt, St.square())
    dxy = (cp[..., :2] - ct[..., :2]).unsqueeze(-1)
    dz = cp[..., 2] - ct[..., 2]
    d_c = 0.5 * _quad(dxy, icov_p)
    d_c = d_c + 0.5 * dz.square() * ep_inv.square()
    d_s = 0.5 * _dg(icov_p.bmm(cov_t)).sum(dim=-1)
    d_s = d_s + 0.5 * ep_inv.square() * et.square()
    ld_p = _dg(Sp).log().sum(dim=-1) + ep.log()
    ld_t = _dg(St).log().sum(dim=-1) + et.log()
    d_s = d_s + (ld_p - ld_t)
    d_s = d_s - 1.5
    dist = (d_c / (alpha * alpha) + d_s)
    if sqrt:
        dist = dist.clamp(0).sqrt()
    return _post(dist, fun, tau)


def _lit_bd(P, T, fun, tau, alpha, sqrt=True):                              # :144-186
    cp, Rp, Sp, ep = P
    ct, Rt, St, et = T
    cov_p = _sandwich(Rp, Sp.square())
    cov_t = _sandwich(Rt, St.square())
    cov = 0.5 * (cov_p + cov_t)
    cov_l = 0.5 * (ep.square() + et.square())
    det = _det2(cov).clamp(min=1e-7)
    adj = torch.stack((cov[..., 1, 1], -cov[..., 0, 1], -cov[..., 1, 0], cov[..., 0, 0]), dim=-1).reshape(-1, 2, 2)
    inv = adj * det.reciprocal().unsqueeze(-1).unsqueeze(-1)
    dxy = (cp[..., :2] - ct[..., :2]).unsqueeze(-1)
    dz = cp[..., 2] - ct[..., 2]
    d_c = 0.125 * _quad(dxy, inv)
    d_c = d_c + 0.125 * dz.square() * cov_l.reciprocal()
    d_s = 0.5 * (det.log() + cov_l.log())
    d_s = d_s - 0.25 * (_dg(Sp.square()).log().sum(dim=-1) + ep.square().log())
    d_s = d_s - 0.25 * (_dg(St.square()).log().sum(dim=-1) + et.square().log())
    dist = (d_c / (alpha * alpha) + d_s)
    if sqrt:
        dist = dist.clamp(0).sqrt()
    return _post(dist, fun, tau)


def _lit_kfiou(P, T, fun='expm1', tau=0.0, alpha=1.0, sqrt=False):         # :227-248 (tau, alpha, sqrt unused there too)
    cp, Rp, Sp, ep = P
    ct, Rt, St, et = T
    tot = _sandwich(Rp, Sp.square()) + _sandwich(Rt, St.square())
    det = _det2(tot) * (ep.square() + et.square())
    vol_p = _dg(Sp).prod(dim=-1) * ep
    vol_t = _dg(St).prod(dim=-1) * et
    inter = vol_p * vol_t / det.clamp(min=1e-7).sqrt()
    union = (vol_p + vol_t - inter).clamp(min=1e-7)
    return _post(1 - _SQRT2_TERM * (inter / union), fun, 0.0)


def _mmdet_weighted(fn):
    """mmdet's `@weighted_loss` (third-party): loss = fn(pred, target, **kw); `* weight`; then
    weight_reduce_loss: avg_factor None -> none | mean | sum; else mean -> sum / avg_factor, none -> as is, sum -> ValueError."""
    def wrapped(pred, target, weight=None, reduction='mean', avg_factor=None, **kw):
        loss = fn(pred, target, **kw)
        if weight is not None:
            loss = loss * weight
        if avg_factor is None:
            if reduction == 'mean':
                loss = loss.mean()
            elif reduction == 'sum':
                loss = loss.sum()
        elif reduction == 'mean':
            loss = loss.sum() / avg_factor
        elif reduction != 'none':
            raise ValueError('avg_factor can not be used with reduction="sum"')
        return loss
    return wrapped


_w_gwd, _w_kld, _w_bd, _w_kfiou = (_mmdet_weighted(f) for f in (_lit_gwd, _lit_kld, _lit_bd, _lit_kfiou))


@_mmdet_weighted
def _w_jd(P, T, fun, tau, alpha, sqrt=True):                                # :189-198
    d = _w_kld(P, T, fun='none', tau=0, alpha=alpha, sqrt=False, reduction='none')
    d = d + _w_kld(T, P, fun='none', tau=0, alpha=alpha, sqrt=False, reduction='none')
    d = d * 0.5
    if sqrt:
        d = d.clamp(0).sqrt()
    return _post(d, fun, tau)


def _sym(pick):
    @_mmdet_weighted
    def loss(P, T, fun, tau, alpha, sqrt=True):                             # :201-224
        a = _w_kld(P, T, fun='none', tau=0, alpha=alpha, sqrt=sqrt, reduction='none')
        b = _w_kld(T, P, fun='none', tau=0, alpha=alpha, sqrt=sqrt, reduction='none')
        return _post(pick(a, b), fun, tau)
    return loss


_LITERAL = {'gwd3d': _w_gwd, 'kld3d': _w_kld, 'bd3d': _w_bd, 'jd3d': _w_jd, 'kld3d_symmax': _sym(torch.max),
            'kld3d_symmin': _sym(torch.min), 'kfiou3d': _w_kfiou}


def literal_gd_loss(pred, target, loss_type, weight=None, avg_factor=None, reduction='mean', loss_weight=1.0, fun='log1p',
                    tau=1.0, alpha=1.0, center_offset=(0, 0, 0.5), **kw):
    """GDLoss.forward (:280-310) over the literal chain: the host-side early-out with its device-to-host wait (:290-292), the
    (N,7) weight mean (:295-296), preprocess x2 (:298-299), the wrapped loss function, `* loss_weight` (:310)."""
    if weight is not None and not torch.any(weight > 0) and reduction != 'none':
        return (pred * weight).sum()
    if weight is not None and weight.dim() > 1:
        weight = weight.mean(dim=-1)
    P = literal_preprocess(pred, center_offset)
    T = literal_preprocess(target, center_offset)
    return _LITERAL[loss_type](P, T, weight=weight, avg_factor=avg_factor, reduction=reduction, fun=fun, tau=tau, alpha=alpha,
                               **kw) * loss_weight


def literal_pair_loss(pred, target, loss_type, fun='log1p', tau=1.0, alpha=1.0, center_offset=(0, 0, 0.5), **kw):
    return literal_gd_loss(pred, target, loss_type, reduction='none', fun=fun, tau=tau, alpha=alpha, center_offset=center_offset, **kw)


def count_top_level_aten_ops(fn):
    """The count SURVEY.md §8a quotes ("top-level ATen ops fwd"): aten:: events without a parent in a torch.profiler run."""
    from torch.profiler import profile
    with profile() as prof:
        fn()
    return sum(1 for e in prof.events() if e.cpu_parent is None and e.name.startswith('aten::'))
