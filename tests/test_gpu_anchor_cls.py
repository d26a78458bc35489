"""GPU parity of the anchor heads' classification + direction loss (csrc/anchor_cls.hip) against the fp64 torch restatement
oracle/anchor_cls_torch.py of gd_anchor3d_head.py:84-92, :143-149 with mmdet's FocalLoss / CrossEntropyLoss.
Floating point: relative tolerance 1e-5 on the losses, 1e-5 * max|grad| + 1e-9 on the gradients (north_star's 1e-5)."""
import importlib

import pytest
import torch

pytestmark = pytest.mark.extras   # frozen extras outside SURVEY.md §8: `pytest -m extras` on a GPU box (conftest.py), not part of `-m gpu`

pkg = importlib.import_module('mmdet3d-gaussian_amd')
from oracle import anchor_cls_torch as ORA  # noqa: E402

FOCAL = dict(type='FocalLoss', use_sigmoid=True, gamma=2.0, alpha=0.25, loss_weight=1.0)
CE = dict(type='CrossEntropyLoss', use_sigmoid=False, loss_weight=0.2)


def make(B, A, C, H, W, seed, pos_frac=0.01, ignore_frac=0.02, scale=3.0):
    g = torch.Generator().manual_seed(seed)
    N = H * W * A
    cls = torch.randn(B, A * C, H, W, generator=g) * scale - 2.0
    dirs = torch.randn(B, A * 2, H, W, generator=g) * scale
    labels = torch.full((B, N), C, dtype=torch.long)
    u = torch.rand(B, N, generator=g)
    pos = u < pos_frac
    labels[pos] = torch.randint(0, C, (int(pos.sum()),), generator=g)
    lw = torch.ones(B, N)
    lw[(u >= pos_frac) & (u < pos_frac + ignore_frac)] = 0.0          # the assigner's ignored anchors: weight 0
    dt = torch.randint(0, 2, (B, N), generator=g)
    dw = pos.float()
    return cls, dirs, labels, lw, dt, dw


def run_both(case, C, avg, focal=FOCAL, ce=CE, up=(1.0, 1.0)):
    cls, dirs, labels, lw, dt, dw = case
    c64, d64 = cls.double().requires_grad_(True), dirs.double().requires_grad_(True)
    lc, ld = ORA.cls_dir_losses(c64, d64, labels, lw.double(), dt, dw.double(), C, avg, gamma=focal.get('gamma', 2.0), alpha=focal.get('alpha', 0.25),
                                cls_weight=focal.get('loss_weight', 1.0), dir_weight=ce.get('loss_weight', 1.0))
    (up[0] * lc + up[1] * ld).backward()
    dev = torch.device('cuda:0')
    cg, dg = cls.to(dev).requires_grad_(True), dirs.to(dev).requires_grad_(True)
    gc, gd = pkg.extras.anchor_head_cls_dir_loss(focal, ce, cg, dg, labels.to(dev), lw.to(dev), dt.to(dev), dw.to(dev), C, avg)
    (up[0] * gc + up[1] * gd).backward()
    return (lc, ld, c64.grad, d64.grad), (gc, gd, cg.grad, dg.grad)


def check(ref, got):
    lc, ld, rc, rd = ref
    gc, gd, hc, hd = got
    assert abs(gc.item() - lc.item()) <= 1e-5 * abs(lc.item()) + 1e-9
    assert abs(gd.item() - ld.item()) <= 1e-5 * abs(ld.item()) + 1e-9
    for r, h in ((rc, hc), (rd, hd)):
        tol = 1e-5 * r.abs().max().item() + 1e-9
        assert (h.cpu().double() - r).abs().max().item() <= tol


@pytest.mark.parametrize('B,A,C,H,W', [(2, 2, 1, 31, 47), (3, 6, 3, 40, 44), (1, 2, 3, 7, 5), (2, 8, 10, 16, 16), (1, 1, 1, 1, 1)])
def test_matches_oracle(B, A, C, H, W):
    case = make(B, A, C, H, W, seed=B * 100 + A, pos_frac=0.05)
    ref, got = run_both(case, C, avg=37.0)
    check(ref, got)


def test_upstream_gradients_and_weights():
    case = make(2, 6, 3, 24, 20, seed=5, pos_frac=0.04)
    case[3].mul_(torch.rand(case[3].shape, generator=torch.Generator().manual_seed(1)) * 2)       # arbitrary label weights
    case[5].mul_(torch.rand(case[5].shape, generator=torch.Generator().manual_seed(2)) * 2)
    focal = dict(FOCAL, loss_weight=1.7)
    ce = dict(CE, loss_weight=0.35)
    ref, got = run_both(case, 3, avg=11.0, focal=focal, ce=ce, up=(0.3, -2.5))
    check(ref, got)


@pytest.mark.parametrize('gamma,alpha', [(1.0, 0.5), (1.5, 0.25), (0.0, 0.75), (3.0, 0.1)])
def test_other_focal_settings(gamma, alpha):
    case = make(2, 2, 3, 20, 12, seed=9, pos_frac=0.05)
    ref, got = run_both(case, 3, avg=5.0, focal=dict(FOCAL, gamma=gamma, alpha=alpha))
    check(ref, got)


def test_extreme_logits_stay_finite():
    case = make(1, 2, 2, 16, 16, seed=3, pos_frac=0.1, scale=30.0)
    ref, got = run_both(case, 2, avg=3.0)
    assert all(torch.isfinite(t).all() for t in got)
    check(ref, got)


def test_no_positive_anchor():
    cls, dirs, labels, lw, dt, dw = make(2, 2, 3, 10, 10, seed=4)
    labels.fill_(3)
    dw.zero_()
    ref, got = run_both((cls, dirs, labels, lw, dt, dw), 3, avg=1.0)
    assert got[1].item() == 0.0 and got[3].abs().max().item() == 0.0            # `pos_dir_cls_preds.sum()` of nothing (:157-158)
    check(ref, got)


def test_negative_labels_are_not_positives():
    cls, dirs, labels, lw, dt, dw = make(2, 2, 3, 10, 10, seed=6, pos_frac=0.1)
    labels[0, :50] = -1           # not a class, not a positive (:101-103); one_hot would refuse it: weight 0 and compare on the rest
    lw[0, :50] = 0.0
    dw[0, :50] = 1.0              # must not enter the direction term
    dev = torch.device('cuda:0')
    cg, dg = cls.to(dev).requires_grad_(True), dirs.to(dev).requires_grad_(True)
    gc, gd = pkg.extras.anchor_head_cls_dir_loss(FOCAL, CE, cg, dg, labels.to(dev), lw.to(dev), dt.to(dev), dw.to(dev), 3, 9.0)
    (gc + gd).backward()
    lab2 = labels.clone()
    lab2[0, :50] = 3
    c64, d64 = cls.double().requires_grad_(True), dirs.double().requires_grad_(True)
    lc, ld = ORA.cls_dir_losses(c64, d64, lab2, lw.double(), dt, dw.double(), 3, 9.0)
    (lc + ld).backward()
    check((lc, ld, c64.grad, d64.grad), (gc, gd, cg.grad, dg.grad))


def test_without_direction_classifier_and_default_avg():
    cls, dirs, labels, lw, dt, dw = make(3, 2, 3, 9, 11, seed=8, pos_frac=0.05)
    dev = torch.device('cuda:0')
    cg = cls.to(dev).requires_grad_(True)
    gc, gd = pkg.extras.anchor_head_cls_dir_loss(FOCAL, None, cg, None, labels.to(dev), lw.to(dev), None, None, 3)
    assert gd is None
    gc.backward()
    c64 = cls.double().requires_grad_(True)
    lc = ORA.sigmoid_focal_loss(c64.permute(0, 2, 3, 1).reshape(-1, 3), labels.reshape(-1), lw.double().reshape(-1), 2.0, 0.25, 3, 1.0)   # avg = B (:85-86)
    lc.backward()
    assert abs(gc.item() - lc.item()) <= 1e-5 * abs(lc.item())
    assert (cg.grad.cpu().double() - c64.grad).abs().max().item() <= 1e-5 * c64.grad.abs().max().item() + 1e-9


def test_deterministic_and_no_grad():
    case = make(2, 6, 3, 32, 32, seed=12, pos_frac=0.03)
    dev = torch.device('cuda:0')
    args = [t.to(dev) for t in case]
    a = pkg.extras.anchor_head_cls_dir_loss(FOCAL, CE, args[0], args[1], args[2], args[3], args[4], args[5], 3, 21.0)
    b = pkg.extras.anchor_head_cls_dir_loss(FOCAL, CE, args[0], args[1], args[2], args[3], args[4], args[5], 3, 21.0)
    assert a[0].item() == b[0].item() and a[1].item() == b[1].item()
    assert not a[0].requires_grad


def test_argument_checks():
    dev = torch.device('cuda:0')
    cls, dirs, labels, lw, dt, dw = [t.to(dev) for t in make(1, 2, 3, 4, 4, seed=1)]
    with pytest.raises(RuntimeError, match='no CPU path'):
        pkg.extras.anchor_head_cls_dir_loss(FOCAL, CE, cls.cpu(), dirs.cpu(), labels.cpu(), lw.cpu(), dt.cpu(), dw.cpu(), 3, 1.0)
    with pytest.raises(RuntimeError, match='is not'):
        pkg.extras.anchor_head_cls_dir_loss(FOCAL, CE, cls, dirs, labels, lw, dt, dw, 4, 1.0)
    with pytest.raises(RuntimeError, match='entries'):
        pkg.extras.anchor_head_cls_dir_loss(FOCAL, CE, cls, dirs, labels[:, :-1], lw, dt, dw, 3, 1.0)
    with pytest.raises(RuntimeError, match='FocalLoss'):
        pkg.extras.anchor_head_cls_dir_loss(dict(type='GaussianFocalLoss'), CE, cls, dirs, labels, lw, dt, dw, 3, 1.0)
    with pytest.raises(RuntimeError, match='num_total_samples'):
        pkg.extras.anchor_head_cls_dir_loss(FOCAL, CE, cls, dirs, labels, lw, dt, dw, 3, 0)


# ---- GDAnchor3DHead.loss_single end to end (gd_anchor3d_head.py:62-161) -------------------------------------------------
SL1 = dict(type='SmoothL1Loss', beta=1.0 / 9.0, loss_weight=2.0)
TRAIN_CFG = dict(code_weight=[1.0] * 7, decode_weight=[1, 1, .5, 1, 2, 1, 1])


def head_case(seed, B=3, A=6, H=10, W=7, C=3):
    from test_gpu_head_loss import _head_inputs
    anchors, bbox_pred, bbox_targets, bbox_weights, _labels, C = _head_inputs(seed, B=B, A=A, H=H, W=W, C=C)
    cls, dirs, labels, lw, dt, dw = make(B, A, C, H, W, seed=seed + 50, pos_frac=0.08)
    return dict(cls=cls, bbox=bbox_pred, dirs=dirs, labels=labels, lw=lw, bt=bbox_targets, bw=bbox_weights, dt=dt, dw=dw, anchors=anchors), C


def oracle_loss_single(c, C, avg, dtype, gd_cfg):
    from oracle import head_torch
    f = lambda t: t.to(dtype) if t.is_floating_point() else t          # noqa: E731
    cls, bbox, dirs = [f(c[k]).clone().requires_grad_(True) for k in ('cls', 'bbox', 'dirs')]
    lc, ld = ORA.cls_dir_losses(cls, dirs, c['labels'], f(c['lw']), c['dt'], f(c['dw']), C, avg, cls_weight=1.0, dir_weight=0.2)
    lb = head_torch.loss_single_bbox(bbox, f(c['bt']), f(c['bw']), c['labels'], f(c['anchors']), C, avg, gd=dict(gd_cfg),
                                     sl1=dict(beta=SL1['beta'], loss_weight=SL1['loss_weight']), code_weight=TRAIN_CFG['code_weight'],
                                     decode_weight=TRAIN_CFG['decode_weight'], diff_rad_by_sin=True)
    (lc + lb + ld).backward()
    return (lc.item(), lb.item(), ld.item()), (cls.grad, bbox.grad, dirs.grad)


@pytest.mark.parametrize('lt,kw', [('kld3d', dict(fun='log1p', tau=1.0)), ('gwd3d', dict(fun='log1p', tau=0.0))])
def test_loss_single_end_to_end(lt, kw):
    c, C = head_case(2)
    avg = 19.0
    dev = torch.device('cuda:0')
    g = {k: v.to(dev) for k, v in c.items()}
    for k in ('cls', 'bbox', 'dirs'):
        g[k].requires_grad_(True)
    mod = pkg.GDLoss(lt, loss_weight=5.0, **kw)
    out = pkg.extras.gd_anchor_head_loss_single(FOCAL, SL1, CE, mod, TRAIN_CFG, C, g['cls'], g['bbox'], g['dirs'], g['labels'], g['lw'], g['bt'],
                                         g['bw'], g['dt'], g['dw'], g['anchors'], avg)
    (out[0] + out[1] + out[2]).backward()
    gd_cfg = dict(loss_type=lt, loss_weight=5.0, **kw)
    l64, g64 = oracle_loss_single(c, C, avg, torch.float64, gd_cfg)
    l32, g32 = oracle_loss_single(c, C, avg, torch.float32, gd_cfg)
    for o, a, b in zip(out, l64, l32):
        assert abs(o.item() - a) <= (1e-5 + 3 * abs(b - a) / (1 + abs(a))) * (1 + abs(a))
    for k, r64, r32 in zip(('cls', 'bbox', 'dirs'), g64, g32):
        sc = r64.abs().max().item()
        tol = (1e-5 + 3 * (r32.double() - r64).abs().max().item() / (1 + sc)) * (1 + sc)
        assert (g[k].grad.cpu().double() - r64).abs().max().item() <= tol, k


def test_loss_single_replays_as_a_hipgraph():
    """static shapes, no host sync: forward + backward of the whole method captured once, replayed with new values: same bits"""
    c, C = head_case(4)
    dev = torch.device('cuda:0')
    g = {k: v.to(dev) for k, v in c.items()}
    for k in ('cls', 'bbox', 'dirs'):
        g[k].requires_grad_(True)
    mod = pkg.GDLoss('kld3d', fun='log1p', tau=1.0, loss_weight=5.0)
    order = ('cls', 'bbox', 'dirs', 'labels', 'lw', 'bt', 'bw', 'dt', 'dw', 'anchors')

    def fn(cls, bbox, dirs, labels, lw, bt, bw, dt, dw, anchors):
        return pkg.extras.gd_anchor_head_loss_single(FOCAL, SL1, CE, mod, TRAIN_CFG, C, cls, bbox, dirs, labels, lw, bt, bw, dt, dw, anchors, 23.0)
    step = pkg.GraphedStep(fn, tuple(g[k] for k in order))
    c2, _ = head_case(7)
    g2 = {k: v.to(dev) for k, v in c2.items()}
    for k in ('cls', 'bbox', 'dirs'):
        g2[k].requires_grad_(True)
    losses, grads = step(*(g2[k] for k in order))
    losses = [x.clone() for x in losses]
    grads = [None if x is None else x.clone() for x in grads]
    eager = fn(*(g2[k] for k in order))
    (eager[0] + eager[1] + eager[2]).backward()
    for a, b in zip(losses, eager):
        assert torch.equal(a, b.detach())
    for k, gr in zip(order[:3], grads[:3]):
        assert torch.equal(gr, g2[k].grad)
    assert all(x is None for x in grads[3:])


def test_loss_single_without_direction_classifier():
    c, C = head_case(6)
    dev = torch.device('cuda:0')
    g = {k: v.to(dev) for k, v in c.items()}
    mod = pkg.GDLoss('kld3d', fun='log1p', tau=1.0, loss_weight=5.0)
    a = pkg.extras.gd_anchor_head_loss_single(FOCAL, SL1, None, mod, None, C, g['cls'], g['bbox'], None, g['labels'], g['lw'], g['bt'], g['bw'], None, None,
                                       g['anchors'], 5.0, use_direction_classifier=False)
    b = pkg.extras.gd_anchor_head_loss_single(FOCAL, SL1, CE, mod, None, C, g['cls'], g['bbox'], g['dirs'], g['labels'], g['lw'], g['bt'], g['bw'], g['dt'], g['dw'],
                                       g['anchors'], 5.0)
    assert a[2] is None and a[0].item() == b[0].item() and a[1].item() == b[1].item()


def test_unit_gradient_constant_is_recognised_by_address():
    """torch.autograd.backward(losses, [unit_grad] * n): the stored gradient maps are handed over as they are (no scaling launch);
    same values as the ordinary backward"""
    c, C = head_case(9)
    dev = torch.device('cuda:0')
    mod = pkg.GDLoss('kld3d', fun='log1p', tau=1.0, loss_weight=5.0)
    gd_loss = importlib.import_module('mmdet3d-gaussian_amd.gd_loss')
    res = []
    for unit in (False, True):
        g = {k: v.to(dev) for k, v in c.items()}
        for k in ('cls', 'bbox', 'dirs'):
            g[k].requires_grad_(True)
        out = pkg.extras.gd_anchor_head_loss_single(FOCAL, SL1, CE, mod, TRAIN_CFG, C, g['cls'], g['bbox'], g['dirs'], g['labels'], g['lw'], g['bt'],
                                             g['bw'], g['dt'], g['dw'], g['anchors'], 13.0)
        if unit:
            torch.autograd.backward(list(out), [gd_loss.unit_grad(dev)] * 3)
        else:
            torch.autograd.backward(list(out), [torch.ones_like(o) for o in out])
        res.append([g[k].grad.clone() for k in ('cls', 'bbox', 'dirs')])
    for a, b in zip(*res):
        assert torch.equal(a, b)


def test_direction_target_outside_its_two_bins_poisons_the_direction_loss():
    """a dir target outside [0, 2) on a positive anchor makes F.cross_entropy raise in the reference; here — no host
    sync to raise from — the direction loss and that anchor's direction gradients are NaN (loud in the first step) instead of a
    silently finite "bin 1"; the classification term and every other anchor's gradient are untouched."""
    cls, dirs, labels, lw, dt, dw = make(2, 2, 3, 9, 11, seed=5, pos_frac=0.2)
    dev = torch.device('cuda:0')
    args = lambda d: (labels.to(dev), lw.to(dev), d.to(dev), dw.to(dev), 3, 11.0)
    cg, dg = cls.to(dev).requires_grad_(True), dirs.to(dev).requires_grad_(True)
    lc0, ld0 = pkg.extras.anchor_head_cls_dir_loss(FOCAL, CE, cg, dg, *args(dt))
    (lc0 + ld0).backward()
    good_c, good_d = cg.grad.clone(), dg.grad.clone()
    assert torch.isfinite(ld0)
    pos = (labels < 3).nonzero()
    bad = dt.clone()
    b, n = int(pos[0, 0]), int(pos[0, 1])
    for wrong in (2, -1, 7):
        bad[b, n] = wrong
        c2, d2 = cls.to(dev).requires_grad_(True), dirs.to(dev).requires_grad_(True)
        lc, ld = pkg.extras.anchor_head_cls_dir_loss(FOCAL, CE, c2, d2, *args(bad))
        (lc + ld).backward()
        assert torch.isnan(ld) and torch.equal(lc, lc0) and torch.equal(c2.grad, good_c)
        nan = torch.isnan(d2.grad)
        assert int(nan.sum()) == 2 and torch.equal(d2.grad[~nan], good_d[~nan])      # that anchor's two direction logits only
    # a bad target on a NON-positive anchor is never read (the reference gathers the positives first, :143-146)
    neg = (labels >= 3).nonzero()
    bad = dt.clone()
    bad[int(neg[0, 0]), int(neg[0, 1])] = 5
    _, ld = pkg.extras.anchor_head_cls_dir_loss(FOCAL, CE, cls.to(dev), dirs.to(dev), *args(bad))
    assert torch.equal(ld, ld0)
