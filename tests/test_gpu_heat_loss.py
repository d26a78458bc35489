"""Heat-map loss of the CenterPoint heads on the MI355X (csrc/heat_focal.hip) against the torch restatement of
gd_centerpoint_head.py:403-411 with mmdet3d's clip_sigmoid and mmdet's GaussianFocalLoss (oracle/heat_focal_torch.py, evaluated in
fp64 with autograd; third-party formulas: unpinned).  Tolerance: 1e-5 relative on the losses, 1e-5 of the largest gradient entry
(plus 1e-6 relative) on the gradients — fp32 evaluation of a sum of ~10^5 terms against fp64."""
import pytest
import torch

import mmdet3d_gaussian_amd as amd
from oracle import heat_focal_torch as hf

pytestmark = pytest.mark.extras   # frozen extras outside SURVEY.md §8: `pytest -m extras` on a GPU box (conftest.py), not part of `-m gpu`


def targets_like(g, shape, peaks):
    """Gaussian-looking targets: values in [0, 1), `peaks` cells exactly 1"""
    t = torch.rand(shape, generator=g) ** 6
    flat = t.view(-1)
    flat[torch.randperm(flat.numel(), generator=g)[:peaks]] = 1.0
    return t


def compare(logits, targets, cfg, upstream=None):
    T = len(logits)
    xs = [x.cuda().requires_grad_(True) for x in logits]
    losses, num_pos = amd.extras.center_head_heatmap_loss(cfg, xs, [t.cuda() for t in targets])
    assert losses.shape == (T,) and num_pos.shape == (T,) and not num_pos.requires_grad
    up = torch.ones(T) if upstream is None else upstream
    (losses * up.cuda()).sum().backward()
    for t in range(T):
        x64 = logits[t].double().requires_grad_(True)
        want, npos = hf.heatmap_loss(x64, targets[t].double(), cfg.get('alpha', 2.0), cfg.get('gamma', 4.0), cfg.get('loss_weight', 1.0))
        (want * float(up[t])).backward()
        assert float(num_pos[t]) == npos
        assert abs(losses[t].item() - want.item()) <= 1e-5 * abs(want.item()) + 1e-7, (t, losses[t].item(), want.item())
        g, gw = xs[t].grad.cpu().double(), x64.grad
        assert g.shape == gw.shape
        err = (g - gw).abs()
        # clip_sigmoid's corners sit at |x| = log(9999) = 9.2102404, where sigmoid is 0.9999 and one fp32 ulp of it (6e-8) spans
        # 6e-4 of x: within 2e-3 of a corner an fp32 evaluation (the reference's own included) may land on the other side of
        # the clamp than fp64, and the gradient is the full value on one side and 0 on the other — not compared there
        edge = (logits[t].double().abs() - 9.210240366975849).abs() < 2e-3
        assert int(edge.sum()) <= 40
        ok = (err <= 1e-5 * gw.abs().max() + 1e-6 * gw.abs()) | edge
        assert bool(ok.all()), (t, float(err[~edge].max()), float(gw.abs().max()))
    return losses


def test_heatmap_loss_nuscenes_geometry():
    g = torch.Generator().manual_seed(41)
    cfg = dict(type='GaussianFocalLoss', reduction='mean', loss_weight=1.0)
    shapes = [(2, c, 128, 128) for c in (1, 2, 2, 1, 2, 2)]
    logits = [torch.randn(s, generator=g) * 2 - 3 for s in shapes]
    targets = [targets_like(g, s, 30 + 7 * i) for i, s in enumerate(shapes)]
    compare(logits, targets, cfg, upstream=torch.tensor([1.0, 0.5, 2.0, 1.0, 0.0, 3.0]))


def test_heatmap_loss_on_real_targets_and_saturated_logits():
    """targets from center_head_get_targets; logits beyond +-9.21 sit in clip_sigmoid's flat part: zero gradient there"""
    from test_gpu_center_targets import NUS, TASKS, scene
    g = torch.Generator().manual_seed(42)
    data = [scene(g, n) for n in (80, 120)]
    hm, _, _ = amd.extras.center_head_get_targets([d[0].cuda() for d in data], [d[1].cuda() for d in data], TASKS, NUS)
    logits = [torch.randn(h.shape, generator=g) * 6 for h in hm]
    losses = compare(logits, [h.cpu() for h in hm], dict(type='GaussianFocalLoss', reduction='mean', loss_weight=1.0))
    assert bool(torch.isfinite(losses).all())
    x = logits[0].cuda().requires_grad_(True)
    l, _ = amd.extras.center_head_heatmap_loss(dict(type='GaussianFocalLoss'), [x], [hm[0]])
    l.sum().backward()
    assert float(x.grad[logits[0].cuda().abs() > 9.3].abs().max()) == 0.0


def test_heatmap_loss_other_exponents_odd_sizes_no_positives_half_inputs():
    g = torch.Generator().manual_seed(43)
    cfg = dict(type='GaussianFocalLoss', reduction='mean', loss_weight=2.5, alpha=1.5, gamma=3.0)
    shapes = [(1, 1, 37, 53), (3, 2, 17, 19), (1, 1, 1, 5)]
    logits = [torch.randn(s, generator=g) * 2 - 1 for s in shapes]
    targets = [targets_like(g, shapes[0], 5), targets_like(g, shapes[1], 0), targets_like(g, shapes[2], 1)]
    compare(logits, targets, cfg)                                   # task 1 has no positive cell: avg_factor max(0, 1) = 1
    half = [x.half().float() for x in logits]
    xs = [x.cuda().half().requires_grad_(True) for x in half]
    l, _ = amd.extras.center_head_heatmap_loss(cfg, xs, [t.cuda() for t in targets])
    l.sum().backward()
    ref = [x.cuda().requires_grad_(True) for x in half]
    l2, _ = amd.extras.center_head_heatmap_loss(cfg, ref, [t.cuda() for t in targets])
    l2.sum().backward()
    assert torch.equal(l, l2) and all(a.grad.dtype == torch.float16 and torch.equal(a.grad, b.grad.half()) for a, b in zip(xs, ref))


def test_heatmap_loss_module_object_and_errors():
    class GaussianFocalLoss:               # what mmdet's module exposes
        alpha, gamma, reduction, loss_weight = 2.0, 4.0, 'mean', 1.0
    g = torch.Generator().manual_seed(44)
    x, t = torch.randn(1, 2, 16, 16, generator=g), targets_like(g, (1, 2, 16, 16), 3)
    a, _ = amd.extras.center_head_heatmap_loss(GaussianFocalLoss(), [x.cuda()], [t.cuda()])
    b, _ = amd.extras.center_head_heatmap_loss(dict(type='GaussianFocalLoss'), [x.cuda()], [t.cuda()])
    assert torch.equal(a, b)
    with pytest.raises(RuntimeError, match='no CPU path'):
        amd.extras.center_head_heatmap_loss(dict(type='GaussianFocalLoss'), [x], [t])
    with pytest.raises(RuntimeError, match='GaussianFocalLoss'):
        amd.extras.center_head_heatmap_loss(dict(type='FocalLoss'), [x.cuda()], [t.cuda()])
    with pytest.raises(RuntimeError, match='vs targets'):
        amd.extras.center_head_heatmap_loss(dict(type='GaussianFocalLoss'), [x.cuda()], [t.cuda()[:, :1]])
    xs = x.cuda().requires_grad_(True)
    l, _ = amd.extras.center_head_heatmap_loss(dict(type='GaussianFocalLoss'), [xs], [t.cuda()])
    # a second backward under retain_graph RECOMPUTES the gradient maps (the first scaled them in place and handed them over), as
    # the loss / anchor-head / centre-head nodes do; a released graph raises torch's own message
    l.sum().backward(retain_graph=True)
    g1 = xs.grad.clone()
    xs.grad = None
    (2.0 * l.sum()).backward(retain_graph=True)
    assert torch.allclose(xs.grad, 2.0 * g1, rtol=1e-6, atol=0) and g1.abs().sum() > 0
    xs.grad = None
    l.sum().backward()
    assert torch.equal(xs.grad, g1)


def test_full_head_loss_is_the_sum_of_its_pieces():
    """center_gd_head_loss (CenterGDHead.loss :390-441) = get_targets -> heat-map loss + regression losses with num_pos as their
    avg_factor: the dict it returns equals the pieces called by hand, and backward reaches every head map"""
    from test_gpu_center_targets import NUS, TASKS, scene
    g = torch.Generator().manual_seed(45)
    data = [scene(g, n, spread=50.0) for n in (70, 90)]
    boxes, labels = [d[0].cuda() for d in data], [d[1].cuda() for d in data]
    cfg = dict(NUS, code_weights=[1.0, 1.0, 0.2, 0.2])
    coder = amd.CenterPointBBoxYawCoder(pc_range=[-51.2, -51.2], out_size_factor=4, voxel_size=[0.2, 0.2], norm_bbox=True)
    gd = amd.GDLoss('gwd3d', fun='log1p', tau=0.0, loss_weight=5.0)
    l1 = dict(type='L1Loss', reduction='mean', loss_weight=0.25)
    cls = dict(type='GaussianFocalLoss', reduction='mean')
    chans = (('heatmap', None), ('reg', 2), ('height', 1), ('dim', 3), ('yaw', 1), ('dir', 2), ('vel', 2))

    def maps():
        gg = torch.Generator().manual_seed(46)
        return [{k: (torch.randn(2, c if c else len(names), 128, 128, generator=gg) * 0.5 - (2.0 if c is None else 0.0)).cuda().requires_grad_(True)
                 for k, c in chans} for names in TASKS]
    pds = maps()
    out = amd.extras.center_gd_head_loss(cls, l1, gd, coder, TASKS, cfg, boxes, labels, tuple([p] for p in pds))
    assert sorted(out) == sorted(f'task{t}.{k}' for t in range(6) for k in ('loss_heatmap', 'loss_l1', 'loss_gd'))
    sum(out.values()).backward()
    ref = maps()
    hm, an, pi = amd.extras.center_head_get_targets(boxes, labels, TASKS, cfg)
    hl, npos = amd.extras.center_head_heatmap_loss(cls, [p['heatmap'] for p in ref], hm)
    reg = amd.center_head_losses(gd, l1, coder, ref, pi, an, npos.tolist(), cfg['code_weights'])
    (hl.sum() + sum(a + b for a, b in reg)).backward()
    for t in range(6):
        assert torch.equal(out[f'task{t}.loss_heatmap'], hl[t]) and torch.equal(out[f'task{t}.loss_l1'], reg[t][0])
        assert torch.equal(out[f'task{t}.loss_gd'], reg[t][1])
        for k, _ in chans:
            assert pds[t][k].grad is not None and torch.equal(pds[t][k].grad, ref[t][k].grad), (t, k)


def test_static_head_loss_equals_the_dynamic_one_and_replays_as_a_hipgraph():
    """center_gd_head_loss(static=True): row offsets and num_pos never leave the device — same losses and gradients as the form
    with two read-backs, bit for bit; with ground truth padded to a fixed row count (label -1) the whole method is captured by
    GraphedStep and replayed on another batch"""
    from test_gpu_center_targets import NUS, TASKS, scene
    cfg = dict(NUS, code_weights=[1.0, 1.0, 0.2, 0.2])
    coder = amd.CenterPointBBoxYawCoder(pc_range=[-51.2, -51.2], out_size_factor=4, voxel_size=[0.2, 0.2], norm_bbox=True)
    gd = amd.GDLoss('gwd3d', fun='log1p', tau=0.0, loss_weight=5.0)
    l1 = dict(type='L1Loss', reduction='mean', loss_weight=0.25)
    cls = dict(type='GaussianFocalLoss', reduction='mean')
    names = ('heatmap', 'reg', 'height', 'dim', 'yaw', 'dir', 'vel')
    width = dict(reg=2, height=1, dim=3, yaw=1, dir=2, vel=2)
    B, ROWS = 2, 128

    def batch(seed):
        g = torch.Generator().manual_seed(seed)
        boxes, labels = [], []
        for n in (70, 100):
            b, l = scene(g, n, spread=50.0)
            pad_b = torch.zeros(ROWS, 9)
            pad_l = torch.full((ROWS,), -1, dtype=torch.int64)
            pad_b[:n], pad_l[:n] = b, l
            boxes.append(pad_b.cuda())
            labels.append(pad_l.cuda())
        maps = [(torch.randn(B, len(TASKS[t]) if k == 'heatmap' else width[k], 128, 128, generator=g) * 0.5 - (2.0 if k == 'heatmap' else 0.0)).cuda()
                for t in range(len(TASKS)) for k in names]
        return boxes, labels, maps

    def loss_fn(static):
        def fn(b0, b1, l0, l1_, *flat):
            pds = [dict(zip(names, flat[7 * t:7 * t + 7])) for t in range(len(TASKS))]
            out = amd.extras.center_gd_head_loss(cls, l1, gd, coder, TASKS, cfg, [b0, b1], [l0, l1_], pds, static=static)
            return [out[k] for k in sorted(out)]
        return fn
    boxes, labels, maps = batch(50)
    res = {}
    for static in (False, True):
        leaves = [m.clone().requires_grad_(True) for m in maps]
        out = loss_fn(static)(boxes[0], boxes[1], labels[0], labels[1], *leaves)
        sum(out).backward()
        res[static] = ([o.detach().clone() for o in out], [x.grad.clone() for x in leaves])
    assert all(torch.equal(a, b) for a, b in zip(res[False][0], res[True][0]))
    assert all(torch.equal(a, b) for a, b in zip(res[False][1], res[True][1]))
    assert all(float(o) != 0.0 for o in res[True][0])
    step = amd.GraphedStep(loss_fn(True), [boxes[0], boxes[1], labels[0], labels[1]] + [m.clone().requires_grad_(True) for m in maps])
    for seed in (51, 50):
        b2, l2, m2 = batch(seed)
        losses, grads = step(b2[0], b2[1], l2[0], l2[1], *m2)
        got_l, got_g = [x.clone() for x in losses], [g.clone() for g in grads[4:]]
        leaves = [m.clone().requires_grad_(True) for m in m2]
        want = loss_fn(False)(b2[0], b2[1], l2[0], l2[1], *leaves)
        sum(want).backward()
        assert all(torch.equal(a, b.detach()) for a, b in zip(got_l, want))
        assert all(torch.equal(a, x.grad) for a, x in zip(got_g, leaves))
