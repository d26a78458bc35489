"""GPU parity of the anchor heads' target assignment (csrc/anchor_targets.hip) against the CPU torch restatement
oracle/anchor_targets_torch.py of mmdet3d's anchor_target_3d chain (called at gd_anchor3d_head.py:206-214).
Integer / decision outputs (labels, weights, direction bins, counts, which anchors carry targets) bit for bit; of the encoded regression
targets the z and yaw columns (+, -, / only) bit for bit, the others to 2e-6: the three size columns go through log (device logf vs the
CPU's: ulps) and x, y through the anchor's diagonal sqrt(l^2 + w^2), where torch's CPU sqrt is NOT correctly rounded (measured: 0.7 % of
random inputs differ from the IEEE result in the build container, 18 % on the GPU box's host) while the device's is."""
import importlib
import math

import pytest
import torch

pytestmark = pytest.mark.extras   # frozen extras outside SURVEY.md §8: `pytest -m extras` on a GPU box (conftest.py), not part of `-m gpu`

pkg = importlib.import_module('mmdet3d-gaussian_amd')
from oracle import anchor_targets_torch as ORA  # noqa: E402

KITTI_ASSIGNERS = [dict(type='MaxIoUAssigner', iou_calculator=dict(type='BboxOverlapsNearest3D'), pos_iou_thr=0.5, neg_iou_thr=0.35, min_pos_iou=0.35,
                        ignore_iof_thr=-1)] * 2 + \
                  [dict(type='MaxIoUAssigner', iou_calculator=dict(type='BboxOverlapsNearest3D'), pos_iou_thr=0.6, neg_iou_thr=0.45, min_pos_iou=0.45,
                        ignore_iof_thr=-1)]
KITTI_RANGES = [[0.08, -39.60, -0.6, 68.88, 39.44, -0.6], [0.08, -39.60, -0.6, 68.88, 39.44, -0.6], [0.08, -39.60, -1.78, 68.88, 39.44, -1.78]]
KITTI_SIZES = [[0.8, 0.6, 1.73], [1.76, 0.6, 1.73], [3.9, 1.6, 1.56]]


def kitti_anchors(H, W):
    return ORA.range_anchors((H, W), KITTI_RANGES, KITTI_SIZES, [0, 1.57])[0]


def random_gt(n, seed, num_classes=3, x=(0, 69), y=(-39, 39), with_ignored=True):
    g = torch.Generator().manual_seed(seed)
    labels = torch.randint(0, num_classes, (n,), generator=g)
    size = torch.tensor(KITTI_SIZES)[labels] * (0.8 + 0.4 * torch.rand(n, 3, generator=g))
    z = torch.where(labels == 2, torch.tensor(-1.78), torch.tensor(-0.6)) + 0.2 * torch.randn(n, generator=g)
    boxes = torch.stack([torch.rand(n, generator=g) * (x[1] - x[0]) + x[0], torch.rand(n, generator=g) * (y[1] - y[0]) + y[0], z,
                         size[:, 0], size[:, 1], size[:, 2], (torch.rand(n, generator=g) * 2 - 1) * math.pi], dim=-1)
    if with_ignored and n > 3:
        labels[1] = -1                      # a DontCare box: no assigner owns it under assign_per_class
    return boxes, labels


def run_both(anchors, gts, labels, assigners, num_classes=3, **kw):
    ref = ORA.anchor_target_3d(anchors, gts, labels, assigners, num_classes, **kw)
    dev = torch.device('cuda:0')
    got = pkg.extras.anchor_head_get_targets(anchors.to(dev), [g.to(dev) for g in gts], [l.to(dev) for l in labels], assigners, num_classes, **kw)
    return ref, got


def check(ref, got, min_pos=1):
    names = ('labels', 'label_weights', 'bbox_targets', 'bbox_weights', 'dir_targets', 'dir_weights')
    for k in (0, 1, 3, 4, 5):
        assert torch.equal(got[k].cpu(), ref[k]), names[k]
    bt, rt = got[2].cpu(), ref[2]
    assert torch.equal(bt[..., [2, 6]], rt[..., [2, 6]])                            # +, -, / only: the same bits
    assert torch.allclose(bt[..., [0, 1, 3, 4, 5]], rt[..., [0, 1, 3, 4, 5]], rtol=2e-6, atol=2e-7)
    assert got[6] == ref[6] and got[7] == ref[7]
    assert int((ref[0] < 3).sum()) >= min_pos


@pytest.mark.parametrize('H,W,ngt', [(62, 54, [12, 0, 30]), (31, 27, [5]), (124, 108, [25, 40])])
def test_kitti_geometry_per_class_assigners(H, W, ngt):
    anchors = kitti_anchors(H, W)
    pairs = [random_gt(n, seed=10 + i) if n else (torch.zeros(0, 7), torch.zeros(0, dtype=torch.long)) for i, n in enumerate(ngt)]
    ref, got = run_both(anchors, [p[0] for p in pairs], [p[1] for p in pairs], KITTI_ASSIGNERS)
    check(ref, got)


def test_full_kitti_grid_one_sample():
    anchors = kitti_anchors(248, 216)
    b, l = random_gt(20, seed=3)
    ref, got = run_both(anchors, [b], [l], KITTI_ASSIGNERS, dir_offset=0.7854, pos_weight=2.5)
    check(ref, got, min_pos=20)
    assert float(got[1].max()) == 2.5


def test_boxes_on_anchor_centres_give_exact_ties():
    """Boxes copied from anchors (overlap exactly 1 with their anchor, equal overlaps with its mirror neighbours), duplicated boxes
    (every overlap of the pair ties: the later box takes the low-quality matches, the first the argmax), and a box far outside the
    grid (its best overlap is 0 < min_pos_iou: nothing assigned)."""
    anchors = kitti_anchors(40, 36)
    flat = anchors.reshape(-1, 7)
    pick = flat[[2 * 3 * (36 * 7 + 5) + 4, 2 * 3 * (36 * 20 + 11) + 0, 2 * 3 * (36 * 30 + 30) + 3]].clone()     # a car, a pedestrian, a cyclist (rot 1.57)
    boxes = torch.cat([pick, pick[:1], pick[:1] + torch.tensor([0.37, 0.21, 0, 0, 0, 0, 0.3]), torch.tensor([[500., 500., -1, 3.9, 1.6, 1.5, 0.]])])
    labels = torch.tensor([2, 0, 1, 2, 2, 2])
    ref, got = run_both(anchors, [boxes], [labels], KITTI_ASSIGNERS)
    check(ref, got, min_pos=3)


@pytest.mark.parametrize('all_', [True, False])
def test_low_quality_matching_modes(all_):
    cfgs = [dict(c, gt_max_assign_all=all_) for c in KITTI_ASSIGNERS]
    anchors = kitti_anchors(50, 44)
    b, l = random_gt(30, seed=21)
    ref, got = run_both(anchors, [b, b[:7]], [l, l[:7]], cfgs)
    check(ref, got)
    off = [dict(c, match_low_quality=False) for c in KITTI_ASSIGNERS]
    ref2, got2 = run_both(anchors, [b], [l], off)
    check(ref2, got2, min_pos=0)
    assert int((ref2[0] < 3).sum()) < int((ref[0][0] < 3).sum())                  # the low-quality matches were doing something


def test_min_pos_iou_zero_assigns_every_zero_overlap_anchor():
    """mmdet's default min_pos_iou = 0: a box that no anchor overlaps has best overlap 0 >= 0, so EVERY anchor with overlap 0 is
    'its best anchor' and becomes its positive — the reference's literal behaviour, reproduced."""
    cfgs = [dict(type='MaxIoUAssigner', pos_iou_thr=0.6, neg_iou_thr=0.45, min_pos_iou=0.0)] * 3
    anchors = kitti_anchors(12, 10)
    boxes = torch.tensor([[30., 0., -1.7, 3.9, 1.6, 1.5, 0.2], [900., 900., -1.7, 3.9, 1.6, 1.5, 0.]])
    ref, got = run_both(anchors, [boxes], [torch.tensor([2, 2])], cfgs)
    check(ref, got)
    assert int((ref[0] == 2).sum()) > 200


def test_single_assigner_and_list_without_per_class():
    anchors = kitti_anchors(40, 36)
    b, l = random_gt(24, seed=5, with_ignored=False)
    one = dict(type='MaxIoUAssigner', iou_calculator=dict(type='BboxOverlapsNearest3D'), pos_iou_thr=0.55, neg_iou_thr=0.4, min_pos_iou=0.4)
    ref, got = run_both(anchors, [b, b[:3]], [l, l[:3]], one)
    check(ref, got)
    ref, got = run_both(anchors, [b, b[:3]], [l, l[:3]], KITTI_ASSIGNERS, assign_per_class=False)
    check(ref, got)


def test_no_boxes_at_all_and_padded_form():
    anchors = kitti_anchors(20, 18)
    empty = (torch.zeros(0, 7), torch.zeros(0, dtype=torch.long))
    ref, got = run_both(anchors, [empty[0], empty[0]], [empty[1], empty[1]], KITTI_ASSIGNERS)
    check(ref, got, min_pos=0)
    assert got[6] == 2 and got[7] == 2 * anchors.numel() // 7                   # max(0, 1) per sample; every anchor a negative
    dev = torch.device('cuda:0')
    b, l = random_gt(9, seed=2)
    res = pkg.extras.anchor_head_get_targets(anchors.to(dev), [b.to(dev), empty[0].to(dev)], [l.to(dev), empty[1].to(dev)], KITTI_ASSIGNERS, 3, padded=True)
    ref = ORA.anchor_target_3d(anchors, [b, empty[0]], [l, empty[1]], KITTI_ASSIGNERS, 3)
    counts = res[6].cpu()
    assert counts.dtype == torch.int32 and counts.shape == (2, 2)
    assert int(counts[0, 0]) == int((ref[0][0] < 3).sum()) and int(counts[1, 0]) == 0 and int(counts[1, 1]) == anchors.numel() // 7
    assert torch.equal(res[0].cpu(), ref[0])


def test_the_maximum_number_of_boxes_in_a_sample():
    """1024 boxes in one sample (the LDS stage's capacity; the tile shrinks to make room), next to a sample with 3"""
    anchors = kitti_anchors(24, 20)
    b, l = random_gt(1024, seed=31, x=(0, 8), y=(-39, -30))
    b2, l2 = random_gt(3, seed=32, x=(0, 8), y=(-39, -30), with_ignored=False)
    ref, got = run_both(anchors, [b, b2], [l, l2], KITTI_ASSIGNERS)
    check(ref, got)
    ref, got = run_both(anchors, [b, b2], [l, l2], KITTI_ASSIGNERS, assign_per_class=False)
    check(ref, got)


def test_deterministic_and_argument_checks():
    anchors = kitti_anchors(30, 30)
    dev = torch.device('cuda:0')
    b, l = random_gt(40, seed=8)
    a = pkg.extras.anchor_head_get_targets(anchors.to(dev), [b.to(dev)] * 3, [l.to(dev)] * 3, KITTI_ASSIGNERS, 3)
    c = pkg.extras.anchor_head_get_targets(anchors.to(dev), [b.to(dev)] * 3, [l.to(dev)] * 3, KITTI_ASSIGNERS, 3)
    assert all(torch.equal(x, y) for x, y in zip(a[:6], c[:6])) and a[6:] == c[6:]
    assert torch.equal(a[0][0], a[0][2]) and torch.equal(a[2][0], a[2][1])
    with pytest.raises(RuntimeError, match='assigners for'):
        pkg.extras.anchor_head_get_targets(anchors.to(dev), [b.to(dev)], [l.to(dev)], KITTI_ASSIGNERS[:2], 3)
    with pytest.raises(RuntimeError, match='MaxIoUAssigner'):
        pkg.extras.anchor_head_get_targets(anchors.to(dev), [b.to(dev)], [l.to(dev)], dict(type='ATSSAssigner'), 3)
    with pytest.raises(RuntimeError, match='one label each'):
        pkg.extras.anchor_head_get_targets(anchors.to(dev), [b.to(dev)], [l[:-1].to(dev)], KITTI_ASSIGNERS, 3)
    with pytest.raises(RuntimeError, match='boxes in a sample'):
        pkg.extras.anchor_head_get_targets(anchors.to(dev), [b.repeat(30, 1).to(dev)], [l.repeat(30).to(dev)], KITTI_ASSIGNERS, 3)


# ---- GDAnchor3DHead.loss end to end (gd_anchor3d_head.py:167-240) ---------------------------------------------------------
FOCAL = dict(type='FocalLoss', use_sigmoid=True, gamma=2.0, alpha=0.25, loss_weight=1.0)
CE = dict(type='CrossEntropyLoss', use_sigmoid=False, loss_weight=0.2)
SL1 = dict(type='SmoothL1Loss', beta=1.0 / 9.0, loss_weight=2.0)
TRAIN_CFG = dict(assigner=KITTI_ASSIGNERS, allowed_border=0, pos_weight=-1, debug=False, code_weight=[1.0] * 7, decode_weight=[1.0] * 7)


def head_outputs(B, H, W, seed, A=6, C=3):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(B, A * C, H, W, generator=g) * 2 - 3, torch.randn(B, A * 7, H, W, generator=g) * 0.15, torch.randn(B, A * 2, H, W, generator=g))


def oracle_loss(anchors, outs, gts, labels, dtype, gd_cfg, dir_offset=0.0):
    from oracle import anchor_cls_torch, head_torch
    tg = ORA.anchor_target_3d(anchors, gts, labels, KITTI_ASSIGNERS, 3, dir_offset=dir_offset)
    avg = float(tg[6])
    f = lambda t: t.to(dtype)          # noqa: E731
    cls, bbox, dirs = [f(o).clone().requires_grad_(True) for o in outs]
    lc, ld = anchor_cls_torch.cls_dir_losses(cls, dirs, tg[0], f(tg[1]), tg[4], f(tg[5]), 3, avg, cls_weight=1.0, dir_weight=0.2)
    lb = head_torch.loss_single_bbox(bbox, f(tg[2]), f(tg[3]), tg[0], f(anchors.reshape(-1, 7)), 3, avg, gd=dict(gd_cfg),
                                     sl1=dict(beta=SL1['beta'], loss_weight=SL1['loss_weight']), code_weight=TRAIN_CFG['code_weight'],
                                     decode_weight=TRAIN_CFG['decode_weight'], diff_rad_by_sin=True)
    (lc + lb + ld).backward()
    return (lc.item(), lb.item(), ld.item()), (cls.grad, bbox.grad, dirs.grad)


@pytest.mark.parametrize('static', [False, True])
def test_head_loss_end_to_end(static):
    H, W = 62, 54
    anchors = kitti_anchors(H, W)
    pairs = [random_gt(n, seed=40 + i) for i, n in enumerate((14, 9))]
    gts, labels = [p[0] for p in pairs], [p[1] for p in pairs]
    outs = head_outputs(2, H, W, seed=1)
    dev = torch.device('cuda:0')
    g = [o.to(dev).requires_grad_(True) for o in outs]
    mod = pkg.GDLoss('kld3d', fun='log1p', tau=1.0, loss_weight=5.0)
    res = pkg.extras.gd_anchor_head_loss(FOCAL, SL1, CE, mod, TRAIN_CFG, 3, anchors.to(dev), [g[0]], [g[1]], [g[2]], [b.to(dev) for b in gts],
                                  [l.to(dev) for l in labels], static=static)
    assert sorted(res) == ['loss_bbox', 'loss_cls', 'loss_dir'] and all(len(v) == 1 for v in res.values())
    (res['loss_cls'][0] + res['loss_bbox'][0] + res['loss_dir'][0]).backward()
    gd_cfg = dict(loss_type='kld3d', fun='log1p', tau=1.0, loss_weight=5.0)
    l64, g64 = oracle_loss(anchors, outs, gts, labels, torch.float64, gd_cfg)
    l32, g32 = oracle_loss(anchors, outs, gts, labels, torch.float32, gd_cfg)
    for key, a, b in zip(('loss_cls', 'loss_bbox', 'loss_dir'), l64, l32):
        assert abs(res[key][0].item() - a) <= (1e-5 + 3 * abs(b - a) / (1 + abs(a))) * (1 + abs(a)), key
    for got, r64, r32 in zip(g, g64, g32):
        sc = r64.abs().max().item()
        tol = (1e-5 + 3 * (r32.double() - r64).abs().max().item() / (1 + sc)) * (1 + sc)
        assert (got.grad.cpu().double() - r64).abs().max().item() <= tol


def test_head_loss_static_form_replays_as_a_hipgraph():
    """ground truth padded to a fixed number of rows (label -1), no read-back: targets + three losses + backward captured once and
    replayed on another batch: the same bits as the eager static call"""
    H, W, G = 40, 36, 24
    anchors = kitti_anchors(H, W)
    dev = torch.device('cuda:0')
    mod = pkg.GDLoss('gwd3d', fun='log1p', tau=0.0, loss_weight=5.0)

    def padded_batch(seed, counts):
        boxes, labels = [], []
        for i, n in enumerate(counts):
            b, l = random_gt(G, seed=seed + i, with_ignored=False)
            l[n:] = -1
            boxes.append(b)
            labels.append(l)
        return torch.stack(boxes).to(dev), torch.stack(labels).to(dev)
    an = anchors.to(dev)

    def fn(cls, bbox, dirs, gt, gl):
        r = pkg.extras.gd_anchor_head_loss(FOCAL, SL1, CE, mod, TRAIN_CFG, 3, an, cls, bbox, dirs, list(gt.unbind(0)), list(gl.unbind(0)), static=True)
        return r['loss_cls'][0], r['loss_bbox'][0], r['loss_dir'][0]
    o1 = [o.to(dev).requires_grad_(True) for o in head_outputs(2, H, W, seed=3)]
    gt1, gl1 = padded_batch(60, (10, 4))
    step = pkg.GraphedStep(fn, (o1[0], o1[1], o1[2], gt1, gl1))
    o2 = [o.to(dev).requires_grad_(True) for o in head_outputs(2, H, W, seed=4)]
    gt2, gl2 = padded_batch(70, (0, 20))
    losses, grads = step(o2[0], o2[1], o2[2], gt2, gl2)
    losses = [x.clone() for x in losses]
    grads = [None if x is None else x.clone() for x in grads]
    eager = fn(o2[0], o2[1], o2[2], gt2, gl2)
    (eager[0] + eager[1] + eager[2]).backward()
    for a, b in zip(losses, eager):
        assert torch.equal(a, b.detach())
    for gr, t in zip(grads[:3], o2):
        assert torch.equal(gr, t.grad)
    assert eager[1].item() > 0


@pytest.mark.parametrize('seed', range(12))
def test_random_configurations(seed):
    """random grids (1-4 sizes, 1-3 rotations), thresholds (min_pos_iou 0 now and then), assigner modes, box counts up to 150 per sample,
    boxes partly outside the grid, partly copied from anchors: every decision equal to the restatement"""
    rng = torch.Generator().manual_seed(1000 + seed)
    ri = lambda lo, hi: int(torch.randint(lo, hi + 1, (1,), generator=rng))          # noqa: E731
    rf = lambda lo, hi: float(torch.rand(1, generator=rng)) * (hi - lo) + lo          # noqa: E731
    S, R = ri(1, 4), ri(1, 3)
    H, W = ri(5, 60), ri(5, 70)
    sizes = [[rf(0.5, 4.5), rf(0.5, 2.0), rf(1.0, 2.0)] for _ in range(S)]
    ranges = [[0.0, -20.0, -1.0, 0.4 * W, 0.4 * H - 20.0, -1.0]] * S
    rots = [0.0, 1.57, 0.78][:R]
    anchors = ORA.range_anchors((H, W), ranges, sizes, rots)[0]
    mode = seed % 3                                                              # 0: per class, 1: list, all boxes, 2: one assigner
    def one_cfg():
        pos = rf(0.3, 0.7)
        neg = rf(0.1, pos)
        return dict(type='MaxIoUAssigner', pos_iou_thr=pos, neg_iou_thr=neg, min_pos_iou=0.0 if ri(0, 3) == 0 else rf(0.05, neg),
                    gt_max_assign_all=seed % 4 != 3, match_low_quality=True)
    cfgs = [one_cfg() for _ in range(S)]
    for c in cfgs:
        c['gt_max_assign_all'] = cfgs[0]['gt_max_assign_all']
    assigner = cfgs[0] if mode == 2 else cfgs
    B = ri(1, 3)
    gts, labels = [], []
    flat = anchors.reshape(-1, 7)
    for b in range(B):
        n = ri(0, 150) if b else ri(1, 150)
        lab = torch.randint(-1, S + 1, (n,), generator=rng)
        sz = torch.tensor(sizes)[lab.clamp(0, S - 1)] * (0.7 + 0.6 * torch.rand(n, 3, generator=rng))
        box = torch.cat([torch.rand(n, 1, generator=rng) * 0.5 * W - 0.05 * W, torch.rand(n, 1, generator=rng) * 0.5 * H - 20.0 - 0.05 * H,
                         torch.full((n, 1), -1.0), sz, (torch.rand(n, 1, generator=rng) * 2 - 1) * math.pi], dim=-1)
        k = min(n, 5)
        if k:
            box[:k] = flat[torch.randint(0, flat.shape[0], (k,), generator=rng)]          # exact ties with an anchor and its mirror images
        gts.append(box)
        labels.append(lab)
    kw = dict(assign_per_class=(mode == 0), dir_offset=rf(-1, 1), pos_weight=-1 if seed % 2 else 1.5)
    ref = ORA.anchor_target_3d(anchors, gts, labels, assigner, S, **kw)
    dev = torch.device('cuda:0')
    got = pkg.extras.anchor_head_get_targets(anchors.to(dev), [g.to(dev) for g in gts], [l.to(dev) for l in labels], assigner, S, **kw)
    for k in (0, 1, 3, 4, 5):
        assert torch.equal(got[k].cpu(), ref[k]), k
    assert torch.equal(got[2].cpu()[..., [2, 6]], ref[2][..., [2, 6]])
    assert torch.allclose(got[2].cpu()[..., [0, 1, 3, 4, 5]], ref[2][..., [0, 1, 3, 4, 5]], rtol=2e-6, atol=2e-7)
    assert got[6] == ref[6] and got[7] == ref[7]


def test_static_form_equals_the_eager_one_bit_for_bit():
    """the device-resident normaliser (sum_b max(positives_b, 1), divided inside the loss kernels in double and rounded once) gives the
    bits of the eager form, which divides on the host; losses and gradients"""
    H, W = 40, 36
    anchors = kitti_anchors(H, W)
    pairs = [random_gt(n, seed=80 + i) for i, n in enumerate((11, 0, 6))]
    pairs[1] = (torch.zeros(0, 7), torch.zeros(0, dtype=torch.long))
    dev = torch.device('cuda:0')
    mod = pkg.GDLoss('kld3d', fun='log1p', tau=1.0, loss_weight=5.0)
    outs = head_outputs(3, H, W, seed=5)
    res = []
    for static in (False, True):
        g = [o.to(dev).requires_grad_(True) for o in outs]
        r = pkg.extras.gd_anchor_head_loss(FOCAL, SL1, CE, mod, TRAIN_CFG, 3, anchors.to(dev), g[0], g[1], g[2], [p[0].to(dev) for p in pairs],
                                    [p[1].to(dev) for p in pairs], static=static)
        (r['loss_cls'][0] + r['loss_bbox'][0] + r['loss_dir'][0]).backward()
        res.append(([r[k][0].detach().clone() for k in ('loss_cls', 'loss_bbox', 'loss_dir')], [t.grad.clone() for t in g]))
    for a, b in zip(res[0][0] + res[0][1], res[1][0] + res[1][1]):
        assert torch.equal(a, b)
    with pytest.raises(RuntimeError, match='one value'):
        pkg.extras.anchor_head_cls_dir_loss(FOCAL, CE, g[0], g[2], torch.zeros(3, H * W * 6, dtype=torch.long, device=dev), torch.ones(3, H * W * 6, device=dev),
                                     torch.zeros(3, H * W * 6, dtype=torch.long, device=dev), torch.ones(3, H * W * 6, device=dev), 3, torch.ones(2, device=dev))


def test_car_only_config_flat_anchors_single_assigner():
    """configs/kitti/hv_pointpillars_secfpn_6x8_160e_kitti-3d-car.py: one size, reshape_out=True (anchors arrive as a flat (N, 7) list),
    ONE MaxIoUAssigner (0.6 / 0.45 / 0.45), one class"""
    H, W = 62, 54
    grid = ORA.range_anchors((H, W), [[0, -39.68, -1.78, 69.12, 39.68, -1.78]], [[1.6, 3.9, 1.56]], [0, 1.57])[0]      # (H, W, 1, 2, 7)
    flat = grid.reshape(-1, 7)
    one = dict(type='MaxIoUAssigner', iou_calculator=dict(type='BboxOverlapsNearest3D'), pos_iou_thr=0.6, neg_iou_thr=0.45, min_pos_iou=0.45,
               ignore_iof_thr=-1)
    g = torch.Generator().manual_seed(3)
    boxes = [torch.cat([torch.rand(n, 1, generator=g) * 69, torch.rand(n, 1, generator=g) * 78 - 39, torch.full((n, 1), -1.7),
                        torch.tensor([[1.6, 3.9, 1.56]]) * (0.8 + 0.4 * torch.rand(n, 3, generator=g)), (torch.rand(n, 1, generator=g) * 2 - 1) * math.pi], -1)
             for n in (15, 0, 7)]
    labels = [torch.zeros(b.shape[0], dtype=torch.long) for b in boxes]
    ref = ORA.anchor_target_3d(grid, boxes, labels, one, 1)
    dev = torch.device('cuda:0')
    for an in (flat, grid):
        got = pkg.extras.anchor_head_get_targets(an.to(dev), [b.to(dev) for b in boxes], [l.to(dev) for l in labels], one, 1)
        for k in (0, 1, 3, 4, 5):
            assert torch.equal(got[k].cpu(), ref[k])
        assert torch.allclose(got[2].cpu(), ref[2], rtol=2e-6, atol=2e-7) and got[6] == ref[6] and got[7] == ref[7]
    assert int((ref[0] < 1).sum()) >= 10          # a rotated box whose best nearest-BEV overlap stays below min_pos_iou gets no anchor
