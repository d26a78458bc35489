"""The unpinned anchor-head restatements (oracle/anchor_targets_torch.py, oracle/anchor_cls_torch.py are mmdet / mmdet3d code restated from
the published text: the reference tree holds neither the code nor a fixture) checked on the CPU against hand-computed cases, so that a
slip in the restatement itself cannot hide behind GPU == restatement.  CPU only."""
import math

import torch

from oracle import anchor_targets_torch as ORA


def test_nearest_bev_snaps_the_yaw_to_the_nearer_axis():
    boxes = torch.tensor([[10., 5., 0., 4., 2., 1.5, 0.0],            # along x: 4 x 2
                          [10., 5., 0., 4., 2., 1.5, 0.7],            # 0.7 < pi/4: still 4 x 2
                          [10., 5., 0., 4., 2., 1.5, 0.8],            # 0.8 > pi/4: 2 x 4
                          [10., 5., 0., 4., 2., 1.5, math.pi - 0.1],  # limit_period folds it to -0.1: 4 x 2
                          [10., 5., 0., 4., 2., 1.5, -1.6]])          # |.| > pi/4: 2 x 4
    bev = ORA.nearest_bev(boxes)
    want = torch.tensor([[8., 4., 12., 6.], [8., 4., 12., 6.], [9., 3., 11., 7.], [8., 4., 12., 6.], [9., 3., 11., 7.]])
    assert torch.equal(bev, want)


def test_bbox_overlaps_known_values():
    a = torch.tensor([[0., 0., 2., 2.], [0., 0., 2., 2.], [5., 5., 6., 6.]])
    b = torch.tensor([[1., 1., 3., 3.], [0., 0., 2., 2.]])
    iou = ORA.bbox_overlaps(a, b)
    assert abs(iou[0, 0].item() - 1.0 / 7.0) < 1e-7 and iou[0, 1].item() == 1.0 and iou[2, 0].item() == 0.0 and iou[2, 1].item() == 0.0


def test_max_iou_assigner_rules_in_order():
    """5 anchors, 2 boxes; thresholds pos 0.6, neg 0.3, min_pos 0.2.
       anchor 0: best 0.7 with box 0 -> positive of box 0;  anchor 1: best 0.1 -> negative;  anchor 2: best 0.45 -> ignored (-1), but it
       is box 1's best anchor (0.45 >= min_pos) -> positive of box 1;  anchor 3: ties box 1's best (0.45) -> with gt_max_assign_all also
       box 1's, without it stays ignored;  anchor 4: 0.65 with box 0 and 0.45 with box 1 -> first the argmax rule gives box 0, then box
       1's tie rule (gt_max_assign_all) overwrites it."""
    ov = torch.tensor([[0.7, 0.1, 0.05, 0.0, 0.65],
                       [0.2, 0.0, 0.45, 0.45, 0.45]])
    labels = torch.tensor([0, 0])
    a = ORA.max_iou_assign(ov, labels, 0.6, 0.3, 0.2, gt_max_assign_all=True)
    assert a.tolist() == [1, 0, 2, 2, 2]
    a = ORA.max_iou_assign(ov, labels, 0.6, 0.3, 0.2, gt_max_assign_all=False)
    assert a.tolist() == [1, 0, 2, -1, 1]                 # only the FIRST best anchor of a box (argmax) is taken
    a = ORA.max_iou_assign(ov, labels, 0.6, 0.3, 0.2, match_low_quality=False)
    assert a.tolist() == [1, 0, -1, -1, 1]
    # min_pos_iou = 0 (mmdet's default): a box nothing overlaps "best-matches" every zero-overlap anchor
    ov0 = torch.tensor([[0.7, 0.1, 0.0], [0.0, 0.0, 0.0]])
    assert ORA.max_iou_assign(ov0, labels, 0.6, 0.3, 0.0).tolist() == [2, 2, 2]
    assert ORA.max_iou_assign(ov0, labels, 0.6, 0.3, 0.05).tolist() == [1, 0, 0]


def test_delta_encode_and_direction_target_known_values():
    anchor = torch.tensor([[1.0, 2.0, -1.0, 3.0, 4.0, 2.0, 0.0]])            # w 3, l 4: diagonal 5
    box = torch.tensor([[2.0, 4.5, -0.5, 3.0 * math.e, 4.0, 1.0, 2.0]])
    t = ORA.delta_encode(anchor, box)[0]
    # z: box centre -0.5 + 0.5 = 0.0, anchor centre -1 + 1 = 0.0
    want = [0.2, 0.5, 0.0, 1.0, 0.0, math.log(0.5), 2.0]
    assert all(abs(x - y) < 1e-6 for x, y in zip(t.tolist(), want))
    # direction bin: rot_gt = 2.0; offset 0 -> limit_period(2.0, 0, 2 pi) = 2.0 -> floor(2 / pi) = 0; yaw -1 -> 2 pi - 1 = 5.28 -> bin 1
    assert ORA.get_direction_target(anchor, t[None], 0.0).item() == 0
    t2 = t.clone(); t2[6] = -1.0
    assert ORA.get_direction_target(anchor, t2[None], 0.0).item() == 1
    # dir_offset pi/4 (the Waymo head's): yaw 0.5 - 0.7854 < 0 wraps to 6.0 -> bin 1;  yaw 1.0 -> 0.21 -> bin 0
    t3 = t.clone(); t3[6] = 0.5
    assert ORA.get_direction_target(anchor, t3[None], 0.7854).item() == 1
    t4 = t.clone(); t4[6] = 1.0
    assert ORA.get_direction_target(anchor, t4[None], 0.7854).item() == 0


def test_single_sample_layout_and_counts():
    """2 x 1 cells, 2 sizes, 2 rotations; one box identical to the (cell 1, size 1, rotation 0) anchor.  Per class: only size 1's assigner
    sees it; its anchor (index (1 * 2 + 1) * 2 + 0 = 6) is the positive; its rotated twin overlaps by 2x2 / (8 + 8 - 4) = 1/3 -> between
    the thresholds of 0.3 and 0.5 -> ignored (weight 0); everything else is a negative with label num_classes and weight 1."""
    sizes = [[1.0, 1.0, 1.0], [4.0, 2.0, 1.5]]
    anchors = ORA.range_anchors((1, 2), [[0., 0., -1., 20., 0., -1.]] * 2, sizes, [0.0, 1.57])[0]        # (1, 2, 2, 2, 7): x = 0 and 20
    assert anchors.shape == (1, 2, 2, 2, 7) and anchors[0, 1, 1, 0].tolist() == [20.0, 0.0, -1.0, 4.0, 2.0, 1.5, 0.0]
    box = anchors[0, 1, 1, 0].clone()[None]
    cfg = [dict(pos_iou_thr=0.5, neg_iou_thr=0.3, min_pos_iou=0.3)] * 2
    lab, lw, bt, bw, dt, dw, pos, neg = ORA.anchor_target_3d_single(anchors, box, torch.tensor([1]), cfg, 2)
    assert lab.tolist() == [2, 2, 2, 2, 2, 2, 1, 2]
    assert lw.tolist() == [1, 1, 1, 1, 1, 1, 1, 0]
    assert bw[:, 0].tolist() == [0, 0, 0, 0, 0, 0, 1, 0] and dw.tolist() == [0, 0, 0, 0, 0, 0, 1, 0]
    assert bt[6].abs().max().item() == 0.0 and dt.tolist() == [0] * 8
    assert pos.numel() == 1 and neg.numel() == 6
    out = ORA.anchor_target_3d(anchors, [box, box[:0]], [torch.tensor([1]), torch.zeros(0, dtype=torch.long)], cfg, 2)
    assert out[6] == 2 and out[7] == 6 + 8            # sum_b max(positives, 1), sum_b max(negatives, 1)


# ---- oracle/anchor_infer_torch.py (mmdet3d's inherited inference path) ---------------------------------------------------------------
def test_delta_decode_inverts_the_encode_and_limit_period_values():
    from oracle import anchor_infer_torch as AIT
    g = torch.Generator().manual_seed(0)
    anchors = torch.rand(50, 7, generator=g, dtype=torch.float64) * torch.tensor([70, 80, 1, 1.5, 3, 0.5, 1.5]) + torch.tensor([0, -40, -2, .6, .9, 1.4, 0])
    boxes = anchors + torch.randn(50, 7, generator=g, dtype=torch.float64) * 0.2
    boxes[:, 3:6] = boxes[:, 3:6].abs() + 0.3
    enc = ORA.delta_encode(anchors, boxes)
    assert (AIT.delta_decode(anchors, enc) - boxes).abs().max().item() < 1e-12          # the two coders are inverses
    v = torch.tensor([0.0, 1.0, 3.0, -3.0, 4.0, 7.0], dtype=torch.float64)
    # limit_period(v, 0.5, pi): into [-pi/2, pi/2);  (v, 0, 2 pi): into [0, 2 pi);  (v, 1, pi): into [-pi, 0)
    assert torch.allclose(AIT.limit_period(v, 0.5, math.pi), torch.tensor([0.0, 1.0, 3.0 - math.pi, math.pi - 3.0, 4.0 - math.pi, 7.0 - 2 * math.pi], dtype=torch.float64))
    assert torch.allclose(AIT.limit_period(v, 0.0, 2 * math.pi), torch.tensor([0.0, 1.0, 3.0, 2 * math.pi - 3.0, 4.0, 7.0 - 2 * math.pi], dtype=torch.float64))
    assert torch.allclose(AIT.limit_period(v, 1.0, math.pi), torch.tensor([-math.pi, 1.0 - math.pi, 3.0 - math.pi, -3.0, 4.0 - 2 * math.pi, 7.0 - 3 * math.pi], dtype=torch.float64))


def test_get_bboxes_single_on_a_hand_case():
    """1 x 2 cells, 2 anchors per cell, 2 classes: four far-apart unit boxes with zero deltas.  Scores (after the sigmoid) by anchor and class:
    a0 (.9, .2)  a1 (.1, .8)  a2 (.6, .7)  a3 (.05, .05);  nms_pre 3 keeps a0, a1, a2 (best class .9, .8, .7);  score_thr .3: class 0 has
    a0 (.9), a2 (.6); class 1 has a1 (.8), a2 (.7) — a2 is reported once per class;  no overlaps, so NMS keeps all; max_num 3 cuts the
    concatenation [a0 .9, a2 .6 | a1 .8, a2 .7] to the three best: .9 (class 0), .8 (class 1), .7 (class 1).  Direction bins 0, 1, 1 with
    dir_offset 0, dir_limit_offset 1: yaw = limit_period(0, 1, pi) + pi * bin = -pi + pi * bin."""
    from oracle import anchor_infer_torch as AIT
    logit = lambda p: math.log(p / (1 - p))          # noqa: E731
    s = [[.9, .2], [.1, .8], [.6, .7], [.05, .05]]
    cls = torch.zeros(4, 1, 2)                       # (A * C, H, W), channel = a * C + c
    for cell in range(2):
        for a in range(2):
            for c in range(2):
                cls[a * 2 + c, 0, cell] = logit(s[cell * 2 + a][c])
    bbox = torch.zeros(14, 1, 2)
    dirs = torch.zeros(4, 1, 2)                      # channel = a * 2 + bin
    for k, b in enumerate([0, 1, 1, 0]):             # anchor k = cell * 2 + a prefers bin b
        dirs[(k % 2) * 2 + b, 0, k // 2] = 1.0
    anchors = torch.tensor([[10.0 * k, 0.0, -1.0, 1.0, 1.0, 1.0, 0.0] for k in range(4)])
    cfg = dict(use_rotate_nms=True, nms_pre=3, nms_thr=0.5, score_thr=0.3, max_num=3)
    boxes, scores, labels = AIT.get_bboxes_single([cls], [bbox], [dirs], [anchors], cfg, 2, dir_offset=0.0, dir_limit_offset=1.0)
    assert labels.tolist() == [0, 1, 1]
    assert torch.allclose(scores, torch.tensor([0.9, 0.8, 0.7]), atol=1e-6)
    assert boxes[:, 0].tolist() == [0.0, 10.0, 20.0]
    assert torch.allclose(boxes[:, 6], torch.tensor([-math.pi, 0.0, 0.0]), atol=1e-6)
    assert torch.allclose(boxes[:, 1:6], torch.tensor([[0.0, -1.0, 1.0, 1.0, 1.0]]).expand(3, 5))


def test_product_anchor_grid_equals_the_restatement_and_aligned_form_sits_on_cell_centres():
    """mmdet3d-gaussian_amd/anchors.py (plain torch ops, any device) against oracle range_anchors bit for bit, and the aligned generator's
    centres against the closed form (cell centres of the range)"""
    import mmdet3d_gaussian_amd as amd
    R = [[0.08, -39.60, -0.6, 68.88, 39.44, -0.6]] * 2 + [[0.08, -39.60, -1.78, 68.88, 39.44, -1.78]]
    S = [[0.8, 0.6, 1.73], [1.76, 0.6, 1.73], [3.9, 1.6, 1.56]]
    a = amd.extras.anchor3d_range_anchors((20, 18), R, S, [0, 1.57], 'cpu')
    assert a.shape == (1, 20, 18, 3, 2, 7) and torch.equal(a, ORA.range_anchors((20, 18), R, S, [0, 1.57]))
    w = amd.extras.anchor3d_range_anchors((4, 8), [[-8., -4., 0.5, 8., 4., 0.5]], [[1., 2., 3.], [2., 2., 2.]], [0., 1.57], 'cpu', aligned=True)
    assert w.shape == (1, 4, 8, 2, 2, 7)
    assert w[0, :, 0, 0, 0, 1].tolist() == [-3.0, -1.0, 1.0, 3.0] and w[0, 0, :, 1, 1, 0].tolist() == [-7.0, -5.0, -3.0, -1.0, 1.0, 3.0, 5.0, 7.0]
    assert w[0, 2, 3, 1, 1, :6].tolist() == [-1.0, 1.0, 0.5, 2.0, 2.0, 2.0] and abs(w[0, 2, 3, 1, 1, 6].item() - 1.57) < 1e-6


def test_pvrcnn_decode_restatement_on_a_hand_case():
    """oracle/pvrcnn_torch.py decode_rois: a residual of one diagonal along the roi's own x axis puts the centre at
    roi + diagonal * (cos ry, sin ry) counter-clockwise (mmdet3d 1.0), at (cos ry, -sin ry) clockwise; sizes exp-scaled, yaw added;
    multi_class_nms keeps class after class and returns [] when nothing reaches the thresholds"""
    from oracle import pvrcnn_torch as PV
    rois = torch.tensor([[0., 1.0, 2.0, -1.0, 3.0, 4.0, 2.0, 0.5]])
    pred = torch.tensor([[1.0, 0.0, 0.25, math.log(2.0), 0.0, math.log(0.5), 0.1]])
    for cw, sgn in ((False, 1.0), (True, -1.0)):
        b = PV.decode_rois(rois, pred, clockwise=cw)[0]
        # z: anchor centre 0 + 2/2 = 1, decoded centre 0.25 * 2 + 1 = 1.5, new height 1 -> bottom 1.0; + roi z -1 -> 0.0
        want = [1.0 + 5.0 * math.cos(0.5), 2.0 + sgn * 5.0 * math.sin(0.5), 0.0, 6.0, 4.0, 1.0, 0.6]
        assert all(abs(x - y) < 1e-5 for x, y in zip(b.tolist(), want)), (b.tolist(), want)
    boxes = torch.tensor([[0., 0., 0., 2., 2., 1., 0.], [0.1, 0., 0., 2., 2., 1., 0.], [10., 0., 0., 2., 2., 1., 0.]])
    probs = torch.tensor([[0.9, 0.1], [0.8, 0.6], [0.2, 0.7]])
    sel = PV.multi_class_nms(probs, boxes, [0.5, 0.5], 0.1)
    assert sel.tolist() == [0, 2, 1]          # class 0: boxes 0, 1 overlap -> 0; class 1: 2 (0.7) then 1 (0.6), far apart
    assert PV.multi_class_nms(probs, boxes, 0.95, 0.1) == []
