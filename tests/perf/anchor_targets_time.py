#!/usr/bin/env python3
"""Anchor heads' target assignment and the whole GDAnchor3DHead.loss at the reference's KITTI PointPillars training geometry
(248 x 216 cells x 3 sizes x 2 rotations = 321 408 anchors per sample, batch 6, 3 classes, the config's three MaxIoUAssigners),
us per call (synchronised):
  targets : ours = anchor_head_get_targets (two launches + one read-back of the counts)
            eager = mmdet3d's anchor_target_3d chain (oracle/anchor_targets_torch.py's statement) on device tensors
  loss    : ours = gd_anchor_head_loss eager, static, and static as one hipGraph (forward + backward)
            eager = the restated chain + the reference's loss ops on device tensors (forward + backward)
Then the target assignment alone at the Waymo geometry (BASELINE configs[4]; hv_pointpillars_secfpn_waymo.py:46-57, :74-96: 468 x 468 cells x
3 sizes x 2 rotations = 1 314 144 anchors per sample, three assigners that each see ALL boxes — the head's default assign_per_class=False),
batch 2, 80 boxes per sample.  Asserts equal labels / counts first."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch  # noqa: E402

import mmdet3d_gaussian_amd as amd  # noqa: E402
from oracle import anchor_cls_torch, head_torch  # noqa: E402
from oracle import anchor_targets_torch as ORA  # noqa: E402
from test_gpu_anchor_targets import CE, FOCAL, KITTI_ASSIGNERS, SL1, TRAIN_CFG, head_outputs, kitti_anchors, random_gt  # noqa: E402

dev = torch.device('cuda:0')


def timeit(fn, it, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(it):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / it * 1e6


def main():
    B, H, W, G = 6, 248, 216, 24
    anchors = kitti_anchors(H, W).to(dev)
    pairs = [random_gt(G, seed=100 + i, with_ignored=False) for i in range(B)]
    gts, labels = [p[0].to(dev) for p in pairs], [p[1].to(dev) for p in pairs]
    ours = lambda: amd.extras.anchor_head_get_targets(anchors, gts, labels, KITTI_ASSIGNERS, 3)          # noqa: E731
    eager = lambda: ORA.anchor_target_3d(anchors, gts, labels, KITTI_ASSIGNERS, 3)                # noqa: E731
    a, e = ours(), eager()
    assert torch.equal(a[0], e[0]) and torch.equal(a[1], e[1]) and a[6] == e[6] and a[7] == e[7]
    t_o, t_e = timeit(ours, 30), timeit(eager, 3, warm=1)
    t_p = timeit(lambda: amd.extras.anchor_head_get_targets(anchors, gts, labels, KITTI_ASSIGNERS, 3, padded=True), 30)
    print(json.dumps(dict(step='anchor_target_3d', geometry='kitti', batch=B, anchors_per_sample=H * W * 6, boxes_per_sample=G, positives=a[6],
                          ours_us=round(t_o, 1), ours_no_readback_us=round(t_p, 1), reference_ops_on_gpu_us=round(t_e, 1),
                          speedup=round(t_e / t_o, 1))), flush=True)

    outs = [o.to(dev).requires_grad_(True) for o in head_outputs(B, H, W, seed=1)]
    mod = amd.GDLoss('kld3d', fun='log1p', tau=1.0, loss_weight=5.0)
    gt_t, gl_t = torch.stack(gts), torch.stack(labels)

    def fn(cls, bbox, dirs, gt, gl, static=True):
        r = amd.extras.gd_anchor_head_loss(FOCAL, SL1, CE, mod, TRAIN_CFG, 3, anchors, cls, bbox, dirs, list(gt.unbind(0)), list(gl.unbind(0)), static=static)
        return r['loss_cls'][0], r['loss_bbox'][0], r['loss_dir'][0]

    def run(static):
        for o in outs:
            o.grad = None
        l = fn(outs[0], outs[1], outs[2], gt_t, gl_t, static)
        (l[0] + l[1] + l[2]).backward()

    def ref():
        for o in outs:
            o.grad = None
        tg = ORA.anchor_target_3d(anchors, gts, labels, KITTI_ASSIGNERS, 3)
        avg = float(tg[6])
        lc, ld = anchor_cls_torch.cls_dir_losses(outs[0], outs[2], tg[0], tg[1], tg[4], tg[5], 3, avg)
        lb = head_torch.loss_single_bbox(outs[1], tg[2], tg[3], tg[0], anchors.reshape(-1, 7), 3, avg,
                                         gd=dict(loss_type='kld3d', fun='log1p', tau=1.0, loss_weight=5.0),
                                         sl1=dict(beta=SL1['beta'], loss_weight=SL1['loss_weight']), code_weight=TRAIN_CFG['code_weight'],
                                         decode_weight=TRAIN_CFG['decode_weight'], diff_rad_by_sin=True)
        (lc + lb + ld).backward()
    step = amd.GraphedStep(fn, (outs[0], outs[1], outs[2], gt_t, gl_t))
    t_eager, t_static = timeit(lambda: run(False), 30), timeit(lambda: run(True), 30)
    t_graph = timeit(lambda: step(outs[0], outs[1], outs[2], gt_t, gl_t), 30)
    t_ref = timeit(ref, 3, warm=1)
    print(json.dumps(dict(step='GDAnchor3DHead.loss fwd+bwd', geometry='kitti', batch=B, ours_eager_us=round(t_eager, 1), ours_static_us=round(t_static, 1),
                          ours_graph_us=round(t_graph, 1), reference_ops_on_gpu_us=round(t_ref, 1), speedup_eager=round(t_ref / t_eager, 1),
                          speedup_graph=round(t_ref / t_graph, 1))), flush=True)


def waymo():
    B, H, W, G = 2, 468, 468, 80
    rng = [[-74.88, -74.88, -0.0345, 74.88, 74.88, -0.0345], [-74.88, -74.88, -0.1188, 74.88, 74.88, -0.1188], [-74.88, -74.88, 0.0, 74.88, 74.88, 0.0]]
    sizes = [[4.73, 2.08, 1.77], [1.81, 0.84, 1.77], [0.91, 0.84, 1.74]]
    cfgs = [dict(type='MaxIoUAssigner', iou_calculator=dict(type='BboxOverlapsNearest3D'), pos_iou_thr=p, neg_iou_thr=n, min_pos_iou=n, ignore_iof_thr=-1)
            for p, n in ((0.55, 0.4), (0.5, 0.3), (0.5, 0.3))]
    anchors = ORA.range_anchors((H, W), rng, sizes, [0, 1.57])[0].to(dev)
    g = torch.Generator().manual_seed(7)
    gts, labels = [], []
    for b in range(B):
        lab = torch.randint(0, 3, (G,), generator=g)
        sz = torch.tensor(sizes)[lab] * (0.8 + 0.4 * torch.rand(G, 3, generator=g))
        box = torch.cat([torch.rand(G, 2, generator=g) * 140 - 70, torch.zeros(G, 1), sz, (torch.rand(G, 1, generator=g) * 2 - 1) * 3.14159], dim=-1)
        gts.append(box.to(dev))
        labels.append(lab.to(dev))
    ours = lambda: amd.extras.anchor_head_get_targets(anchors, gts, labels, cfgs, 3, assign_per_class=False, dir_offset=0.7854)          # noqa: E731
    eager = lambda: ORA.anchor_target_3d(anchors, gts, labels, cfgs, 3, assign_per_class=False, dir_offset=0.7854)                # noqa: E731
    a, e = ours(), eager()
    assert torch.equal(a[0], e[0]) and torch.equal(a[1], e[1]) and torch.equal(a[4], e[4]) and a[6] == e[6] and a[7] == e[7]
    t_o, t_e = timeit(ours, 30), timeit(eager, 2, warm=1)
    print(json.dumps(dict(step='anchor_target_3d', geometry='waymo', batch=B, anchors_per_sample=H * W * 6, boxes_per_sample=G, positives=a[6],
                          ours_us=round(t_o, 1), reference_ops_on_gpu_us=round(t_e, 1), speedup=round(t_e / t_o, 1))), flush=True)

    # the whole loss at this geometry: GWD as the config has it (configs/waymo/hv_pointpillars_secfpn_gwd5_...: loss_weight 5), code_weight only
    outs = [o.to(dev).requires_grad_(True) for o in head_outputs(B, H, W, seed=2)]
    mod = amd.GDLoss('gwd3d', fun='log1p', tau=0.0, loss_weight=5.0)
    tcfg = dict(assigner=cfgs, allowed_border=0, code_weight=[1.0] * 7, pos_weight=-1, debug=False)
    gt_t, gl_t = torch.stack(gts), torch.stack(labels)

    def fn(cls, bbox, dirs, gt, gl, static=True):
        r = amd.extras.gd_anchor_head_loss(FOCAL, SL1, CE, mod, tcfg, 3, anchors, cls, bbox, dirs, list(gt.unbind(0)), list(gl.unbind(0)), assign_per_class=False,
                                    dir_offset=0.7854, static=static)
        return r['loss_cls'][0], r['loss_bbox'][0], r['loss_dir'][0]

    def run(static):
        for o in outs:
            o.grad = None
        l = fn(outs[0], outs[1], outs[2], gt_t, gl_t, static)
        (l[0] + l[1] + l[2]).backward()

    def ref():
        for o in outs:
            o.grad = None
        tg = ORA.anchor_target_3d(anchors, gts, labels, cfgs, 3, assign_per_class=False, dir_offset=0.7854)
        avg = float(tg[6])
        lc, ld = anchor_cls_torch.cls_dir_losses(outs[0], outs[2], tg[0], tg[1], tg[4], tg[5], 3, avg)
        lb = head_torch.loss_single_bbox(outs[1], tg[2], tg[3], tg[0], anchors.reshape(-1, 7), 3, avg, gd=dict(loss_type='gwd3d', fun='log1p', tau=0.0, loss_weight=5.0),
                                         sl1=dict(beta=SL1['beta'], loss_weight=SL1['loss_weight']), code_weight=[1.0] * 7, decode_weight=None, diff_rad_by_sin=True)
        (lc + lb + ld).backward()
    step = amd.GraphedStep(fn, (outs[0], outs[1], outs[2], gt_t, gl_t))
    t_eager, t_graph, t_ref = timeit(lambda: run(False), 30), timeit(lambda: step(outs[0], outs[1], outs[2], gt_t, gl_t), 30), timeit(ref, 2, warm=1)
    print(json.dumps(dict(step='GDAnchor3DHead.loss fwd+bwd', geometry='waymo', batch=B, ours_eager_us=round(t_eager, 1), ours_graph_us=round(t_graph, 1),
                          reference_ops_on_gpu_us=round(t_ref, 1), speedup_eager=round(t_ref / t_eager, 1), speedup_graph=round(t_ref / t_graph, 1))), flush=True)


if __name__ == '__main__':
    main()
    waymo()
