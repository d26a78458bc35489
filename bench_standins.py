"""bench_standins.py — what bench.py adds to its JSON line AFTER the timed regions (nothing here is inside one).

SURVEY.md §8d's metric is "M box-pairs/s … plus achieved HBM GB/s and fraction of roofline; plus max-abs-error vs oracle", and
BASELINE.md asks for NMS boxes/s; BASELINE.json's other configs are parity-test cases (tests/test_gpu_configs.py), but the
driver only ever runs bench.py — so the line also carries, measured on the same box in the same process:

  parity : max-abs-error of the per-pair loss and of grad_pred against the fp64 CPU oracle (oracle/gd_oracle.c, the
           reference's formulas, pinned by tests/golden) on a strided >= 100 k-row sample of the very buffers the timed
           region ran on — grad_pred rows are read from the gradient the LAST TIMED STEP left in HBM.  The oracle is the
           CHECKER here, as in smoke(); nothing measured goes through it.
  nms    : configs[4] stand-in (gd_centerpoint_head.py:336-345 / pvrcnn_bbox_head.py:438-464): 3 classes x 4096 Waymo-like
           boxes, thr 0.25, post_max_size 500: us for one class, us for the three as one batched call, boxes/s, and
           `keep_bit_exact` = every keep list equal to the `_cpu` twin's (csrc/rbox_cpu.cpp).
  head   : configs[1] stand-in (gd_anchor3d_head.py:133-141: 6 x 321 408 KITTI anchors, KLD tau = 0) and configs[3] stand-in
           (gd_centerpoint_head.py:402-441: 6 nuScenes CenterPoint tasks x 8 x 128 x 128 maps, BCD), us per step launched
           eagerly and replayed as a hipGraph.
Bounded: ~10 s in all.
"""
import time

import numpy as np
import torch


def _timeit(fn, iters, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e6


def parity(amd, losses, preds, tgt, n, loss_weight, idx, grad_rows):
    """Per loss: max |ours - fp64 oracle| of the per-pair loss (reduction='none' on the sampled rows `idx` of the timed buffers) and
    of grad_pred (`grad_rows`: those rows of the gradient the last timed step wrote, copied right after the region; per-pair
    scale: loss_weight / n undone), with the tolerance the tests use: 1e-5 x (1 + row scale) (tests/gd_golden.py)."""
    import oracle
    stride = int(idx[1] - idx[0]) if idx.numel() > 1 else 1
    t_s = tgt.index_select(0, idx).contiguous()
    out = {'rows': int(idx.numel()), 'stride': stride, 'oracle': 'oracle/gd_oracle.c fp64 (the reference\'s formulas; checker only)',
           'tolerance': '1e-5 x (1 + |row|): tests/gd_golden.py'}
    ok = True
    for lt in losses:
        p = preds[lt]
        p_s = p.detach().index_select(0, idx).contiguous()
        g_s = grad_rows[lt] * (n / loss_weight)          # per-pair gradient as the last timed step left it
        mod = amd.build_loss(dict(type='GDLoss', loss_type=lt, fun='log1p', tau=1.0, alpha=1.0, reduction='none', loss_weight=1.0))
        with torch.no_grad():
            l_s = mod(p_s, t_s)
        ref = oracle.gd_loss(p_s.cpu().numpy(), t_s.cpu().numpy(), oracle.make_params(lt, fun='log1p', tau=1.0), scale=1.0)
        el = np.abs(l_s.cpu().numpy().astype(np.float64) - ref['loss'])
        eg = np.abs(g_s.cpu().numpy().astype(np.float64) - ref['grad_pred'])
        rl = el / (1.0 + np.abs(ref['loss']))
        rg = eg / (1.0 + np.abs(ref['grad_pred']).max(-1, keepdims=True))
        good = bool(rl.max() <= 1e-5 and rg.max() <= 1e-5)
        ok &= good
        out[lt] = {'loss_max_abs_err': float(el.max()), 'grad_pred_max_abs_err': float(eg.max()),
                   'loss_max_rel': float(rl.max()), 'grad_pred_max_rel': float(rg.max()),
                   'grad_pred_max_abs': float(np.abs(ref['grad_pred']).max()), 'within_1e-5': good}
    out['all_within_1e-5'] = ok
    return out


def _waymo_like_boxes(n, seed, extent=74.88):
    """BEV boxes [x1, y1, x2, y2, ry] + scores: car / ped / cyc sizes (hv_pointpillars_secfpn_waymo.py:51-55) in clusters of
    jittered duplicates, as a dense head produces them (the generator of tests/rbox_inputs.py)."""
    rng = np.random.default_rng(seed)
    sizes = np.array([[4.73, 2.08], [0.91, 0.84], [1.81, 0.84]], np.float32)
    nc = max(1, n // 8)
    cx = rng.uniform(-extent, extent, nc); cy = rng.uniform(-extent, extent, nc)
    cls = rng.integers(0, 3, nc); yaw = rng.uniform(-np.pi, np.pi, nc)
    idx = rng.integers(0, nc, n)
    x = cx[idx] + rng.normal(0, 0.3, n); y = cy[idx] + rng.normal(0, 0.3, n)
    wl = sizes[cls[idx]] * rng.uniform(0.9, 1.1, (n, 2))
    r = yaw[idx] + rng.normal(0, 0.1, n)
    boxes = np.stack([x - wl[:, 0] / 2, y - wl[:, 1] / 2, x + wl[:, 0] / 2, y + wl[:, 1] / 2, r], -1)
    return boxes.astype(np.float32), rng.uniform(0, 1, n).astype(np.float32)


def nms(amd, dev, classes=3, n=4096, thr=0.25, post=500):
    cls, exact = [], True
    for c in range(classes):
        b, s = _waymo_like_boxes(n, seed=100 + c)
        bt, st = torch.from_numpy(b), torch.from_numpy(s)
        cls.append((bt.to(dev), st.to(dev)))
        want = amd.nms_gpu(bt, st, thr, post_max_size=post)            # CPU tensors: the `_cpu` twin
        got = amd.nms_gpu(cls[-1][0], cls[-1][1], thr, post_max_size=post)
        exact &= bool(torch.equal(got.cpu(), want))
    allb = torch.cat([b for b, _ in cls])
    alls = torch.zeros(classes, classes * n, device=dev)
    allv = torch.zeros(classes, classes * n, dtype=torch.bool, device=dev)
    for c in range(classes):
        alls[c, c * n:(c + 1) * n] = cls[c][1]
        allv[c, c * n:(c + 1) * n] = True

    def batched():
        return amd.nms_gpu_batched(allb, alls, thr, allv, pre_max_size=n, post_max_size=post)
    res = batched()
    kept = []
    for c in range(classes):
        one = amd.nms_gpu(cls[c][0], cls[c][1], thr, post_max_size=post)
        exact &= bool(torch.equal(res[c] - c * n, one))
        kept.append(int(one.numel()))
    us_one = _timeit(lambda: amd.nms_gpu(cls[0][0], cls[0][1], thr, post_max_size=post), 50)
    us_batched = _timeit(batched, 50)
    return {'workload': f'{classes} classes x {n} boxes, thr {thr}, post_max_size {post} (BASELINE configs[4] stand-in)',
            'us_single_class': round(us_one, 1), 'us_batched': round(us_batched, 1),
            'boxes_per_s_single_class': round(n / us_one * 1e6), 'boxes_per_s_batched': round(classes * n / us_batched * 1e6),
            'kept': kept, 'keep_bit_exact': exact, 'keep_checked_against': '_cpu twin (rnms_bev_cpu), end to end incl. the host read of the count'}


def head(amd, dev):
    g = torch.Generator(device=dev).manual_seed(0)
    out = {}
    # ---- configs[1]: PointPillars KITTI 3-class, KLD tau = 0: anchor-head decoded-box loss from the raw NCHW output
    B, A, H, W, C = 6, 6, 248, 216, 3
    n_per = H * W * A
    anchors = (torch.rand(n_per, 7, generator=g, device=dev) * torch.tensor([70, 80, 1, 1.5, 3, .5, 1.5], device=dev)
               + torch.tensor([0, -40, -2, .6, .9, 1.4, 0], device=dev))
    bbox_pred = (torch.randn(B, A * 7, H, W, generator=g, device=dev) * 0.1).requires_grad_(True)
    bbox_targets = torch.randn(B, n_per, 7, generator=g, device=dev) * 0.2
    bbox_weights = torch.ones(B, n_per, 7, device=dev)
    labels = torch.full((B, n_per), C, device=dev, dtype=torch.long)
    labels.view(-1)[torch.randperm(B * n_per, generator=g, device=dev)[:60 * B]] = 0
    mod = amd.GDLoss('kld3d', fun='log1p', tau=0.0, loss_weight=5.0)

    def fn1(bp, lb):
        return amd.anchor_head_decoded_loss_fused(mod, bp, bbox_targets, bbox_weights, lb, anchors, C, 360.0, [1.0] * 7)

    def step1():
        bbox_pred.grad = None
        fn1(bbox_pred, labels).backward()
    us_e = _timeit(step1, 50)
    gs = amd.GraphedStep(fn1, (bbox_pred, labels))
    sbp, slb = gs.static_inputs()
    us_g = _timeit(lambda: gs(sbp, slb), 100)
    out['config1_kitti_kld'] = {'workload': f'{B} x {n_per} anchors, {60 * B} positives, KLD tau=0, fwd+bwd from NCHW (gd_anchor3d_head.py:133-141)',
                                'us_eager': round(us_e, 1), 'us_graph': round(us_g, 1)}
    del gs, bbox_pred, bbox_targets, bbox_weights, labels
    # ---- configs[3]: nuScenes CenterPoint (nearest shipped), BCD: all regression losses of 6 tasks from the raw head maps
    coder = amd.CenterPointBBoxYawCoder(pc_range=[-51.2, -51.2], out_size_factor=4, voxel_size=[0.2, 0.2], norm_bbox=True)
    Bs, K, tasks = 8, 500, 6
    modb = amd.GDLoss('bd3d', fun='log1p', tau=0.0, loss_weight=5.0)
    poss, annos, maps = [], [], []
    names = ('reg', 'height', 'dim', 'yaw', 'dir', 'vel')
    for _ in range(tasks):
        P = Bs * K
        pos = torch.stack([torch.randint(0, Bs, (P,), generator=g, device=dev), torch.randint(0, 128, (P,), generator=g, device=dev),
                           torch.randint(0, 128, (P,), generator=g, device=dev)], -1)
        xy = (pos[:, 1:].float() + torch.rand(P, 2, generator=g, device=dev)) * 0.8 - 51.2
        annos.append(torch.cat([xy, torch.rand(P, 1, generator=g, device=dev) * 4 - 3, torch.rand(P, 3, generator=g, device=dev) * 2 + 0.5,
                                torch.rand(P, 1, generator=g, device=dev) * 6 - 3, torch.randn(P, 2, generator=g, device=dev)], -1))
        poss.append(pos)
        maps.append({k: (torch.randn(Bs, c, 128, 128, generator=g, device=dev) * 0.3).requires_grad_(True)
                     for k, c in zip(names, (2, 1, 3, 1, 2, 2))})
    l1cfg = dict(type='L1Loss', reduction='mean', loss_weight=0.25)
    cw = [1.0, 1.0, 0.2, 0.2]

    def step3():
        for d in maps:
            for v in d.values():
                v.grad = None
        res = amd.center_head_losses(modb, l1cfg, coder, maps, poss, annos, [Bs * K] * tasks, cw)
        sum(a + b for a, b in res).backward()
    us_e = _timeit(step3, 30)
    flat = [d[k] for d in maps for k in names]

    def fn3(*fl):
        ds = [dict(zip(names, fl[6 * i:6 * i + 6])) for i in range(tasks)]
        return amd.center_head_losses(modb, l1cfg, coder, ds, poss, annos, [Bs * K] * tasks, cw)
    g3 = amd.GraphedStep(fn3, flat)
    s3 = g3.static_inputs()
    us_g = _timeit(lambda: g3(*s3), 100)
    out['config3_nuscenes_bcd'] = {'workload': f'{tasks} CenterPoint tasks x {Bs} x 128 x 128 maps x {Bs * K} objects, BCD, loss_l1 + loss_gd fwd+bwd '
                                               '(gd_centerpoint_head.py:402-441)',
                                   'us_eager': round(us_e, 1), 'us_graph': round(us_g, 1)}
    return out
