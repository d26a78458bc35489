#!/usr/bin/env python3
"""The heat-map loss of the CenterPoint heads and the whole CenterGDHead.loss (gd_centerpoint_head.py:390-441), nuScenes geometry
(6 tasks, batch 8, 128 x 128 maps).  JSON lines, us per step (forward + backward, synchronised):
  heatmap_loss : center_head_heatmap_loss (all tasks in one pass) vs the torch op sequence of clip_sigmoid + GaussianFocalLoss per
                 task on the GPU (oracle/heat_focal_torch.py, with the reference's `.item()` per task)
  full_loss    : center_gd_head_loss (targets + heat-map loss + regression losses) vs the reference's statement: get_targets loops
                 on the CPU + the eager losses on the GPU (oracle/center_targets_torch.py, heat_focal_torch.py, head_torch.py)"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch  # noqa: E402

import mmdet3d_gaussian_amd as amd  # noqa: E402
from oracle import center_targets_torch as ct, head_torch, heat_focal_torch as hf  # noqa: E402
from test_gpu_center_targets import NUS, TASKS, scene  # noqa: E402

dev = torch.device('cuda:0')


def timeit(fn, it, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(it):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / it * 1e6


def main():
    torch.set_num_threads(1)
    g = torch.Generator().manual_seed(7)
    B = 8
    data = [scene(g, 150, spread=50.0) for _ in range(B)]
    boxes, labels = [d[0] for d in data], [d[1] for d in data]
    gb, gl = [b.to(dev) for b in boxes], [l.to(dev) for l in labels]
    cfg = dict(NUS, code_weights=[1.0, 1.0, 0.2, 0.2])
    coder = amd.CenterPointBBoxYawCoder(pc_range=[-51.2, -51.2], out_size_factor=4, voxel_size=[0.2, 0.2], norm_bbox=True)
    gd = amd.GDLoss('bd3d', fun='log1p', tau=0.0, loss_weight=5.0)
    l1 = dict(type='L1Loss', reduction='mean', loss_weight=0.25)
    cls = dict(type='GaussianFocalLoss', reduction='mean')
    chans = (('heatmap', None), ('reg', 2), ('height', 1), ('dim', 3), ('yaw', 1), ('dir', 2), ('vel', 2))
    pds = [{k: (torch.randn(B, c if c else len(names), 128, 128, generator=g) * 0.5 - (2.0 if c is None else 0.0)).to(dev).requires_grad_(True)
            for k, c in chans} for names in TASKS]
    hm, an, pi = amd.extras.center_head_get_targets(gb, gl, TASKS, cfg)

    def zero():
        for p in pds:
            for v in p.values():
                v.grad = None

    def heat_ours():
        zero()
        l, _ = amd.extras.center_head_heatmap_loss(cls, [p['heatmap'] for p in pds], hm)
        l.sum().backward()

    def heat_eager():
        zero()
        tot = 0
        for p, t in zip(pds, hm):
            l, _ = hf.heatmap_loss(p['heatmap'], t)
            tot = tot + l
        tot.backward()
    us_a, us_b = timeit(heat_ours, 50), timeit(heat_eager, 10)
    cells = sum(p['heatmap'].numel() for p in pds)
    print(json.dumps(dict(what='heatmap_loss fwd+bwd, 6 tasks, batch 8, 128x128', cells=cells, ours_us=round(us_a, 1),
                          eager_torch_us=round(us_b, 1))), flush=True)

    # the same kernel at batch 64 (12.6 M cells, 151 MB moved per call): the device time of the pass, forward only
    big_x = [torch.randn(64, len(names), 128, 128, generator=g).to(dev).requires_grad_(True) for names in TASKS]
    big_t = [torch.rand(64, len(names), 128, 128, generator=g).to(dev) ** 6 for names in TASKS]
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(3):
        amd.extras.center_head_heatmap_loss(cls, big_x, big_t)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(20):
        amd.extras.center_head_heatmap_loss(cls, big_x, big_t)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    cells_big = sum(x.numel() for x in big_x)
    print(json.dumps(dict(what='heatmap_loss forward call at batch 64 (logit + target read, raw gradient written: 12 B per cell)', cells=cells_big,
                          us_per_call=round(ms * 1e3, 1), GBps_of_12B_per_cell=round(cells_big * 12 / (ms * 1e-3) / 1e9, 1))), flush=True)

    def full_ours():
        zero()
        out = amd.extras.center_gd_head_loss(cls, l1, gd, coder, TASKS, cfg, gb, gl, pds)
        sum(out.values()).backward()

    def full_eager():
        zero()
        counts = [len(t) for t in TASKS]
        hmc, anc, pic = ct.get_targets(boxes, labels, counts, cfg)          # the reference's loops (CPU statement)
        tot = 0
        for t, p in enumerate(pds):
            lh, npos = hf.heatmap_loss(p['heatmap'], hmc[t].to(dev))
            a, b = head_torch.center_head_task_losses(p, pic[t].to(dev), anc[t].to(dev), max(npos, 1), dict(pc_range=[-51.2, -51.2], out_size_factor=4,
                                                      voxel_size=[0.2, 0.2], norm_bbox=True),
                                                      dict(loss_type='bd3d', fun='log1p', tau=0.0, loss_weight=5.0), 0.25, cfg['code_weights'])
            tot = tot + lh + a + b
        tot.backward()
    us_a, us_b = timeit(full_ours, 30), timeit(full_eager, 3, warm=1)

    # static form: nothing is read back (row offsets and num_pos stay on the device), so the whole method can be captured
    def full_static():
        zero()
        out = amd.extras.center_gd_head_loss(cls, l1, gd, coder, TASKS, cfg, gb, gl, pds, static=True)
        sum(out.values()).backward()
    us_s = timeit(full_static, 30)
    keys = ('heatmap', 'reg', 'height', 'dim', 'yaw', 'dir', 'vel')

    def fn(*args):
        bx, lb, flat = args[:B], args[B:2 * B], args[2 * B:]
        out = amd.extras.center_gd_head_loss(cls, l1, gd, coder, TASKS, cfg, list(bx), list(lb),
                                      [dict(zip(keys, flat[7 * t:7 * t + 7])) for t in range(len(TASKS))], static=True)
        return [out[k] for k in sorted(out)]
    step = amd.GraphedStep(fn, gb + gl + [p[k] for p in pds for k in keys])
    st = step.static_inputs()
    us_g = timeit(lambda: step(*st), 100)
    print(json.dumps(dict(what='CenterGDHead.loss end to end (targets + heat-map loss + regression losses) fwd+bwd, 6 tasks, batch 8 x 150 boxes',
                          ours_us=round(us_a, 1), ours_static_no_readback_us=round(us_s, 1), ours_static_as_a_hipgraph_us=round(us_g, 1),
                          reference_statement_us=round(us_b, 1))), flush=True)


if __name__ == '__main__':
    main()
