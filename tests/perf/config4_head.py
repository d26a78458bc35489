#!/usr/bin/env python3
"""BASELINE configs[3] (nuScenes CenterPoint geometry, BCD), regression losses of the head as a MODEL sees them: the 36 head
maps are NON-LEAF outputs of convolutions (per task and head one 1x1 conv from a shared 64-channel feature: the role of
SeparateHead's final convs, whose outputs gd_centerpoint_head.py:402-441 consumes), so backward hands the gradient maps to
the convs' backward instead of to 36 leaf AccumulateGrad nodes (round 2's number, 639 us, timed those).
Prints JSON lines:
  model_step     : 36 convs forward + center_head_losses + backward through the convs, us per step, ours vs the op-for-op
                   eager restatement (oracle/head_torch.py); the conv work is the same in both and is timed alone as well
  loss_only      : the same maps as leaves (round 2's harness), for continuity
  sorted_finish  : batch-64 sizes (32 000 objects per task in 64 cells x 6 tasks): sorted accumulate vs the scanning one
Run under `rocprofv3 --kernel-trace --stats` for the per-kernel table (profiles/r03_config4_kernel_stats.csv).
Asserts that the losses of the model-shaped step equal those of the leaf-shaped step bit for bit."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import mmdet3d_gaussian_amd as amd  # noqa: E402
from mmdet3d_gaussian_amd import head_loss  # noqa: E402
from oracle import head_torch  # noqa: E402

dev = torch.device('cuda:0')
HEADS = (('reg', 2), ('height', 1), ('dim', 3), ('yaw', 1), ('dir', 2), ('vel', 2))


def timeit(fn, it, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(it):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / it * 1e6


def objects(g, Bs, P, hw=128, cells=None):
    if cells is None:
        pos = torch.stack([torch.randint(0, Bs, (P,), generator=g, device=dev), torch.randint(0, hw, (P,), generator=g, device=dev),
                           torch.randint(0, hw, (P,), generator=g, device=dev)], -1)
    else:   # everything into `cells` cells of two samples
        pos = torch.stack([torch.randint(0, 2, (P,), generator=g, device=dev), torch.randint(10, 14, (P,), generator=g, device=dev),
                           torch.randint(20, 20 + cells // 8, (P,), generator=g, device=dev)], -1)
    xy = (pos[:, 1:].float() + torch.rand(P, 2, generator=g, device=dev)) * 0.8 - 51.2
    anno = torch.cat([xy, torch.rand(P, 1, generator=g, device=dev) * 4 - 3, torch.rand(P, 3, generator=g, device=dev) * 2 + 0.5,
                      torch.rand(P, 1, generator=g, device=dev) * 6 - 3, torch.randn(P, 2, generator=g, device=dev)], -1)
    return pos, anno


def main():
    g = torch.Generator(device=dev).manual_seed(0)
    Bs, K, T = 8, 500, 6
    P = Bs * K
    coder = amd.CenterPointBBoxYawCoder(pc_range=[-51.2, -51.2], out_size_factor=4, voxel_size=[0.2, 0.2], norm_bbox=True)
    cfg = dict(pc_range=[-51.2, -51.2], out_size_factor=4, voxel_size=[0.2, 0.2], norm_bbox=True)
    gdcfg = dict(loss_type='bd3d', fun='log1p', tau=0.0, loss_weight=5.0)
    mod = amd.GDLoss('bd3d', fun='log1p', tau=0.0, loss_weight=5.0)
    l1cfg = dict(type='L1Loss', reduction='mean', loss_weight=0.25)
    cw = [1.0, 1.0, 0.2, 0.2]
    feat = torch.randn(Bs, 64, 128, 128, generator=g, device=dev)
    convs = [{h: torch.nn.Conv2d(64, n, 1).to(dev) for h, n in HEADS} for _ in range(T)]
    with torch.no_grad():
        for cs in convs:
            for c in cs.values():
                c.weight.mul_(0.3)
    data = [objects(g, Bs, P) for _ in range(T)]
    pos, ann = [p for p, _ in data], [a for _, a in data]

    def head_maps():
        return [{h: c(feat) for h, c in cs.items()} for cs in convs]

    def zero():
        for cs in convs:
            for c in cs.values():
                c.weight.grad = c.bias.grad = None

    def step_ours():
        zero()
        out = amd.center_head_losses(mod, l1cfg, coder, head_maps(), pos, ann, [P] * T, cw)
        sum(a + b for a, b in out).backward()
        return out

    def step_eager():
        zero()
        tot = 0
        for d, p, a in zip(head_maps(), pos, ann):
            l1, gd = head_torch.center_head_task_losses(d, p, a, P, cfg, gdcfg, 0.25, cw)
            tot = tot + l1 + gd
        tot.backward()

    def step_convs_only():
        zero()
        sum(sum(v.sum() for v in d.values()) for d in head_maps()).backward()

    us_o, us_e, us_c = timeit(step_ours, 30), timeit(step_eager, 10), timeit(step_convs_only, 30)
    # forward call alone (host + device), maps precomputed
    with torch.no_grad():
        fixed = [{k: v.contiguous() for k, v in d.items()} for d in head_maps()]
    us_f = timeit(lambda: amd.center_head_losses(mod, l1cfg, coder, fixed, pos, ann, [P] * T, cw), 50)
    print(json.dumps(dict(what='model_step: 36 1x1 convs (64 -> 1..3 channels, 8x128x128) + all regression losses of 6 CenterPoint tasks + backward',
                          objects_per_task=P, ours_us=round(us_o, 1), eager_torch_us=round(us_e, 1),
                          convs_and_a_sum_only_us=round(us_c, 1), loss_forward_call_only_us=round(us_f, 1))), flush=True)

    # continuity with round 2: the same maps as 36 leaves
    leaves = [{k: v.detach().clone().requires_grad_(True) for k, v in d.items()} for d in fixed]

    def step_leaf():
        for d in leaves:
            for v in d.values():
                v.grad = None
        out = amd.center_head_losses(mod, l1cfg, coder, leaves, pos, ann, [P] * T, cw)
        sum(a + b for a, b in out).backward()
        return out
    us_l = timeit(step_leaf, 50)
    keep = head_loss.CENTER_SORT_MIN_N
    head_loss.CENTER_SORT_MIN_N = 0          # what the sort between the two launches would cost at this size
    try:
        us_ls = timeit(step_leaf, 50)
    finally:
        head_loss.CENTER_SORT_MIN_N = keep
    print(json.dumps(dict(what='loss_only: the same maps as 36 leaf tensors (round 2 harness)', us_per_step=round(us_l, 1),
                          us_per_step_with_the_sorted_finish=round(us_ls, 1), sort_threshold_n=keep)), flush=True)
    a = amd.center_head_losses(mod, l1cfg, coder, fixed, pos, ann, [P] * T, cw)
    b = step_leaf()
    assert all(torch.equal(x[0], y[0]) and torch.equal(x[1], y[1]) for x, y in zip(a, b)), 'model-shaped and leaf-shaped losses differ'

    # batch-64 sizes, the pathological placement: 32 000 objects of every task in 64 cells
    P64 = 32_000
    big = [objects(g, 64, P64, cells=64) for _ in range(T)]
    maps64 = [{h: (torch.randn(64, n, 128, 128, generator=g, device=dev) * 0.3).requires_grad_(True) for h, n in HEADS} for _ in range(T)]

    def step64():
        for d in maps64:
            for v in d.values():
                v.grad = None
        out = amd.center_head_losses(mod, l1cfg, coder, maps64, [p for p, _ in big], [a for _, a in big], [P64] * T, cw)
        sum(x + y for x, y in out).backward()
        return [v.grad for d in maps64 for v in d.values()]
    keep = head_loss.CENTER_SORT_MIN_N
    try:
        head_loss.CENTER_SORT_MIN_N = 0
        gs = [x.clone() for x in step64()]
        us_s = timeit(step64, 10, warm=2)
        head_loss.CENTER_SORT_MIN_N = 10 ** 9
        gq = [x.clone() for x in step64()]
        us_q = timeit(step64, 3, warm=1)
    finally:
        head_loss.CENTER_SORT_MIN_N = keep
    assert all(torch.equal(x, y) for x, y in zip(gs, gq)), 'sorted and scanning accumulate differ'
    print(json.dumps(dict(what='sorted_finish: 6 tasks x 32 000 objects in 64 cells (batch 64), fwd+bwd from leaf maps',
                          sorted_us=round(us_s, 1), scanning_us=round(us_q, 1))), flush=True)


if __name__ == '__main__':
    main()
