#!/usr/bin/env python3
"""cProfile of the host side of center_head_losses (6 tasks, 36 head maps as leaves, forward + backward): where the ~600 us per
step go that are not kernel time (~30 us).  usage: tools/center_head_host_profile.py"""
import cProfile
import os
import pstats
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import mmdet3d_gaussian_amd as amd  # noqa: E402

dev = torch.device('cuda:0')
g = torch.Generator(device=dev).manual_seed(0)
coder = amd.CenterPointBBoxYawCoder(pc_range=[-51.2, -51.2], out_size_factor=4, voxel_size=[0.2, 0.2], norm_bbox=True)
Bs, K, tasks = 8, 500, 6
modb = amd.GDLoss('bd3d', fun='log1p', tau=0.0, loss_weight=5.0)
data = []
for _ in range(tasks):
    P = Bs * K
    pos = torch.stack([torch.randint(0, Bs, (P,), generator=g, device=dev), torch.randint(0, 128, (P,), generator=g, device=dev),
                       torch.randint(0, 128, (P,), generator=g, device=dev)], -1)
    xy = (pos[:, 1:].float() + torch.rand(P, 2, generator=g, device=dev)) * 0.8 - 51.2
    anno = torch.cat([xy, torch.rand(P, 1, generator=g, device=dev) * 4 - 3, torch.rand(P, 3, generator=g, device=dev) * 2 + 0.5,
                      torch.rand(P, 1, generator=g, device=dev) * 6 - 3, torch.randn(P, 2, generator=g, device=dev)], -1)
    data.append((pos, anno))
maps4 = [{k: (torch.randn(Bs, c, 128, 128, generator=g, device=dev) * 0.3).requires_grad_(True)
          for k, c in (('reg', 2), ('height', 1), ('dim', 3), ('yaw', 1), ('dir', 2), ('vel', 2))} for _ in range(tasks)]
l1cfg = dict(type='L1Loss', reduction='mean', loss_weight=0.25)
cw4 = [1.0, 1.0, 0.2, 0.2]


def fwd():
    return amd.center_head_losses(modb, l1cfg, coder, maps4, [p for p, _ in data], [a for _, a in data], [Bs * K] * tasks, cw4)


def step():
    for d in maps4:
        for v in d.values():
            v.grad = None
    out = fwd()
    sum(a + b for a, b in out).backward()


for _ in range(20):
    step()
torch.cuda.synchronize()
for name, fn in (('forward only', fwd), ('forward + backward', step)):
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(200):
        fn()
    torch.cuda.synchronize()
    pr.disable()
    print('====', name)
    pstats.Stats(pr).sort_stats('tottime').print_stats(12)
