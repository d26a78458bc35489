#!/usr/bin/env python3
"""2000 calls each of the CenterPoint and anchor-head slices with FRESH tensors every time (new addresses defeat the descriptor caches): device
memory in use and host RSS before / after.  Asserts no growth beyond noise."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import psutil  # noqa: E402
import torch  # noqa: E402

import mmdet3d_gaussian_amd as amd  # noqa: E402
from test_gpu_center_infer import NUS, NUS_TEST, make_tasks  # noqa: E402
from test_gpu_center_targets import NUS as TCFG, TASKS, scene  # noqa: E402
from test_gpu_anchor_targets import CE, FOCAL, SL1, TRAIN_CFG, head_outputs, kitti_anchors, random_gt  # noqa: E402

dev = torch.device('cuda:0')
g = torch.Generator().manual_seed(0)
classes = [1, 2, 2, 1, 2, 2]
cpu_tasks = make_tasks(g, 1, 64, 64, classes, 'yaw')
coder = amd.CenterPointBBoxYawCoder(**NUS)
cfg = dict(NUS_TEST, max_per_img=200)
boxes, labels = scene(g, 80)
tcfg = dict(TCFG, grid_size=[256, 256, 1], code_weights=[1.0, 1.0, 0.2, 0.2])
gd = amd.GDLoss('gwd3d', fun='log1p', tau=0.0, loss_weight=5.0)
proc = psutil.Process()
AH, AW = 40, 36
a_anchors = kitti_anchors(AH, AW)
a_outs = head_outputs(2, AH, AW, seed=1)
a_gt = [random_gt(12, seed=5), random_gt(7, seed=6)]
a_infer_cfg = dict(use_rotate_nms=True, nms_pre=512, nms_thr=0.01, score_thr=0.05, max_num=50)
kld = amd.GDLoss('kld3d', fun='log1p', tau=1.0, loss_weight=5.0)


def once():
    tasks = [{k: v.to(dev) + 0.0 for k, v in pd.items()} for pd in cpu_tasks]          # fresh tensors: new addresses
    amd.extras.center_head_get_bboxes(tasks, coder, cfg, classes)
    hm, an, pi = amd.extras.center_head_get_targets([boxes.to(dev)], [labels.to(dev)], TASKS, tcfg)
    pds = [{k: v.requires_grad_(True) for k, v in t.items()} for t in tasks]
    out = amd.extras.center_gd_head_loss(dict(type='GaussianFocalLoss'), dict(type='L1Loss', loss_weight=0.25), gd, coder, TASKS, tcfg,
                                  [boxes.to(dev)], [labels.to(dev)], pds, static=True)
    sum(out.values()).backward()
    an = a_anchors.to(dev) + 0.0
    o = [t.to(dev).requires_grad_(True) for t in a_outs]
    res = amd.extras.gd_anchor_head_loss(FOCAL, SL1, CE, kld, TRAIN_CFG, 3, an, [o[0]], [o[1]], [o[2]], [b.to(dev) for b, _ in a_gt], [l.to(dev) for _, l in a_gt])
    (res['loss_cls'][0] + res['loss_bbox'][0] + res['loss_dir'][0]).backward()
    amd.extras.anchor_head_get_bboxes([o[0].detach()], [o[1].detach()], [o[2].detach()], [an.reshape(-1, 7)], a_infer_cfg, 3, 0.0, 1.0)


for _ in range(200):
    once()
torch.cuda.synchronize()
m0, r0 = torch.cuda.memory_allocated(), proc.memory_info().rss
for _ in range(2000):
    once()
torch.cuda.synchronize()
m1, r1 = torch.cuda.memory_allocated(), proc.memory_info().rss
print(f'device bytes in use {m0} -> {m1}; host RSS {r0 / 1e6:.1f} MB -> {r1 / 1e6:.1f} MB')
assert m1 - m0 < 8 << 20, 'device memory grows'
assert r1 - r0 < 64 << 20, 'host memory grows'
