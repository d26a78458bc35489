"""The CenterPoint head's three device-side slices working together on a toy head: target assignment -> loss (heat-map + regression)
-> backward through convolutions -> optimiser steps -> inference (top-K, decode, NMS, merge).  An integration check of the call
surfaces (shapes, dtypes, devices, autograd), not a parity test: every slice has its own."""
import pytest
import torch

import mmdet3d_gaussian_amd as amd

pytestmark = pytest.mark.extras   # frozen extras outside SURVEY.md §8: `pytest -m extras` on a GPU box (conftest.py), not part of `-m gpu`


def test_train_a_toy_center_head_then_detect():
    dev = torch.device('cuda:0')
    torch.manual_seed(0)
    tasks = [['car'], ['truck', 'bus'], ['pedestrian']]
    B, H, W = 2, 64, 64
    train_cfg = dict(grid_size=[256, 256, 1], point_cloud_range=[-25.6, -25.6, -5.0, 25.6, 25.6, 3.0], voxel_size=[0.2, 0.2, 8],
                     out_size_factor=4, gaussian_overlap=0.1, min_radius=2, code_weights=[1.0, 1.0, 0.2, 0.2])
    test_cfg = dict(post_center_limit_range=[-30.0, -30.0, -10.0, 30.0, 30.0, 10.0], max_per_img=50, score_threshold=0.3,
                    nms_type='rotate', nms_thr=0.2, pre_max_size=100, post_max_size=20)
    coder = amd.CenterPointBBoxYawCoder(pc_range=[-25.6, -25.6], out_size_factor=4, voxel_size=[0.2, 0.2], norm_bbox=True)
    loss_gd = amd.build_loss(dict(type='GDLoss', loss_type='gwd3d', fun='log1p', tau=0.0, loss_weight=2.0))
    loss_l1 = dict(type='L1Loss', reduction='mean', loss_weight=0.25)
    loss_cls = dict(type='GaussianFocalLoss', reduction='mean')
    heads = (('heatmap', None), ('reg', 2), ('height', 1), ('dim', 3), ('yaw', 1), ('dir', 2), ('vel', 2))
    convs = torch.nn.ModuleList([torch.nn.ModuleDict({k: torch.nn.Conv2d(16, c if c else len(names), 3, padding=1) for k, c in heads})
                                 for names in tasks]).to(dev)
    for m in convs:
        torch.nn.init.constant_(m['heatmap'].bias, -2.19)                      # the usual focal-loss prior
    feat = torch.randn(B, 16, H, W, device=dev)
    g = torch.Generator().manual_seed(1)
    gt_boxes, gt_labels = [], []
    for _ in range(B):
        n = 12
        xy = torch.rand(n, 2, generator=g) * 44 - 22
        box = torch.cat([xy, torch.rand(n, 1, generator=g) - 1.5, torch.rand(n, 3, generator=g) * 2 + 1.0,
                         torch.rand(n, 1, generator=g) * 6.28 - 3.14, torch.zeros(n, 2)], 1)
        gt_boxes.append(box.to(dev))
        gt_labels.append(torch.randint(0, 4, (n,), generator=g).to(dev))

    def forward():
        return [{k: m[k](feat) for k, _ in heads} for m in convs]
    opt = torch.optim.Adam(convs.parameters(), lr=2e-2)
    history = []
    for it in range(60):
        opt.zero_grad()
        loss_dict = amd.extras.center_gd_head_loss(loss_cls, loss_l1, loss_gd, coder, tasks, train_cfg, gt_boxes, gt_labels,
                                            tuple([p] for p in forward()), static=bool(it % 2))
        total = sum(loss_dict.values())
        assert torch.isfinite(total)
        total.backward()
        opt.step()
        history.append(total.item())
    assert history[-1] < 0.35 * history[0], (history[0], history[-1])
    with torch.no_grad():
        dets = amd.extras.center_head_get_bboxes(tuple([p] for p in forward()), coder, test_cfg, [len(t) for t in tasks])
    assert len(dets) == B
    found = 0
    for b, (boxes, scores, labels) in enumerate(dets):
        assert boxes.shape[1] == 9 and boxes.shape[0] == scores.shape[0] == labels.shape[0]
        assert labels.dtype == torch.int32 and int(labels.min()) >= 0 and int(labels.max()) <= 3
        # the toy head has memorised its two samples: most ground-truth centres are detected within a cell
        d = torch.cdist(gt_boxes[b][:, :2], boxes[:, :2])
        found += int((d.min(dim=1)[0] < 1.0).sum())
    assert found >= 16, found
