#!/usr/bin/env python3
"""A short driver for rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE: separate runs): four launches each of the anchor heads' classification
kernel and target-assignment kernels at KITTI geometry (6 x 321 408 anchors), nothing else.  tools/collect_profiles.sh runs it and
summarises the counters with tools/pmc_summary.py into <round>_anchor_pmc.txt."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
import mmdet3d_gaussian_amd as amd
from test_gpu_anchor_cls import CE, FOCAL, make
from test_gpu_anchor_targets import KITTI_ASSIGNERS, kitti_anchors, random_gt
dev = torch.device('cuda:0')
B, A, C, H, W = 6, 6, 3, 248, 216
cls, dirs, labels, lw, dt, dw = [t.to(dev) for t in make(B, A, C, H, W, seed=1, pos_frac=0.002)]
cls.requires_grad_(True); dirs.requires_grad_(True)
anchors = kitti_anchors(H, W).to(dev)
pairs = [random_gt(24, seed=100 + i, with_ignored=False) for i in range(B)]
gts, gl = [p[0].to(dev) for p in pairs], [p[1].to(dev) for p in pairs]
for _ in range(4):
    amd.anchor_head_cls_dir_loss(FOCAL, CE, cls, dirs, labels, lw, dt, dw, C, 100.0)
    amd.anchor_head_get_targets(anchors, gts, gl, KITTI_ASSIGNERS, 3, padded=True)
torch.cuda.synchronize()
