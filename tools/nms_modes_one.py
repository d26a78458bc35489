"""Axis-aligned and circle NMS, 40 calls each at n = 4096 (workload for rocprofv3 + tools/nms_kstats.py): which kernels, how long."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch
import mmdet3d_gaussian_amd as amd
from rbox_inputs import nms_boxes
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
boxes, scores = nms_boxes(n, seed=n, clutter=True)
b, s = torch.from_numpy(boxes).cuda(), torch.from_numpy(scores).cuda()
ctr = ((b[:, :2] + b[:, 2:4]) / 2).contiguous()
for _ in range(40):
    k1 = amd.nms_normal_gpu(b, s, 0.25)
for _ in range(40):
    k2 = amd.circle_nms(torch.cat([ctr, s[:, None]], 1), 1.0, 4096)
torch.cuda.synchronize()
print('normal kept', len(k1), 'circle kept', None if k2 is None else len(k2))
