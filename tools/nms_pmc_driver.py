#!/usr/bin/env python3
"""A short driver for rocprofv3 --pmc passes over the rotated NMS: five rnms_bev calls (prep + mask + scan kernels) on the
BASELINE configs[4] stand-in (n = 4096 clustered Waymo-like boxes, thr 0.25), nothing else.  tools/nms_pmc.sh runs it."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
import mmdet3d_gaussian_amd as amd
from rbox_inputs import nms_boxes

lib = amd.load_library()
n, thr = 4096, 0.25
boxes, scores = nms_boxes(n, seed=n, clutter=len(sys.argv) < 2 or sys.argv[1] != 'sparse')
b, s = torch.from_numpy(boxes).cuda(), torch.from_numpy(scores).cuda()
sb = b[s.sort(0, descending=True)[1]].contiguous()
keep = torch.empty(n, dtype=torch.int64, device='cuda'); num = torch.zeros(1, dtype=torch.int64, device='cuda')
ws = torch.empty(lib.rnms_workspace_bytes(n), dtype=torch.uint8, device='cuda')
for _ in range(5):
    lib.rnms_bev(sb.data_ptr(), n, thr, keep.data_ptr(), num.data_ptr(), ws.data_ptr(), None)
torch.cuda.synchronize()
