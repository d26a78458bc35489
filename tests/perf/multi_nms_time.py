import sys, time; sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import numpy as np, torch
import mmdet3d_gaussian_amd as amd
from rbox_inputs import nms_boxes
bl, sl = [], []
for k in range(6):
    b, s = nms_boxes(500, seed=k, extent=30.0)
    bl.append(torch.from_numpy(b).cuda()); sl.append(torch.from_numpy(s).cuda())
def loop(): return [amd.nms_gpu(b, s, 0.2, pre_max_size=1000, post_max_size=83) for b, s in zip(bl, sl)]
def multi(): return amd.nms_gpu_multi(bl, sl, 0.2, pre_max_size=1000, post_max_size=83)
for f in (loop, multi):
    for _ in range(5): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(50): f()
    torch.cuda.synchronize(); print(f.__name__, f'{(time.perf_counter()-t0)/50*1e6:.1f} us for 6 tasks x 500 boxes')
