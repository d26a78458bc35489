#!/usr/bin/env python3
"""cProfile of the host side of center_head_get_bboxes (6 tasks, nuScenes geometry) and the per-call time of the padded form.
usage: tests/perf/center_infer_host_profile.py   (lives under tests/: its inputs come from the test module, which uses the oracle)"""
import sys, cProfile, pstats
import os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch, mmdet3d_gaussian_amd as amd
from test_gpu_center_infer import NUS, NUS_TEST, make_tasks
dev = torch.device('cuda:0')
g = torch.Generator().manual_seed(3)
classes = [1, 2, 2, 1, 2, 2]
tasks = [{k: v.to(dev) for k, v in pd.items()} for pd in make_tasks(g, 1, 128, 128, classes, 'yaw')]
coder = amd.CenterPointBBoxYawCoder(**NUS)
for _ in range(50): amd.extras.center_head_get_bboxes(tasks, coder, NUS_TEST, classes)
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(500): amd.extras.center_head_get_bboxes(tasks, coder, NUS_TEST, classes)
pr.disable()
pstats.Stats(pr).sort_stats('tottime').print_stats(14)
import time
t=time.perf_counter()
for _ in range(500): amd.extras.center_head_get_bboxes(tasks, coder, NUS_TEST, classes, padded=True)
torch.cuda.synchronize(); print('padded us/call', (time.perf_counter()-t)/500*1e6)
