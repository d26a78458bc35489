#!/usr/bin/env python3
"""Rotated BEV NMS timing: GPU (C ABI kernels only, and nms_gpu end to end incl. sort + count sync) vs the CPU oracle.
SURVEY.md §8d config 5 stand-in: Waymo-like boxes, thr 0.25; nuScenes: n=1000 thr 0.2."""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch
import mmdet3d_gaussian_amd as amd, oracle
from rbox_inputs import nms_boxes
lib = amd.load_library()
vp = lambda t: ctypes.c_void_p(t.data_ptr())
for n, thr, clutter in ((1000, 0.2, True), (4096, 0.25, True), (4096, 0.25, False), (9000, 0.7, True)):
    boxes, scores = nms_boxes(n, seed=n, clutter=clutter)
    b, s = torch.from_numpy(boxes).cuda(), torch.from_numpy(scores).cuda()
    order = s.sort(0, descending=True)[1]; sb = b[order].contiguous()
    keep = torch.empty(n, dtype=torch.int64, device='cuda'); num = torch.zeros(1, dtype=torch.int64, device='cuda')
    ws = torch.empty(lib.rnms_workspace_bytes(n), dtype=torch.uint8, device='cuda')
    call = lambda: lib.rnms_bev(vp(sb), n, thr, vp(keep), vp(num), vp(ws), None)
    for _ in range(3): call()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    it = 20
    e0.record()
    for _ in range(it): call()
    e1.record(); torch.cuda.synchronize()
    k_us = e0.elapsed_time(e1) / it * 1e3
    for _ in range(3): amd.nms_gpu(b, s, thr)     # (the first call loads the C++ glue module)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(it): k = amd.nms_gpu(b, s, thr)
    torch.cuda.synchronize(); e2e_us = (time.perf_counter() - t0) / it * 1e6
    t0 = time.perf_counter(); want = oracle.nms_gpu_oracle(boxes, scores, thr); cpu_us = (time.perf_counter() - t0) * 1e6
    ok = np.array_equal(k.cpu().numpy(), want)
    print(f'n={n:5d} thr={thr} clutter={clutter}: kept {len(want):5d}  GPU kernels {k_us:8.1f} us ({n / k_us:7.2f} Mboxes/s)  '
          f'nms_gpu e2e {e2e_us:8.1f} us  CPU oracle(1 thread) {cpu_us:10.0f} us  keep bit-exact={ok}', flush=True)
    assert ok, 'keep indices differ from the oracle'
