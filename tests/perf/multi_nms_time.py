#!/usr/bin/env python3
"""nms_gpu_multi (one segmented call) vs the per-entry loop of nms_gpu, G entries x 500 boxes (nuScenes test_cfg sizes:
B samples x 6 tasks).  ADVICE r01: the dense (G, G n) form grew with G^2; the segmented form must not."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
import mmdet3d_gaussian_amd as amd
from rbox_inputs import nms_boxes
for G in (6, 24, 48):
    bl, sl = [], []
    for k in range(G):
        b, s = nms_boxes(500, seed=k, extent=30.0)
        bl.append(torch.from_numpy(b).cuda()); sl.append(torch.from_numpy(s).cuda())
    def loop(): return [amd.nms_gpu(b, s, 0.2, pre_max_size=1000, post_max_size=83) for b, s in zip(bl, sl)]
    def multi(): return amd.nms_gpu_multi(bl, sl, 0.2, pre_max_size=1000, post_max_size=83)
    same = all(torch.equal(a, b) for a, b in zip(loop(), multi()))
    res = {}
    for f in (loop, multi):
        for _ in range(5): f()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(30): f()
        torch.cuda.synchronize(); res[f.__name__] = (time.perf_counter() - t0) / 30 * 1e6
    print(json.dumps(dict(entries=G, boxes_per_entry=500, per_entry_loop_us=round(res['loop'], 1),
                          one_segmented_call_us=round(res['multi'], 1), identical=same)), flush=True)
    assert same, 'segmented call and per-entry loop disagree'
