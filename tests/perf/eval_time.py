#!/usr/bin/env python3
"""Evaluation helpers on the device: affinity (iou_3d) + match_coco, vs the CPU oracle (and the
compiled reference when oracle/_ref is present).  Context numbers, not bench lines."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import mmdet3d_gaussian_amd as amd  # noqa: E402
import oracle  # noqa: E402
from rbox_inputs import eval_boxes  # noqa: E402

ev = oracle.load_ref_eval()
thrs = [0.3, 0.5, 0.7]
for D, G in ((300, 40), (3000, 60), (20000, 200)):
    det = eval_boxes(D, seed=1, spread=40.0); gt = eval_boxes(G, seed=2, spread=40.0)
    ign = np.zeros(G, bool); ign[::5] = True; crowd = np.zeros(G, bool)
    d, g = torch.from_numpy(det).cuda(), torch.from_numpy(gt).cuda()
    ig, cr = torch.from_numpy(ign).cuda(), torch.from_numpy(crowd).cuda()
    nthr = -torch.tensor(thrs, dtype=torch.float32, device='cuda')

    def gpu():
        return amd.match_coco(-amd.iou_3d(d, g, 0.5), nthr, ig, cr)
    for _ in range(5):
        gpu()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20):
        out = gpu()
    torch.cuda.synchronize(); t_gpu = (time.perf_counter() - t0) / 20 * 1e6
    t0 = time.perf_counter()
    aff = oracle.eval_iou_3d(det, gt, 0.5)
    want = oracle.match_coco(-aff, -np.array(thrs, np.float32), ign, crowd)
    t_cpu = (time.perf_counter() - t0) * 1e6
    row = dict(dets=D, gts=G, thresholds=len(thrs), gpu_us=round(t_gpu, 1), cpu_oracle_us=round(t_cpu, 1),
               matches_equal=bool(np.array_equal(out.cpu().numpy(), want)))
    if ev is not None and hasattr(ev, 'match_coco'):
        t0 = time.perf_counter()
        a = ev.iou_3d(det, gt, 0.5); ev.match_coco(-a, -np.array(thrs, np.float32), ign, crowd)
        row['cpu_reference_us'] = round((time.perf_counter() - t0) * 1e6, 1)
    print(json.dumps(row), flush=True)
    assert row['matches_equal'], 'GPU matches differ from the CPU oracle'
