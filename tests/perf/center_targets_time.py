#!/usr/bin/env python3
"""CenterPoint target assignment (gd_centerpoint_head.py:65-156), nuScenes geometry, 6 tasks: us per call of
center_head_get_targets (two launches + one read-back) against the reference's statement run on the CPU
(oracle/center_targets_torch.py: the per-sample / per-task / per-box Python loops with a numpy Gaussian per box; the reference runs
the same loops with device tensors and a host copy per box, which is slower still).  Asserts equal outputs.  JSON lines."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch  # noqa: E402

import mmdet3d_gaussian_amd as amd  # noqa: E402
from oracle import center_targets_torch as ct  # noqa: E402
from test_gpu_center_targets import NUS, TASKS, scene  # noqa: E402


def main():
    torch.set_num_threads(1)
    counts = [len(t) for t in TASKS]
    for B, n in ((4, 60), (8, 150), (8, 500)):
        g = torch.Generator().manual_seed(B * 1000 + n)
        data = [scene(g, n) for _ in range(B)]
        boxes, labels = [d[0] for d in data], [d[1] for d in data]
        gb, gl = [b.cuda() for b in boxes], [l.cuda() for l in labels]
        got = amd.extras.center_head_get_targets(gb, gl, TASKS, NUS)
        want = ct.get_targets(boxes, labels, counts, NUS)
        for t in range(len(TASKS)):
            assert torch.equal(got[0][t].cpu(), want[0][t]) and torch.equal(got[1][t].cpu(), want[1][t]) and torch.equal(got[2][t].cpu(), want[2][t])
        for _ in range(10):
            amd.extras.center_head_get_targets(gb, gl, TASKS, NUS)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(100):
            amd.extras.center_head_get_targets(gb, gl, TASKS, NUS)
        torch.cuda.synchronize()
        us = (time.perf_counter() - t0) / 100 * 1e6
        t0 = time.perf_counter()
        for _ in range(3):
            ct.get_targets(boxes, labels, counts, NUS)
        us_cpu = (time.perf_counter() - t0) / 3 * 1e6
        print(json.dumps(dict(what=f'get_targets, batch {B} x {n} boxes, 6 tasks, 128 x 128 maps', valid_boxes=int(sum(a.shape[0] for a in got[1])),
                              ours_us=round(us, 1), reference_statement_on_cpu_us=round(us_cpu, 1))), flush=True)


if __name__ == '__main__':
    main()
