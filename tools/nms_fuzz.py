#!/usr/bin/env python3
"""Fuzz of the NMS keep lists against the CPU oracle, and a determinism check (the list scan synchronises waves through LDS flags:
a race would show as a rare, timing-dependent difference).
  part 1: random problems — n in [768, 16384] (list scan) and a few below / above, random threshold, clustered or sparse, rotated or
          axis-aligned, pre-sorted C entry and nms_gpu;
  part 2: the same problem 300 times on the same buffers: every keep list identical to the first.
usage: tools/nms_fuzz.py [problems=200] > profiles/rNN_nms_fuzz.txt"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch
import mmdet3d_gaussian_amd as amd
import oracle
from rbox_inputs import nms_boxes

P = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(2026)
lib = amd.load_library()
bad = 0
t0 = time.time()
sizes = []
for p in range(P):
    u = rng.uniform()
    n = int(rng.integers(768, 6000)) if u < 0.8 else (int(rng.integers(6000, 16385)) if u < 0.93 else int(rng.integers(1, 768)) if u < 0.97 else int(rng.integers(16385, 20000)))
    thr = float(np.round(rng.choice([rng.uniform(0.05, 0.95), 0.1, 0.25, 0.5, 0.7]), 3))
    clutter = bool(rng.integers(0, 2))
    normal = bool(rng.uniform() < 0.2)
    boxes, scores = nms_boxes(n, seed=int(rng.integers(1 << 30)), clutter=clutter)
    if rng.uniform() < 0.3:   # ties in the scores
        scores = np.round(scores, 2)
    b, s = torch.from_numpy(boxes).cuda(), torch.from_numpy(scores).cuda()
    want = oracle.nms_gpu_oracle(boxes, scores, thr, normal=normal)
    got = (amd.nms_normal_gpu(b, s, thr) if normal else amd.nms_gpu(b, s, thr)).cpu().numpy()
    ok = np.array_equal(got, want)
    if ok and n <= 65536:
        order = torch.sort(s, dim=0, descending=True, stable=True)[1]
        sb = b[order].contiguous()
        keep = torch.empty(n, dtype=torch.int64, device='cuda'); num = torch.zeros(1, dtype=torch.int64, device='cuda')
        ws = torch.full((lib.rnms_workspace_bytes(n),), 0xA5, dtype=torch.uint8, device='cuda')
        f = lib.rnms_normal_bev if normal else lib.rnms_bev
        assert f(sb.data_ptr(), n, thr, keep.data_ptr(), num.data_ptr(), ws.data_ptr(), None) == 0
        ok = np.array_equal(order[keep[:int(num.item())]].cpu().numpy(), want)
    sizes.append(n)
    if not ok:
        bad += 1
        print(f'MISMATCH problem {p}: n={n} thr={thr} clutter={clutter} normal={normal}', flush=True)
print(f'part 1: {P} random problems (n {min(sizes)}..{max(sizes)}, {sum(768 <= x <= 16384 for x in sizes)} in the list scan\'s range), '
      f'{bad} mismatches against the CPU oracle, {time.time() - t0:.0f} s', flush=True)

rep_bad = 0
for n, thr, clutter in ((9000, 0.7, True), (4096, 0.25, True), (16384, 0.5, True), (4096, 0.25, False)):
    boxes, scores = nms_boxes(n, seed=n, clutter=clutter)
    b, s = torch.from_numpy(boxes).cuda(), torch.from_numpy(scores).cuda()
    order = torch.sort(s, dim=0, descending=True, stable=True)[1]
    sb = b[order].contiguous()
    keep = torch.empty(n, dtype=torch.int64, device='cuda'); num = torch.zeros(1, dtype=torch.int64, device='cuda')
    ws = torch.empty(lib.rnms_workspace_bytes(n), dtype=torch.uint8, device='cuda')
    first = None
    for r in range(300):
        lib.rnms_bev(sb.data_ptr(), n, thr, keep.data_ptr(), num.data_ptr(), ws.data_ptr(), None)
        k = keep[:int(num.item())].clone()
        if first is None:
            first = k
            assert np.array_equal(order[k].cpu().numpy(), oracle.nms_gpu_oracle(boxes, scores, thr))
        elif not torch.equal(k, first):
            rep_bad += 1
    print(f'part 2: n={n} thr={thr} clutter={clutter}: 300 calls on the same buffers, {rep_bad} differ from the first (which equals the oracle)', flush=True)
print('RESULT:', 'clean' if bad == 0 and rep_bad == 0 else 'FAILURES')
sys.exit(0 if bad == 0 and rep_bad == 0 else 1)
