#!/usr/bin/env python3
"""Per-phase cycle stamps of the NMS greedy scan (profiling build: tools/build_variants.py prof="-DSCAN_PROFILE=1",
GD3D_LIB=tools/variants/libgd3d_prof.so).  The scan writes clock64() stamps into the (dead) OBox part of the workspace."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch
import mmdet3d_gaussian_amd as amd
from rbox_inputs import nms_boxes
lib = amd.load_library()
for n, thr, clutter in ((1000, 0.2, True), (4096, 0.25, True), (4096, 0.25, False), (9000, 0.7, True)):
    b, s = nms_boxes(n, seed=n, clutter=clutter)
    boxes = torch.from_numpy(b).cuda(); order = torch.from_numpy(s).cuda().sort(descending=True)[1].contiguous()
    keep = torch.empty(n, dtype=torch.int64, device='cuda'); num = torch.empty(1, dtype=torch.int64, device='cuda')
    ws = torch.zeros(lib.rnms_workspace_bytes(n), dtype=torch.uint8, device='cuda')
    for _ in range(3):
        lib.rnms_bev_ordered(boxes.data_ptr(), order.data_ptr(), n, thr, keep.data_ptr(), num.data_ptr(), ws.data_ptr(), None)
    torch.cuda.synchronize()
    cb = (n + 63) // 64
    st = ws[:cb * 16 * 8 + 16 * 8].view(torch.int64).cpu().numpy().reshape(-1, 16)[:cb].astype(np.float64)
    iv = np.diff(st[:, 0])                       # resolver: interval start -> next interval start
    d = lambda a, b, rows=slice(1, -1): np.median(st[rows, b] - st[rows, a])
    grp = slice(3, cb - 6, 3)                    # wave 1 (group 0, rank 0) stamps live at rows t0 = 0, 3, 6, ...
    print(f'n={n} clutter={clutter} kept={int(num)} blocks={cb}: interval {np.median(iv):.0f} cyc | resolver: lds-read {d(0,1):.0f} '
          f'solve {d(1,2):.0f} store+or {d(2,3):.0f} barrier {d(3,5):.0f} | group wave: issue {d(8,10,grp):.0f} 2 barriers {d(10,11,grp):.0f} '
          f'wait+consume {d(11,9,grp):.0f} barrier {d(9,12,grp):.0f} || issue split: lds {d(8,13,grp):.0f} fields+rows {d(13,15,grp):.0f} rest {d(15,10,grp):.0f}', flush=True)
