#!/usr/bin/env python3
"""Per-phase cycle stamps of the NMS greedy scan (profiling build: tools/build_variants.py prof="-DSCAN_PROFILE=1",
GD3D_LIB=tools/variants/libgd3d_prof.so).  The scan writes clock64() stamps into the (dead) OBox part of the workspace.

Default: the LIST scan (every call of <= 16384 boxes per group): resolver stamps per block (top, state bytes read, in-block fixed
point done, victims marked + next fields fetched), how often it had to wait for the ring / for the far victims, the helper waves'
phases per block, and the HW_ID of the workgroup's sixteen waves (which waves share a SIMD).
RNMS_LIST_MIN_THR=2 in the environment switches the list scan off: the CLASSIC scan's stamps (round 4 / 5 layout)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch
import mmdet3d_gaussian_amd as amd
from rbox_inputs import nms_boxes
lib = amd.load_library()
classic = float(os.environ.get('RNMS_LIST_MIN_THR', '0')) > 1.0
for n, thr, clutter in ((1000, 0.2, True), (4096, 0.25, True), (4096, 0.25, False), (9000, 0.7, True)):
    b, s = nms_boxes(n, seed=n, clutter=clutter)
    boxes = torch.from_numpy(b).cuda(); order = torch.from_numpy(s).cuda().sort(descending=True)[1].contiguous()
    keep = torch.empty(n, dtype=torch.int64, device='cuda'); num = torch.empty(1, dtype=torch.int64, device='cuda')
    ws = torch.zeros(lib.rnms_workspace_bytes(n), dtype=torch.uint8, device='cuda')
    for _ in range(3):
        lib.rnms_bev_ordered(boxes.data_ptr(), order.data_ptr(), n, thr, keep.data_ptr(), num.data_ptr(), ws.data_ptr(), None)
    torch.cuda.synchronize()
    cb = (n + 63) // 64
    allst = ws[:(2 * cb + 17) * 16 * 8].view(torch.int64).cpu().numpy().reshape(-1, 16).astype(np.float64)
    st = allst[:cb]
    iv = np.diff(st[:, 0])                       # resolver: block start -> next block start
    if classic:
        d = lambda a, b, rows=slice(1, -1): np.median(st[rows, b] - st[rows, a])
        grp = slice(3, cb - 6, 3)                # wave 1 (group 0, rank 0) stamps live at rows t0 = 0, 3, 6, ...
        print(f'CLASSIC n={n} clutter={clutter} kept={int(num)} blocks={cb}: interval {np.median(iv):.0f} cyc | resolver: lds-read {d(0,1):.0f} '
              f'solve {d(1,2):.0f} store+or {d(2,3):.0f} barrier {d(3,5):.0f} | group wave: issue {d(8,10,grp):.0f} 2 barriers {d(10,11,grp):.0f} '
              f'wait+consume {d(11,9,grp):.0f} barrier {d(9,12,grp):.0f} || issue split: lds {d(8,13,grp):.0f} fields+rows {d(13,15,grp):.0f} rest {d(15,10,grp):.0f}', flush=True)
        continue
    hs = allst[cb + 8:cb + 8 + cb]
    hw = allst[2 * cb + 16].astype(np.int64)
    d = lambda a, b: np.median(st[8:-2, b] - st[8:-2, a])
    ring, fdw = st[:, 6], st[:, 7]
    steady = iv[8:]
    print(f'LIST n={n} thr={thr} clutter={clutter} kept={int(num)} blocks={cb}: block median {np.median(steady):.0f} mean {steady.mean():.0f} cyc '
          f'(first 8 blocks: {" ".join(f"{x:.0f}" for x in iv[:8])}) | resolver: state bytes {d(0,1):.0f} in-block {d(1,2):.0f} '
          f'victims+fetch {d(2,3):.0f} tail {np.median(st[9:-1, 0] - st[8:-2, 3]):.0f} | waited for the ring in {(ring > 0).sum()} blocks, '
          f'for far victims in {(fdw > 0).sum()} blocks', flush=True)
    hd = lambda a, b: np.median(hs[8:-4, b] - hs[8:-4, a])
    print(f'     helper wave per block: loads issued -> block resolved {hd(0,1):.0f} | far victims {hd(1,2):.0f} | count {hd(2,3):.0f} | ring store {hd(3,4):.0f}'
          f' || SIMD of waves 0..15: {[(int(h) >> 4) & 3 for h in hw]}', flush=True)
