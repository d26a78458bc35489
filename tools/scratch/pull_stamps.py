import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch
import mmdet3d_gaussian_amd as amd
from rbox_inputs import nms_boxes
lib = amd.load_library()
for n, thr in ((9000, 0.7), (4096, 0.25)):
    b, s = nms_boxes(n, seed=n, clutter=True)
    boxes = torch.from_numpy(b).cuda(); order = torch.from_numpy(s).cuda().sort(descending=True)[1].contiguous()
    keep = torch.empty(n, dtype=torch.int64, device='cuda'); num = torch.empty(1, dtype=torch.int64, device='cuda')
    ws = torch.zeros(lib.rnms_workspace_bytes(n), dtype=torch.uint8, device='cuda')
    for _ in range(3):
        lib.rnms_bev_ordered(boxes.data_ptr(), order.data_ptr(), n, thr, keep.data_ptr(), num.data_ptr(), ws.data_ptr(), None)
    torch.cuda.synchronize()
    cb = (n + 63) // 64
    hw = ws[(2 * cb + 16) * 16 * 8:(2 * cb + 17) * 16 * 8].view(torch.int64).cpu().numpy()
    print('  HW_ID per wave (wave_id, simd, pipe, cu, sh, se):', [(int(h) & 15, (int(h) >> 4) & 3, (int(h) >> 6) & 3, (int(h) >> 8) & 15, (int(h) >> 12) & 1, (int(h) >> 13) & 7) for h in hw])
    allst = ws[:(2 * cb + 16) * 16 * 8].view(torch.int64).cpu().numpy().reshape(-1, 16).astype(np.float64)
    st = allst[:cb]; hs = allst[cb + 8:cb + 8 + cb]
    iv = np.diff(st[:, 0])
    d = lambda a, b: np.median(st[8:-2, b] - st[8:-2, a])
    ring = st[:, 6]; fdw = st[:, 7]
    print(f'n={n} thr={thr} kept={int(num)} blocks={cb}: interval median {np.median(iv):.0f} mean {iv.mean():.0f} (min {iv.min():.0f} max {iv.max():.0f}) | state {d(0,1):.0f} fixedpoint {d(1,2):.0f} victims {d(2,3):.0f} | ring-wait blocks {(ring > 0).sum()} spins {ring.sum():.0f} | fdone-wait blocks {(fdw > 0).sum()} spins {fdw.sum():.0f}')
    print('  intervals:', ' '.join(f'{x:.0f}' for x in iv[:48]))
