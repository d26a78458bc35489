import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch
import mmdet3d_gaussian_amd as amd
from rbox_inputs import nms_boxes
lib = amd.load_library()
for n, thr in ((9000, 0.7), (4096, 0.5)):
    b, s = nms_boxes(n, seed=n, clutter=True)
    boxes = torch.from_numpy(b).cuda(); order = torch.from_numpy(s).cuda().sort(descending=True)[1].contiguous()
    keep = torch.empty(n, dtype=torch.int64, device='cuda'); num = torch.empty(1, dtype=torch.int64, device='cuda')
    ws = torch.zeros(lib.rnms_workspace_bytes(n), dtype=torch.uint8, device='cuda')
    for _ in range(3):
        lib.rnms_bev_ordered(boxes.data_ptr(), order.data_ptr(), n, thr, keep.data_ptr(), num.data_ptr(), ws.data_ptr(), None)
    torch.cuda.synchronize()
    cb = (n + 63) // 64
    st = ws[:cb * 16 * 8 + 16 * 8].view(torch.int64).cpu().numpy().reshape(-1, 16)[:cb].astype(np.float64)
    iv = np.diff(st[:, 0])
    d = lambda a, b: np.median(st[8:-2, b] - st[8:-2, a])
    print(f'n={n} thr={thr} kept={int(num)} blocks={cb}: interval median {np.median(iv):.0f} (min {iv.min():.0f} max {iv.max():.0f}) cyc | gather {d(0,1):.0f} fixedpoint {d(1,2):.0f} publish {d(2,3):.0f} ringwait {d(3,4):.0f} | spins total {st[:, 6].sum():.0f} blocks with spins {(st[:, 6] > 0).sum()}')
