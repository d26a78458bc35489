"""The configs[4] stand-in under rocprofv3: 3 classes x 4096 boxes, thr 0.25, 60 calls of ONE form.
usage: tools/nms_batched_one.py single|batched|batched_c   (GD3D_LIB selects a library variant)
  single    : nms_gpu on one class (scored path: rank_place + circle + clip + scan)
  batched   : nms_gpu_batched over the 12 288 shared boxes with a (3, 12288) validity mask (pvrcnn_bbox_head.py:438-464)
  batched_c : the same through the C ABI only (rnms_batched_scored; no Python between the launches)"""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch
import mmdet3d_gaussian_amd as amd
from rbox_inputs import nms_boxes
dev = torch.device('cuda:0')
form = sys.argv[1]
C, n = 3, 4096
cls = []
for c in range(C):
    b, s = nms_boxes(n, seed=100 + c)
    cls.append((torch.from_numpy(b).to(dev), torch.from_numpy(s).to(dev)))
allb = torch.cat([b for b, _ in cls]); alls = torch.zeros(C, C * n, device=dev); allv = torch.zeros(C, C * n, dtype=torch.bool, device=dev)
for c in range(C):
    alls[c, c * n:(c + 1) * n] = cls[c][1]; allv[c, c * n:(c + 1) * n] = True
if form == 'single':
    fn = lambda: amd.nms_gpu(cls[0][0], cls[0][1], 0.25, post_max_size=500)
elif form == 'multi':     # diagnostic: the same three problems as separate entries (rnms_segmented_scored: every workgroup has boxes of its group)
    fn = lambda: amd.nms_gpu_multi([b for b, _ in cls], [s for _, s in cls], 0.25, pre_max_size=n, post_max_size=500)
elif form == 'batched':
    fn = lambda: amd.nms_gpu_batched(allb, alls, 0.25, allv, pre_max_size=n, post_max_size=500)
else:
    lib = amd.load_library()
    vp = lambda t: ctypes.c_void_p(t.data_ptr())
    th = torch.full((C,), 0.25, device=dev); keep = torch.empty(C, n, dtype=torch.int64, device=dev); num = torch.empty(C, dtype=torch.int64, device=dev)
    ws = torch.empty(lib.rnms_batched_scored_workspace_bytes(C, C * n, n), dtype=torch.uint8, device=dev)
    vb = allv.contiguous()
    def fn():
        rc = lib.rnms_batched_scored(0, vp(allb), vp(alls), vp(vb), C, C * n, n, vp(th), vp(keep), vp(num), vp(ws), None)
        assert rc == 0
for _ in range(10): fn()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(50): fn()
torch.cuda.synchronize()
print(form, f'{(time.perf_counter() - t0) / 50 * 1e6:.1f} us per call', flush=True)
