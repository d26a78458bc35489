"""One NMS problem, 60 calls of rnms_bev (pre-sorted C ABI entry): the workload tools/nms_ab.sh puts under rocprofv3.
usage: tools/nms_one.py n thresh clutter(0|1)   (GD3D_LIB selects a library variant from tools/build_variants.py)"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch
import mmdet3d_gaussian_amd as amd
from rbox_inputs import nms_boxes
lib = amd.load_library()
vp = lambda t: ctypes.c_void_p(t.data_ptr())
n, thr, clutter = int(sys.argv[1]), float(sys.argv[2]), sys.argv[3] == '1'
boxes, scores = nms_boxes(n, seed=n, clutter=clutter)
b, s = torch.from_numpy(boxes).cuda(), torch.from_numpy(scores).cuda()
order = s.sort(0, descending=True)[1]; sb = b[order].contiguous()
keep = torch.empty(n, dtype=torch.int64, device='cuda'); num = torch.zeros(1, dtype=torch.int64, device='cuda')
ws = torch.empty(lib.rnms_workspace_bytes(n), dtype=torch.uint8, device='cuda')
for _ in range(60): lib.rnms_bev(vp(sb), n, thr, vp(keep), vp(num), vp(ws), None)
torch.cuda.synchronize()
