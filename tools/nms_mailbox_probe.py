"""Probe: what does reading nms_gpu's data-dependent count back cost, and does a count written straight into pinned host
memory (polled by the host, no HIP call) beat the blocking 8-byte copy?  n = 4096, thr 0.25 (configs[4] stand-in, one class)."""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch
import mmdet3d_gaussian_amd as amd
from rbox_inputs import nms_boxes
lib = amd.load_library()
dev = torch.device('cuda:0')
n = 4096
b, s = nms_boxes(n, seed=100)
bt, st = torch.from_numpy(b).to(dev), torch.from_numpy(s).to(dev)
vp = lambda t: ctypes.c_void_p(t.data_ptr())
keep = torch.empty(n, dtype=torch.int64, device=dev); num = torch.empty(1, dtype=torch.int64, device=dev)
ws = torch.empty(lib.rnms_scored_workspace_bytes(n, n), dtype=torch.uint8, device=dev)
box = torch.empty(8, dtype=torch.int64).pin_memory()
box_np = box.numpy()
stream = torch.cuda.current_stream().cuda_stream
def launch(dst):
    rc = lib.rnms_scored(0, vp(bt), vp(st), n, n, 0.25, vp(keep), ctypes.c_void_p(dst), vp(ws), stream)
    assert rc == 0
def a():   # launches only, then a device synchronize (no result on the host)
    launch(num.data_ptr()); torch.cuda.synchronize()
def bb():  # the shipped form: blocking 8-byte copy
    launch(num.data_ptr()); return int(num.item())
def c():   # count written to pinned host memory by the scan kernel, host polls
    box_np[0] = -(1 << 62)
    launch(box.data_ptr())
    spins = 0
    while box_np[0] == -(1 << 62):
        spins += 1
    return int(box_np[0])
want = bb()
assert c() == want, (c(), want)
for name, fn in (('launch + device synchronize', a), ('launch + num.item()', bb), ('launch + pinned mailbox poll', c), ('launch + num.item()', bb), ('launch + pinned mailbox poll', c)):
    for _ in range(20): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(200): fn()
    torch.cuda.synchronize()
    print(f'{name:32s} {(time.perf_counter() - t0) / 200 * 1e6:7.1f} us per call', flush=True)
print('nms_gpu (module surface)        ', end='')
for _ in range(20): amd.nms_gpu(bt, st, 0.25)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(200): amd.nms_gpu(bt, st, 0.25)
print(f'{(time.perf_counter() - t0) / 200 * 1e6:7.1f} us per call')
