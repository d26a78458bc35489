import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mmdet3d_gaussian_amd as amd
lib = amd.load_library()
n = 10_000_000
g = torch.randn(n, 7, device='cuda'); s = torch.tensor([0.5], device='cuda')
def t(fn, it=30):
    for _ in range(5): fn()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / it * 1e3
print('scale pass (g != 1): %.1f us' % t(lambda: lib.gd3d_scale_rows(g.data_ptr(), s.data_ptr(), 0, n, None)))
one = torch.ones(1, device='cuda')
print('early exit (g == 1): %.1f us' % t(lambda: lib.gd3d_scale_rows(g.data_ptr(), one.data_ptr(), 0, n, None)))
