#!/usr/bin/env python3
"""Is the 129-140 us spread of the fused kernel a property of WHERE its buffers sit in physical HBM?
The same kernel (bd3d, 10 M pairs, kernel-only timing) on K independently allocated buffer sets in one process — all sets
stay alive, so every set is backed by different physical pages — and on one set obtained with
hipExtMallocWithFlags(hipDeviceMallocContiguous) where the runtime offers it."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mmdet3d_gaussian_amd as amd
from mmdet3d_gaussian_amd import gd_loss as gdl
lib = amd.load_library()
dev = torch.device('cuda:0')
n = 10_000_000
nb = n * 28
g = torch.Generator(device=dev).manual_seed(0)
src_t = torch.rand(n, 7, generator=g, device=dev) * 2 + 0.5
src_p = src_t + torch.randn(n, 7, generator=g, device=dev) * 0.1
ws = torch.empty(lib.gd3d_loss_workspace_bytes(n), dtype=torch.uint8, device=dev)
prm = gdl.make_params('bd3d', 'log1p', 1.0, 1.0, (0, 0, 0.5), {})


def time_ptrs(p, t, gp):
    call = lambda: lib.gd3d_loss_fused(prm, p, t, None, n, 5.0 / n, None, None, gp, None, ws.data_ptr(), None)
    for _ in range(5): call()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): assert call() == 0
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 20 * 1e3


def time_add(p, t, gp):
    """the same three buffers through a plain elementwise kernel (torch.add, 2 streams read : 1 written)"""
    for _ in range(5): torch.add(p, t, out=gp)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): torch.add(p, t, out=gp)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 20 * 1e3


keep = []
for k in range(8):
    p, t, gp = src_p.clone(), src_t.clone(), torch.empty_like(src_p)
    keep.append((p, t, gp))
    tk = time_ptrs(p.data_ptr(), t.data_ptr(), gp.data_ptr())
    ta = time_add(p, t, gp)
    p.copy_(src_p); t.copy_(src_t)
    print(f'torch allocation set {k}: {tk:6.1f} us   (torch.add on the same buffers: {ta:6.1f} us)', flush=True)
# re-time the first sets: is the time a stable property of the set?
for k in (0, 1, 2):
    p, t, gp = keep[k]
    print(f'torch allocation set {k} again: {time_ptrs(p.data_ptr(), t.data_ptr(), gp.data_ptr()):6.1f} us', flush=True)

hip = ctypes.CDLL('libamdhip64.so')
if hasattr(hip, 'hipExtMallocWithFlags'):
    hip.hipExtMallocWithFlags.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t, ctypes.c_uint]
    hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
    for flag, name in ((0x4, 'hipDeviceMallocContiguous'), (0x0, 'hipDeviceMallocDefault')):
        ptrs = []
        ok = True
        for _ in range(3):
            q = ctypes.c_void_p()
            rc = hip.hipExtMallocWithFlags(ctypes.byref(q), nb, flag)
            if rc != 0:
                print(f'{name}: hipExtMallocWithFlags -> {rc}', flush=True)
                ok = False
                break
            ptrs.append(q.value)
        if ok:
            hip.hipMemcpy(ptrs[0], src_p.data_ptr(), nb, 3)
            hip.hipMemcpy(ptrs[1], src_t.data_ptr(), nb, 3)
            print(f'{name} set: {time_ptrs(*ptrs):6.1f} us', flush=True)
