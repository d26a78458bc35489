#!/usr/bin/env python3
"""A/B of experimental libgd3d builds (tools/variants/libgd3d_<name>.so) on the SAME buffer sets in one process: the fused
bd3d kernel at 10 M pairs, kernel-only timing, on 8 independently allocated sets.  usage: variant_probe.py base <name> ..."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import mmdet3d_gaussian_amd as amd
from mmdet3d_gaussian_amd import _lib, gd_loss as gdl
names = sys.argv[1:]
libs = {}
for nm in names:
    L = ctypes.CDLL(os.path.join(ROOT, 'tools', 'variants', f'libgd3d_{nm}.so'))
    res, args = _lib.SYMBOLS['gd3d_loss_fused']
    L.gd3d_loss_fused.restype, L.gd3d_loss_fused.argtypes = res, args
    libs[nm] = L
base = amd.load_library()
dev = torch.device('cuda:0')
n = 10_000_000
g = torch.Generator(device=dev).manual_seed(0)
src_t = torch.rand(n, 7, generator=g, device=dev) * 2 + 0.5
src_p = src_t + torch.randn(n, 7, generator=g, device=dev) * 0.1
ws = torch.empty(base.gd3d_loss_workspace_bytes(n), dtype=torch.uint8, device=dev)
out = torch.zeros(4, device=dev)
prm = gdl.make_params('bd3d', 'log1p', 1.0, 1.0, (0, 0, 0.5), {})
def t_one(L, p, t, gp):
    call = lambda: L.gd3d_loss_fused(prm, p.data_ptr(), t.data_ptr(), None, n, 5.0 / n, None, out.data_ptr(), gp.data_ptr(), None, ws.data_ptr(), None)
    for _ in range(3): call()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): assert call() == 0
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 20 * 1e3, out[0].item()
keep = []
for k in range(8):
    p, t, gp = src_p.clone(), src_t.clone(), torch.empty_like(src_p)
    keep.append((p, t, gp))
    r = {nm: t_one(L, p, t, gp) for nm, L in libs.items()}
    print(f'set {k}: ' + '  '.join(f'{nm} {v[0]:6.1f} us' for nm, v in r.items()) + f'   (loss {list(r.values())[0][1]:.6f} / {list(r.values())[-1][1]:.6f})', flush=True)
