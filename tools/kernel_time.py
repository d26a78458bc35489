#!/usr/bin/env python3
"""Time gd3d_loss_fused directly through the C ABI (no autograd, no Python in the loop) for one or more
builds of the library.  usage: tools/kernel_time.py [lib.so ...]   (default: the in-tree libgd3d.so)
Prints per loss type: us per launch (HIP events over back-to-back launches, incl. the reduce stage) and the
88 B/pair credited GB/s."""
import ctypes, math, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import mmdet3d_gaussian_amd as amd
from mmdet3d_gaussian_amd import _lib

def load(path):
    L = ctypes.CDLL(path)
    for name, (res, args) in _lib.SYMBOLS.items():
        fn = getattr(L, name); fn.restype = res; fn.argtypes = args
    return L

def main():
    libs = sys.argv[1:] or [amd.lib_path()]
    n = int(os.environ.get('PAIRS', 10_000_000)); iters = int(os.environ.get('ITERS', 50))
    dev = torch.device('cuda:0')
    g = torch.Generator(device=dev).manual_seed(0)
    lo = torch.tensor([0, -40, -3, 0.5, 0.5, 0.5, -math.pi], device=dev); hi = torch.tensor([70, 40, 1, 2.5, 4.5, 2.0, math.pi], device=dev)
    tgt = (torch.rand(n, 7, generator=g, device=dev) * (hi - lo) + lo).contiguous()
    pred = (tgt + torch.randn(n, 7, generator=g, device=dev) * torch.tensor([0.3, 0.3, 0.1, 0.1, 0.1, 0.1, 0.1], device=dev)).contiguous()
    gp = torch.empty_like(pred); total = torch.zeros((), device=dev)
    vp = lambda t: ctypes.c_void_p(t.data_ptr())
    s = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    rounds = int(os.environ.get('ROUNDS', 5))
    libs_loaded = [(os.path.basename(p), load(p)) for p in libs]
    res = {name: {lt: [] for lt in ('gwd3d', 'kld3d', 'bd3d')} for name, _ in libs_loaded}
    loss_val = {}
    for r in range(rounds):                      # interleaved rounds in ONE process (cdna guide §5.4 rule 24)
        for name, L in libs_loaded:
            ws = torch.empty(L.gd3d_loss_workspace_bytes(n), dtype=torch.uint8, device=dev)
            for lt in ('gwd3d', 'kld3d', 'bd3d'):
                prm = amd.make_params(lt, 'log1p', 1.0, 1.0, (0, 0, 0.5), {})
                call = lambda: L.gd3d_loss_fused(ctypes.byref(prm), vp(pred), vp(tgt), None, n, 5.0 / n, None, vp(total), vp(gp), None, vp(ws), s)
                for _ in range(3): assert call() == 0
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(iters): call()
                e1.record(); torch.cuda.synchronize()
                res[name][lt].append(e0.elapsed_time(e1) / iters * 1e3)
                loss_val[(name, lt)] = total.item()
    for name, _ in libs_loaded:
        out = []
        for lt in ('gwd3d', 'kld3d', 'bd3d'):
            v = sorted(res[name][lt]); med = v[len(v) // 2]
            out.append(f'{lt} min {v[0]:6.1f} med {med:6.1f} us ({88 * n / med / 1e3:6.0f} GB/s) loss={loss_val[(name, lt)]:.6f}')
        print(f'{name:24s} ' + ' | '.join(out), flush=True)

if __name__ == '__main__':
    main()
