#!/usr/bin/env python3
"""A/B of the loss-sum finish: separate reduce launch (shipped) vs in-kernel ticket tree (-DGD_TICKET=1 build).
usage: tools/ticket_ab.py [variant.so ...]   (first the in-tree library, then every variant given)
Times the bench step's device work without autograd: gwd3d, kld3d, bd3d fused (+ reduce) launches over three separate
pred / grad buffer sets and one target, back to back, HIP events around ROUNDS x ITERS steps, interleaved rounds in one
process.  Also checks that every library returns the same three sums, run after run (bitwise)."""
import ctypes, math, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import mmdet3d_gaussian_amd as amd
from mmdet3d_gaussian_amd import _lib


def main():
    libs = [amd.lib_path()] + sys.argv[1:]
    n = int(os.environ.get('PAIRS', 10_000_000)); iters = int(os.environ.get('ITERS', 30)); rounds = int(os.environ.get('ROUNDS', 5))
    dev = torch.device('cuda:0')
    g = torch.Generator(device=dev).manual_seed(0)
    lo = torch.tensor([0, -40, -3, 0.5, 0.5, 0.5, -math.pi], device=dev); hi = torch.tensor([70, 40, 1, 2.5, 4.5, 2.0, math.pi], device=dev)
    tgt = (torch.rand(n, 7, generator=g, device=dev) * (hi - lo) + lo).contiguous()
    pred0 = (tgt + torch.randn(n, 7, generator=g, device=dev) * torch.tensor([0.3, 0.3, 0.1, 0.1, 0.1, 0.1, 0.1], device=dev)).contiguous()
    lts = ('gwd3d', 'kld3d', 'bd3d')
    preds = [pred0.clone() for _ in lts]; grads = [torch.empty_like(pred0) for _ in lts]
    totals = torch.zeros(3, device=dev)
    vp = lambda t: ctypes.c_void_p(t.data_ptr())
    s = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    loaded = [(os.path.basename(p), _lib._bind(p)) for p in libs]
    prms = [amd.make_params(lt, 'log1p', 1.0, 1.0, (0, 0, 0.5), {}) for lt in lts]
    res = {name: [] for name, _ in loaded}; sums = {}
    for r in range(rounds):
        for name, L in loaded:
            wss = [torch.zeros(L.gd3d_loss_workspace_bytes(n), dtype=torch.uint8, device=dev) for _ in lts]   # zeroed ONCE
            def step():
                for k in range(3):
                    rc = L.gd3d_loss_fused(ctypes.byref(prms[k]), vp(preds[k]), vp(tgt), None, n, 5.0 / n, None,
                                           ctypes.c_void_p(totals.data_ptr() + 4 * k), vp(grads[k]), None, vp(wss[k]), s)
                    assert rc == 0, rc
            for _ in range(3): step()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(iters): step()
            e1.record(); torch.cuda.synchronize()
            res[name].append(e0.elapsed_time(e1) / iters * 1e3)
            got = tuple(totals.cpu().tolist())
            assert sums.setdefault(name, got) == got, (name, 'sums changed between runs', sums[name], got)
    base = None
    for name, _ in loaded:
        v = sorted(res[name]); med = v[len(v) // 2]
        base = base or med
        print(f'{name:28s} step (3 losses) min {v[0]:7.1f} med {med:7.1f} us  ({3 * n / med:8.0f} M pairs/s, {med / base:5.3f} x base)  sums {sums[name]}', flush=True)
    ref = sums[loaded[0][0]]
    for name, _ in loaded[1:]:
        print(name, 'sums equal to base:', sums[name] == ref, [abs(a - b) for a, b in zip(sums[name], ref)])


if __name__ == '__main__':
    main()
