#!/usr/bin/env python3
"""A/B of experimental libgd3d builds (tools/variants/libgd3d_<name>.so, tools/build_variants.py) on the SAME buffer sets in
one process: the fused kernels at 10 M pairs, events bound to the dispatch, variants interleaved in rounds, on several
independently allocated buffer sets.  usage: variant_probe.py [--sets K] base <name> ..."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import mmdet3d_gaussian_amd as amd
from mmdet3d_gaussian_amd import _lib, gd_loss as gdl
import bench
argv = sys.argv[1:]
sets = 3
if argv and argv[0] == '--sets':
    sets = int(argv[1]); argv = argv[2:]
names = argv
libs = {}
for nm in names:
    L = ctypes.CDLL(os.path.join(ROOT, 'tools', 'variants', f'libgd3d_{nm}.so'))
    res, args = _lib.SYMBOLS['gd3d_loss_fused_timed']
    L.gd3d_loss_fused_timed.restype, L.gd3d_loss_fused_timed.argtypes = res, args
    libs[nm] = L
base = amd.load_library()
dev = torch.device('cuda:0')
n = 10_000_000
lts = ('gwd3d', 'kld3d', 'bd3d')
prm = {lt: gdl.make_params(lt, 'log1p', 1.0, 1.0, (0, 0, 0.5), {}) for lt in lts}
ws = torch.empty(base.gd3d_loss_workspace_bytes(n), dtype=torch.uint8, device=dev)
out = torch.zeros(4, device=dev)
stream = torch.cuda.current_stream().cuda_stream
def run(L, lt, p, t, gp, iters):
    tms = []
    for _ in range(iters):
        tm = gdl.DispatchTimer()
        rc = L.gd3d_loss_fused_timed(prm[lt], None, p.data_ptr(), t.data_ptr(), None, None, n, 5.0 / n, None, out.data_ptr(),
                                     gp.data_ptr(), None, ws.data_ptr(), stream, tm.start, tm.stop)
        assert rc == 0
        tms.append(tm)
    torch.cuda.synchronize()
    d = sorted(x.elapsed_ms() for x in tms)
    return sum(d) / len(d) * 1e3
keep = []
for k in range(sets):
    p, t = bench.synthetic_pairs(n, k, dev)
    gp = torch.empty_like(p)
    keep.append((p, t, gp))
    res = {(nm, lt): [] for nm in names for lt in lts}
    for nm in names:
        for lt in lts:
            run(libs[nm], lt, p, t, gp, 5)
    for r in range(4):
        for nm in names:
            for lt in lts:
                res[(nm, lt)].append(run(libs[nm], lt, p, t, gp, 10))
    if k == 0:   # every variant must produce the shipped library's bits (sum and gradient)
        for lt in lts:
            want_out = torch.zeros(4, device=dev); want_gp = torch.empty_like(p)
            rc = base.gd3d_loss_fused_decoded(prm[lt], None, p.data_ptr(), t.data_ptr(), None, None, n, 5.0 / n, None,
                                              want_out.data_ptr(), want_gp.data_ptr(), None, ws.data_ptr(), stream)
            assert rc == 0
            torch.cuda.synchronize()
            for nm in names:
                out.zero_(); gp.zero_()
                run(libs[nm], lt, p, t, gp, 1)
                same = bool(torch.equal(out[:1], want_out[:1])) and bool(torch.equal(gp, want_gp))
                if not same:
                    print(f'  !! {nm} {lt}: result differs from the shipped library (sum {out[0].item()!r} vs {want_out[0].item()!r}, '
                          f'{int((gp != want_gp).sum())} gradient entries)', flush=True)
    print(f'buffer set {k}   ' + '   '.join(f'{lt:>7s}' for lt in lts))
    for nm in names:
        print(f'  {nm:<10s}  ' + '   '.join(f'{sum(res[(nm, lt)]) / len(res[(nm, lt)]):7.1f}' for lt in lts), flush=True)
