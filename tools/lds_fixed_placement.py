#!/usr/bin/env python3
"""Occupancy cap of the fused kernel at FIXED buffer placement: one process, one set of buffers per loss (allocated the way
bench.py does), the cap switched per launch (experiment build -DGD_LDS_ENV reads GD3D_MIN_LDS on every launch), dispatch-
bound events, interleaved rounds.  Separates the cap's effect from the 5 % that the physical placement of a process's
buffers moves every kernel and the copy probe by.  usage: GD3D_LIB=tools/variants/libgd3d_ldsenv.so tools/lds_fixed_placement.py"""
import ctypes, os, sys, math
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import mmdet3d_gaussian_amd as amd
from mmdet3d_gaussian_amd import gd_loss as gdl
import bench
dev = torch.device('cuda:0')
n = 10_000_000
lib = amd.load_library()
pred0, tgt = bench.synthetic_pairs(n, 0, dev)
lts = ('gwd3d', 'kld3d', 'bd3d')
preds = {lt: pred0.clone() for lt in lts}; grads = {lt: torch.empty_like(pred0) for lt in lts}
del pred0
total = torch.zeros((), device=dev)
ws = torch.empty(16 * lib.gd3d_loss_workspace_bytes(n), dtype=torch.uint8, device=dev)   # 16x: room for the partial-layout experiment builds
prm = {lt: amd.make_params(lt, 'log1p', 1.0, 1.0, (0, 0, 0.5), {}) for lt in lts}
stream = torch.cuda.current_stream().cuda_stream
NOSUM = '--nosum' in sys.argv         # no loss sum: the kernel's partial-sum tail is skipped (gradient only)
WEIGHTED = '--weighted' in sys.argv   # (N,7) weights: a third tile per workgroup (21.6 KB of LDS: 7 workgroups per CU uncapped)
w7 = torch.rand(n, 7, device=dev) if WEIGHTED else None
caps = {7: 0, 6: 27300, 5: 32768, 4: 40960, 3: 54600} if WEIGHTED else {8: 0, 7: 23400, 6: 27300, 5: 32768, 4: 40960}
if os.environ.get('GD_CAPS'):   # restrict the sweep, e.g. GD_CAPS=6,5
    caps = {int(c): caps[int(c)] for c in os.environ['GD_CAPS'].split(',')}
def run(lt, iters):
    tms = []
    for _ in range(iters):
        tm = gdl.DispatchTimer()
        rc = lib.gd3d_loss_fused_timed(prm[lt], None, preds[lt].data_ptr(), tgt.data_ptr(), None, w7.data_ptr() if WEIGHTED else None, n, 5.0 / n, None,
                                       None if NOSUM else total.data_ptr(), grads[lt].data_ptr(), None, ws.data_ptr(), stream, tm.start, tm.stop)
        assert rc == 0
        tms.append(tm)
    torch.cuda.synchronize()
    d = sorted(t.elapsed_ms() for t in tms)
    return sum(d) / len(d) * 1e3
def probe(lt):
    tms = []
    for _ in range(12):
        tm = gdl.DispatchTimer()
        lib.gd3d_probe_stream(preds[lt].data_ptr(), tgt.data_ptr(), grads[lt].data_ptr(), 7 * n, stream, tm.start, tm.stop)
        tms.append(tm)
    torch.cuda.synchronize()
    d = sorted(t.elapsed_ms() for t in tms)[2:]
    return sum(d) / len(d) * 1e3
for lt in lts:
    run(lt, 30)
res = {(lt, c): [] for lt in lts for c in caps}
for r in range(4):
    for c, b in caps.items():
        os.environ['GD3D_MIN_LDS'] = str(b)
        for lt in lts:      # cycle the three losses as a step does (three different buffer sets: no cache reuse)
            run(lt, 3)
        for lt in lts:
            res[(lt, c)].append(run(lt, 12))
print('copy probe per buffer set (us):', {lt: round(probe(lt), 1) for lt in lts})
print('WG/CU  ' + '   '.join(f'{lt:>8s}' for lt in lts))
for c in caps:
    print(f'{c:5d}  ' + '   '.join(f'{sum(res[(lt, c)]) / len(res[(lt, c)]):8.1f}' for lt in lts))
