"""Probe: can the 3-loss fwd+bwd step be captured in a hipGraph (torch.cuda.CUDAGraph) and can HIP events recorded
inside the capture (external=True) time the fused kernel on replay?"""
import math, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mmdet3d_gaussian_amd as amd
from mmdet3d_gaussian_amd import gd_loss as gdl
dev = torch.device('cuda:0')
n = 10_000_000
g = torch.Generator(device=dev).manual_seed(0)
tgt = torch.rand(n, 7, generator=g, device=dev) * 2 + 0.5
pred = (tgt + torch.randn(n, 7, generator=g, device=dev) * 0.1).requires_grad_(True)
mods = [amd.GDLoss(lt, loss_weight=5.0) for lt in ('gwd3d', 'kld3d', 'bd3d')]
def step():
    outs = []
    for m in mods:
        pred.grad = None
        l = m(pred, tgt); l.backward(); outs.append(l.detach())
    return outs
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3): step()
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
graph = torch.cuda.CUDAGraph()
try:
    evs = []
    class Ext(list):
        pass
    with torch.cuda.graph(graph):
        outs = step()
    print('capture OK')
except Exception as e:
    print('capture FAILED', repr(e)); sys.exit(0)
for _ in range(5): graph.replay()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
t0 = time.perf_counter(); e0.record()
for _ in range(50): graph.replay()
e1.record(); host = time.perf_counter() - t0
torch.cuda.synchronize(); wall = time.perf_counter() - t0
print(f'graph replay: {e0.elapsed_time(e1)/50*1e3:.1f} us/step (events) wall {wall/50*1e6:.1f} us host-enqueue {host/50*1e6:.1f} us; losses', [o.item() for o in outs])
# external timing events inside capture
try:
    g2 = torch.cuda.CUDAGraph()
    a = torch.cuda.Event(enable_timing=True, external=True); b = torch.cuda.Event(enable_timing=True, external=True)
    with torch.cuda.graph(g2):
        a.record()
        l = mods[0](pred.detach(), tgt)
        b.record()
    g2.replay(); g2.replay(); torch.cuda.synchronize()
    print('external events inside graph:', a.elapsed_time(b) * 1e3, 'us')
except Exception as e:
    print('external events FAILED', repr(e))
