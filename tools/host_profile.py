"""cProfile of the eager host path of one loss fwd+bwd (where do the ~147 us/loss of host time go?)."""
import cProfile, pstats, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mmdet3d_gaussian_amd as amd
dev = torch.device('cuda:0')
n = 1_000_000
tgt = torch.rand(n, 7, device=dev) * 2 + 0.5
pred = (tgt + torch.randn(n, 7, device=dev) * 0.1).requires_grad_(True)
m = amd.GDLoss('gwd3d', loss_weight=5.0)
def step():
    pred.grad = None
    l = m(pred, tgt); l.backward()
for _ in range(20): step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(500): step()
t1 = time.perf_counter(); torch.cuda.synchronize()
print(f'host per fwd+bwd: {(t1 - t0) / 500 * 1e6:.1f} us')
t0 = time.perf_counter()
for _ in range(500):
    pred.grad = None
    l = m(pred, tgt)
t1 = time.perf_counter(); torch.cuda.synchronize()
print(f'host per fwd only: {(t1 - t0) / 500 * 1e6:.1f} us')
pr = cProfile.Profile(); pr.enable()
for _ in range(500): step()
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats('tottime').print_stats(18)
