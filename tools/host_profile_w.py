"""cProfile of the eager host path at training size: GDLoss(pred, target, weight(P,7), avg_factor) fwd+bwd, P = 4096."""
import cProfile, pstats, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mmdet3d_gaussian_amd as amd
dev = torch.device('cuda:0')
n = 4096
tgt = torch.rand(n, 7, device=dev) * 2 + 0.5
pred = (tgt + torch.randn(n, 7, device=dev) * 0.1).requires_grad_(True)
w = torch.ones(n, 7, device=dev)
m = amd.GDLoss('kld3d', fun='log1p', tau=0.0, loss_weight=5.0)
def step():
    pred.grad = None
    m(pred, tgt, w, avg_factor=float(n)).backward()
def step_now():
    pred.grad = None
    m(pred, tgt).backward()
for name, fn in (('weight(P,7)+avg_factor', step), ('no weight', step_now)):
    for _ in range(50): fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(1000): fn()
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f'{name}: host enqueue per fwd+bwd {(t1 - t0) / 1000 * 1e6:.1f} us, incl. drain {(t2 - t0) / 1000 * 1e6:.1f} us', flush=True)
t0 = time.perf_counter()
for _ in range(1000):
    pred.grad = None
    l = m(pred, tgt, w, avg_factor=float(n))
t1 = time.perf_counter(); torch.cuda.synchronize()
print(f'forward only: {(t1 - t0) / 1000 * 1e6:.1f} us')
pr = cProfile.Profile(); pr.enable()
for _ in range(1000): step()
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats('tottime').print_stats(22)
