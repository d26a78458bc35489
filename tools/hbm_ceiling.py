"""Measured HBM ceilings on this box (context for roofline.frac): a plain device copy and a
read-mostly reduction with the same 56:32 read:write byte mix as the fused kernel is not available
in torch, so report copy (1:1) and sum (read only)."""
import torch, json
dev = torch.device('cuda:0')
n = 110_000_000  # 440 MB per buffer (> 256 MiB infinity cache)
a = torch.rand(n, device=dev); b = torch.empty_like(a)
def timeit(fn, it=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it * 1e-3
t = timeit(lambda: b.copy_(a)); copy = 2 * n * 4 / t / 1e9
t = timeit(lambda: a.sum()); rd = n * 4 / t / 1e9
t = timeit(lambda: b.fill_(1.0)); wr = n * 4 / t / 1e9
print(json.dumps({'copy_GBps': round(copy, 1), 'read_sum_GBps': round(rd, 1), 'write_fill_GBps': round(wr, 1)}))
