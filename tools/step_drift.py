#!/usr/bin/env python3
"""Does the per-step time of the 3-loss step drift over a long run?  Prints wall time per 200 steps (GPU-synchronised)
and the host enqueue share."""
import os, sys, time, gc
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import mmdet3d_gaussian_amd as amd
dev = torch.device('cuda:0')
n = 10_000_000
t = torch.rand(n, 7, device=dev) + 0.5
p = (t + torch.randn(n, 7, device=dev) * 0.1).requires_grad_(True)
mods = [amd.GDLoss(lt, fun='log1p', tau=1.0, loss_weight=5.0) for lt in ('gwd3d', 'kld3d', 'bd3d')]
def step():
    for m in mods:
        p.grad = None
        l = m(p, t); l.backward()
mode = sys.argv[1] if len(sys.argv) > 1 else 'plain'
if mode == 'nogc': gc.disable()
for blk in range(12):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(200): step()
    th = time.perf_counter() - t0
    torch.cuda.synchronize(); tt = time.perf_counter() - t0
    print(f'{mode} block {blk}: {tt / 200 * 1e3:.3f} ms/step (host enqueue {th / 200 * 1e3:.3f}) gc={gc.get_count()}', flush=True)

# --- mimic bench.py's pre-warm: bursts of 20 steps + sync for ~1 s, then 60 steps with / without event recording
from mmdet3d_gaussian_amd import gd_loss as gdl
def burst(seconds):
    t0 = time.perf_counter(); k = 0
    while time.perf_counter() - t0 < seconds:
        for _ in range(20): step()
        torch.cuda.synchronize(); k += 20
    return k
for rec in (False, True, False):
    k = burst(1.0)
    ev = []
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(60):
        gdl.PROFILE_EVENTS = ev if rec else None
        step()
    gdl.PROFILE_EVENTS = None
    th = time.perf_counter() - t0
    torch.cuda.synchronize(); tt = time.perf_counter() - t0
    print(f'after {k} burst steps, record={rec}: {tt / 60 * 1e3:.3f} ms/step (host enqueue {th / 60 * 1e3:.3f})', flush=True)

# --- the same with events recorded DURING the bursts as well (fresh list per burst)
def burst_rec(seconds):
    t0 = time.perf_counter(); k = 0
    while time.perf_counter() - t0 < seconds:
        gdl.PROFILE_EVENTS = []
        for _ in range(20): step()
        torch.cuda.synchronize(); k += 20
    gdl.PROFILE_EVENTS = None
    return k
for rec in (True, True):
    k = burst_rec(1.0)
    ev = []
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(60):
        gdl.PROFILE_EVENTS = ev if rec else None
        step()
    gdl.PROFILE_EVENTS = None
    th = time.perf_counter() - t0
    torch.cuda.synchronize(); tt = time.perf_counter() - t0
    d = [tm.elapsed_ms() for tm in ev]
    print(f'after {k} recorded burst steps, record={rec}: {tt / 60 * 1e3:.3f} ms/step (host enqueue {th / 60 * 1e3:.3f}) kernel {sum(d) / len(d) * 1e3:.1f} us', flush=True)
