"""Timing of match_coco on problems the GENERIC matcher kernel takes (more than 256 gts): sparse and dense cost matrices.
usage: tools/match_wide_time.py   (GD3D_LIB selects a library variant)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import mmdet3d_gaussian_amd as amd
from mmdet3d_gaussian_amd import evaluation as ev
torch.manual_seed(0)
for D, G, frac in ((20000, 600, 0.002), (20000, 600, 0.2), (5000, 3000, 0.001)):
    cost = torch.rand(D, G, device='cuda')
    cost = torch.where(torch.rand(D, G, device='cuda') < frac, cost * 0.5, cost + 1.0)   # `frac` of the pairs under the thresholds
    thrs = torch.tensor([0.3, 0.5, 0.5], device='cuda')
    ign = (torch.rand(G, device='cuda') < 0.1); crowd = (torch.rand(G, device='cuda') < 0.05)
    for _ in range(3): m = ev.match_coco(cost, thrs, ign, crowd)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): m = ev.match_coco(cost, thrs, ign, crowd)
    torch.cuda.synchronize()
    print(f'D={D} G={G} frac={frac}: {(time.perf_counter() - t0) / 5 * 1e3:.2f} ms  matched {(m >= 0).sum().item()}  checksum {int(m.to(torch.int64).sum())}', flush=True)
