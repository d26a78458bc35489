#!/usr/bin/env python3
"""Loss/NMS-level stand-ins for BASELINE.json configs 2, 4 and 5 on ONE MI355X (SURVEY.md §8d maps each config to the
nearest shipped reference config).  Per-GPU work of one training / inference step of the hot path only (the network body
is out of scope).  TIMING ONLY: the oracle checks of these three geometries are the -m gpu tests in
tests/test_gpu_configs.py; here only config 5's NMS keep indices are re-checked (the `keep_bit_exact` field).
  config 2: PointPillars KITTI 3-class, KLD (tau=0, log1p): anchor head decoded-box branch from raw NCHW output,
            B=6 samples x 321 408 anchors, ~60 positives per sample (gd_anchor3d_head.py:95-141).
  config 4: nuScenes CenterPoint (nearest shipped; BASELINE says PointPillars), BCD: 6 tasks x samples_per_gpu=8 x <=500
            objects: center_head_gd_loss per task (gd_centerpoint_head.py:413-434).
  config 5: Waymo PointPillars, GWD + rotated NMS: loss at 4096 positives + nms_gpu(4096 boxes, thr 0.25, max 500) x 3
            classes, keep indices checked bit-exact."""
import json, math, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch
import mmdet3d_gaussian_amd as amd, oracle
from rbox_inputs import nms_boxes
dev = torch.device('cuda:0')
def timeit(fn, it):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(it): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / it * 1e6

# ---- config 2
B, A, H, W, C = 6, 6, 248, 216, 3
n_per = H * W * A
g = torch.Generator(device=dev).manual_seed(0)
anchors = torch.rand(n_per, 7, generator=g, device=dev) * torch.tensor([70, 80, 1, 1.5, 3, .5, 1.5], device=dev) + torch.tensor([0, -40, -2, .6, .9, 1.4, 0], device=dev)
bbox_pred = (torch.randn(B, A * 7, H, W, generator=g, device=dev) * 0.1).requires_grad_(True)
bbox_targets = torch.randn(B, n_per, 7, generator=g, device=dev) * 0.2
bbox_weights = torch.ones(B, n_per, 7, device=dev)
labels = torch.full((B, n_per), C, device=dev, dtype=torch.long)
labels.view(-1)[torch.randperm(B * n_per, generator=g, device=dev)[:60 * B]] = 0
mod = amd.GDLoss('kld3d', fun='log1p', tau=0.0, loss_weight=5.0)
def step2():
    bbox_pred.grad = None
    amd.anchor_head_decoded_loss_fused(mod, bbox_pred, bbox_targets, bbox_weights, labels, anchors, C, 360.0, [1.0] * 7).backward()
us = timeit(step2, 100)
print(json.dumps(dict(config=2, what='anchor-head decoded-box loss slice fwd+bwd from NCHW, KITTI geometry', batch=B,
                      anchors=B * n_per, positives=60 * B, us_per_step=round(us, 1))), flush=True)

# the same step captured once and replayed as a hipGraph (the dense form has static shapes and no host sync): the package's
# GraphedStep; `replay only` = new values written straight into the graph's input buffers, `with input copies` = through its
# call interface (one device copy of the 108 MB head output per step)
gstep = amd.GraphedStep(lambda bp, lb: amd.anchor_head_decoded_loss_fused(mod, bp, bbox_targets, bbox_weights, lb, anchors, C, 360.0, [1.0] * 7),
                        (bbox_pred, labels))
sbp, slb = gstep.static_inputs()
usg = timeit(lambda: gstep(sbp, slb), 200)
usc = timeit(lambda: gstep(bbox_pred, labels), 100)
lg, gg = gstep(bbox_pred, labels)
step2()
torch.cuda.synchronize()
assert torch.equal(gg[0], bbox_pred.grad), 'graph replay and eager step disagree'
print(json.dumps(dict(config=2, what='the same step replayed as a hipGraph (GraphedStep: static shapes, no sync)', us_per_step=round(usg, 1),
                      us_per_step_with_input_copies=round(usc, 1), eager_us_per_step=round(us, 1))), flush=True)

# ---- config 4
coder = amd.CenterPointBBoxYawCoder(pc_range=[-51.2, -51.2], out_size_factor=4, voxel_size=[0.2, 0.2], norm_bbox=True)
Bs, K, tasks = 8, 500, 6
modb = amd.GDLoss('bd3d', fun='log1p', tau=0.0, loss_weight=5.0)
data = []
for _ in range(tasks):
    P = Bs * K
    pos = torch.stack([torch.randint(0, Bs, (P,), generator=g, device=dev), torch.randint(0, 128, (P,), generator=g, device=dev),
                       torch.randint(0, 128, (P,), generator=g, device=dev)], -1)
    xy = (pos[:, 1:].float() + torch.rand(P, 2, generator=g, device=dev)) * 0.8 - 51.2
    anno = torch.cat([xy, torch.rand(P, 1, generator=g, device=dev) * 4 - 3, torch.rand(P, 3, generator=g, device=dev) * 2 + 0.5,
                      torch.rand(P, 1, generator=g, device=dev) * 6 - 3, torch.randn(P, 2, generator=g, device=dev)], -1)
    pred = torch.cat([torch.rand(P, 2, generator=g, device=dev), anno[:, 2:3], anno[:, 3:6].log(), anno[:, 6:7],
                      anno[:, 6:7].sin(), anno[:, 6:7].cos(), anno[:, 7:9]], -1)
    pred = (pred + torch.randn(P, 11, generator=g, device=dev) * 0.1).requires_grad_(True)
    data.append((pos, pred, anno))
def step4():
    for pos, pred, anno in data:
        pred.grad = None
        amd.center_head_gd_loss(modb, coder, pos, pred, anno, num_pos=float(Bs * K)).backward()
us = timeit(step4, 50)
print(json.dumps(dict(config=4, what='6 CenterPoint tasks x center_head_gd_loss (bd3d) fwd+bwd, samples_per_gpu=8',
                      positives_per_task=Bs * K, us_per_step=round(us, 1), us_per_task=round(us / tasks, 1))), flush=True)

# the same step from the RAW head maps: all regression losses (loss_l1 + loss_gd) of the 6 tasks in one launch, vs the
# op-for-op eager restatement of gd_centerpoint_head.py:409-434 per task (oracle/head_torch.py)
from oracle import head_torch  # noqa: E402
cfg4 = dict(pc_range=[-51.2, -51.2], out_size_factor=4, voxel_size=[0.2, 0.2], norm_bbox=True)
maps4 = []
for _ in range(tasks):
    maps4.append({k: (torch.randn(Bs, c, 128, 128, generator=g, device=dev) * 0.3).requires_grad_(True)
                  for k, c in (('reg', 2), ('height', 1), ('dim', 3), ('yaw', 1), ('dir', 2), ('vel', 2))})
l1cfg = dict(type='L1Loss', reduction='mean', loss_weight=0.25)
cw4 = [1.0, 1.0, 0.2, 0.2]
def step4_maps():
    for d in maps4:
        for v in d.values(): v.grad = None
    out = amd.center_head_losses(modb, l1cfg, coder, maps4, [p for p, _, _ in data], [a for _, _, a in data], [Bs * K] * tasks, cw4)
    sum(a + b for a, b in out).backward()
def step4_eager():
    for d in maps4:
        for v in d.values(): v.grad = None
    tot = 0
    for d, (pos, _, anno) in zip(maps4, data):
        l1, gd = head_torch.center_head_task_losses(d, pos, anno, Bs * K, cfg4, dict(loss_type='bd3d', fun='log1p', tau=0.0, loss_weight=5.0), 0.25, cw4)
        tot = tot + l1 + gd
    tot.backward()
us_m, us_e = timeit(step4_maps, 50), timeit(step4_eager, 10)
names4 = ('reg', 'height', 'dim', 'yaw', 'dir', 'vel')
flat4 = [d[k] for d in maps4 for k in names4]
def fn4(*flat):
    ds = [dict(zip(names4, flat[6 * i:6 * i + 6])) for i in range(tasks)]
    return amd.center_head_losses(modb, l1cfg, coder, ds, [p for p, _, _ in data], [a for _, _, a in data], [Bs * K] * tasks, cw4)
g4 = amd.GraphedStep(fn4, flat4)
s4 = g4.static_inputs()
us_g4 = timeit(lambda: g4(*s4), 100)
print(json.dumps(dict(config=4, what='the same 6-task head-loss step replayed as a hipGraph (GraphedStep)', us_per_step=round(us_g4, 1),
                      eager_one_launch_us=round(us_m, 1))), flush=True)
print(json.dumps(dict(config=4, what='all regression losses (loss_l1 + loss_gd) of 6 CenterPoint tasks from the raw head maps (8 x 128 x 128), fwd+bwd',
                      positives_per_task=Bs * K, one_launch_us=round(us_m, 1), eager_torch_us=round(us_e, 1),
                      speedup=round(us_e / us_m, 1))), flush=True)

# ---- config 5
P = 4096
t = torch.rand(P, 7, generator=g, device=dev) * torch.tensor([150, 150, 4, 2, 4, 1.5, 6.28], device=dev) + torch.tensor([-75, -75, -3, .5, .5, .5, -3.14], device=dev)
p = (t + torch.randn(P, 7, generator=g, device=dev) * 0.1).requires_grad_(True)
modg = amd.GDLoss('gwd3d', fun='log1p', tau=0.0, loss_weight=5.0)
def loss5():
    p.grad = None; modg(p, t, avg_factor=float(P)).backward()
us_loss = timeit(loss5, 100)
cls = []
ok = True
for c in range(3):
    b, s = nms_boxes(4096, seed=100 + c)
    cls.append((torch.from_numpy(b).to(dev), torch.from_numpy(s).to(dev)))
    want = oracle.nms_gpu_oracle(b, s, 0.25, post_max_size=500)
    got = amd.nms_gpu(cls[-1][0], cls[-1][1], 0.25, post_max_size=500).cpu().numpy()
    ok &= bool(np.array_equal(got, want))
def nms5():
    for b, s in cls: amd.nms_gpu(b, s, 0.25, post_max_size=500)
us_nms = timeit(nms5, 20)
# the same three problems as ONE batched call (blockIdx.y = class; group sizes read on the device; one sync)
allb = torch.cat([b for b, _ in cls]); alls = torch.zeros(3, 3 * 4096, device=dev); allv = torch.zeros(3, 3 * 4096, dtype=torch.bool, device=dev)
for c in range(3):
    alls[c, c * 4096:(c + 1) * 4096] = cls[c][1]; allv[c, c * 4096:(c + 1) * 4096] = True
def nms5b():
    return amd.nms_gpu_batched(allb, alls, 0.25, allv, pre_max_size=4096, post_max_size=500)
res = nms5b()
for c in range(3):
    ok &= bool(torch.equal(res[c] - c * 4096, amd.nms_gpu(cls[c][0], cls[c][1], 0.25, post_max_size=500)))
us_nmsb = timeit(nms5b, 20)
print(json.dumps(dict(config=5, what='GWD loss fwd+bwd at 4096 positives + nms_gpu(4096, thr .25, max 500) x 3 classes',
                      loss_us=round(us_loss, 1), nms_us_3_classes=round(us_nms, 1),
                      nms_us_3_classes_one_batched_call=round(us_nmsb, 1), keep_bit_exact=ok)), flush=True)
assert ok, 'config 5: NMS keep indices differ from the oracle'
