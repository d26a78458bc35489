"""CenterPoint target assignment on the MI355X (csrc/center_targets.hip, center_targets.py) against the CPU restatement of
gd_centerpoint_head.py:65-156 with mmdet3d's Gaussian helpers (oracle/center_targets_torch.py; third-party parts unpinned).
Positions, boxes and their order bit for bit; heat maps bit for bit (the window is evaluated in fp64 and rounded once, as numpy
does — a device exp that differed by an fp64 ulp exactly at an fp32 rounding boundary would show as one ulp: not seen)."""
import numpy as np
import pytest
import torch

import mmdet3d_gaussian_amd as amd
from oracle import center_targets_torch as ct

pytestmark = pytest.mark.extras   # frozen extras outside SURVEY.md §8: `pytest -m extras` on a GPU box (conftest.py), not part of `-m gpu`
NUS = dict(grid_size=[512, 512, 1], point_cloud_range=[-51.2, -51.2, -5.0, 51.2, 51.2, 3.0], voxel_size=[0.2, 0.2, 8],
           out_size_factor=4, gaussian_overlap=0.1, min_radius=2)
TASKS = [['car'], ['truck', 'construction_vehicle'], ['bus', 'trailer'], ['barrier'], ['motorcycle', 'bicycle'],
         ['pedestrian', 'traffic_cone']]


def scene(g, n, spread=58.0, ignore=0.1):
    xy = torch.rand(n, 2, generator=g) * 2 * spread - spread          # some centres fall outside the 51.2 m range
    dims = torch.rand(n, 3, generator=g) * torch.tensor([3.0, 9.0, 3.0]) + torch.tensor([0.3, 0.4, 0.5])
    box = torch.cat([xy, torch.rand(n, 1, generator=g) * 4 - 3, dims, torch.rand(n, 1, generator=g) * 6.28 - 3.14,
                     torch.randn(n, 2, generator=g)], 1)
    lab = torch.randint(0, 10, (n,), generator=g)
    lab[torch.rand(n, generator=g) < ignore] = -1
    return box, lab


def check(boxes, labels, tasks, cfg, objects=False):
    counts = [len(t) for t in tasks]

    class Obj:                       # what the head receives: bottom-centred rows behind `.tensor`
        def __init__(self, t):
            self.tensor = t
    if objects:
        gpu_boxes = [Obj(b.cuda()) for b in boxes]
        grav = [torch.cat([b[:, :2], (b[:, 2] + b[:, 5] * 0.5).unsqueeze(1), b[:, 3:]], 1) for b in boxes]
    else:
        gpu_boxes, grav = [b.cuda() for b in boxes], boxes
    hm, an, pi = amd.extras.center_head_get_targets(gpu_boxes, [l.cuda() for l in labels], tasks, cfg)
    hw, aw, pw = ct.get_targets(grav, labels, counts, cfg)
    assert len(hm) == len(hw) == len(tasks)
    for t in range(len(tasks)):
        assert pi[t].dtype == torch.int64 and torch.equal(pi[t].cpu(), pw[t]), t
        assert torch.equal(an[t].cpu(), aw[t]), t
        assert hm[t].shape == hw[t].shape
        assert torch.equal(hm[t].cpu(), hw[t]), (t, float((hm[t].cpu() - hw[t]).abs().max()))
    return hm, an, pi


def test_targets_nuscenes_geometry():
    g = torch.Generator().manual_seed(31)
    data = [scene(g, n) for n in (140, 3, 260, 75)]
    hm, an, pi = check([d[0] for d in data], [d[1] for d in data], TASKS, NUS)
    assert sum(a.shape[0] for a in an) > 250 and all(float(h.max()) == 1.0 for h in hm)
    check([d[0] for d in data], [d[1] for d in data], TASKS, NUS, objects=True)


def test_targets_edge_cases():
    g = torch.Generator().manual_seed(32)
    b0, l0 = scene(g, 60)
    b0[:5, 3] = 0.0                         # zero width: invalid (:124)
    b0[5:8, 4] = -1.0                       # negative length
    b0[8, 0], b0[8, 1] = -51.3, 10.0        # just left of the range: `.long()` truncates -0.125 to cell 0 -> still valid
    b0[9, 0], b0[9, 1] = 51.19, -51.19      # last / first cell
    b0[10, 0] = 51.2                        # first cell outside
    b0[11:14, :2] = torch.tensor([[-51.0, -51.0], [51.0, 51.0], [0.0, 51.0]])   # Gaussians clipped at corners / an edge
    b0[11:14, 3:5] = torch.tensor([[8.0, 20.0], [10.0, 25.0], [6.0, 14.0]])     # big boxes: radius well above min_radius
    l0[11:14] = torch.tensor([3, 3, 4])
    b0[14:20, :2] = torch.tensor([1.0, 2.0])                                     # six boxes of one class in ONE cell
    l0[14:20] = 0
    b1, l1 = scene(g, 0)                    # a sample without boxes
    b2, l2 = scene(g, 30)
    l2[:] = 9                               # one class only: five tasks get nothing from this sample
    check([b0, b1, b2], [l0, l1, l2], TASKS, NUS)
    # nothing valid at all
    hm, an, pi = check([b1, b1], [l1, l1], TASKS, NUS)
    assert all(a.shape[0] == 0 for a in an) and all(float(h.abs().max()) == 0.0 for h in hm)
    # other geometry: one task of three classes, rectangular... the reference mixes the two map extents only for non-square
    # grids (rows = grid_size[0] // osf); keep it square but change every other setting
    cfg = dict(grid_size=[1440, 1440, 40], point_cloud_range=[-54.0, -54.0, -5.0, 54.0, 54.0, 3.0], voxel_size=[0.075, 0.075, 0.2],
               out_size_factor=8, gaussian_overlap=0.35, min_radius=1)
    data = [scene(g, n, spread=56.0) for n in (90, 41)]
    for d in data:
        d[1].clamp_(max=2)
    check([d[0] for d in data], [d[1] for d in data], [['a', 'b', 'c']], cfg)
    # the reference's KITTI CenterPoint geometry (configs/_base_/models/pillarmvf_centerpoint_016pillar_second_secfpn_kitti.py):
    # grid_size = [496, 432, 1] -> maps of 248 rows x 216 columns, tasks Pedestrian | Cyclist | Car
    kitti = dict(grid_size=[496, 432, 1], point_cloud_range=[0, -39.68, -3, 69.12, 39.68, 1], voxel_size=[0.16, 0.16, 4],
                 out_size_factor=2, gaussian_overlap=0.1, min_radius=2)
    data = []
    for n in (40, 25):
        b, l = scene(g, n)
        b[:, 0] = torch.rand(n, generator=g) * 75 - 3           # x in [-3, 72): a few outside [0, 69.12)
        b[:, 1] = torch.rand(n, generator=g) * 84 - 42
        data.append((b, l.clamp(min=-1, max=2)))
    hm, an, pi = check([d[0] for d in data], [d[1] for d in data], [['Pedestrian'], ['Cyclist'], ['Car']], kitti)
    assert hm[0].shape == (2, 1, 248, 216) and int(torch.cat(pi)[:, 1].max()) < 216 and int(torch.cat(pi)[:, 2].max()) > 216


def test_targets_feed_the_head_losses():
    """the outputs are what center_head_losses takes: boxes (n, 9) and [batch, x, y] rows on the device"""
    g = torch.Generator().manual_seed(33)
    data = [scene(g, n, spread=50.0, ignore=0.0) for n in (50, 60)]
    hm, an, pi = amd.extras.center_head_get_targets([d[0].cuda() for d in data], [d[1].cuda() for d in data], TASKS, NUS)
    coder = amd.CenterPointBBoxYawCoder(pc_range=[-51.2, -51.2], out_size_factor=4, voxel_size=[0.2, 0.2], norm_bbox=True)
    maps = [{k: (torch.randn(2, c, 128, 128, generator=g) * 0.3).cuda().requires_grad_(True)
             for k, c in (('reg', 2), ('height', 1), ('dim', 3), ('yaw', 1), ('dir', 2), ('vel', 2))} for _ in TASKS]
    out = amd.center_head_losses(amd.GDLoss('bd3d', fun='log1p', tau=0.0, loss_weight=5.0), dict(type='L1Loss', reduction='mean', loss_weight=0.25),
                                 coder, maps, pi, an, [max(a.shape[0], 1) for a in an], [1.0, 1.0, 0.2, 0.2])
    total = sum(a + b for a, b in out)
    total.backward()
    assert torch.isfinite(total) and all(torch.isfinite(m['dim'].grad).all() for m in maps)


def test_targets_errors():
    g = torch.Generator().manual_seed(34)
    b, l = scene(g, 10)
    with pytest.raises(RuntimeError, match='no CPU path'):
        amd.extras.center_head_get_targets([b], [l], TASKS, NUS)
    with pytest.raises(RuntimeError, match='one label each'):
        amd.extras.center_head_get_targets([b.cuda()], [l[:5].cuda()], TASKS, NUS)
    big = scene(g, 9000)
    with pytest.raises(RuntimeError, match='sorts at most'):
        amd.extras.center_head_get_targets([big[0].cuda()], [big[1].cuda()], TASKS, NUS)


def test_targets_random_batches():
    """25 random batches (sample counts, box counts incl. empty samples, task layouts, geometries, overlaps): everything bit for bit"""
    g = torch.Generator().manual_seed(88)
    rng = np.random.default_rng(88)
    for it in range(25):
        B = int(rng.integers(1, 6))
        layout = [['c'] * int(rng.integers(1, 4)) for _ in range(int(rng.integers(1, 7)))]
        ncls = sum(len(t) for t in layout)
        osf = int(rng.choice([1, 2, 4, 8]))
        vs = float(rng.choice([0.1, 0.16, 0.2, 0.32]))
        nx, ny = int(rng.integers(4, 40)) * osf * 4, int(rng.integers(4, 40)) * osf * 4
        cfg = dict(grid_size=[ny, nx, 1], point_cloud_range=[-nx * vs / 2, -ny * vs / 3, -5.0, nx * vs / 2, ny * vs * 2 / 3, 3.0],
                   voxel_size=[vs, vs, 8], out_size_factor=osf, gaussian_overlap=float(rng.choice([0.1, 0.3, 0.5])),
                   min_radius=int(rng.integers(0, 4)))
        boxes, labels = [], []
        for b in range(B):
            n = int(rng.integers(0, 120))
            bx, _ = scene(g, n)
            bx[:, 0] = (torch.rand(n, generator=g) * 1.2 - 0.1) * nx * vs - nx * vs / 2
            bx[:, 1] = (torch.rand(n, generator=g) * 1.2 - 0.1) * ny * vs - ny * vs / 3
            boxes.append(bx)
            labels.append(torch.randint(-1, ncls + 1, (n,), generator=g))     # -1 and ncls belong to no task
        check(boxes, labels, layout, cfg, objects=bool(it % 2))
