"""BASELINE.json configs[1], [3], [4] at their REAL geometry on the MI355X, against the oracle.

The network body is out of scope; what a config contributes to the hot path is the shape of the head
tensors and the loss / NMS settings, so each test builds synthetic head outputs of exactly that shape:

  config 2  PointPillars KITTI 3-class, KLD, tau=0, log1p: B=6 samples x 248x216 cells x 6 anchors = 6 x 321 408 anchors
            (/root/reference/configs/_base_/models/hv_pointpillars_secfpn_kitti.py:40-49, code_weight=[0]*7,
            decode_weight=1 from configs/kitti/hv_pointpillars_secfpn_kld5tau1_*.py:9-12), `loss_bbox` of
            GDAnchor3DHead.loss_single (models/dense_heads/gd_anchor3d_head.py:95-161).
  config 4  nuScenes, BCD: 6 tasks x (8, c, 128, 128) head maps x 4000 objects per task, `loss_l1` + `loss_gd` of
            CenterGDHead.loss (models/dense_heads/gd_centerpoint_head.py:402-441).
  config 5  Waymo dense head (>= 100 k anchors per sample), GWD, + rotated NMS of 3 classes x 4096 boxes
            (gd_centerpoint_head.py:340-345 call shape; thr 0.25, max 500), keep indices bit-exact.

Oracles: oracle.gd_loss_decoded (C, fp64; the GD term is pinned by the reference golden files, the anchor coder is
restated), oracle/head_torch.py (fp64 torch restatement of the head lines, autograd backward), oracle.nms_gpu_oracle.
Tolerance: 1e-5 relative + 3 x the fp32 evaluation noise of the same oracle (tests/gd_golden.py policy), stated per assert.
"""
import numpy as np
import pytest
import torch

import oracle
from gd_golden import check_close, grad_bound
from rbox_inputs import nms_boxes

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def amd():
    import mmdet3d_gaussian_amd as m
    assert torch.cuda.is_available(), 'these tests need the MI355X'
    m.load_library()
    return m


def range_anchors(H, W, ranges, sizes, rotations):
    """(H*W*A, 7) anchors in mmdet3d Anchor3DRangeGenerator order (y, x, size, rotation), A = len(sizes)*len(rotations);
    box = [x, y, z, w, l, h, r]."""
    per_size = []
    for (x0, y0, z0, x1, y1, _), sz in zip(ranges, sizes):
        xs = torch.linspace(x0, x1, W)
        ys = torch.linspace(y0, y1, H)
        rot = torch.tensor(rotations, dtype=torch.float32)
        yy, xx, rr = torch.meshgrid(ys, xs, rot, indexing='ij')                       # (H, W, R)
        a = torch.stack([xx, yy, torch.full_like(xx, z0), torch.full_like(xx, sz[0]), torch.full_like(xx, sz[1]),
                         torch.full_like(xx, sz[2]), rr], -1)                          # (H, W, R, 7)
        per_size.append(a)
    return torch.stack(per_size, 2).reshape(-1, 7).contiguous()                       # (H, W, S, R, 7) -> rows


def head_case(B, H, W, anchors, num_pos, C, seed, weights='ones'):
    """Synthetic head output + assigner output of the given geometry: `num_pos` positives at random anchors plus the
    corner cases (first / last anchor of the batch, a run of neighbours that shares a 256-row tile)."""
    g = torch.Generator().manual_seed(seed)
    n_per = anchors.shape[0]
    A = n_per // (H * W)
    M = B * n_per
    bbox_pred = torch.randn(B, A * 7, H, W, generator=g) * 0.1
    labels = torch.full((M,), C, dtype=torch.long)
    pos = torch.randperm(M, generator=g)[:num_pos - 8]
    pos = torch.cat([pos, torch.tensor([0, M - 1, 1000, 1001, 1002, 1003, n_per - 1, n_per])]).unique()
    labels[pos] = torch.randint(0, C, (pos.numel(),), generator=g)
    labels[M // 2 + 7] = -1                                                          # an ignored anchor (label < 0)
    bbox_targets = torch.zeros(M, 7)
    bbox_targets[pos] = torch.randn(pos.numel(), 7, generator=g) * 0.2
    if weights == 'ones':                                                            # the assigner's pos weight is 1.0
        bbox_weights = torch.zeros(M, 7)
        bbox_weights[pos] = 1.0
    else:
        bbox_weights = torch.rand(M, 7, generator=g)
    return bbox_pred, bbox_targets.reshape(B, n_per, 7), bbox_weights.reshape(B, n_per, 7), labels.reshape(B, n_per)


def gathered_oracle(bbox_pred, bbox_targets, bbox_weights, labels, anchors, C, lt, kw, dw, scale):
    """fp64 / fp32 C oracle of the decoded-box GD term on the numpy-gathered positives."""
    n_per = anchors.shape[0]
    lab = labels.reshape(-1).numpy()
    pos = np.nonzero((lab >= 0) & (lab < C))[0]
    bp = bbox_pred.permute(0, 2, 3, 1).reshape(-1, 7).numpy()[pos]
    bt = bbox_targets.reshape(-1, 7).numpy()[pos]
    an = anchors.numpy()[pos % n_per]
    w = bbox_weights.reshape(-1, 7).numpy()[pos].astype(np.float64) * np.asarray(dw, np.float64)
    prm = oracle.make_params(lt, **kw)
    r64 = oracle.gd_loss_decoded(bp, bt, prm, oracle.PRO_ANCHOR_DELTA, an, row_weight=w.mean(-1), scale=scale)
    r32 = oracle.gd_loss_decoded(bp, bt, prm, oracle.PRO_ANCHOR_DELTA, an, row_weight=w.astype(np.float32).mean(-1),
                                 scale=scale, dtype=np.float32)
    return pos, r64, r32


KITTI_RANGES = [[0.08, -39.60, -0.6, 68.88, 39.44, -0.6], [0.08, -39.60, -0.6, 68.88, 39.44, -0.6],
                [0.08, -39.60, -1.78, 68.88, 39.44, -1.78]]
KITTI_SIZES = [[0.8, 0.6, 1.73], [1.76, 0.6, 1.73], [3.9, 1.6, 1.56]]


@pytest.mark.parametrize('dense', [True, False])
def test_config2_kitti_pointpillars_kld_tau0(amd, dense):
    """configs[1]: 6 x 321 408 anchors, KLD tau=0 log1p loss_weight=5, code_weight=[0]*7, decode_weight=1,
    SmoothL1(beta=1/9, loss_weight=2), diff_rad_by_sin.  With the shipped zero code weights the SmoothL1 term is
    exactly 0, so loss_bbox == the GD term: checked against the C oracle on the gathered positives (loss, the NCHW
    gradient AT the positives, exact zeros elsewhere); dense (label test in the kernel) and list (nonzero) forms."""
    B, H, W, C = 6, 248, 216, 3
    anchors = range_anchors(H, W, KITTI_RANGES, KITTI_SIZES, [0, 1.57])
    assert anchors.shape[0] == 321408
    bbox_pred, bbox_targets, bbox_weights, labels = head_case(B, H, W, anchors, 360, C, seed=2)
    lt, kw = 'kld3d', dict(fun='log1p', tau=0.0)
    mod = amd.GDLoss(lt, loss_weight=5.0, **kw)
    sl1 = dict(type='SmoothL1Loss', beta=1.0 / 9.0, loss_weight=2.0)
    num_total = 355.0
    bp = bbox_pred.cuda().requires_grad_(True)
    out = amd.anchor_head_bbox_loss(mod, sl1, bp, bbox_targets.cuda(), bbox_weights.cuda(), labels.cuda(), anchors.cuda(),
                                    C, num_total, code_weight=[0.0] * 7, decode_weight=1, diff_rad_by_sin=True, dense=dense)
    out.backward()
    pos, r64, r32 = gathered_oracle(bbox_pred, bbox_targets, bbox_weights, labels, anchors, C, lt, kw, [1.0] * 7,
                                    5.0 / num_total)
    tol = 1e-5 + 3 * abs(r32['loss_sum'] - r64['loss_sum']) / (1 + abs(r64['loss_sum']))
    assert abs(out.item() - r64['loss_sum']) <= tol * (1 + abs(r64['loss_sum'])), (out.item(), r64['loss_sum'])
    gflat = bp.grad.permute(0, 2, 3, 1).reshape(-1, 7)
    posd = torch.from_numpy(pos).cuda()
    check_close('config2.grad_at_positives', gflat[posd].cpu().numpy(), r64['grad_pred'],
                grad_bound(r64['grad_pred'], r32['grad_pred']))
    assert int((bp.grad != 0).sum().item()) <= 7 * len(pos)                          # nothing outside the positives
    mask = torch.ones(gflat.shape[0], dtype=torch.bool, device='cuda')
    mask[posd] = False
    assert gflat[mask].abs().max().item() == 0.0


def test_config2_kitti_full_regression_term_vs_torch_fp64(amd):
    """Same geometry with NON-zero code weights and fractional bbox weights, so that the SmoothL1 / add_sin_difference
    term contributes: whole loss_bbox vs the fp64 torch restatement of gd_anchor3d_head.py:95-161 with autograd."""
    from oracle import head_torch
    B, H, W, C = 6, 248, 216, 3
    anchors = range_anchors(H, W, KITTI_RANGES, KITTI_SIZES, [0, 1.57])
    bbox_pred, bbox_targets, bbox_weights, labels = head_case(B, H, W, anchors, 360, C, seed=3, weights='rand')
    lt, kw = 'kld3d', dict(fun='log1p', tau=0.0)
    cw, dw, avg = [1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 0.5], 1, 355.0
    mod = amd.GDLoss(lt, loss_weight=5.0, **kw)
    bp = bbox_pred.cuda().requires_grad_(True)
    out = amd.anchor_head_bbox_loss(mod, dict(type='SmoothL1Loss', beta=1.0 / 9.0, loss_weight=2.0), bp, bbox_targets.cuda(),
                                    bbox_weights.cuda(), labels.cuda(), anchors.cuda(), C, avg, code_weight=cw,
                                    decode_weight=dw, diff_rad_by_sin=True)
    out.backward()

    def ref(dtype):
        p = bbox_pred.to(dtype).requires_grad_(True)
        r = head_torch.loss_single_bbox(p, bbox_targets.to(dtype), bbox_weights.to(dtype), labels, anchors.to(dtype), C, avg,
                                        gd=dict(loss_type=lt, loss_weight=5.0, **kw), sl1=dict(beta=1.0 / 9.0, loss_weight=2.0),
                                        code_weight=cw, decode_weight=[1.0] * 7, diff_rad_by_sin=True)
        r.backward()
        return r.item(), p.grad
    l64, g64 = ref(torch.float64)
    l32, g32 = ref(torch.float32)
    tol_l = 1e-5 + 3 * abs(l32 - l64) / (1 + abs(l64))
    assert abs(out.item() - l64) <= tol_l * (1 + abs(l64)), (out.item(), l64, l32)
    sc = g64.abs().max().item()
    tol_g = 1e-5 + 3 * (g32.double() - g64).abs().max().item() / (1 + sc)
    err = (bp.grad.cpu().double() - g64).abs().max().item()
    assert err <= tol_g * (1 + sc), (err, tol_g)


def test_config4_nuscenes_centerpoint_bcd_six_tasks(amd):
    """configs[3]: samples_per_gpu=8, 6 tasks, (8, c, 128, 128) head maps, 4000 objects per task (500 x 8), BCD
    (tau=0, log1p, loss_weight=5) + L1Loss(0.25) on dir / vel with the nuScenes code weights; every (loss_l1, loss_gd)
    pair and every head-map gradient vs the fp64 torch restatement of gd_centerpoint_head.py:402-441."""
    from oracle import head_torch
    g = torch.Generator().manual_seed(4)
    Bs, K, T, HW = 8, 500, 6, 128
    cfg = dict(pc_range=[-51.2, -51.2], out_size_factor=4, voxel_size=[0.2, 0.2], norm_bbox=True)
    coder = amd.CenterPointBBoxYawCoder(pc_range=cfg['pc_range'], out_size_factor=4, voxel_size=cfg['voxel_size'], norm_bbox=True)
    gd = dict(loss_type='bd3d', fun='log1p', tau=0.0, loss_weight=5.0)
    mod = amd.GDLoss('bd3d', fun='log1p', tau=0.0, loss_weight=5.0)
    cw = [1.0, 1.0, 0.2, 0.2]
    tasks = []
    for t in range(T):
        P = Bs * K
        maps = {k: torch.randn(Bs, c, HW, HW, generator=g) * 0.3
                for k, c in (('reg', 2), ('height', 1), ('dim', 3), ('yaw', 1), ('dir', 2), ('vel', 2))}
        pos = torch.stack([torch.randint(0, Bs, (P,), generator=g), torch.randint(0, HW, (P,), generator=g),
                           torch.randint(0, HW, (P,), generator=g)], -1)
        pos[1::97] = pos[0:-1:97]                                                     # objects that share a cell
        xy = (pos[:, 1:].float() + torch.rand(P, 2, generator=g)) * 0.8 - 51.2
        anno = torch.cat([xy, torch.rand(P, 1, generator=g) * 4 - 3, torch.rand(P, 3, generator=g) * 2 + 0.5,
                          torch.rand(P, 1, generator=g) * 6 - 3, torch.randn(P, 2, generator=g)], -1)
        tasks.append((maps, pos, anno))
    dev_maps = [{k: v.cuda().requires_grad_(True) for k, v in m.items()} for m, _, _ in tasks]
    out = amd.center_head_losses(mod, dict(type='L1Loss', reduction='mean', loss_weight=0.25), coder, dev_maps,
                                 [p.cuda() for _, p, _ in tasks], [a.cuda() for _, _, a in tasks], [Bs * K] * T, cw)
    sum(a + b for a, b in out).backward()
    for t, (maps, pos, anno) in enumerate(tasks):
        res = {}
        for dtype in (torch.float64, torch.float32):
            dd = {k: v.to(dtype).requires_grad_(True) for k, v in maps.items()}
            l1, lg = head_torch.center_head_task_losses(dd, pos, anno.to(dtype), Bs * K, cfg, gd, 0.25, cw)
            (l1 + lg).backward()
            res[dtype] = (l1.item(), lg.item(), {k: v.grad for k, v in dd.items()})
        r64, r32 = res[torch.float64], res[torch.float32]
        for j in range(2):
            tol = 1e-5 + 3 * abs(r32[j] - r64[j]) / (1 + abs(r64[j]))
            assert abs(out[t][j].item() - r64[j]) <= tol * (1 + abs(r64[j])), (t, j, out[t][j].item(), r64[j])
        for k, v in dev_maps[t].items():
            g64 = r64[2][k]
            sc = g64.abs().max().item()
            tol = 1e-5 + 3 * (r32[2][k].double() - g64).abs().max().item() / (1 + sc)
            err = (v.grad.cpu().double() - g64).abs().max().item()
            assert err <= tol * (1 + sc), (t, k, err, tol)


def test_config4_gradient_is_bitwise_reproducible(amd):
    """Many objects per cell (4000 objects into 64 cells): the head-map gradients of two runs are bit-identical and the
    accumulation order does not depend on the launch (deterministic reduction, no float atomics)."""
    g = torch.Generator().manual_seed(5)
    Bs, HW, P = 8, 128, 4000
    coder = amd.CenterPointBBoxYawCoder(pc_range=[-51.2, -51.2], out_size_factor=4, voxel_size=[0.2, 0.2], norm_bbox=True)
    mod = amd.GDLoss('bd3d', fun='log1p', tau=0.0, loss_weight=5.0)
    maps = {k: (torch.randn(Bs, c, HW, HW, generator=g) * 0.3).cuda()
            for k, c in (('reg', 2), ('height', 1), ('dim', 3), ('yaw', 1), ('dir', 2), ('vel', 2))}
    pos = torch.stack([torch.randint(0, 2, (P,), generator=g), torch.randint(10, 14, (P,), generator=g),
                       torch.randint(20, 28, (P,), generator=g)], -1)
    xy = (pos[:, 1:].float() + torch.rand(P, 2, generator=g)) * 0.8 - 51.2
    anno = torch.cat([xy, torch.rand(P, 1, generator=g) * 4 - 3, torch.rand(P, 3, generator=g) * 2 + 0.5,
                      torch.rand(P, 1, generator=g) * 6 - 3, torch.randn(P, 2, generator=g)], -1).cuda()
    pos = pos.cuda()
    runs = []
    for _ in range(4):
        d = {k: v.clone().requires_grad_(True) for k, v in maps.items()}
        out = amd.center_head_losses(mod, dict(type='L1Loss', loss_weight=0.25), coder, [d], [pos], [anno], [P], [1.0, 1.0, 0.2, 0.2])
        (out[0][0] + out[0][1]).backward()
        runs.append({k: v.grad.clone() for k, v in d.items()})
    for r in runs[1:]:
        for k in r:
            assert torch.equal(r[k], runs[0][k]), k
    # a permutation of the object list leaves every cell's set of contributions unchanged; with a fixed accumulation
    # order per cell (ascending object index after a stable sort by cell) the sums may differ in the last bits only
    perm = torch.randperm(P, generator=g).cuda()
    d = {k: v.clone().requires_grad_(True) for k, v in maps.items()}
    out = amd.center_head_losses(mod, dict(type='L1Loss', loss_weight=0.25), coder, [d], [pos[perm]], [anno[perm]], [P], [1.0, 1.0, 0.2, 0.2])
    (out[0][0] + out[0][1]).backward()
    for k in d:
        sc = runs[0][k].abs().max().item()
        assert (d[k].grad - runs[0][k]).abs().max().item() <= 1e-5 * (1 + sc), k


def test_config4_batch64_objects_in_few_cells_sorted_accumulate(amd):
    """32 000 objects of ONE task in 64 cells (batch-64 sizes; the scanning accumulate is quadratic
    there).  Above head_loss.CENTER_SORT_MIN_N objects the finish step walks the cell keys in sorted order
    (gd3d_center_head_stage -> one stable batched sort -> gd3d_center_head_finish): every head-map gradient and both
    losses against the fp64 torch restatement, bit-identical run to run, and — on a size both forms handle quickly —
    bit-identical to the scanning form (same accumulation order: ascending object index per cell)."""
    from oracle import head_torch
    from mmdet3d_gaussian_amd import head_loss
    g = torch.Generator().manual_seed(6)
    Bs, HW = 64, 128
    cfg = dict(pc_range=[-51.2, -51.2], out_size_factor=4, voxel_size=[0.2, 0.2], norm_bbox=True)
    coder = amd.CenterPointBBoxYawCoder(pc_range=cfg['pc_range'], out_size_factor=4, voxel_size=cfg['voxel_size'], norm_bbox=True)
    gd = dict(loss_type='bd3d', fun='log1p', tau=0.0, loss_weight=5.0)
    mod = amd.GDLoss('bd3d', fun='log1p', tau=0.0, loss_weight=5.0)
    cw = [1.0, 1.0, 0.2, 0.2]
    l1 = dict(type='L1Loss', loss_weight=0.25)

    def problem(P):
        maps = {k: torch.randn(Bs, c, HW, HW, generator=g) * 0.3
                for k, c in (('reg', 2), ('height', 1), ('dim', 3), ('yaw', 1), ('dir', 2), ('vel', 2))}
        pos = torch.stack([torch.randint(3, 5, (P,), generator=g), torch.randint(10, 14, (P,), generator=g),
                           torch.randint(20, 28, (P,), generator=g)], -1)           # 2 x 4 x 8 = 64 cells
        xy = (pos[:, 1:].float() + torch.rand(P, 2, generator=g)) * 0.8 - 51.2
        anno = torch.cat([xy, torch.rand(P, 1, generator=g) * 4 - 3, torch.rand(P, 3, generator=g) * 2 + 0.5,
                          torch.rand(P, 1, generator=g) * 6 - 3, torch.randn(P, 2, generator=g)], -1)
        return maps, pos, anno

    def run(maps, pos, anno, P):
        d = {k: v.cuda().requires_grad_(True) for k, v in maps.items()}
        out = amd.center_head_losses(mod, l1, coder, [d], [pos.cuda()], [anno.cuda()], [P], cw)
        (out[0][0] + out[0][1]).backward()
        torch.cuda.synchronize()
        return out[0][0].item(), out[0][1].item(), {k: v.grad.clone() for k, v in d.items()}

    P = 32_000
    assert P > head_loss.CENTER_SORT_MIN_N
    maps, pos, anno = problem(P)
    a = run(maps, pos, anno, P)
    b = run(maps, pos, anno, P)
    assert a[0] == b[0] and a[1] == b[1] and all(torch.equal(a[2][k], b[2][k]) for k in a[2])
    res = {}
    for dtype in (torch.float64, torch.float32):
        dd = {k: v.to(dtype).requires_grad_(True) for k, v in maps.items()}
        r1, rg = head_torch.center_head_task_losses(dd, pos, anno.to(dtype), P, cfg, gd, 0.25, cw)
        (r1 + rg).backward()
        res[dtype] = (r1.item(), rg.item(), {k: v.grad for k, v in dd.items()})
    r64, r32 = res[torch.float64], res[torch.float32]
    for j in range(2):
        tol = 1e-5 + 3 * abs(r32[j] - r64[j]) / (1 + abs(r64[j]))
        assert abs(a[j] - r64[j]) <= tol * (1 + abs(r64[j])), (j, a[j], r64[j])
    for k in maps:
        g64 = r64[2][k]
        sc = g64.abs().max().item()
        tol = 1e-5 + 3 * (r32[2][k].double() - g64).abs().max().item() / (1 + sc)
        err = (a[2][k].cpu().double() - g64).abs().max().item()
        assert err <= tol * (1 + sc), (k, err, tol)
        assert int((a[2][k] != 0).sum()) <= 64 * maps[k].shape[1]     # nothing outside the 64 cells
    # sorted form == scanning form, bit for bit (6000 objects, 64 cells + a task without shared cells + an empty task)
    P2 = 6000
    m2, p2, a2 = problem(P2)
    m3, p3, a3 = problem(300)
    p3 = torch.stack([torch.zeros(300, dtype=torch.long), torch.arange(300) % HW, torch.arange(300) // HW], -1)    # all cells distinct

    def run3(min_n):
        keep = head_loss.CENTER_SORT_MIN_N
        head_loss.CENTER_SORT_MIN_N = min_n
        try:
            ds = [{k: v.cuda().requires_grad_(True) for k, v in m.items()} for m in (m2, m3, m3)]
            out = amd.center_head_losses(mod, l1, coder, ds, [p2.cuda(), p3.cuda(), p3[:0].cuda()],
                                         [a2.cuda(), a3.cuda(), a3[:0].cuda()], [P2, 300, 0], cw)
            sum(x + y for x, y in out).backward()
            torch.cuda.synchronize()
            return [(x.item(), y.item()) for x, y in out], [{k: v.grad.clone() for k, v in d.items()} for d in ds]
        finally:
            head_loss.CENTER_SORT_MIN_N = keep
    ls, gs = run3(0)            # sorted
    lq, gq = run3(10 ** 9)      # scanning
    assert ls == lq
    for x, y in zip(gs, gq):
        for k in x:
            assert torch.equal(x[k], y[k]), k
    assert ls[2] == (0.0, 0.0) and all(float(v.abs().max()) == 0.0 for v in gs[2].values())


WAYMO_RANGES = [[-74.88, -74.88, -0.0345, 74.88, 74.88, -0.0345], [-74.88, -74.88, -0.1188, 74.88, 74.88, -0.1188],
                [-74.88, -74.88, 0, 74.88, 74.88, 0]]
WAYMO_SIZES = [[2.08, 4.73, 1.77], [0.84, 1.81, 1.77], [0.84, 0.91, 1.74]]


def test_config5_waymo_dense_head_gwd(amd):
    """configs[4], loss half: dense anchor head with 2 x 104 544 anchors (132 x 132 cells x 3 sizes x 2 rotations, Waymo
    anchor sizes hv_pointpillars_secfpn_waymo.py:51-55), GWD tau=0 log1p loss_weight=5, ~2000 positives, dense form,
    vs the C oracle on the gathered positives."""
    B, H, W, C = 2, 132, 132, 3
    anchors = range_anchors(H, W, WAYMO_RANGES, WAYMO_SIZES, [0, 1.57])
    assert anchors.shape[0] >= 100_000
    bbox_pred, bbox_targets, bbox_weights, labels = head_case(B, H, W, anchors, 2000, C, seed=6, weights='rand')
    lt, kw = 'gwd3d', dict(fun='log1p', tau=0.0)
    dw = [1.0, 1.0, 0.5, 1.0, 2.0, 1.0, 1.0]
    mod = amd.GDLoss(lt, loss_weight=5.0, **kw)
    bp = bbox_pred.cuda().requires_grad_(True)
    out = amd.anchor_head_decoded_loss_fused(mod, bp, bbox_targets.cuda(), bbox_weights.cuda(), labels.cuda(), anchors.cuda(),
                                             C, 1987.0, dw, dense=True)
    out.backward()
    pos, r64, r32 = gathered_oracle(bbox_pred, bbox_targets, bbox_weights, labels, anchors, C, lt, kw, dw, 5.0 / 1987.0)
    tol = 1e-5 + 3 * abs(r32['loss_sum'] - r64['loss_sum']) / (1 + abs(r64['loss_sum']))
    assert abs(out.item() - r64['loss_sum']) <= tol * (1 + abs(r64['loss_sum']))
    gflat = bp.grad.permute(0, 2, 3, 1).reshape(-1, 7)
    posd = torch.from_numpy(pos).cuda()
    check_close('config5.grad_at_positives', gflat[posd].cpu().numpy(), r64['grad_pred'],
                grad_bound(r64['grad_pred'], r32['grad_pred']))
    assert int((bp.grad != 0).sum().item()) <= 7 * len(pos)


def test_config5_waymo_nms_three_classes_bit_exact(amd):
    """configs[4], NMS half: 3 classes x 4096 score-unsorted boxes, thr 0.25, max 500 — per-class nms_gpu calls and the
    ONE batched call (nms_gpu_batched) both equal the CPU restatement's keep indices exactly."""
    cls = [nms_boxes(4096, seed=200 + c) for c in range(3)]
    want = [oracle.nms_gpu_oracle(b, s, 0.25, pre_max_size=4096, post_max_size=500) for b, s in cls]
    dev = [(torch.from_numpy(b).cuda(), torch.from_numpy(s).cuda()) for b, s in cls]
    for c in range(3):
        got = amd.nms_gpu(dev[c][0], dev[c][1], 0.25, pre_max_size=4096, post_max_size=500)
        assert got.dtype == torch.int64 and np.array_equal(got.cpu().numpy(), want[c]), c
    allb = torch.cat([b for b, _ in dev])
    alls = torch.zeros(3, 3 * 4096, device='cuda')
    allv = torch.zeros(3, 3 * 4096, dtype=torch.bool, device='cuda')
    for c in range(3):
        alls[c, c * 4096:(c + 1) * 4096] = dev[c][1]
        allv[c, c * 4096:(c + 1) * 4096] = True
    res = amd.nms_gpu_batched(allb, alls, 0.25, allv, pre_max_size=4096, post_max_size=500)
    for c in range(3):
        assert np.array_equal((res[c] - c * 4096).cpu().numpy(), want[c]), c


@pytest.mark.parametrize('name', ['kitti', 'waymo'])
def test_anchor_head_loss_at_the_configs_real_geometry(amd, name):
    """BASELINE configs[1] / [4] as the anchor head sees them in training: `GDAnchor3DHead.loss` (target assignment + classification,
    regression, direction losses, forward + backward) at the real grid — KITTI 248 x 216 x 6 anchors, batch 2, per-class assigners;
    Waymo 468 x 468 x 6, batch 1, every assigner sees every box (the head's default), dir_offset pi/4, aligned anchors.  No CPU
    restatement finishes at this size in seconds for the whole method, so the check is by properties: the target counts equal the
    restatement's on the first sample (labels bit for bit), the read-back-free form equals the eager one bit for bit, the gradients of
    the regression maps are zero exactly off the positives, and the three losses are finite and positive."""
    from oracle import anchor_targets_torch as ORA
    from test_gpu_anchor_targets import CE, FOCAL, KITTI_ASSIGNERS, KITTI_RANGES, KITTI_SIZES, SL1, head_outputs, random_gt
    dev = torch.device('cuda:0')
    if name == 'kitti':
        B, H, W, per_class, dir_offset = 2, 248, 216, True, 0.0
        anchors = amd.extras.anchor3d_range_anchors((H, W), KITTI_RANGES, KITTI_SIZES, [0, 1.57], dev)[0]
        assigners, sizes = KITTI_ASSIGNERS, KITTI_SIZES
        gts = [random_gt(18 + 5 * b, seed=200 + b, with_ignored=False) for b in range(B)]
        mod = amd.GDLoss('kld3d', fun='log1p', tau=1.0, loss_weight=5.0)
    else:
        B, H, W, per_class, dir_offset = 1, 468, 468, False, 0.7854
        rng = [[-74.88, -74.88, -0.0345, 74.88, 74.88, -0.0345], [-74.88, -74.88, -0.1188, 74.88, 74.88, -0.1188], [-74.88, -74.88, 0.0, 74.88, 74.88, 0.0]]
        sizes = [[4.73, 2.08, 1.77], [1.81, 0.84, 1.77], [0.91, 0.84, 1.74]]
        anchors = amd.extras.anchor3d_range_anchors((H, W), rng, sizes, [0, 1.57], dev, aligned=True)[0]
        assigners = [dict(type='MaxIoUAssigner', iou_calculator=dict(type='BboxOverlapsNearest3D'), pos_iou_thr=p, neg_iou_thr=n, min_pos_iou=n, ignore_iof_thr=-1)
                     for p, n in ((0.55, 0.4), (0.5, 0.3), (0.5, 0.3))]
        g = torch.Generator().manual_seed(9)
        lab = torch.randint(0, 3, (60,), generator=g)
        box = torch.cat([torch.rand(60, 2, generator=g) * 140 - 70, torch.zeros(60, 1), torch.tensor(sizes)[lab] * (0.8 + 0.4 * torch.rand(60, 3, generator=g)),
                         (torch.rand(60, 1, generator=g) * 2 - 1) * 3.14159], dim=-1)
        gts = [(box, lab)]
        mod = amd.GDLoss('gwd3d', fun='log1p', tau=0.0, loss_weight=5.0)
    tcfg = dict(assigner=assigners, allowed_border=0, code_weight=[1.0] * 7, pos_weight=-1, debug=False)
    boxes, labels = [p[0].to(dev) for p in gts], [p[1].to(dev) for p in gts]
    tg = amd.extras.anchor_head_get_targets(anchors, boxes, labels, assigners, 3, assign_per_class=per_class, dir_offset=dir_offset)
    ref = ORA.anchor_target_3d_single(anchors.cpu(), gts[0][0], gts[0][1], assigners, 3, assign_per_class=per_class, dir_offset=dir_offset)
    assert torch.equal(tg[0][0].cpu(), ref[0]) and torch.equal(tg[4][0].cpu(), ref[4]) and int((ref[0] < 3).sum()) >= len(gts[0][1])
    outs = head_outputs(B, H, W, seed=3)
    res = []
    for static in (False, True):
        g3 = [o.to(dev).requires_grad_(True) for o in outs]
        r = amd.extras.gd_anchor_head_loss(FOCAL, SL1, CE, mod, tcfg, 3, anchors, g3[0], g3[1], g3[2], boxes, labels, assign_per_class=per_class,
                                    dir_offset=dir_offset, static=static)
        losses = [r[k][0] for k in ('loss_cls', 'loss_bbox', 'loss_dir')]
        (losses[0] + losses[1] + losses[2]).backward()
        res.append(([l.detach().clone() for l in losses], [t.grad for t in g3]))
    for a, b in zip(res[0][0] + res[0][1], res[1][0] + res[1][1]):
        assert torch.equal(a, b)
    assert all(torch.isfinite(l).item() and l.item() > 0 for l in res[0][0])
    pos = (tg[0] < 3).reshape(B, H, W, 6)                                      # (B, H, W, A)
    gb = res[0][1][1].reshape(B, 6, 7, H, W).permute(0, 3, 4, 1, 2)              # (B, H, W, A, 7)
    assert gb[~pos].abs().max().item() == 0.0 and gb[pos].abs().max().item() > 0.0
