"""CenterPoint inference slice on the MI355X (csrc/center_infer.hip, center_infer.py) against
  * vectors of the REAL reference coders (tests/golden/center_infer.npz: select_best + decode, tie-free maps), and
  * the CPU restatement of get_bboxes (oracle/center_infer_torch.py), stage by stage: what enters the NMS within fp32 rounding
    of the reference's decode (exp / atan2 / sigmoid differ by an ulp between libraries), and from those very candidates
    on — NMS, cuts, merge, labels — bit for bit."""
import os

import numpy as np
import pytest
import torch

import mmdet3d_gaussian_amd as amd
from oracle import center_infer_torch as cit

pytestmark = pytest.mark.extras   # frozen extras outside SURVEY.md §8: `pytest -m extras` on a GPU box (conftest.py), not part of `-m gpu`
GOLD = os.path.join(os.path.dirname(__file__), 'golden', 'center_infer.npz')
NUS = dict(pc_range=[-51.2, -51.2], out_size_factor=4, voxel_size=[0.2, 0.2], norm_bbox=True)


def dev():
    return torch.device('cuda:0')


def load(name):
    z = np.load(GOLD)
    d = {k.split('.', 1)[1]: torch.from_numpy(z[k]) for k in z.files if k.startswith(name + '.')}
    cfg = dict(pc_range=z['cfg_pc_range'].tolist(), voxel_size=z['cfg_voxel_size'].tolist(),
               out_size_factor=int(z['cfg_out_size_factor']), norm_bbox=True)
    return d, cfg


def lex_topk(scores, K):
    """descending score, equal scores by ascending flat index — the selection rule of the kernel, in numpy"""
    B = scores.shape[0]
    flat = scores.reshape(B, -1)
    out = []
    for b in range(B):
        v = flat[b].numpy()
        key = np.where(np.isnan(v), np.inf, v)
        order = np.lexsort((np.arange(v.size), -key))
        nan_first = np.concatenate([np.flatnonzero(np.isnan(v)), order[~np.isnan(v[order])]])
        out.append(nan_first[:K])
    return np.stack(out)


@pytest.mark.parametrize('name', ['rev_c1', 'rev_c3', 'yaw_c2', 'yaw_c1_k500'])
def test_select_best_equals_the_reference_coder(name):
    d, cfg = load(name)
    kind = 'rev' if name.startswith('rev') else 'yaw'
    pd = {k: d[k] for k in ('reg', 'height', 'dim', 'rot', 'yaw', 'dir', 'vel') if k in d}
    coder = (amd.CenterPointBBoxCoderRev if kind == 'rev' else amd.CenterPointBBoxYawCoder)(**cfg)
    sig = d['heat'].sigmoid().to(dev())                   # the very scores the reference saw
    pred = cit.reconstruct(pd, kind).to(dev())
    s, c, xy, p = coder.select_best(sig, pred, int(d['K']))
    assert c.dtype == torch.int64 and xy.dtype == torch.int64
    assert torch.equal(s.cpu(), d['scores']) and torch.equal(c.cpu(), d['clses'])
    assert torch.equal(xy.cpu(), d['locs']) and torch.equal(p.cpu(), d['preds'])
    boxes = coder.decode(xy, p)
    torch.testing.assert_close(boxes.cpu(), d['boxes'], rtol=2e-6, atol=2e-6)   # exp / atan2 of two libraries


def test_select_best_with_equal_scores_takes_the_lowest_indices():
    d, cfg = load('yaw_ties')
    sig = d['heat'].sigmoid()
    pred = cit.reconstruct({k: d[k] for k in ('reg', 'height', 'dim', 'yaw', 'dir', 'vel')}, 'yaw')
    K = int(d['K'])
    s, c, xy, p = amd.extras.select_best(sig.to(dev()), pred.to(dev()), K)
    assert torch.equal(s.cpu(), d['scores'])              # the score VALUES are unique whatever the tie order
    B, C, H, W = sig.shape
    idx = torch.from_numpy(lex_topk(sig, K))
    assert torch.equal(c.cpu(), idx // (H * W))
    cell = idx % (H * W)
    assert torch.equal(xy.cpu(), torch.stack((cell % W, cell // W), -1))
    want = pred.permute(0, 2, 3, 1).reshape(B, H * W, -1)
    assert torch.equal(p.cpu(), torch.stack([want[b][cell[b]] for b in range(B)]))


@pytest.mark.parametrize('shape,K', [((2, 1, 128, 128), 500), ((1, 3, 300, 300), 4096), ((3, 2, 40, 24), 960), ((1, 1, 8, 8), 64)])
def test_select_best_degenerate_maps(shape, K):
    """constant maps (every key equal: the ordered tie path when the map is larger than the sort buffer), two-valued maps,
    NaN and infinities, K == H*W"""
    B, C, H, W = shape
    g = torch.Generator().manual_seed(5)
    pred = torch.randn(B, 4, H, W, generator=g)
    maps = [torch.zeros(shape), torch.full(shape, -3.25)]
    two = torch.where(torch.rand(shape, generator=g) < 0.3, torch.tensor(0.75), torch.tensor(0.25))
    maps.append(two)
    wild = torch.randn(shape, generator=g)
    flat = wild.view(-1)
    flat[torch.randint(0, flat.numel(), (7,), generator=g)] = float('nan')
    flat[torch.randint(0, flat.numel(), (5,), generator=g)] = float('inf')
    flat[torch.randint(0, flat.numel(), (5,), generator=g)] = float('-inf')
    flat[torch.randint(0, flat.numel(), (9,), generator=g)] = -0.0
    flat[torch.randint(0, flat.numel(), (9,), generator=g)] = 0.0
    maps.append(wild)
    for m in maps:
        s, c, xy, p = amd.extras.select_best(m.to(dev()), pred.to(dev()), K)
        idx = torch.from_numpy(lex_topk(m, K))
        cell = idx % (H * W)
        assert torch.equal(c.cpu(), idx // (H * W))
        assert torch.equal(xy.cpu(), torch.stack((cell % W, cell // W), -1))
        want = torch.stack([m.view(B, -1)[b][idx[b]] for b in range(B)])
        assert torch.equal(s.cpu().nan_to_num(nan=7.0), want.nan_to_num(nan=7.0))


@pytest.mark.parametrize('shape,K', [((2, 3, 468, 468), 500), ((1, 2, 320, 320), 2000), ((1, 3, 468, 468), 4096), ((3, 1, 400, 330), 64)])
def test_select_best_wide_maps(shape, K):
    """above 131072 cells per group the threshold and the filtering pass are chip-wide launches of their own; candidates that
    fit the LDS buffer directly, candidates thinned by a second threshold, and (equal keys everywhere) the exact fallback"""
    B, C, H, W = shape
    g = torch.Generator().manual_seed(9)
    pred = torch.randn(B, 3, H, W, generator=g)
    for kind in ('normal', 'uniform', 'quantised'):
        m = torch.randn(shape, generator=g) if kind == 'normal' else torch.rand(shape, generator=g)
        if kind == 'quantised':
            m = (m * 16).floor() / 16
        s, c, xy, p = amd.extras.select_best(m.to(dev()), pred.to(dev()), K)
        idx = torch.from_numpy(lex_topk(m, K))
        cell = idx % (H * W)
        assert torch.equal(c.cpu(), idx // (H * W)), kind
        assert torch.equal(xy.cpu(), torch.stack((cell % W, cell // W), -1)), kind
        assert torch.equal(s.cpu(), torch.stack([m.view(B, -1)[b][idx[b]] for b in range(B)])), kind


def distinct_heat(shape, g, lo=0.002, hi=0.7):
    """Logits whose sigmoid scores are pairwise distinct per sample and stay so through fp32 rounding (a shuffled regular grid,
    spacing >= 1e-6): torch.topk then has ONE answer and the kernel's tie rule cannot show.  (Scores of a trained head that
    collide after the sigmoid are ordered by their logits here, arbitrarily in the reference: DESIGN.md.)"""
    B = shape[0]
    n = int(np.prod(shape[1:]))
    assert (hi - lo) / n >= 1e-6
    rows = []
    for _ in range(B):
        sc = lo + (hi - lo) * (torch.randperm(n, generator=g).double() + 0.5) / n
        rows.append(torch.log(sc / (1 - sc)).float())
    heat = torch.stack(rows).view(shape)
    top = heat.sigmoid().view(B, -1).sort(dim=1)[0]
    assert bool((top[:, 1:] > top[:, :-1]).all())
    return heat


def midgap(tasks, K, q):
    """a score threshold in the middle of a gap of the candidates' scores, near their q-quantile: no score within half a grid
    step of it, so the two sigmoid implementations cannot disagree about the mask"""
    top = torch.cat([pd['heatmap'].sigmoid().view(pd['heatmap'].shape[0], -1).topk(K)[0].reshape(-1) for pd in tasks]).unique()
    top = top[top > 1e-3]                  # a sample pushed far below every threshold does not take part
    i = min(max(int(q * top.numel()), 1), top.numel() - 1)
    return float((top[i - 1].double() + top[i].double()) / 2)


def make_tasks(g, B, H, W, classes, kind, with_reg=True, with_vel=True):
    tasks = []
    for C in classes:
        heat = distinct_heat((B, C, H, W), g)
        yaw = (torch.rand(B, 1, H, W, generator=g) * 2 - 1) * 3.3
        pd = dict(heatmap=heat, height=torch.rand(B, 1, H, W, generator=g) * 4 - 3,
                  dim=torch.randn(B, 3, H, W, generator=g) * 0.3 + 0.5)
        if with_reg:
            pd['reg'] = torch.rand(B, 2, H, W, generator=g)
        if kind == 'rev':
            pd['rot'] = torch.cat([yaw.sin(), yaw.cos()], 1) + torch.randn(B, 2, H, W, generator=g) * 0.1
        else:
            turn = torch.randint(-2, 3, (B, 1, H, W), generator=g).float() * (np.pi / 2)
            pd['yaw'] = yaw
            pd['dir'] = torch.cat([(yaw + turn).sin(), (yaw + turn).cos()], 1) + torch.randn(B, 2, H, W, generator=g) * 0.1
        if with_vel:
            pd['vel'] = torch.randn(B, 2, H, W, generator=g)
        tasks.append(pd)
    return tasks


def run_and_check(tasks, kind, cfg, test_cfg, classes, wrap=False):
    coder = (amd.CenterPointBBoxCoderRev if kind == 'rev' else amd.CenterPointBBoxYawCoder)(**cfg)
    gpu = [{k: v.to(dev()) for k, v in pd.items()} for pd in tasks]
    arg = tuple([pd] for pd in gpu) if wrap else gpu
    out, cands = amd.extras.center_head_get_bboxes(arg, coder, test_cfg, classes, return_candidates=True)
    stage = {}
    cit.get_bboxes(tasks, kind, cfg, test_cfg, classes, stage=stage)
    B = tasks[0]['heatmap'].shape[0]
    K = test_cfg['max_per_img']
    rets = []
    for t, cd in enumerate(cands):
        st = stage[t]
        counts = cd['counts'].cpu()
        # scores within 2 ulp of the threshold may legitimately fall on either side; the generators keep clear of it
        margin = (st['scores'] - test_cfg['score_threshold']).abs().min()
        assert margin > 2.5e-7, 'test input has a score on the threshold'
        assert torch.equal(counts.long(), st['mask'].sum(1)), (t, counts, st['mask'].sum(1))
        ret_task = []
        for b in range(B):
            n = int(counts[b])
            m = st['mask'][b]
            bx, sc, lb = cd['boxes'][b, :n].cpu(), cd['scores'][b, :n].cpu(), cd['labels'][b, :n].cpu()
            assert torch.equal(lb.long(), st['clses'][b][m])
            torch.testing.assert_close(sc, st['scores'][b][m], rtol=0, atol=2e-7)
            torch.testing.assert_close(bx, st['boxes'][b][m], rtol=3e-6, atol=3e-6)
            # from these candidates on everything is exact: the restated NMS on the kernel's own boxes
            if test_cfg['nms_type'] == 'circle':
                from oracle import circle_nms
                dets = torch.cat([bx[:, [0, 1]], sc.view(-1, 1)], 1).numpy()
                keep = np.asarray(circle_nms(dets, test_cfg['min_radius'][t], post_max_size=test_cfg['post_max_size']), np.int64)
            elif n > 0:
                from oracle import nms_gpu_oracle
                keep = nms_gpu_oracle(cit.bev_xyxyr(bx).numpy(), sc.numpy(), test_cfg['nms_thr'],
                                      pre_max_size=test_cfg.get('pre_max_size'), post_max_size=test_cfg.get('post_max_size'))
            else:
                keep = np.zeros(0, np.int64)
            keep = torch.from_numpy(np.asarray(keep, np.int64))
            ret_task.append((bx[keep], sc[keep], lb[keep]))
        rets.append(ret_task)
    assert len(out) == B
    flag = np.concatenate([[0], np.cumsum(classes)])
    for b in range(B):
        want_b = torch.cat([r[b][0] for r in rets]).clone()
        want_b[:, 2] = want_b[:, 2] - want_b[:, 5] * 0.5
        want_s = torch.cat([r[b][1] for r in rets])
        want_l = torch.cat([(r[b][2] + int(flag[t])) for t, r in enumerate(rets)]).int()
        got_b, got_s, got_l = out[b]
        assert got_l.dtype == torch.int32
        assert torch.equal(got_b.cpu(), want_b), (b, got_b.shape, want_b.shape)
        assert torch.equal(got_s.cpu(), want_s) and torch.equal(got_l.cpu(), want_l)
    return out


NUS_TEST = dict(post_center_limit_range=[-61.2, -61.2, -10.0, 61.2, 61.2, 10.0], max_per_img=500, score_threshold=0.1,
                nms_type='rotate', nms_thr=0.2, pre_max_size=1000, post_max_size=83, min_radius=[4, 12, 10, 1, 0.85, 0.175])


@pytest.mark.parametrize('kind', ['yaw', 'rev'])
def test_get_bboxes_nuscenes_geometry(kind):
    """6 tasks (classes 1,2,2,1,2,2), 128 x 128 maps, batch 2, test_cfg of configs/_base_/models/centerpoint_02pillar_second_secfpn_nus.py"""
    g = torch.Generator().manual_seed(21)
    classes = [1, 2, 2, 1, 2, 2]
    tasks = make_tasks(g, 2, 128, 128, classes, kind)
    out = run_and_check(tasks, kind, NUS, dict(NUS_TEST, score_threshold=midgap(tasks, 500, 0.4)), classes, wrap=True)
    assert all(0 < o[0].shape[0] <= 6 * 83 and o[0].shape[1] == 9 for o in out)
    out = run_and_check(tasks, kind, NUS, NUS_TEST, classes)      # the config's own 0.1: every candidate passes
    assert all(o[0].shape[0] == 6 * 83 for o in out)


def test_get_bboxes_circle_nms_and_no_reg_no_vel():
    g = torch.Generator().manual_seed(22)
    classes = [1, 2, 1]
    tasks = make_tasks(g, 3, 64, 80, classes, 'yaw', with_reg=False, with_vel=False)
    cfg = dict(NUS_TEST, nms_type='circle', max_per_img=300, score_threshold=midgap(tasks, 300, 0.3), min_radius=[4, 0.85, 0.175],
               post_max_size=40)
    out = run_and_check(tasks, 'yaw', NUS, cfg, classes)
    assert out[0][0].shape[1] == 7


def test_get_bboxes_cuts_thresholds_and_empty_groups():
    g = torch.Generator().manual_seed(23)
    classes = [2, 2]
    tasks = make_tasks(g, 2, 48, 48, classes, 'rev')
    tasks[1]['heatmap'][1] -= 20.0                           # one (task, sample) group loses every candidate
    for pre, post, q in ((40, 10, 0.5), (None, None, 0.7), (7, 50, 0.2)):
        cfg = dict(NUS_TEST, max_per_img=200, score_threshold=midgap(tasks, 200, q), pre_max_size=pre, post_max_size=post, nms_thr=0.1)
        run_and_check(tasks, 'rev', NUS, cfg, classes)
    # the limit-range expression of the reference: an upper limit below 1 changes its meaning (boolean <= hi)
    cfg = dict(NUS_TEST, max_per_img=200, post_center_limit_range=[-61.2, 0.0, -10.0, 61.2, 0.5, 10.0])
    run_and_check(tasks, 'rev', NUS, cfg, classes)
    cfg = dict(NUS_TEST, max_per_img=200, post_center_limit_range=[-61.2, -61.2, -10.0, 61.2, 61.2, -1.0])
    out = run_and_check(tasks, 'rev', NUS, cfg, classes)
    assert all(o[0].shape[0] == 0 for o in out)
    cfg = dict(NUS_TEST, max_per_img=200, post_center_limit_range=None)
    run_and_check(tasks, 'rev', NUS, cfg, classes)


def test_get_bboxes_many_tasks_take_two_launches():
    g = torch.Generator().manual_seed(24)
    classes = [1] * 13
    tasks = make_tasks(g, 1, 32, 32, classes, 'yaw')
    cfg = dict(NUS_TEST, max_per_img=100, score_threshold=midgap(tasks, 100, 0.5))
    run_and_check(tasks, 'yaw', NUS, cfg, classes)


def test_get_bboxes_waymo_sized_map_and_k():
    """468 x 468 x 3 classes, max_per_img 4096 (configs/_base_/models/centerpoint_*_waymo): the multi-level select"""
    g = torch.Generator().manual_seed(25)
    tasks = make_tasks(g, 1, 468, 468, [3], 'yaw')
    cfg = dict(NUS_TEST, max_per_img=4096, score_threshold=midgap(tasks, 4096, 0.25), nms_thr=0.25, pre_max_size=4096, post_max_size=500,
               post_center_limit_range=[-80, -80, -10.0, 80, 80, 10.0])
    wcfg = dict(pc_range=[-74.88, -74.88], out_size_factor=1, voxel_size=[0.32, 0.32], norm_bbox=True)
    out = run_and_check(tasks, 'yaw', wcfg, cfg, [3])
    assert out[0][0].shape[0] == 500


def test_errors():
    g = torch.Generator().manual_seed(26)
    tasks = make_tasks(g, 1, 16, 16, [1], 'yaw')
    coder = amd.CenterPointBBoxYawCoder(**NUS)
    with pytest.raises(RuntimeError, match='no CPU path'):
        amd.extras.center_head_get_bboxes(tasks, coder, NUS_TEST, [1])
    gpu = [{k: v.to(dev()) for k, v in tasks[0].items()}]
    with pytest.raises(RuntimeError, match='out of range'):
        amd.extras.center_head_get_bboxes(gpu, coder, NUS_TEST, [1])          # 500 > 16 * 16: torch.topk raises in the reference
    with pytest.raises(AssertionError):
        amd.extras.center_head_get_bboxes(gpu, coder, dict(NUS_TEST, max_per_img=50, nms_type='soft'), [1])
    del gpu[0]['dir']
    with pytest.raises(RuntimeError, match='dir'):
        amd.extras.center_head_get_bboxes(gpu, coder, dict(NUS_TEST, max_per_img=50), [1])


def test_get_bboxes_replays_as_a_hipgraph():
    """padded=True: nothing is read back, so the slice can be captured once and replayed on new head outputs in the same
    buffers; the replayed detections equal the eager call's"""
    g = torch.Generator().manual_seed(27)
    classes = [1, 2, 2, 1, 2, 2]
    coder = amd.CenterPointBBoxYawCoder(**NUS)
    first = make_tasks(g, 2, 128, 128, classes, 'yaw')
    second = make_tasks(g, 2, 128, 128, classes, 'yaw')
    static = [{k: v.to(dev()).clone() for k, v in pd.items()} for pd in first]
    cfg = dict(NUS_TEST, score_threshold=midgap(first + second, 500, 0.4))
    amd.extras.center_head_get_bboxes(static, coder, cfg, classes, padded=True)          # warm-up outside the capture
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        out = amd.extras.center_head_get_bboxes(static, coder, cfg, classes, padded=True)
    for src in (second, first):
        for pd_s, pd_n in zip(static, src):
            for k in pd_s:
                pd_s[k].copy_(pd_n[k])
        graph.replay()
        torch.cuda.synchronize()
        want = amd.extras.center_head_get_bboxes([{k: v.to(dev()) for k, v in pd.items()} for pd in src], coder, cfg, classes)
        n = out['counts'].tolist()
        for b in range(2):
            assert n[b] == want[b][0].shape[0] > 0
            assert torch.equal(out['bboxes'][b, :n[b]], want[b][0]) and torch.equal(out['scores'][b, :n[b]], want[b][1])
            assert torch.equal(out['labels'][b, :n[b]], want[b][2])


def test_get_bboxes_kitti_centerpoint_geometry():
    """configs/_base_/models/pillarmvf_centerpoint_016pillar_second_secfpn_kitti.py: maps of 248 rows x 216 columns, tasks with
    1 / 1 / 2 heat-map classes, CenterPointBBoxCoderRev (rot = atan2(sin, cos)), its test_cfg"""
    g = torch.Generator().manual_seed(28)
    classes = [1, 1, 2]
    tasks = make_tasks(g, 2, 248, 216, classes, 'rev')
    cfg = dict(pc_range=[0, -39.68], out_size_factor=2, voxel_size=[0.16, 0.16], norm_bbox=True)
    test_cfg = dict(post_center_limit_range=[-10, -49.68, -10, 79.12, 49.68, 10], max_per_img=500, score_threshold=midgap(tasks, 500, 0.3),
                    nms_type='rotate', nms_thr=0.2, pre_max_size=1000, post_max_size=83, min_radius=[0.85, 0.175, 4])
    out = run_and_check(tasks, 'rev', cfg, test_cfg, classes)
    assert all(o[0].shape[1] == 9 and o[0].shape[0] > 0 for o in out)
    run_and_check(tasks, 'rev', cfg, dict(test_cfg, nms_type='circle'), classes)


def test_select_best_random_shapes_and_k():
    """40 random (B, C, H, W, K) with heat maps of different value distributions (dense ties, heavy tails, plateaus, a few cells):
    the kernel's rule (score descending, equal scores by ascending class, y, x) against numpy"""
    g = torch.Generator().manual_seed(77)
    rng = np.random.default_rng(77)
    for it in range(40):
        B, C = int(rng.integers(1, 4)), int(rng.integers(1, 4))
        H, W = int(rng.integers(1, 200)), int(rng.integers(1, 200))
        K = int(rng.integers(1, min(H * W, 1200) + 1))
        kind = it % 5
        m = torch.randn(B, C, H, W, generator=g)
        if kind == 1:
            m = (m * 3).round() / 3                       # a few dozen distinct values
        elif kind == 2:
            m = torch.where(m > 1.5, m * 50, torch.full_like(m, -7.0))   # a plateau with a sparse heavy tail
        elif kind == 3:
            m = m.exp() * 1e-3                            # everything squeezed into two exponent bins
        elif kind == 4:
            m = -m.abs() - 100.0                          # all negative
        pred = torch.randn(B, 2, H, W, generator=g)
        s, c, xy, p = amd.extras.select_best(m.to(dev()), pred.to(dev()), K)
        idx = torch.from_numpy(lex_topk(m, K))
        cell = idx % (H * W)
        assert torch.equal(c.cpu(), idx // (H * W)), (it, B, C, H, W, K, kind)
        assert torch.equal(xy.cpu(), torch.stack((cell % W, cell // W), -1)), (it, B, C, H, W, K, kind)
        assert torch.equal(s.cpu(), torch.stack([m.view(B, -1)[b][idx[b]] for b in range(B)])), (it, B, C, H, W, K, kind)


def test_get_bboxes_random_configurations():
    """10 random heads (tasks, classes, map shapes, batch, K, coder, cuts, NMS kind, with / without reg and vel): stage-wise check"""
    g = torch.Generator().manual_seed(99)
    rng = np.random.default_rng(99)
    for it in range(10):
        B = int(rng.integers(1, 4))
        classes = [int(rng.integers(1, 4)) for _ in range(int(rng.integers(1, 5)))]
        H, W = int(rng.integers(12, 90)), int(rng.integers(12, 90))
        K = int(rng.integers(8, min(H * W, 400)))
        kind = 'rev' if it % 2 else 'yaw'
        tasks = make_tasks(g, B, H, W, classes, kind, with_reg=bool(rng.integers(0, 2)), with_vel=bool(rng.integers(0, 2)))
        osf, vs = int(rng.choice([2, 4])), float(rng.choice([0.16, 0.2]))
        cfg = dict(pc_range=[-W * osf * vs / 2, -H * osf * vs / 2], out_size_factor=osf, voxel_size=[vs, vs], norm_bbox=True)
        circle = bool(it % 3 == 2)
        test_cfg = dict(post_center_limit_range=None if it % 4 == 3 else [-100.0, -100.0, -10.0, 100.0, 100.0, 10.0], max_per_img=K,
                        score_threshold=midgap(tasks, K, float(rng.uniform(0.05, 0.8))), nms_type='circle' if circle else 'rotate',
                        nms_thr=float(rng.choice([0.1, 0.2, 0.5])), pre_max_size=None if circle else int(rng.integers(4, 2 * K)),
                        post_max_size=int(rng.integers(1, K + 10)), min_radius=[float(rng.uniform(0.2, 6.0)) for _ in classes])
        run_and_check(tasks, kind, cfg, test_cfg, classes, wrap=bool(it % 2))
