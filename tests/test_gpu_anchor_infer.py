"""Anchor-head inference slice on the MI355X (csrc/anchor_infer.hip, anchor_infer.py) against the CPU restatement of mmdet3d's
Anchor3DHead.get_bboxes_single + box3d_multiclass_nms (oracle/anchor_infer_torch.py; third party, absent: unpinned), stage by
stage: what enters the NMS within fp32 rounding of the restatement (two libraries' exp / sigmoid), and from the kernel's own
candidates on — class NMS, concatenation, max_num cut, direction correction — bit for bit."""
import math

import numpy as np
import pytest
import torch

import mmdet3d_gaussian_amd as amd
from oracle import anchor_infer_torch as ait

pytestmark = pytest.mark.extras   # frozen extras outside SURVEY.md §8: `pytest -m extras` on a GPU box (conftest.py), not part of `-m gpu`


def distinct_logits(shape, g, lo=0.001, hi=0.6):
    """logits whose sigmoid scores are pairwise distinct per sample (a shuffled grid): topk and the NMS order have one answer"""
    B = shape[0]
    n = int(np.prod(shape[1:]))
    rows = []
    for _ in range(B):
        sc = lo + (hi - lo) * (torch.randperm(n, generator=g).double() + 0.5) / n
        rows.append(torch.log(sc / (1 - sc)).float())
    return torch.stack(rows).view(shape)


def head_outputs(g, B, A, C, H, W, scene=70.0):
    cls = distinct_logits((B, A * C, H, W), g)
    bbox = torch.randn(B, A * 7, H, W, generator=g) * 0.2
    dirs = torch.randn(B, A * 2, H, W, generator=g)
    ys, xs = torch.meshgrid(torch.arange(H).float(), torch.arange(W).float(), indexing='ij')
    centers = torch.stack([xs * scene / W - scene / 2, ys * scene / H - scene / 2], -1)      # (H, W, 2)
    sizes = torch.tensor([[1.6, 3.9, 1.56], [0.6, 0.8, 1.73], [0.6, 1.76, 1.73]])[:max(A // 2, 1)]
    an = []
    for a in range(A):
        sz = sizes[(a // 2) % sizes.shape[0]]
        rot = 0.0 if a % 2 == 0 else math.pi / 2
        an.append(torch.cat([centers, torch.full((H, W, 1), -1.0), sz.expand(H, W, 3), torch.full((H, W, 1), rot)], -1))
    anchors = torch.stack(an, 2).reshape(-1, 7)                                          # (H, W, A, 7) -> (H*W*A, 7)
    return cls, bbox, dirs, anchors


def midgap_scores(cls_list, C, q):
    s = torch.cat([c.sigmoid().reshape(-1) for c in cls_list]).unique()
    i = min(max(int(q * s.numel()), 1), s.numel() - 1)
    return float((s[i - 1].double() + s[i].double()) / 2)


def run_and_check(levels, cfg, C, dir_offset, dir_limit_offset):
    """levels: list of (cls, bbox, dirs, anchors) CPU tensors"""
    B = levels[0][0].shape[0]
    gpu = [[t.cuda() for t in lv] for lv in levels]
    out, cands = amd.extras.anchor_head_get_bboxes([lv[0] for lv in gpu], [lv[1] for lv in gpu], [lv[2] for lv in gpu], [lv[3] for lv in gpu],
                                            cfg, C, dir_offset, dir_limit_offset, return_candidates=True)
    assert len(out) == B
    from oracle import nms_gpu_oracle
    for b in range(B):
        stage = {}
        ait.get_bboxes_single([lv[0][b] for lv in levels], [lv[1][b] for lv in levels], [lv[2][b] for lv in levels],
                              [lv[3] for lv in levels], cfg, C, 7, dir_offset, dir_limit_offset, stage=stage)
        cb, cs, cd = cands['boxes'][b].cpu(), cands['scores'][b].cpu(), cands['dirs'][b].cpu()
        K = cb.shape[0]
        assert stage['boxes'].shape[0] == K
        torch.testing.assert_close(cs.t(), stage['scores'], rtol=0, atol=2e-7)
        torch.testing.assert_close(cb, stage['boxes'], rtol=3e-6, atol=3e-6)
        assert torch.equal(cd.long(), stage['dirs'])
        assert float((stage['scores'] - cfg['score_thr']).abs().min()) > 2.5e-7, 'test input has a score on the threshold'
        # from the kernel's own candidates on: exact
        bx, sc, lb, dr = [], [], [], []
        for i in range(C):
            m = cs[i] > cfg['score_thr']
            if not bool(m.any()):
                continue
            idx = m.nonzero().view(-1)
            sel = idx[torch.from_numpy(np.asarray(nms_gpu_oracle(ait.bev_xyxyr(cb[idx]).numpy(), cs[i][idx].numpy(), cfg['nms_thr'],
                                                                  normal=not cfg.get('use_rotate_nms', True)), np.int64))]
            bx.append(cb[sel]); sc.append(cs[i][sel]); lb.append(torch.full((sel.numel(),), i, dtype=torch.long)); dr.append(cd[sel])
        if bx:
            bx, sc, lb, dr = torch.cat(bx), torch.cat(sc), torch.cat(lb), torch.cat(dr)
            if bx.shape[0] > cfg['max_num']:
                inds = sc.sort(descending=True, stable=True)[1][:cfg['max_num']]
                bx, sc, lb, dr = bx[inds], sc[inds], lb[inds], dr[inds]
            bx = bx.clone()
            rot = ait.limit_period(bx[:, 6] - dir_offset, dir_limit_offset, math.pi)
            bx[:, 6] = rot + dir_offset + math.pi * dr.to(bx.dtype)
        else:
            bx, sc, lb = torch.zeros(0, 7), torch.zeros(0), torch.zeros(0, dtype=torch.long)
        gb, gs, gl = out[b]
        assert gl.dtype == torch.int64
        assert torch.equal(gb.cpu(), bx), (b, gb.shape, bx.shape)
        assert torch.equal(gs.cpu(), sc) and torch.equal(gl.cpu(), lb)
    return out


def test_anchor_head_get_bboxes_waymo_config():
    """BASELINE configs[4]: nms_pre 4096, 3 classes x 2 rotations, nms_thr 0.25, score_thr 0.1, max_num 500 (map cut to 124 x 124)"""
    g = torch.Generator().manual_seed(61)
    lv = head_outputs(g, 2, 6, 3, 124, 124)
    cfg = dict(use_rotate_nms=True, nms_across_levels=False, nms_pre=4096, nms_thr=0.25, score_thr=0.1, min_bbox_size=0, max_num=500)
    out = run_and_check([lv], cfg, 3, 0.0, 1.0)
    assert all(o[0].shape == (500, 7) for o in out)


def test_anchor_head_get_bboxes_kitti_config_and_small_maps():
    """KITTI: nms_pre 100, nms_thr 0.01, score_thr 0.1, max_num 50, dir_offset 0.7854, dir_limit_offset 0; then a map smaller than
    nms_pre (every anchor enters, no selection), axis-aligned NMS, thresholds that empty a class, and two levels"""
    g = torch.Generator().manual_seed(62)
    lv = head_outputs(g, 2, 6, 3, 62, 54)
    kitti = dict(use_rotate_nms=True, nms_pre=100, nms_thr=0.01, score_thr=0.1, max_num=50)
    run_and_check([lv], kitti, 3, 0.7854, 0.0)
    small = head_outputs(g, 3, 2, 1, 10, 12)
    run_and_check([small], dict(use_rotate_nms=False, nms_pre=1000, nms_thr=0.3, score_thr=midgap_scores([small[0]], 1, 0.7), max_num=20), 1, 0.0, 1.0)
    run_and_check([small], dict(use_rotate_nms=True, nms_pre=-1, nms_thr=0.3, score_thr=0.9, max_num=20), 1, 0.0, 1.0)      # nothing passes
    two = [head_outputs(g, 2, 4, 2, 40, 40), head_outputs(g, 2, 4, 2, 20, 20, scene=60.0)]
    run_and_check(two, dict(use_rotate_nms=True, nms_pre=300, nms_thr=0.2, score_thr=midgap_scores([t[0] for t in two], 2, 0.995), max_num=120),
                  2, 0.3, 0.5)


def test_anchor_head_get_bboxes_padded_replays_as_a_hipgraph_and_errors():
    g = torch.Generator().manual_seed(63)
    a = head_outputs(g, 2, 6, 3, 62, 54)
    b = head_outputs(g, 2, 6, 3, 62, 54)
    cfg = dict(use_rotate_nms=True, nms_pre=1000, nms_thr=0.1, score_thr=0.05, max_num=80)
    static = [t.cuda().clone() for t in a]
    call = lambda: amd.extras.anchor_head_get_bboxes([static[0]], [static[1]], [static[2]], [static[3]], cfg, 3, 0.7854, 0.0, padded=True)  # noqa: E731
    call()
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        out = call()
    for src in (b, a):
        for s, t in zip(static, src):
            s.copy_(t)
        graph.replay()
        torch.cuda.synchronize()
        want = amd.extras.anchor_head_get_bboxes([src[0].cuda()], [src[1].cuda()], [src[2].cuda()], [src[3].cuda()], cfg, 3, 0.7854, 0.0)
        n = out['counts'].tolist()
        for i in range(2):
            assert n[i] == want[i][0].shape[0] > 0 and torch.equal(out['bboxes'][i, :n[i]], want[i][0])
            assert torch.equal(out['scores'][i, :n[i]], want[i][1]) and torch.equal(out['labels'][i, :n[i]], want[i][2])
    with pytest.raises(RuntimeError, match='no CPU path'):
        amd.extras.anchor_head_get_bboxes([a[0]], [a[1]], [a[2]], [a[3]], cfg, 3)
    with pytest.raises(RuntimeError, match='unsupported configuration'):
        amd.extras.anchor_head_get_bboxes([static[0]], [static[1]], [static[2]], [static[3]], dict(cfg, nms_pre=-1), 3)     # 20 088 anchors into the NMS
    with pytest.raises(RuntimeError, match='do not describe'):
        amd.extras.anchor_head_get_bboxes([static[0]], [static[1][:, :35]], [static[2]], [static[3]], cfg, 3)


def test_anchor_head_get_bboxes_random_configurations():
    g = torch.Generator().manual_seed(64)
    rng = np.random.default_rng(64)
    for it in range(8):
        B, A, C = int(rng.integers(1, 4)), int(rng.choice([2, 4, 6])), int(rng.integers(1, 5))
        H, W = int(rng.integers(6, 70)), int(rng.integers(6, 70))
        lv = head_outputs(g, B, A, C, H, W, scene=float(rng.uniform(20, 120)))
        n = H * W * A
        nms_pre = int(rng.choice([-1, 50, 400, 5000])) if n <= 4096 else int(rng.choice([60, 700, 4096]))
        cfg = dict(use_rotate_nms=bool(it % 3), nms_pre=nms_pre, nms_thr=float(rng.choice([0.01, 0.2, 0.5])),
                   score_thr=midgap_scores([lv[0]], C, float(rng.uniform(0.9, 0.999))), max_num=int(rng.integers(1, 300)))
        run_and_check([lv], cfg, C, float(rng.choice([0.0, 0.7854])), float(rng.choice([0.0, 0.5, 1.0])))


def test_saturated_scores_tie_and_the_nms_pre_cut_goes_by_index():
    """the nms_pre selection ranks the fp32 SIGMOID of the best class logit, as the reference does (`scores.max(dim=1)`
    then topk): logits beyond the saturation point (x > ~17 -> 1.0f) are ties there, and ties go by index.  Here 300 anchors carry
    saturated logits of DIFFERENT sizes (20 .. 60, shuffled); with nms_pre = 100 the candidates must be the 100 saturated anchors of
    lowest index — ranking on the raw logit would have picked the 100 largest logits instead."""
    g = torch.Generator().manual_seed(7)
    B, A, C, H, W = 1, 2, 1, 20, 25
    cls, bbox, dirs, anchors = head_outputs(g, B, A, C, H, W)
    N = H * W * A
    flat = torch.full((N,), -4.0)                                   # (cell, a) order, as the reference flattens the maps
    sat = torch.randperm(N, generator=g)[:300]
    flat[sat] = 20.0 + 40.0 * torch.rand(300, generator=g)
    assert bool((torch.sigmoid(flat[sat]) == 1.0).all())
    cls = flat.view(H, W, A).permute(2, 0, 1).reshape(1, A * C, H, W).contiguous()
    cfg = dict(use_rotate_nms=True, nms_pre=100, nms_thr=0.01, score_thr=0.1, max_num=100)
    _, cands = amd.extras.anchor_head_get_bboxes([cls.cuda()], [bbox.cuda()], [dirs.cuda()], [anchors.cuda()], cfg, C, 0.0, 1.0, return_candidates=True)
    want = torch.sort(sat)[0][:100]
    dec = ait.delta_decode(anchors[want], bbox[0].permute(1, 2, 0).reshape(-1, 7)[want])
    got_boxes = cands['boxes'][0].cpu()
    assert got_boxes.shape[0] == 100 and bool((cands['scores'][0].cpu() == 1.0).all())
    torch.testing.assert_close(got_boxes, dec.float(), rtol=3e-6, atol=3e-6)            # the 100 lowest-index saturated anchors, in index order
    by_logit = sat[torch.sort(flat[sat], descending=True)[1][:100]]
    assert set(by_logit.tolist()) != set(want.tolist())                                 # the two rankings really differ on this input
