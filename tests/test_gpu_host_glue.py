"""The two host glues above the C ABI on the GPU — _pynode.py (torch.autograd.Function + ctypes) and the optional C++ node
(csrc/torch_node.cpp) — must be interchangeable: the same extern "C" calls with the same arguments, so VALUES AND GRADIENTS ARE
BIT-EQUAL between them, for each of the four things that go through the glue (GDLoss's reduced forms, nms_gpu's scored path, the
anchor-head regression slice, scatter_reduce).  The per-function suites (test_gpu_gd_loss / test_gpu_rbox / test_gpu_head_loss /
test_voxel_scatter) run their expectations once per glue (`host_glue` fixture); this file compares the two directly.
Reference shape of the Python glue: /root/reference/mmdet3d_gaussian/ops/voxel/scatter.py:29-72 (a Function over a native op)."""
import numpy as np
import pytest
import torch

from mmdet3d_gaussian_amd import _lib

pytestmark = pytest.mark.gpu
LOSSES = ('gwd3d', 'kld3d', 'bd3d', 'jd3d', 'kld3d_symmax', 'kld3d_symmin', 'kfiou3d')


@pytest.fixture(scope='module')
def amd():
    import mmdet3d_gaussian_amd as m
    assert torch.cuda.is_available(), 'these tests need the MI355X'
    m.load_library()
    return m


def _both(fn):
    """fn() under each glue -> {'python': ..., 'cpp': ...}"""
    out = {}
    try:
        for mode in _lib.HOST_GLUE_MODES:
            _lib.set_host_glue(mode)
            assert _lib.host_glue() == mode
            out[mode] = fn()
    finally:
        _lib.set_host_glue(None)
    return out


def _same(a, b, what):
    assert len(a) == len(b), what
    for i, (x, y) in enumerate(zip(a, b)):
        if x is None or y is None:
            assert x is None and y is None, (what, i)
        else:
            assert x.dtype == y.dtype and x.shape == y.shape and torch.equal(x, y), (what, i)


def _synthetic(n, seed):
    g = torch.Generator().manual_seed(seed)
    lo = torch.tensor([0, -40, -3, 0.5, 0.5, 0.5, -np.pi])
    hi = torch.tensor([70, 40, 1, 2.5, 4.5, 2.0, np.pi])
    tgt = torch.rand(n, 7, generator=g) * (hi - lo) + lo
    pred = tgt + torch.randn(n, 7, generator=g) * torch.tensor([0.3, 0.3, 0.1, 0.1, 0.1, 0.1, 0.1])
    return pred.float().contiguous(), tgt.float().contiguous()


@pytest.mark.parametrize('n', [1, 300, 4096, 70_001])       # one-launch form (<= gd3d_one_launch_max_n) and two-stage form
@pytest.mark.parametrize('lt', LOSSES)
def test_reduced_gdloss_is_bit_equal_between_the_glues(amd, lt, n):
    pred, tgt = _synthetic(n, seed=n + len(lt))
    g = torch.Generator().manual_seed(5)
    w1 = torch.rand(n, generator=g).cuda()
    w7 = torch.rand(n, 7, generator=g).cuda()
    w7[::3] = 0
    fun = 'expm1' if lt == 'kfiou3d' else 'log1p'
    calls = [dict(), dict(weight=w1), dict(weight=w7, avg_factor=37.0), dict(weight=w7, avg_factor=torch.tensor(37.0).cuda()),
             dict(weight=torch.zeros(n, 7).cuda()), dict(weight=w1, reduction_override='sum')]

    def run():
        res = []
        for kw in calls:
            p, t = pred.cuda().requires_grad_(True), tgt.cuda().requires_grad_(True)
            out = amd.GDLoss(lt, fun=fun, loss_weight=5.0)(p, t, **kw)
            (out * 0.75).backward(retain_graph=True)          # grad_finish path
            first = (p.grad.clone(), None if t.grad is None else t.grad.clone())
            p.grad = t.grad = None
            out.backward()                                     # the retain_graph replay
            res += [out.detach().clone(), *first, p.grad.clone(), None if t.grad is None else t.grad.clone()]
        # the unit gradient: no finish launch, the forward launch's buffers as they are
        p = pred.cuda().requires_grad_(True)
        out = amd.GDLoss(lt, fun=fun)(p, tgt.cuda())
        torch.autograd.backward([out], grad_tensors=[amd.gd_loss.unit_grad('cuda')])
        return res + [out.detach().clone(), p.grad.clone()]
    r = _both(run)
    _same(r['python'], r['cpp'], (lt, n))


def test_reduced_gdloss_with_the_decode_prologue_is_bit_equal(amd):
    from mmdet3d_gaussian_amd.head_loss import anchor_decoded_gd_loss
    g = torch.Generator().manual_seed(3)
    P = 2000
    anchors = (torch.rand(P, 7, generator=g) * torch.tensor([70, 80, 1, 1.5, 3, 0.5, 1.5]) + torch.tensor([0, -40, -2, .6, .9, 1.4, 0])).cuda()
    enc_t = (torch.randn(P, 7, generator=g) * 0.2).cuda()
    enc_p = (torch.randn(P, 7, generator=g) * 0.2)
    w = torch.rand(P, 7, generator=g).cuda()

    def run():
        p = enc_p.cuda().requires_grad_(True)
        out = anchor_decoded_gd_loss(amd.GDLoss('kld3d', loss_weight=5.0), anchors, p, enc_t, w, avg_factor=torch.tensor(77.0).cuda())
        (out * 2.0).backward()
        return [out.detach().clone(), p.grad.clone()]
    r = _both(run)
    _same(r['python'], r['cpp'], 'prologue')


@pytest.mark.parametrize('n,thr,normal', [(1, 0.25, False), (777, 0.25, False), (4096, 0.01, False), (4096, 0.25, True), (9000, 0.7, False)])
def test_nms_scored_is_bit_equal_between_the_glues(amd, n, thr, normal):
    rng = np.random.default_rng(n)
    c = rng.uniform(-40, 40, (n, 2)); wl = rng.uniform(0.8, 4.5, (n, 2))
    boxes = torch.from_numpy(np.concatenate([c - wl / 2, c + wl / 2, rng.uniform(-3, 3, (n, 1))], 1).astype(np.float32)).cuda()
    scores = torch.from_numpy(rng.uniform(0, 1, n).astype(np.float32)).cuda()

    def run():
        from mmdet3d_gaussian_amd.iou3d import _nms
        keep = amd.nms_normal_gpu(boxes, scores, thr) if normal else amd.nms_gpu(boxes, scores, thr)
        cut = _nms(boxes, scores, thr, max(1, n // 2), 5, normal)
        pk, pn = _nms(boxes, scores, thr, None, 50, normal, padded=True)
        k = int(pn.item())
        return [keep, cut, pk[:k], pn]
    r = _both(run)
    _same(r['python'], r['cpp'], (n, thr, normal))
    assert r['python'][0].dtype == torch.int64 and r['python'][1].numel() <= 5


def _head_inputs(seed, B=3, A=6, H=10, W=7, C=3):
    g = torch.Generator().manual_seed(seed)
    n_per = H * W * A
    anchors = torch.rand(n_per, 7, generator=g) * torch.tensor([70, 80, 1, 1.5, 3, 0.5, 1.5]) + torch.tensor([0, -40, -2, .6, .9, 1.4, 0])
    bbox_pred = torch.randn(B, A * 7, H, W, generator=g) * 0.15
    bbox_targets = torch.randn(B, n_per, 7, generator=g) * 0.2
    bbox_weights = torch.rand(B, n_per, 7, generator=g)
    labels = torch.randint(-1, C + 2, (B, n_per), generator=g)
    return anchors, bbox_pred, bbox_targets, bbox_weights, labels, C


@pytest.mark.parametrize('form', ['sparse', 'dense', 'dense_dyn', 'decoded_only'])
def test_anchor_head_slice_is_bit_equal_between_the_glues(amd, form):
    anchors, bbox_pred, bbox_targets, bbox_weights, labels, C = _head_inputs(21)
    mod = amd.GDLoss('kld3d', fun='log1p', tau=1.0, loss_weight=5.0)
    sl1 = dict(type='SmoothL1Loss', beta=1.0 / 9.0, loss_weight=2.0)
    bt, bw, lb, an = bbox_targets.cuda(), bbox_weights.cuda(), labels.cuda(), anchors.cuda()

    def run():
        bp = bbox_pred.cuda().requires_grad_(True)
        if form == 'decoded_only':
            out = amd.anchor_head_decoded_loss_fused(mod, bp, bt, bw, lb, an, C, 19.0, decode_weight=[1, 1, .5, 1, 2, 1, 1], dense=True)
        else:
            nts = torch.tensor(19.0).cuda() if form == 'dense_dyn' else 19.0
            out = amd.anchor_head_bbox_loss(mod, sl1, bp, bt, bw, lb, an, C, nts, code_weight=[1.0] * 7, decode_weight=1,
                                            dense=form != 'sparse')
        (out * 1.25).backward(retain_graph=True)
        first = bp.grad.clone()
        bp.grad = None
        torch.autograd.backward([out], grad_tensors=[amd.gd_loss.unit_grad('cuda')])   # replay + hand-over without scaling
        return [out.detach().clone(), first, bp.grad.clone()]
    r = _both(run)
    _same(r['python'], r['cpp'], form)
    assert r['python'][1].abs().sum() > 0


@pytest.mark.parametrize('c', [3, 10, 64, 130])
@pytest.mark.parametrize('red', ['sum', 'mean', 'max'])
def test_scatter_reduce_is_bit_equal_between_the_glues(amd, c, red):
    from mmdet3d_gaussian_amd.scatter import Scatter
    rng = np.random.default_rng(c)
    n = 30_000
    coors = np.stack([rng.integers(0, g, n) for g in (40, 40, 2)], -1).astype(np.int32)
    coors[rng.random(n) < 0.03, rng.integers(0, 3)] = -1
    feats = torch.from_numpy(rng.normal(0, 1, (n, c)).astype(np.float32))
    sc = Scatter(torch.from_numpy(coors).cuda())
    gv = torch.from_numpy(rng.normal(0, 1, (sc.voxel_coors.shape[0], c)).astype(np.float32)).cuda()

    def run():
        res = []
        for dtype in (torch.float32, torch.float16):
            f = feats.to(dtype).cuda().requires_grad_(True)
            out, _ = sc.reduce(f, red)
            out.backward(gv.to(dtype))
            res += [out.detach().clone(), f.grad.clone()]
        return res
    r = _both(run)
    _same(r['python'], r['cpp'], (c, red))


def test_scatter_node_guards_in_both_glues(amd, host_glue):
    """the scatter node saves its index tensors version-checked (an in-place edit of the map or of a cached grouping
    between forward and backward raises), puts the gradient behind an error node under create_graph, and refuses a grouping
    whose `order` does not cover the points."""
    from mmdet3d_gaussian_amd.scatter import Scatter, scatter_reduce
    rng = np.random.default_rng(1)
    n = 5000
    coors = torch.from_numpy(np.stack([rng.integers(0, g, n) for g in (20, 20, 2)], -1).astype(np.int32)).cuda()
    sc = Scatter(coors)
    f = torch.randn(n, 16, device='cuda', requires_grad=True)
    out, _ = sc.reduce(f, 'max')
    assert out.grad_fn.name() == 'GDScatterReduceBackward'
    (g,) = torch.autograd.grad(out.sum(), f, create_graph=True)
    with pytest.raises(RuntimeError, match='differentiate twice'):
        g.sum().backward()
    out, _ = sc.reduce(f, 'sum')
    sc.pts_voxel_maps.add_(0)                                   # an in-place write bumps the version
    with pytest.raises(RuntimeError, match='modified by an inplace operation'):
        out.sum().backward()
    out, _ = sc.reduce(f, 'mean')
    out.sum().backward()
    with pytest.raises(RuntimeError, match='second time'):
        out.sum().backward()
    order, seg = sc._grouping
    with pytest.raises(RuntimeError, match='inconsistent index tensors'):
        scatter_reduce(f, sc.pts_voxel_maps, sc.voxel_pts_counts, 'sum', grouping=(order[:-1].contiguous(), seg))
