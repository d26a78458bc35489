"""GPU robustness of the host layer around the fused kernel: dtypes, layouts, which inputs need gradients,
hipGraph capture, non-default streams, the reference's `(pred * weight).sum()` early-out."""
import numpy as np
import pytest
import torch

import oracle

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def amd():
    import mmdet3d_gaussian_amd as m
    m.load_library()
    return m


def _pairs(n, seed=0):
    rng = np.random.default_rng(seed)
    t = np.stack([rng.uniform(0, 70, n), rng.uniform(-40, 40, n), rng.uniform(-3, 1, n), rng.uniform(.5, 2.5, n),
                  rng.uniform(.5, 4.5, n), rng.uniform(.5, 2, n), rng.uniform(-3, 3, n)], -1).astype(np.float32)
    p = (t + rng.normal(0, 0.15, (n, 7))).astype(np.float32)
    return p, t


@pytest.mark.parametrize('dtype,tol', [(torch.float64, 2e-6), (torch.float16, 3e-2), (torch.bfloat16, 2e-1)])
def test_other_float_dtypes_are_computed_in_fp32_and_cast_back(amd, dtype, tol):
    p, t = _pairs(500)
    pred = torch.from_numpy(p).cuda().to(dtype).requires_grad_(True)
    tgt = torch.from_numpy(t).cuda().to(dtype)
    out = amd.GDLoss('kld3d', loss_weight=5.0)(pred, tgt)
    assert out.dtype == dtype
    out.backward()
    assert pred.grad.dtype == dtype and pred.grad.shape == pred.shape
    ref = oracle.gd_loss(pred.detach().float().cpu().numpy(), tgt.float().cpu().numpy(), oracle.make_params('kld3d'), scale=5.0 / 500)
    assert abs(out.item() - ref['loss_sum']) <= tol * (1 + abs(ref['loss_sum']))


def test_non_contiguous_and_extra_columns(amd):
    """Heads slice `[..., :7]` out of wider tensors (gd_centerpoint_head.py:414-423): strided views must work and the
    gradient must land in the right columns of the base tensor."""
    p, t = _pairs(300, 1)
    base = torch.zeros(300, 9).cuda()
    base[:, :7] = torch.from_numpy(p).cuda()
    base.requires_grad_(True)
    out = amd.GDLoss('gwd3d', reduction='sum')(base[..., :7], torch.from_numpy(t).cuda())
    out.backward()
    ref = oracle.gd_loss(p, t, oracle.make_params('gwd3d'))
    assert abs(out.item() - ref['loss_sum']) <= 1e-5 * (1 + abs(ref['loss_sum']))
    g = base.grad.cpu().numpy()
    assert np.abs(g[:, 7:]).max() == 0
    np.testing.assert_allclose(g[:, :7], ref['grad_pred'], rtol=0, atol=3e-5 * (1 + np.abs(ref['grad_pred']).max()))
    # transposed storage
    pt = torch.from_numpy(p).cuda().t().contiguous().t().requires_grad_(True)
    assert not pt.is_contiguous()
    out2 = amd.GDLoss('gwd3d', reduction='sum')(pt, torch.from_numpy(t).cuda())
    assert abs(out2.item() - out.item()) <= 1e-6 * (1 + abs(out.item()))


@pytest.mark.parametrize('lt', ['gwd3d', 'kld3d', 'bd3d', 'jd3d', 'kfiou3d'])
def test_gradient_only_wrt_target_or_both(amd, lt):
    p, t = _pairs(400, 2)
    fun = 'expm1' if lt == 'kfiou3d' else 'log1p'
    ref = oracle.gd_loss(p, t, oracle.make_params(lt, fun=fun), scale=1.0)
    mod = amd.GDLoss(lt, fun=fun, reduction='sum')
    tg = torch.from_numpy(t).cuda().requires_grad_(True)
    out = mod(torch.from_numpy(p).cuda(), tg)
    out.backward()
    sc = 1 + np.abs(ref['grad_target']).max()
    np.testing.assert_allclose(tg.grad.cpu().numpy(), ref['grad_target'], rtol=0, atol=5e-5 * sc)
    pp = torch.from_numpy(p).cuda().requires_grad_(True); tg2 = torch.from_numpy(t).cuda().requires_grad_(True)
    mod(pp, tg2).backward()
    np.testing.assert_allclose(pp.grad.cpu().numpy(), ref['grad_pred'], rtol=0, atol=5e-5 * (1 + np.abs(ref['grad_pred']).max()))
    assert torch.equal(tg2.grad, tg.grad)
    # no gradient requested at all (inference / no_grad)
    with torch.no_grad():
        v = mod(torch.from_numpy(p).cuda(), torch.from_numpy(t).cuda())
    assert not v.requires_grad and abs(v.item() - ref['loss_sum']) <= 1e-5 * (1 + abs(ref['loss_sum']))


def test_hipgraph_capture_of_forward_backward_matches_eager(amd):
    p, t = _pairs(70_000, 3)
    pred = torch.from_numpy(p).cuda().requires_grad_(True); tgt = torch.from_numpy(t).cuda()
    mod = amd.GDLoss('bd3d', loss_weight=5.0)
    def step():
        pred.grad = None
        l = mod(pred, tgt); l.backward()
        return l.detach(), pred.grad
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3):
            l_eager, g_eager = step()
    torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
    l_eager, g_eager = l_eager.clone(), g_eager.clone()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        l_g, g_g = step()
    pred.data.add_(0.01)          # new input values in the static buffers
    graph.replay(); torch.cuda.synchronize()
    l_new, g_new = step()
    assert torch.equal(l_g, l_new) and torch.equal(g_g, g_new)
    assert not torch.equal(l_g, l_eager)


def test_non_default_stream(amd):
    p, t = _pairs(200_000, 4)
    pred = torch.from_numpy(p).cuda().requires_grad_(True); tgt = torch.from_numpy(t).cuda()
    mod = amd.GDLoss('kld3d')
    want = mod(pred, tgt); want.backward(); g_want = pred.grad.clone(); pred.grad = None
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        out = mod(pred, tgt); out.backward()
    s.synchronize()
    assert torch.equal(out, want) and torch.equal(pred.grad, g_want)


def test_zero_weight_early_out_keeps_graph(amd):
    """ref :290-292: all weights <= 0 and reduction != 'none' -> (pred * weight).sum() (0 with a graph)."""
    p, t = _pairs(64, 5)
    pred = torch.from_numpy(p).cuda().requires_grad_(True)
    out = amd.GDLoss('gwd3d')(pred, torch.from_numpy(t).cuda(), torch.zeros(64, 7).cuda(), avg_factor=10.0)
    assert out.item() == 0.0 and out.requires_grad
    out.backward()
    assert pred.grad.abs().max().item() == 0.0
    # reduction 'none' does NOT take the early-out
    v = amd.GDLoss('gwd3d', reduction='none')(pred, torch.from_numpy(t).cuda(), torch.zeros(64).cuda())
    assert v.shape == (64,) and v.abs().max().item() == 0.0


def test_avg_factor_as_device_tensor_no_host_sync(amd):
    p, t = _pairs(128, 6)
    pred = torch.from_numpy(p).cuda().requires_grad_(True)
    mod = amd.GDLoss('bd3d', loss_weight=2.0)
    a = mod(pred, torch.from_numpy(t).cuda(), avg_factor=torch.tensor(37.0).cuda())
    a.backward(); g1 = pred.grad.clone(); pred.grad = None
    b = mod(pred, torch.from_numpy(t).cuda(), avg_factor=37.0)
    b.backward()
    assert abs(a.item() - b.item()) <= 1e-6 * (1 + abs(b.item()))
    assert torch.allclose(g1, pred.grad, rtol=1e-5, atol=1e-8)


def test_dispatch_bound_events_time_the_fused_kernel_and_change_nothing(amd):
    """gd3d_loss_fused_timed (bench.py's in-region timing): identical bits to the plain call, and the event pair bound to
    the dispatch yields a positive duration that is no longer than an outer event bracket around the same launches."""
    import time
    from mmdet3d_gaussian_amd import gd_loss as gdl
    p, t = _pairs(1_000_003, 7)
    pred = torch.from_numpy(p).cuda().requires_grad_(True)
    tgt = torch.from_numpy(t).cuda()
    mod = amd.GDLoss('bd3d', loss_weight=5.0)
    plain = mod(pred, tgt)
    plain.backward()
    g_plain = pred.grad.clone()
    pred.grad = None
    timers = []
    gdl.PROFILE_EVENTS = timers
    try:
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        timed = mod(pred, tgt)
        e1.record()
    finally:
        gdl.PROFILE_EVENTS = None
    timed.backward()
    torch.cuda.synchronize()
    assert len(timers) == 1
    assert timed.item() == plain.item()
    assert torch.equal(pred.grad, g_plain)
    ms = timers[0].elapsed_ms()
    assert 0.0 < ms <= e0.elapsed_time(e1)
    # 1 M pairs move >= 84 MB: at the chip's 8 TB/s peak that is 10.5 us — a shorter reading would be a broken timer
    assert ms * 1e-3 >= 84.0 * 1_000_003 / 8e12
    del timers[:]
    time.sleep(0)  # timers destroyed here: hipEventDestroy through the ABI must not raise


def test_anchor_head_entry_points_take_half_precision_and_strided_inputs(amd):
    """the round-3 head slices under AMP-like conditions: bf16 / fp16 head maps (computed in fp32, gradients cast back to the maps' dtype),
    channels-last (non-contiguous) maps, int32 labels, ground truth on another dtype: the same values as the plain fp32 call within the
    input rounding"""
    from test_gpu_anchor_targets import CE, FOCAL, SL1, TRAIN_CFG, head_outputs, kitti_anchors, random_gt
    dev = torch.device('cuda:0')
    H, W = 24, 20
    anchors = kitti_anchors(H, W).to(dev)
    b, l = random_gt(9, seed=3, with_ignored=False)
    outs = [o.to(dev) for o in head_outputs(2, H, W, seed=2)]
    mod = amd.GDLoss('kld3d', fun='log1p', tau=1.0, loss_weight=5.0)

    def run(maps, boxes, labels):
        maps = [m.clone().requires_grad_(True) for m in maps]
        r = amd.extras.gd_anchor_head_loss(FOCAL, SL1, CE, mod, TRAIN_CFG, 3, anchors, maps[0], maps[1], maps[2], boxes, labels)
        tot = r['loss_cls'][0] + r['loss_bbox'][0] + r['loss_dir'][0]
        tot.backward()
        return [r[k][0].detach().float() for k in ('loss_cls', 'loss_bbox', 'loss_dir')], [m.grad for m in maps]
    base_l, base_g = run(outs, [b.to(dev)] * 2, [l.to(dev)] * 2)
    for dtype, tol in ((torch.bfloat16, 3e-2), (torch.float16, 4e-3)):
        ls, gs = run([o.to(dtype) for o in outs], [b.to(dev).double()] * 2, [l.to(dev).int()] * 2)
        assert all(g.dtype == dtype for g in gs)
        for x, y in zip(ls, base_l):
            assert abs(x.item() - y.item()) <= tol * (1 + abs(y.item()))
    cl = [o.to(memory_format=torch.channels_last) for o in outs]
    assert not cl[0].is_contiguous()
    ls, gs = run(cl, [b.to(dev)] * 2, [l.to(dev)] * 2)
    for x, y in zip(ls, base_l):
        assert x.item() == y.item()
    for g, y in zip(gs, base_g):
        assert torch.equal(g.contiguous(), y)
    # inference side: bf16 maps in, fp32 boxes out, same detections as from the maps rounded to bf16 and widened again
    cfg = dict(use_rotate_nms=True, nms_pre=256, nms_thr=0.01, score_thr=0.05, max_num=50)
    flat = anchors.reshape(-1, 7)
    wide = [o.to(torch.bfloat16).float() for o in outs]
    a = amd.extras.anchor_head_get_bboxes([outs[0].to(torch.bfloat16)], [outs[1].to(torch.bfloat16)], [outs[2].to(torch.bfloat16)], [flat], cfg, 3, 0.0, 1.0)
    c = amd.extras.anchor_head_get_bboxes([wide[0]], [wide[1]], [wide[2]], [flat], cfg, 3, 0.0, 1.0)
    for x, y in zip(a, c):
        assert x[0].shape == y[0].shape and torch.allclose(x[0].float(), y[0], rtol=1e-2, atol=1e-2) and torch.equal(x[2], y[2])
