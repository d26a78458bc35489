"""GPU parity tests proper: the HIP fused kernels, through the C ABI / GDLoss, against
  (a) golden vectors produced by the REAL reference (fp32 and fp64 runs), and
  (b) the fp64 CPU oracle on seeded inputs (sizes the oracle finishes in seconds), and
  (c) size-independent properties at BASELINE.json's full size (10 M pairs).
Tolerance: see tests/gd_golden.py (1e-5 relative + the reference's own fp32 noise as yardstick)."""
import ctypes

import numpy as np
import pytest
import torch

import oracle
from gd_golden import (GRAD_TOL, LOSS_TOL, check_close, families, grad_bound, index, loss_bound, module,
                       oracle32_bounds, pair_case_names, pairs)

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def amd():
    import mmdet3d_gaussian_amd as m
    assert torch.cuda.is_available(), 'these tests need the MI355X'
    m.load_library()
    return m


def _dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


@pytest.mark.parametrize('case', pair_case_names())
def test_pairs_against_reference_golden(amd, case):
    """Per-pair loss + grad wrt pred AND target for every loss type / parameter set / input family."""
    c = index()['pairs']['cases'][case]
    g = pairs()
    for fam in families():
        pred = _dev(g[f'in.{fam}.pred']).requires_grad_(True)
        tgt = _dev(g[f'in.{fam}.target']).requires_grad_(True)
        kw = {k: (tuple(v) if isinstance(v, list) else v) for k, v in c['kwargs'].items()}
        mod = amd.GDLoss(c['loss_type'], reduction='none', loss_weight=1.0, **kw)
        loss = mod(pred, tgt)
        key = f'{case}.{fam}'
        l64, l32 = g[key + '.loss64'], g[key + '.loss32']
        if fam == 'ident':
            # identical boxes: the value is sqrt(cancellation noise); the reference's fp32 gives 0 .. ~1e-3
            bound = np.maximum(loss_bound(l64, l32), 2e-3)
            check_close(key + '.loss', loss.detach().cpu().numpy(), l64, bound)
            continue
        loss.sum().backward()
        check_close(key + '.loss', loss.detach().cpu().numpy(), l64, loss_bound(l64, l32))
        check_close(key + '.gp', pred.grad.cpu().numpy(), g[key + '.gp64'], grad_bound(g[key + '.gp64'], g[key + '.gp32']))
        check_close(key + '.gt', tgt.grad.cpu().numpy(), g[key + '.gt64'], grad_bound(g[key + '.gt64'], g[key + '.gt32']))


@pytest.mark.parametrize('case', sorted(index()['module']))
def test_module_glue_against_reference_golden(amd, case):
    """GDLoss.forward host logic + reduction kernels against the reference module's outputs."""
    spec = index()['module'][case]
    m = module()
    g = pairs()
    ctor = dict(spec['ctor'])
    call = spec['call']
    n = g['in.kitti.pred'].shape[0]
    pred = _dev(g['in.kitti.pred'])
    tgt = _dev(g['in.kitti.target'])
    if call.get('reshape'):
        pred = pred.reshape(call['reshape'])
        tgt = tgt.reshape(call['reshape'])
    pred.requires_grad_(True)
    kwargs = {}
    if 'weight' in call:
        kwargs['weight'] = {'w1': _dev(m['w1']), 'w7': _dev(m['w7']), 'w0': torch.zeros(n).cuda(),
                            'w07': torch.zeros(n, 7).cuda()}[call['weight']]
    for k in ('avg_factor', 'reduction_override'):
        if k in call:
            kwargs[k] = call[k]
    kwargs.update(call.get('call_kwargs', {}))
    mod = amd.GDLoss(spec['loss_type'], **ctor)
    if spec.get('raises'):
        with pytest.raises(RuntimeError):
            mod(pred, tgt, **kwargs)
        return
    res = mod(pred, tgt, **kwargs)
    out64, out32 = m[case + '.out64'], m[case + '.out32']
    assert tuple(res.shape) == tuple(out64.shape)
    if res.dim() == 0:
        res.backward()
    else:
        res.backward(_dev(m['up']).reshape(res.shape))
    check_close(case + '.out', res.detach().cpu().numpy(), out64, loss_bound(out64, out32))
    check_close(case + '.gp', pred.grad.cpu().numpy().reshape(-1, 7), m[case + '.gp64'],
                grad_bound(m[case + '.gp64'], m[case + '.gp32']))


def _synthetic(n, seed):
    """SURVEY.md §8d recipe."""
    g = torch.Generator().manual_seed(seed)
    lo = torch.tensor([0, -40, -3, 0.5, 0.5, 0.5, -np.pi])
    hi = torch.tensor([70, 40, 1, 2.5, 4.5, 2.0, np.pi])
    tgt = torch.rand(n, 7, generator=g) * (hi - lo) + lo
    sigma = torch.tensor([0.3, 0.3, 0.1, 0.1, 0.1, 0.1, 0.1])
    pred = tgt + torch.randn(n, 7, generator=g) * sigma
    return pred.float().contiguous(), tgt.float().contiguous()


@pytest.mark.parametrize('lt', ['gwd3d', 'kld3d', 'bd3d', 'jd3d', 'kld3d_symmax', 'kld3d_symmin', 'kfiou3d'])
@pytest.mark.parametrize('n', [1, 255, 256, 257, 511, 512, 513, 100_003])
def test_against_fp64_oracle_ragged_sizes(amd, lt, n):
    """Seeded synthetic pairs vs the fp64 oracle, incl. tile-boundary sizes (tile = 512 pairs, DMA pieces per 256).
    Shipped KITTI setting: fun=log1p, tau=1, loss_weight=5; reduction 'sum' keeps grads O(1)."""
    pred, tgt = _synthetic(n, seed=n)
    fun = 'expm1' if lt == 'kfiou3d' else 'log1p'
    prm = oracle.make_params(lt, fun=fun, tau=1.0)
    ref = oracle.gd_loss(pred.numpy(), tgt.numpy(), prm, scale=5.0)
    lb, gb = oracle32_bounds(pred.numpy(), tgt.numpy(), prm, ref, 5.0)
    p = pred.cuda().requires_grad_(True)
    out = amd.GDLoss(lt, fun=fun, tau=1.0, reduction='sum', loss_weight=5.0)(p, tgt.cuda())
    out.backward()
    assert abs(out.item() - ref['loss_sum']) <= LOSS_TOL * (1 + abs(ref['loss_sum']))
    check_close(f'{lt}.{n}.gp', p.grad.cpu().numpy(), ref['grad_pred'], gb)
    per = amd.GDLoss(lt, fun=fun, tau=1.0, reduction='none', loss_weight=5.0)(p.detach(), tgt.cuda())
    check_close(f'{lt}.{n}.loss', per.cpu().numpy(), ref['loss'], lb)


def test_unaligned_pointers_take_scalar_path_same_result(amd):
    """C ABI called on 4-byte-aligned (not 16-byte-aligned) device pointers: identical bits to the fast path."""
    from mmdet3d_gaussian_amd import _lib
    lib = amd.load_library()
    n = 1000
    pred, tgt = _synthetic(n, seed=3)
    prm = amd.make_params('bd3d', 'log1p', 1.0, 1.0, (0, 0, 0.5), {})
    res = []
    for off in (0, 1):
        bp = torch.zeros(n * 7 + 4, device='cuda'); bt = torch.zeros(n * 7 + 4, device='cuda')
        bg = torch.zeros(n * 7 + 4, device='cuda')
        bp[off:off + n * 7] = pred.cuda().reshape(-1); bt[off:off + n * 7] = tgt.cuda().reshape(-1)
        loss = torch.empty(n, device='cuda'); total = torch.empty((), device='cuda')
        ws = torch.empty(lib.gd3d_loss_workspace_bytes(n), dtype=torch.uint8, device='cuda')
        vp = lambda t, o=0: ctypes.c_void_p(t.data_ptr() + 4 * o)
        rc = lib.gd3d_loss_fused(ctypes.byref(prm), vp(bp, off), vp(bt, off), None, n, 1.0, vp(loss), vp(total),
                                 vp(bg, off), None, vp(ws), None)
        assert rc == 0
        torch.cuda.synchronize()
        res.append((loss.cpu(), total.cpu(), bg[off:off + n * 7].cpu()))
        assert bg[:off].abs().sum() == 0 and bg[off + n * 7:].abs().sum() == 0   # no out-of-range write
    for a, b in zip(res[0], res[1]):
        assert torch.equal(a, b)


def test_c_abi_argument_validation(amd):
    lib = amd.load_library()
    prm = amd.make_params('gwd3d', 'log1p', 1.0, 1.0, (0, 0, 0.5), {})
    x = torch.zeros(4, 7, device='cuda')
    vp = lambda t: ctypes.c_void_p(t.data_ptr())
    assert lib.gd3d_loss_fused(None, vp(x), vp(x), None, 4, 1.0, None, None, None, None, None, None) == 10001
    assert lib.gd3d_loss_fused(ctypes.byref(prm), None, vp(x), None, 4, 1.0, None, None, None, None, None, None) == 10001
    assert lib.gd3d_loss_fused(ctypes.byref(prm), vp(x), vp(x), None, -1, 1.0, None, None, None, None, None, None) == 10001
    prm.fun = 2  # expm1 is not legal for gwd3d (ref :267-270)
    assert lib.gd3d_loss_fused(ctypes.byref(prm), vp(x), vp(x), None, 4, 1.0, None, None, None, None, None, None) == 10001
    prm.fun = 1; prm.loss_type = 9
    assert lib.gd3d_loss_fused(ctypes.byref(prm), vp(x), vp(x), None, 4, 1.0, None, None, None, None, None, None) == 10001
    # empty input: legal, loss_sum = 0
    prm.loss_type = 0
    total = torch.ones((), device='cuda')
    ws = torch.empty(16, dtype=torch.uint8, device='cuda')
    assert lib.gd3d_loss_fused(ctypes.byref(prm), None, None, None, 0, 1.0, None, vp(total), None, None, vp(ws), None) == 0
    torch.cuda.synchronize()
    assert total.item() == 0.0


def test_empty_and_cpu_inputs(amd):
    mod = amd.GDLoss('kld3d', reduction='sum')
    out = mod(torch.zeros(0, 7, device='cuda', requires_grad=True), torch.zeros(0, 7, device='cuda'))
    assert out.item() == 0.0
    assert torch.isnan(amd.GDLoss('kld3d', reduction='mean')(torch.zeros(0, 7, device='cuda'),
                                                             torch.zeros(0, 7, device='cuda')))
    with pytest.raises(RuntimeError):
        mod(torch.zeros(2, 7), torch.zeros(2, 7))


def test_backward_scaling_and_retain_graph(amd):
    """grad_output != 1 (e.g. fp16 loss scaling) and a second backward on a retained graph."""
    pred, tgt = _synthetic(513, seed=11)
    p = pred.cuda().requires_grad_(True)
    mod = amd.GDLoss('gwd3d', loss_weight=5.0)
    out = mod(p, tgt.cuda())
    out.backward(retain_graph=True)
    g1 = p.grad.clone(); p.grad = None
    (out * 128.0).backward(retain_graph=True)
    g128 = p.grad.clone(); p.grad = None
    out.backward()
    g1b = p.grad.clone()
    assert torch.allclose(g128, g1 * 128.0, rtol=1e-6, atol=0)
    assert torch.equal(g1, g1b)


def test_determinism_bitwise(amd):
    pred, tgt = _synthetic(300_000, seed=5)
    p, t = pred.cuda(), tgt.cuda()
    mod = amd.GDLoss('bd3d', loss_weight=5.0)
    outs = []
    for _ in range(3):
        pp = p.clone().requires_grad_(True)
        o = mod(pp, t); o.backward()
        outs.append((o.detach().clone(), pp.grad.clone()))
    for o, g in outs[1:]:
        assert torch.equal(o, outs[0][0]) and torch.equal(g, outs[0][1])


@pytest.mark.parametrize('lt', ['gwd3d', 'kld3d', 'bd3d'])
def test_full_size_properties_10m(amd, lt):
    """BASELINE.json config 3 size (10 M pairs): size-independent properties.
      * mean loss equals the mean of per-pair losses (reduction kernel vs 'none' path), and a 1 M-pair sample of
        rows equals the fp64 oracle (loss and grads);
      * linearity of the reduction: loss over the whole = sum over two halves (avg_factor fixed);
      * symmetry: yaw + pi leaves every loss value unchanged (Gaussian of a box is pi-periodic in yaw)."""
    n = 10_000_000
    pred, tgt = _synthetic(n, seed=0)
    p = pred.cuda().requires_grad_(True)
    t = tgt.cuda()
    mod = amd.GDLoss(lt, fun='log1p', tau=1.0, reduction='mean', loss_weight=5.0)
    out = mod(p, t)
    out.backward()
    per = amd.GDLoss(lt, fun='log1p', tau=1.0, reduction='none', loss_weight=5.0)(p.detach(), t)
    mean64 = per.double().mean().item()
    assert abs(out.item() - mean64) <= 2e-6 * (1 + abs(mean64))
    # halves
    h = n // 2
    a = mod(p.detach()[:h], t[:h], avg_factor=n).item() + mod(p.detach()[h:], t[h:], avg_factor=n).item()
    assert abs(a - out.item()) <= 2e-6 * (1 + abs(a))
    # oracle on a strided 1 M sample
    idx = torch.arange(0, n, 10)
    prm = oracle.make_params(lt, fun='log1p', tau=1.0)
    ref = oracle.gd_loss(pred[idx].numpy(), tgt[idx].numpy(), prm, scale=5.0, nthreads=8)
    lb, gb = oracle32_bounds(pred[idx].numpy(), tgt[idx].numpy(), prm, ref, 5.0)
    check_close(lt + '.loss', per[idx.cuda()].cpu().numpy(), ref['loss'], lb)
    gp = p.grad[idx.cuda()].cpu().numpy() * n   # undo the 1/n of 'mean'
    check_close(lt + '.gp', gp, ref['grad_pred'], gb)
    # yaw + pi
    p2 = p.detach().clone(); p2[:, 6] += np.pi
    out2 = mod(p2, t)
    assert abs(out2.item() - out.item()) <= 1e-5 * (1 + abs(out.item()))


def _ref_forward(pred, target, weight, mod_kw, avg_factor, reduction, lt):
    """Reference GDLoss.forward control flow (gaussian_distance_loss.py:286-310) on top of the eager torch restatement
    of the loss functions (oracle/gd_torch.py, pinned by test_oracle_torch.py), fp64, autograd backward."""
    from oracle import gd_torch
    if (weight is not None) and (not torch.any(weight > 0)) and (reduction != 'none'):
        return (pred * weight).sum()
    return gd_torch.gd_loss(pred, target, lt, weight=weight, avg_factor=avg_factor, reduction=reduction, **mod_kw)


@pytest.mark.parametrize('wkind', ['zero', 'negative', 'nan', 'one_positive', 'mixed'])
@pytest.mark.parametrize('n', [300, 70_001])
def test_weight7_early_out_resolved_on_device(amd, wkind, n):
    """`if not torch.any(weight > 0): return (pred * weight).sum()` (ref :290-292) for (N,7) weights is decided inside the
    fused launch (no host sync): value, gradient wrt pred and the missing gradient wrt target equal the reference's
    control flow for all-zero, all-negative, NaN-only, exactly-one-positive (in the last tile) and ordinary weights,
    also under an upstream gradient != 1."""
    pred, tgt = _synthetic(n, seed=7)
    g = torch.Generator().manual_seed(n)
    if wkind == 'zero':
        w = torch.zeros(n, 7)
    elif wkind == 'negative':
        w = -torch.rand(n, 7, generator=g)
    elif wkind == 'nan':
        w = torch.full((n, 7), float('nan'))
    elif wkind == 'one_positive':
        w = -torch.rand(n, 7, generator=g)
        w[n - 2, 3] = 0.25
    else:
        w = torch.rand(n, 7, generator=g) - 0.3
    lt, kw = 'kld3d', dict(fun='log1p', tau=1.0, loss_weight=5.0)
    mod = amd.GDLoss(lt, **kw)
    p = pred.cuda().requires_grad_(True)
    t = tgt.cuda().requires_grad_(True)
    out = mod(p, t, w.cuda(), avg_factor=37.0)
    (out * 3.0).backward()
    p64 = pred.double().requires_grad_(True)
    t64 = tgt.double().requires_grad_(True)
    ref = _ref_forward(p64, t64, w.double(), kw, 37.0, 'mean', lt)
    (ref * 3.0).backward()
    if wkind == 'nan':
        assert torch.isnan(out).item() and torch.isnan(ref).item()
        assert torch.isnan(p.grad).all()
        return
    assert abs(out.item() - ref.item()) <= 2e-5 * (1 + abs(ref.item())), (out.item(), ref.item())
    sc = p64.grad.abs().max().item()
    assert (p.grad.cpu().double() - p64.grad).abs().max().item() <= 5e-5 * (1 + sc)
    gt_ref = t64.grad if t64.grad is not None else torch.zeros_like(t64)
    gt_got = t.grad if t.grad is not None else torch.zeros_like(t)
    assert (gt_got.cpu().double() - gt_ref).abs().max().item() <= 5e-5 * (1 + gt_ref.abs().max().item())
    if wkind in ('zero', 'negative'):          # early-out branch: the gradient is exactly 3 * weight
        assert torch.equal(p.grad.cpu(), 3.0 * w)


def test_weight7_early_out_with_tensor_avg_factor_and_decode_prologue(amd):
    """Early-out value is NOT divided by a tensor avg_factor, and with a fused bbox-coder decode `pred` of the early-out
    is the DECODED row (gradient chained back to the encoded prediction)."""
    n = 515
    pred, tgt = _synthetic(n, seed=9)
    w = -torch.rand(n, 7)
    mod = amd.GDLoss('gwd3d', loss_weight=5.0)
    p = pred.cuda().requires_grad_(True)
    out = mod(p, tgt.cuda(), w.cuda(), avg_factor=torch.tensor(11.0, device='cuda'))
    out.backward()
    want = (pred.double() * w.double()).sum().item()
    assert abs(out.item() - want) <= 1e-5 * (1 + abs(want))
    assert torch.equal(p.grad.cpu(), w)
    # ordinary weights with the same tensor avg_factor: divided as usual
    w2 = torch.rand(n, 7)
    a = mod(pred.cuda(), tgt.cuda(), w2.cuda(), avg_factor=torch.tensor(11.0, device='cuda')).item()
    b = mod(pred.cuda(), tgt.cuda(), w2.cuda(), avg_factor=11.0).item()
    assert abs(a - b) <= 1e-6 * (1 + abs(b))
    # anchor-delta prologue
    rng = np.random.default_rng(0)
    anchors = np.stack([rng.uniform(0, 70, n), rng.uniform(-40, 40, n), rng.uniform(-2, 0, n), rng.uniform(.6, 2, n),
                        rng.uniform(.8, 4, n), rng.uniform(1.4, 1.8, n), rng.choice([0, np.pi / 2], n)], -1).astype(np.float32)
    enc_t = rng.normal(0, 0.3, (n, 7)).astype(np.float32)
    enc_p = (enc_t + rng.normal(0, 0.1, (n, 7))).astype(np.float32)
    pe = torch.from_numpy(enc_p).cuda().requires_grad_(True)
    out = amd.anchor_decoded_gd_loss(mod, torch.from_numpy(anchors).cuda(), pe, torch.from_numpy(enc_t).cuda(), w.cuda(),
                                     avg_factor=5.0)
    out.backward()
    from oracle import head_torch
    pe64 = torch.from_numpy(enc_p).double().requires_grad_(True)
    ref = (head_torch.delta_decode(torch.from_numpy(anchors).double(), pe64) * w.double()).sum()
    ref.backward()
    assert abs(out.item() - ref.item()) <= 1e-5 * (1 + abs(ref.item()))
    assert (pe.grad.cpu().double() - pe64.grad).abs().max().item() <= 1e-5 * (1 + pe64.grad.abs().max().item())


def test_weighted_call_path_has_no_host_sync(amd):
    """The (N,7)-weight path must not wait for the device: the forward + backward calls return while a long kernel
    queued in front of them is still running."""
    n = 4096
    pred, tgt = _synthetic(n, seed=1)
    p = pred.cuda().requires_grad_(True)
    t, w = tgt.cuda(), torch.rand(n, 7).cuda()
    mod = amd.GDLoss('bd3d', loss_weight=5.0)
    mod(p, t, w, avg_factor=float(n)).backward()       # warm (library load, allocator)
    torch.cuda.synchronize()
    big = torch.empty(1 << 28, device='cuda')           # 1 GiB of fills: tens of milliseconds of queued GPU work
    done = torch.cuda.Event()
    for _ in range(40):
        big.fill_(1.0)
    out = mod(p, t, w, avg_factor=float(n))
    out.backward()
    done.record()
    assert not done.query(), 'the weighted GDLoss call waited for the device (host sync on the call path)'
    torch.cuda.synchronize()
    assert torch.isfinite(out).item()
