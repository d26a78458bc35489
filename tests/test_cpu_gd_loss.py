"""The `_cpu` twins of the loss entry points (gd3d_loss_fused_cpu & co., csrc/gd3d_cpu.cpp) through GDLoss on CPU tensors —
the path the reference's device-agnostic module takes on a machine without a GPU (gaussian_distance_loss.py:280-310,
BASELINE configs[0]).  Same data and the same tolerance policy (tests/gd_golden.py) as the GPU suite
tests/test_gpu_gd_loss.py: golden vectors written by the REAL reference (fp32 and fp64 runs), the fp64 oracle on seeded
inputs, NaN / inf rows, module glue, autograd behaviour, C-ABI argument rules.  No GPU involved."""
import ctypes
import os

import numpy as np
import pytest
import torch

import oracle
from gd_golden import (LOSS_TOL, NONFINITE_CASES, check_close, check_golden, check_nonfinite, check_nonfinite_grad_rows, families,
                       grad_bound, index, loss_bound, module, nonfinite, oracle32_bounds, pair_case_names, pairs)

import mmdet3d_gaussian_amd as amd

HERE = os.path.dirname(os.path.abspath(__file__))
ALL_LOSSES = ('gwd3d', 'kld3d', 'bd3d', 'jd3d', 'kld3d_symmax', 'kld3d_symmin', 'kfiou3d')


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def _synthetic(n, seed):
    """SURVEY.md §8d recipe."""
    g = torch.Generator().manual_seed(seed)
    lo = torch.tensor([0, -40, -3, 0.5, 0.5, 0.5, -np.pi])
    hi = torch.tensor([70, 40, 1, 2.5, 4.5, 2.0, np.pi])
    tgt = torch.rand(n, 7, generator=g) * (hi - lo) + lo
    pred = tgt + torch.randn(n, 7, generator=g) * torch.tensor([0.3, 0.3, 0.1, 0.1, 0.1, 0.1, 0.1])
    return pred.float().contiguous(), tgt.float().contiguous()


@pytest.mark.parametrize('case', pair_case_names())
def test_pairs_against_reference_golden(case):
    c = index()['pairs']['cases'][case]
    g = pairs()
    for fam in families():
        pred = _t(g[f'in.{fam}.pred']).requires_grad_(True)
        tgt = _t(g[f'in.{fam}.target']).requires_grad_(True)
        kw = {k: (tuple(v) if isinstance(v, list) else v) for k, v in c['kwargs'].items()}
        mod = amd.GDLoss(c['loss_type'], reduction='none', loss_weight=1.0, **kw)
        with np.errstate(all='ignore'):
            loss = mod(pred, tgt)
        key = f'{case}.{fam}'
        l64, l32 = g[key + '.loss64'], g[key + '.loss32']
        if fam == 'ident':
            check_close(key + '.loss', loss.detach().numpy(), l64, np.maximum(loss_bound(l64, l32), 2e-3), noisy=True)
            continue
        loss.sum().backward()
        check_golden(key + '.loss', loss.detach().numpy(), l64, l32, False)
        check_golden(key + '.gp', pred.grad.numpy(), g[key + '.gp64'], g[key + '.gp32'], True)
        check_golden(key + '.gt', tgt.grad.numpy(), g[key + '.gt64'], g[key + '.gt32'], True)


@pytest.mark.parametrize('case', sorted(index()['module']))
def test_module_glue_against_reference_golden(case):
    """GDLoss.forward host logic (reduction override, weights (N,) / (N,7), avg_factor, early-out, kwargs) on CPU tensors."""
    spec = index()['module'][case]
    m = module()
    g = pairs()
    call = spec['call']
    n = g['in.kitti.pred'].shape[0]
    pred, tgt = _t(g['in.kitti.pred']), _t(g['in.kitti.target'])
    if call.get('reshape'):
        pred, tgt = pred.reshape(call['reshape']), tgt.reshape(call['reshape'])
    pred.requires_grad_(True)
    kwargs = {}
    if 'weight' in call:
        kwargs['weight'] = {'w1': _t(m['w1']), 'w7': _t(m['w7']), 'w0': torch.zeros(n), 'w07': torch.zeros(n, 7)}[call['weight']]
    for k in ('avg_factor', 'reduction_override'):
        if k in call:
            kwargs[k] = call[k]
    kwargs.update(call.get('call_kwargs', {}))
    mod = amd.GDLoss(spec['loss_type'], **dict(spec['ctor']))
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
        res.backward(_t(m['up']).reshape(res.shape))
    check_close(case + '.out', res.detach().numpy(), out64, loss_bound(out64, out32))
    check_close(case + '.gp', pred.grad.numpy().reshape(-1, 7), m[case + '.gp64'], grad_bound(m[case + '.gp64'], m[case + '.gp32']))


@pytest.mark.parametrize('lt', ALL_LOSSES)
@pytest.mark.parametrize('n', [1, 255, 256, 257, 513, 100_003])
def test_against_fp64_oracle_ragged_sizes(lt, n):
    """Tile-boundary sizes (one partial per 256 pairs) and a size that takes the thread team (>= 64 tiles)."""
    pred, tgt = _synthetic(n, seed=n)
    fun = 'expm1' if lt == 'kfiou3d' else 'log1p'
    prm = oracle.make_params(lt, fun=fun, tau=1.0)
    ref = oracle.gd_loss(pred.numpy(), tgt.numpy(), prm, scale=5.0)
    lb, gb = oracle32_bounds(pred.numpy(), tgt.numpy(), prm, ref, 5.0)
    p = pred.clone().requires_grad_(True)
    out = amd.GDLoss(lt, fun=fun, tau=1.0, reduction='sum', loss_weight=5.0)(p, tgt)
    out.backward()
    assert abs(out.item() - ref['loss_sum']) <= LOSS_TOL * (1 + abs(ref['loss_sum']))
    check_close(f'{lt}.{n}.gp', p.grad.numpy(), ref['grad_pred'], gb)
    per = amd.GDLoss(lt, fun=fun, tau=1.0, reduction='none', loss_weight=5.0)(p.detach(), tgt)
    check_close(f'{lt}.{n}.loss', per.numpy(), ref['loss'], lb)


@pytest.mark.parametrize('kind', __import__('gd_stress').FAMILIES)
def test_stress_families_against_fp64_oracle(kind):
    from gd_golden import stress_bounds, stress_report
    from gd_stress import stress_pairs
    p_np, t_np = stress_pairs(2048, kind, seed=1)
    for lt in ALL_LOSSES:
        for fun, tau in ((('none', 0.0), ('expm1', 0.0)) if lt == 'kfiou3d' else (('log1p', 1.0), ('none', 0.0))):
            prm = oracle.make_params(lt, fun=fun, tau=tau)
            with np.errstate(all='ignore'):
                ref = oracle.gd_loss(p_np, t_np, prm, scale=1.0)
                lb, gb = stress_bounds(kind, p_np, t_np, prm, ref, 1.0)
            p = _t(p_np).requires_grad_(True)
            out = amd.GDLoss(lt, fun=fun, tau=tau, reduction='none', loss_weight=1.0)(p, _t(t_np))
            out.sum().backward()
            stress_report('cpu_twin', kind, lt, fun, out.detach().numpy(), p.grad.numpy(), ref)
            check_close(f'{kind}.{lt}.{fun}.loss', out.detach().numpy(), ref['loss'], lb)
            check_close(f'{kind}.{lt}.{fun}.gp', p.grad.numpy(), ref['grad_pred'], gb)


def test_nonfinite_and_degenerate_rows_vs_real_reference():
    """NaN exactly where the real reference has it (tests/golden/gd_nonfinite.npz), and a bad row never touches its neighbours."""
    gold = nonfinite()
    p_np, t_np = gold['pred'], gold['target']
    clean = np.isfinite(p_np).all(1) & np.isfinite(t_np).all(1)
    for lt, kw in NONFINITE_CASES:
        mod = amd.GDLoss(lt, reduction='none', loss_weight=1.0, **kw)
        p, t = _t(p_np).requires_grad_(True), _t(t_np).requires_grad_(True)
        out = mod(p, t)
        out.sum().backward()
        check_nonfinite(lt, out.detach().numpy(), gold[f'{lt}.loss32'], gold[f'{lt}.loss64'])
        check_nonfinite_grad_rows(lt + '.gp', p.grad.numpy(), gold[f'{lt}.gp_nanrow32'], gold[f'{lt}.gp_nanrow64'])
        check_nonfinite_grad_rows(lt + '.gt', t.grad.numpy(), gold[f'{lt}.gt_nanrow32'], gold[f'{lt}.gt_nanrow64'])
        pc, tc = _t(p_np[clean]).requires_grad_(True), _t(t_np[clean]).requires_grad_(True)
        oc = mod(pc, tc)
        oc.sum().backward()
        sel = _t(clean)
        bits = lambda x: x.contiguous().view(torch.int32)   # noqa: E731
        assert torch.equal(bits(out.detach()[sel]), bits(oc.detach())), lt
        assert torch.equal(bits(p.grad[sel]), bits(pc.grad)) and torch.equal(bits(t.grad[sel]), bits(tc.grad)), lt


def test_config0_1k_pairs_forward_and_early_out_vs_real_reference():
    """BASELINE configs[0] as BASELINE.json words it — 1 k synthetic KITTI pairs, forward, CPU — against vectors the REAL
    reference module produced (tests/golden/gd_config0.npz): all 7 losses x tau {0, 1}; plus the non-positive-weight early-out
    (ref :290-292), which on CPU tensors is decided on the host exactly as the reference decides it."""
    g = np.load(os.path.join(HERE, 'golden', 'gd_config0.npz'))
    p, t = _t(g['pred']), _t(g['target'])
    for lt in ALL_LOSSES:
        fun = 'expm1' if lt == 'kfiou3d' else 'log1p'
        for tau in (0, 1):
            l64, l32 = g[f'{lt}.tau{tau}.loss64'], g[f'{lt}.tau{tau}.loss32']
            got = amd.GDLoss(lt, fun=fun, tau=float(tau), reduction='none')(p, t).numpy()
            check_close(f'config0.{lt}.tau{tau}.loss', got, l64, loss_bound(l64, l32))
            mean = amd.GDLoss(lt, fun=fun, tau=float(tau), reduction='mean')(p, t).item()
            assert abs(mean - l64.mean()) <= 1e-5 * (1 + abs(l64.mean()))
    mod = amd.GDLoss('kld3d', fun='log1p', tau=1.0, loss_weight=5.0)
    for kind in ('neg', 'zero', 'mixed'):
        pp = _t(g['pred']).requires_grad_(True)
        out = mod(pp, t, _t(g[f'early.{kind}.w']), avg_factor=37.0)
        out.backward()
        o64, o32 = float(g[f'early.{kind}.out64']), float(g[f'early.{kind}.out32'])
        assert abs(out.item() - o64) <= (1e-5 + 3 * abs(o32 - o64) / (1 + abs(o64))) * (1 + abs(o64)), (kind, out.item(), o64)
        check_close(f'config0.early.{kind}.gp', pp.grad.numpy(), g[f'early.{kind}.gp64'],
                    grad_bound(g[f'early.{kind}.gp64'], g[f'early.{kind}.gp32']))


def test_result_does_not_depend_on_the_thread_count():
    """Tile partials are summed in a fixed order: loss, gradients and the scalar are bit-identical for 1, 3 and 8 threads."""
    pred, tgt = _synthetic(300_001, seed=5)
    keep = torch.get_num_threads()
    outs = []
    try:
        for nt in (1, 3, 8):
            torch.set_num_threads(nt)
            p = pred.clone().requires_grad_(True)
            o = amd.GDLoss('bd3d', loss_weight=5.0)(p, tgt)
            o.backward()
            outs.append((o.detach().clone(), p.grad.clone()))
    finally:
        torch.set_num_threads(keep)
    for o, g in outs[1:]:
        assert torch.equal(o, outs[0][0]) and torch.equal(g, outs[0][1])


def test_backward_scaling_retain_graph_unit_grad_and_target_gradient():
    pred, tgt = _synthetic(513, seed=11)
    p = pred.clone().requires_grad_(True)
    mod = amd.GDLoss('gwd3d', loss_weight=5.0)
    out = mod(p, tgt)
    out.backward(retain_graph=True)
    g1 = p.grad.clone(); p.grad = None
    (out * 128.0).backward(retain_graph=True)
    g128 = p.grad.clone(); p.grad = None
    out.backward()
    assert torch.allclose(g128, g1 * 128.0, rtol=1e-6, atol=0) and torch.equal(g1, p.grad)
    # the library's unit gradient on the CPU: recognised by address, same gradient
    from mmdet3d_gaussian_amd import gd_loss as gdl
    q = pred.clone().requires_grad_(True)
    torch.autograd.backward([mod(q, tgt)], grad_tensors=[gdl.unit_grad('cpu')])
    assert torch.equal(q.grad, g1)
    # only the target wants a gradient, (N,7) weights, avg_factor, upstream factor 2
    t = tgt.clone().requires_grad_(True)
    w = torch.rand(513, 7)
    o = amd.GDLoss('kld3d', loss_weight=5.0)(pred, t, w, avg_factor=50.0)
    (o * 2.0).backward()
    ref = oracle.gd_loss(pred.numpy(), tgt.numpy(), oracle.make_params('kld3d', fun='log1p', tau=1.0),
                         row_weight=w.numpy().astype(np.float64).mean(-1), scale=5.0 / 50.0)
    assert abs(o.item() - ref['loss_sum']) <= 2e-5 * (1 + abs(ref['loss_sum']))
    assert np.abs(t.grad.numpy() / 2.0 - ref['grad_target']).max() <= 5e-5 * (1 + np.abs(ref['grad_target']).max())


def test_empty_inputs_dtypes_and_mixed_devices():
    mod = amd.GDLoss('kld3d', reduction='sum')
    out = mod(torch.zeros(0, 7, requires_grad=True), torch.zeros(0, 7))
    assert out.item() == 0.0
    assert torch.isnan(amd.GDLoss('kld3d', reduction='mean')(torch.zeros(0, 7), torch.zeros(0, 7)))
    pred, tgt = _synthetic(64, seed=2)
    o64 = amd.GDLoss('gwd3d')(pred.double(), tgt.double())      # computed in fp32 (the heads run under force_fp32), returned in the input dtype
    assert o64.dtype == torch.float64 and abs(o64.item() - amd.GDLoss('gwd3d')(pred, tgt).item()) < 1e-6
    nc = torch.zeros(64, 14)
    nc[:, ::2] = pred
    assert torch.equal(amd.GDLoss('gwd3d')(nc[:, ::2], tgt), amd.GDLoss('gwd3d')(pred, tgt))   # non-contiguous rows
    with pytest.raises(RuntimeError, match='GPU-only'):
        mod(pred, tgt, _prologue=object())


def test_c_abi_argument_validation_of_the_cpu_entries():
    from mmdet3d_gaussian_amd import _lib
    lib = amd.load_library()
    prm = amd.make_params('gwd3d', 'log1p', 1.0, 1.0, (0, 0, 0.5), {})
    n = 10
    a = np.random.default_rng(0).random((n, 7)).astype(np.float32) + 0.5
    b = (a + 0.1).astype(np.float32)
    out = np.zeros(1, np.float32); ws = np.zeros(lib.gd3d_loss_workspace_bytes(n) // 4, np.float32)
    vp = lambda x: x.ctypes.data   # noqa: E731
    call = lambda **k: lib.gd3d_loss_fused_cpu(k.get('prm', ctypes.byref(prm)), k.get('pred', vp(a)), vp(b), k.get('w', None),   # noqa: E731
                                               k.get('w7', None), k.get('n', n), 1.0, None, k.get('sum', vp(out)), None, None,
                                               k.get('ws', vp(ws)), 1)
    assert call() == 0 and out[0] > 0
    assert call(prm=None) == 10001 and call(pred=None) == 10001 and call(n=-1) == 10001
    assert call(w=vp(a), w7=vp(a)) == 10001                       # row_weight and weight7 are mutually exclusive
    assert call(ws=None) == 10001                                 # loss_sum without a workspace
    bad = amd.make_params('gwd3d', 'log1p', 1.0, 1.0, (0, 0, 0.5), {}); bad.fun = 2
    assert call(prm=ctypes.byref(bad)) == 10001                   # expm1 is a kfiou3d-only fun (ref :267-270)
    bad.fun, bad.loss_type = 1, 9
    assert call(prm=ctypes.byref(bad)) == 10001
    out[0] = 7.0
    assert call(n=0) == 0 and out[0] == 0.0
    assert lib.gd3d_loss_reduce_cpu(vp(ws), n, vp(out)) == 0 and lib.gd3d_loss_reduce_cpu(None, n, vp(out)) == 10001
    g = np.array([2.0], np.float32); rows = np.ones((n, 7), np.float32)
    assert lib.gd3d_scale_rows_cpu(vp(rows), vp(g), 0, n, 1) == 0 and np.all(rows == 2.0)
    per = np.arange(n, dtype=np.float32)
    assert lib.gd3d_scale_rows_cpu(vp(rows), vp(per), 1, n, 1) == 0 and np.all(rows[:, 0] == 2.0 * per)
    assert lib.gd3d_scale_rows_cpu(vp(rows), None, 0, n, 1) == 10001
    assert _lib.ABI_VERSION == 6


def test_module_survives_deepcopy_pickle_and_torch_save(tmp_path):
    """mmcv builds the loss from its config, but hooks deep-copy models (EMA) and users pickle them: the module — with its cached
    gd3d_params structure populated by a call — must come back working and give the same value; like the reference's module it has
    no parameters or buffers (empty state_dict)."""
    import copy
    import pickle
    m = amd.GDLoss('kld3d', fun='log1p', tau=1.0, loss_weight=5.0, sqrt=True)
    g = torch.Generator().manual_seed(3)
    t = torch.rand(16, 7, generator=g) + 0.5
    p = t + 0.1 * torch.randn(16, 7, generator=g)
    a = m(p, t)
    assert torch.equal(copy.deepcopy(m)(p, t), a)
    assert torch.equal(pickle.loads(pickle.dumps(m))(p, t), a)
    holder = torch.nn.Sequential(torch.nn.Linear(7, 7))
    holder.loss = m
    path = str(tmp_path / 'm.pt')
    torch.save(holder, path)
    assert torch.equal(torch.load(path, weights_only=False).loss(p, t), a)
    assert len(m.state_dict()) == 0


@pytest.mark.parametrize('dtype', [torch.float64, torch.float16, torch.bfloat16])
def test_non_fp32_inputs_are_evaluated_in_fp32_and_cast_back(dtype):
    """INTEGRATION.md §5 / the GDLoss docstring: the reference follows the dtype of its inputs
    (gaussian_distance_loss.py:8-21); this package evaluates in fp32 whatever the dtype and casts the result back.  Pinned: an
    fp64 (fp16, bf16) call returns EXACTLY the fp32 call's value and gradients on the fp32 image of its inputs, carrying the
    input's dtype — i.e. an fp64 label with fp32 accuracy, which is stated, not hidden."""
    g = torch.Generator().manual_seed(5)
    t32 = torch.rand(300, 7, generator=g) * torch.tensor([70, 80, 4, 2, 4, 1.5, 6.28]) + torch.tensor([0, -40, -3, .5, .5, .5, -3.14])
    p32 = t32 + 0.1 * torch.randn(300, 7, generator=g)
    w32 = torch.rand(300, generator=g)
    for lt in ('gwd3d', 'kld3d', 'bd3d'):
        mod = amd.GDLoss(lt, loss_weight=5.0)
        p = p32.to(dtype).requires_grad_(True)
        out = mod(p, t32.to(dtype), w32.to(dtype), avg_factor=77.0)
        out.backward()
        assert out.dtype == dtype and p.grad.dtype == dtype
        q = p32.to(dtype).float().requires_grad_(True)             # the fp32 image of the same inputs
        ref = mod(q, t32.to(dtype).float(), w32.to(dtype).float(), avg_factor=77.0)
        ref.backward()
        assert torch.equal(out, ref.to(dtype)) and torch.equal(p.grad, q.grad.to(dtype))
        per = mod(p32.to(dtype), t32.to(dtype), reduction_override='none')
        assert per.dtype == dtype and torch.equal(per, mod(p32.to(dtype).float(), t32.to(dtype).float(), reduction_override='none').to(dtype))
    if dtype == torch.float64:   # and what that means: the fp64-labelled value is fp32-accurate against the fp64 oracle, not fp64-accurate
        mod = amd.GDLoss('kld3d', reduction='sum')
        got = mod(p32.double(), t32.double()).item()
        want = oracle.gd_loss(p32.numpy(), t32.numpy(), oracle.make_params('kld3d', fun='log1p', tau=1.0), scale=1.0)['loss_sum']
        assert 1e-12 < abs(got - want) <= 1e-5 * (1 + abs(want))


def test_loss_value_backward_starts_from_the_unit_constant_and_changes_nothing_else():
    """gd_loss.LossValue (round 6) on CPU tensors: the reduced forms return it, sums / products with anything stay LossValues,
    `.backward()` without a gradient starts from the library's unit constant (seen by a hook on the summed loss), gradients are
    bit-identical to torch's own path (flag off, functional form, explicit ones), scaled losses and explicit gradients scale,
    other losses in the sum get their gradients, retain_graph replays, non-fp32 values take torch's path, 'none' stays plain."""
    from mmdet3d_gaussian_amd import gd_loss as gdl
    pred, tgt = _synthetic(900, seed=12)
    mods = [amd.GDLoss(lt, loss_weight=5.0) for lt in ('gwd3d', 'kld3d', 'bd3d')]
    unit_ptr = gdl.unit_grad('cpu').data_ptr()

    def step(how, flag=True, extra=False):
        gdl._UNIT_ROOT = flag
        try:
            ps = [pred.clone().requires_grad_(True) for _ in mods]
            q = torch.linspace(-1, 1, 7).requires_grad_(True)
            ls = [m(p, tgt) for m, p in zip(mods, ps)]
            tot = sum(ls) + ((q ** 2).sum() if extra else 0.0)
            seen = []
            tot.register_hook(lambda g: seen.append(g.data_ptr()))
            how(tot)
            return [p.grad.clone() for p in ps], seen[0], q.grad, type(tot), type(ls[0])
        finally:
            gdl._UNIT_ROOT = True
    base, root0, _, ty_tot0, ty_l0 = step(lambda x: x.backward(), flag=False)
    assert ty_tot0 is torch.Tensor and ty_l0 is torch.Tensor and root0 != unit_ptr
    got, root, _, ty_tot, ty_l = step(lambda x: x.backward())
    assert ty_tot is gdl.LossValue and ty_l is gdl.LossValue and root == unit_ptr
    assert all(torch.equal(a, b) for a, b in zip(got, base))
    got, root, *_ = step(lambda x: torch.autograd.backward(x))
    assert root != unit_ptr and all(torch.equal(a, b) for a, b in zip(got, base))
    got, *_ = step(lambda x: (x * 3.0).backward())
    assert all(torch.allclose(a, 3.0 * b, rtol=1e-6, atol=0) for a, b in zip(got, base))
    got, *_ = step(lambda x: x.backward(torch.tensor(0.5)))
    assert all(torch.equal(a, 0.5 * b) for a, b in zip(got, base))
    got, root, qg, *_ = step(lambda x: x.backward(), extra=True)
    assert root == unit_ptr and torch.equal(qg, 2 * torch.linspace(-1, 1, 7)) and all(torch.equal(a, b) for a, b in zip(got, base))
    got, *_ = step(lambda x: (x.backward(retain_graph=True), x.backward()))
    assert all(torch.equal(a, 2 * b) for a, b in zip(got, base))
    assert gdl.unit_grad('cpu').item() == 1.0
    p64 = pred.double().requires_grad_(True)
    out = mods[1](p64, tgt.double())
    out.backward()                                                     # fp64 value: torch's own ones_like path
    assert out.dtype == torch.float64 and torch.equal(p64.grad, base[1].double())
    assert type(mods[0](pred, tgt, reduction_override='none')) is torch.Tensor
    import copy, pickle
    v = mods[0](pred, tgt)
    assert float(copy.deepcopy(v.detach())) == float(v) == float(pickle.loads(pickle.dumps(v.detach())))
