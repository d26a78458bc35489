"""The C++ autograd node of GDLoss's reduced forms (csrc/torch_node.cpp -> _gd3d_node.so) on CPU tensors: it is host plumbing
above the C ABI, so everything but the launch itself can be checked without a GPU — loading and binding, argument checks,
what backward hands out, the retain_graph replay, the double-backward guard, released graphs, in-place edits of saved inputs.
(The GPU tests run every golden case through the same node; tests/test_gpu_gd_loss.py.)"""
import ctypes
import os

import pytest
import torch

import mmdet3d_gaussian_amd as amd
from mmdet3d_gaussian_amd import _lib, gd_loss

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _pair(n=64, seed=0):
    g = torch.Generator().manual_seed(seed)
    t = torch.rand(n, 7, generator=g) + 0.5
    return (t + 0.1 * torch.randn(n, 7, generator=g)), t


def test_node_is_built_in_tree_and_binds_the_loaded_library():
    node = _lib.load_node()
    assert os.path.dirname(node.__file__) == os.path.join(ROOT, 'mmdet3d-gaussian_amd')
    assert node.bind(amd.lib_path()) == _lib.ABI_VERSION == 6          # binding again is harmless
    with pytest.raises(RuntimeError, match='cannot open'):
        node.bind(os.path.join(ROOT, 'no_such_library.so'))
    src = open(os.path.join(ROOT, 'mmdet3d-gaussian_amd', 'csrc', 'torch_node.cpp')).read()
    assert '__global__' not in src and 'hipLaunch' not in src              # plumbing: no kernel, no launch of its own
    assert not _lib._build.node_is_stale()


def test_node_checks_its_operands():
    node = _lib.load_node()
    p, t = _pair()
    params = amd.GDLoss('gwd3d')._params({})
    addr, ws = ctypes.addressof(params), gd_loss._ws_floats(64)
    call = lambda pp, tt, w=None, select=False, pro=0: node.reduced(pp, tt, w, addr, pro, None, 1.0, select, 0, 0, 0, ws, False)
    with pytest.raises(RuntimeError, match='contiguous fp32'):
        call(p.t().contiguous().t(), t)
    with pytest.raises(RuntimeError, match='contiguous fp32'):
        call(p.double(), t)
    with pytest.raises(RuntimeError, match='contiguous fp32'):
        call(p, t[:32])
    with pytest.raises(RuntimeError, match='weight must be'):
        call(p, t, torch.ones(64, 3))
    with pytest.raises(RuntimeError, match='select needs'):
        call(p, t, torch.ones(64, 7), select=True)                      # the on-device selection is GPU-only
    with pytest.raises(RuntimeError, match='GPU-only'):
        call(p, t, pro=addr)                                            # so are the bbox-coder prologues
    out, flag = call(p, t)
    assert out.dim() == 0 and flag is None and not out.requires_grad   # nothing requires grad: no node is attached


def test_backward_hands_over_the_forward_launch_s_buffers():
    p, t = _pair(200, seed=1)
    mod = amd.GDLoss('kld3d', loss_weight=5.0)
    pa = p.clone().requires_grad_(True)
    out = mod(pa, t)
    assert out.grad_fn.name() == 'GDLossReducedBackward'
    out.backward()
    # target gradients, (N,) weights and the unit gradient go the same way
    tb = t.clone().requires_grad_(True)
    pb = p.clone().requires_grad_(True)
    w = torch.rand(200)
    torch.autograd.backward([mod(pb, tb, w)], grad_tensors=[gd_loss.unit_grad('cpu')])
    pc, tc = p.clone().requires_grad_(True), t.clone().requires_grad_(True)
    mod(pc, tc, w).backward()
    assert torch.equal(pb.grad, pc.grad) and torch.equal(tb.grad, tc.grad) and tb.grad.abs().sum() > 0
    # an upstream gradient that is not the constant is applied by the `_cpu` scale twin
    pd = p.clone().requires_grad_(True)
    (mod(pd, t) * 3.0).backward()
    assert torch.allclose(pd.grad, 3.0 * pa.grad, rtol=1e-6, atol=0)
    # no_grad / detached inputs: a plain value
    with torch.no_grad():
        assert not mod(pa, t).requires_grad
    assert mod(p, t).grad_fn is None


def test_retain_graph_replays_and_a_released_graph_raises():
    p, t = _pair(100, seed=2)
    mod = amd.GDLoss('bd3d')
    pa = p.clone().requires_grad_(True)
    out = mod(pa, t)
    out.backward(retain_graph=True)
    g1 = pa.grad.clone()
    pa.grad = None
    (out * 2.0).backward(retain_graph=True)        # second backward: the buffers were handed over -> recomputed, then scaled
    assert torch.allclose(pa.grad, 2.0 * g1, rtol=1e-6, atol=0)
    pa.grad = None
    out.backward()
    assert torch.equal(pa.grad, g1)
    with pytest.raises(RuntimeError, match='second time'):
        out.backward()


def test_differentiating_the_gradient_raises():
    p, t = _pair(50, seed=3)
    pa = p.clone().requires_grad_(True)
    (g,) = torch.autograd.grad(amd.GDLoss('gwd3d')(pa, t), pa, create_graph=True)
    assert g.requires_grad                      # an Error node sits behind it, as with torch's once_differentiable
    with pytest.raises(RuntimeError, match='differentiate twice'):
        g.sum().backward()
    (g2,) = torch.autograd.grad(amd.GDLoss('gwd3d')(pa, t), pa)
    assert torch.equal(g.detach(), g2) and not g2.requires_grad


def test_in_place_edit_of_a_saved_input_is_detected():
    p, t = _pair(50, seed=4)
    pa = p.clone().requires_grad_(True)
    x = pa * 1.0
    out = amd.GDLoss('gwd3d')(x, t)
    with torch.no_grad():
        x.add_(1.0)
    with pytest.raises(RuntimeError, match='modified by an inplace operation'):
        out.backward()
