"""The host glue above the C ABI — mmdet3d-gaussian_amd/_pynode.py (torch.autograd.Function + ctypes, first class) and its optional
C++ twin csrc/torch_node.cpp (-> _gd3d_node.so) — on CPU tensors: both are host plumbing, so everything but the launch itself can
be checked without a GPU: loading and binding, argument checks, what backward hands out, the retain_graph replay, the
double-backward guard, released graphs, in-place edits of saved inputs.  Every behavioural test runs in BOTH modes, and the two
must agree bit for bit (they make the same C ABI calls).  The last tests hide the compiler and the binary: GDLoss must still work.
(The GPU suites run the golden cases, the NMS, the anchor-head slice and the scatter ops through both glues as well.)"""
import ctypes
import os
import subprocess
import sys

import pytest
import torch

import mmdet3d_gaussian_amd as amd
from mmdet3d_gaussian_amd import _lib, gd_loss

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LOSSES = ('gwd3d', 'kld3d', 'bd3d', 'jd3d', 'kld3d_symmax', 'kld3d_symmin', 'kfiou3d')

# tools/sanitize.sh runs this file against INSTRUMENTED builds selected by GD3D_LIB / GD3D_NODE_LIB (which bypass the in-tree
# build, hash and fallback logic by design): the tests of that logic itself are skipped there
in_tree_binaries = pytest.mark.skipif(bool(os.environ.get('GD3D_SANITIZER')), reason='runs against overridden (instrumented) binaries')



@pytest.fixture(params=_lib.HOST_GLUE_MODES)
def glue(request):
    _lib.set_host_glue(request.param)
    assert _lib.host_glue() == request.param
    yield request.param
    _lib.set_host_glue(None)


def _pair(n=64, seed=0):
    g = torch.Generator().manual_seed(seed)
    t = torch.rand(n, 7, generator=g) + 0.5
    return (t + 0.1 * torch.randn(n, 7, generator=g)), t


@in_tree_binaries
def test_cpp_node_is_built_in_tree_and_binds_the_loaded_library():
    _lib.set_host_glue('cpp')
    try:
        node = _lib.load_node()
        assert os.path.dirname(node.__file__) == os.path.join(ROOT, 'mmdet3d-gaussian_amd') and node.__file__.endswith('.so')
        assert node.bind(amd.lib_path()) == _lib.ABI_VERSION == 6          # binding again is harmless
        with pytest.raises(RuntimeError, match='cannot open'):
            node.bind(os.path.join(ROOT, 'no_such_library.so'))
        src = open(os.path.join(ROOT, 'mmdet3d-gaussian_amd', 'csrc', 'torch_node.cpp')).read()
        assert '__global__' not in src and 'hipLaunch' not in src              # plumbing: no kernel, no launch of its own
        assert not _lib._build.node_is_stale()
    finally:
        _lib.set_host_glue(None)


def test_python_glue_needs_no_torch_extension():
    _lib.set_host_glue('python')
    try:
        node = _lib.load_node()
        assert node.__file__.endswith('_pynode.py') and node.IMPLEMENTATION == 'python'
        assert node.bind(amd.lib_path()) == _lib.ABI_VERSION
        with pytest.raises(RuntimeError, match='cannot open'):
            node.bind(os.path.join(ROOT, 'no_such_library.so'))
        src = open(node.__file__).read()
        assert 'cpp_extension' not in src and 'pybind' not in src.split('"""')[2]   # ctypes + autograd.Function only
    finally:
        _lib.set_host_glue(None)


def test_glue_selection_by_environment_and_setter(monkeypatch):
    for mode in _lib.HOST_GLUE_MODES:
        monkeypatch.setenv('GD3D_HOST', mode)
        _lib.set_host_glue(None)
        assert _lib.host_glue() == mode
    monkeypatch.setenv('GD3D_HOST', 'fortran')
    _lib.set_host_glue(None)
    with pytest.raises(RuntimeError, match='GD3D_HOST'):
        _lib.host_glue()
    monkeypatch.delenv('GD3D_HOST')
    _lib.set_host_glue(None)
    assert _lib.host_glue() == 'cpp'            # auto: this tree builds the node, so the accelerator is preferred
    _lib.set_host_glue('python')                # the setter wins over the environment
    monkeypatch.setenv('GD3D_HOST', 'cpp')
    assert _lib.host_glue() == 'python'
    with pytest.raises(RuntimeError):
        _lib.set_host_glue('fortran')
    _lib.set_host_glue(None)


def test_node_checks_its_operands(glue):
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


def test_backward_hands_over_the_forward_launch_s_buffers(glue):
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


def test_retain_graph_replays_and_a_released_graph_raises(glue):
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


def test_differentiating_the_gradient_raises(glue):
    p, t = _pair(50, seed=3)
    pa = p.clone().requires_grad_(True)
    (g,) = torch.autograd.grad(amd.GDLoss('gwd3d')(pa, t), pa, create_graph=True)
    assert g.requires_grad                      # an Error node sits behind it, as with torch's once_differentiable
    with pytest.raises(RuntimeError, match='differentiate twice'):
        g.sum().backward()
    (g2,) = torch.autograd.grad(amd.GDLoss('gwd3d')(pa, t), pa)
    assert torch.equal(g.detach(), g2) and not g2.requires_grad


def test_in_place_edit_of_a_saved_input_is_detected(glue):
    p, t = _pair(50, seed=4)
    pa = p.clone().requires_grad_(True)
    x = pa * 1.0
    out = amd.GDLoss('gwd3d')(x, t)
    with torch.no_grad():
        x.add_(1.0)
    with pytest.raises(RuntimeError, match='modified by an inplace operation'):
        out.backward()


def _module_cases():
    """(name, ctor kwargs, call kwargs builder) over every reduced-form route GDLoss.forward has on CPU tensors."""
    n = 333
    g = torch.Generator().manual_seed(11)
    w1 = torch.rand(n, generator=g)
    w7 = torch.rand(n, 7, generator=g)
    w7[::5] = 0
    return n, [('plain', {}, {}),
               ('sum', dict(reduction='sum'), {}),
               ('w1', {}, dict(weight=w1)),
               ('w7_avg', {}, dict(weight=w7, avg_factor=77.0)),
               ('w7_avg_tensor', {}, dict(weight=w7, avg_factor=torch.tensor(77.0))),
               ('override_sum', {}, dict(weight=w1, reduction_override='sum')),
               ('zero_weight', {}, dict(weight=torch.zeros(n, 7)))]


def _run_all(mode):
    _lib.set_host_glue(mode)
    n, cases = _module_cases()
    p, t = _pair(n, seed=9)
    res = {}
    for lt in LOSSES:
        fun = 'expm1' if lt == 'kfiou3d' else 'log1p'
        for name, ck, kw in cases:
            mod = amd.GDLoss(lt, fun=fun, loss_weight=2.5, **ck)
            pa, ta = p.clone().requires_grad_(True), t.clone().requires_grad_(True)
            out = mod(pa, ta, **kw)
            (out * 1.5).backward(retain_graph=True)
            grads = lambda: tuple(torch.zeros(()) if x.grad is None else x.grad.clone() for x in (pa, ta))   # early-out: no target grad
            g1 = grads()
            pa.grad = ta.grad = None
            out.backward()                                   # the replay path
            res[(lt, name)] = (out.detach().clone(), g1, grads())
    _lib.set_host_glue(None)
    return res


def test_both_glues_agree_bit_for_bit_on_every_reduced_route():
    a, b = _run_all('python'), _run_all('cpp')
    assert a.keys() == b.keys() and len(a) == 7 * 7
    for k in a:
        assert torch.equal(a[k][0], b[k][0]), k
        for i in (1, 2):
            for x, y in zip(a[k][i], b[k][i]):
                assert torch.equal(x, y), (k, i)


_HIDDEN = r'''
import os, sys, logging
sys.path.insert(0, {root!r})
logging.basicConfig(stream=sys.stderr, level=logging.WARNING)
import torch
from mmdet3d_gaussian_amd import _lib
b = _lib._build
# a box without the host compiler and without a usable node binary (e.g. a torch this binary was not built for)
b.host_cxx_path = lambda: None
b.NODE_PATH = os.path.join({tmp!r}, '_gd3d_node.so'); b.NODE_HASH_PATH = b.NODE_PATH + '.srchash'
{extra}
import mmdet3d_gaussian_amd as amd
assert _lib.host_glue() == 'python'
g = torch.Generator().manual_seed(0)
t = torch.rand(128, 7, generator=g) + 0.5
p = (t + 0.1 * torch.randn(128, 7, generator=g)).requires_grad_(True)
out = amd.GDLoss('kld3d', loss_weight=5.0)(p, t, torch.rand(128, 7, generator=g), avg_factor=31.0)
out.backward()
print('RESULT', out.item().hex(), p.grad.double().sum().item().hex())
'''


def _hidden(tmp_path, extra=''):
    env = dict(os.environ)
    env.pop('GD3D_HOST', None)
    r = subprocess.run([sys.executable, '-c', _HIDDEN.format(root=ROOT, tmp=str(tmp_path), extra=extra)], capture_output=True,
                       text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    return [ln for ln in r.stdout.splitlines() if ln.startswith('RESULT')][0], r.stderr


@in_tree_binaries
def test_gdloss_works_with_the_compiler_and_the_node_binary_hidden(tmp_path):
    """auto mode on a box where the C++ node cannot be had: one loud line, then the Python glue — same numbers as here."""
    line, err = _hidden(tmp_path)
    assert err.count('host glue: the optional C++ autograd node is unavailable') == 1 and 'is missing' in err
    _lib.set_host_glue('cpp')
    try:
        g = torch.Generator().manual_seed(0)
        t = torch.rand(128, 7, generator=g) + 0.5
        p = (t + 0.1 * torch.randn(128, 7, generator=g)).requires_grad_(True)
        out = amd.GDLoss('kld3d', loss_weight=5.0)(p, t, torch.rand(128, 7, generator=g), avg_factor=31.0)
        out.backward()
        assert line == f'RESULT {out.item().hex()} {p.grad.double().sum().item().hex()}'
    finally:
        _lib.set_host_glue(None)


@in_tree_binaries
def test_a_node_binary_that_does_not_match_is_never_loaded(tmp_path):
    """A stale _gd3d_node.so (other sources / other torch) with no compiler to rebuild it: refused, Python glue instead — not
    'run the binary as it is'.  GD3D_HOST=cpp turns the same situation into an error."""
    import shutil
    shutil.copy(_lib._build.NODE_PATH, tmp_path / '_gd3d_node.so')
    (tmp_path / '_gd3d_node.so.srchash').write_text('0' * 64)
    line, err = _hidden(tmp_path)
    assert 'does not match' in err and line.startswith('RESULT')
    env = dict(os.environ, GD3D_HOST='cpp')
    r = subprocess.run([sys.executable, '-c', _HIDDEN.format(root=ROOT, tmp=str(tmp_path), extra='')], capture_output=True,
                       text=True, env=env, timeout=300)
    assert r.returncode != 0 and 'does not match' in r.stderr


@in_tree_binaries
def test_a_library_that_does_not_match_is_refused_without_hipcc(tmp_path):
    code = f'''
import os, sys
sys.path.insert(0, {ROOT!r})
from mmdet3d_gaussian_amd import _lib
b = _lib._build
b.hipcc_path = lambda: os.path.join({str(tmp_path)!r}, 'no_hipcc')
b.source_hash = lambda: 'f' * 64
try:
    _lib.load()
except RuntimeError as e:
    print('REFUSED', e)
'''
    r = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, timeout=300)
    assert 'REFUSED' in r.stdout and 'does not match its sources' in r.stdout, r.stdout + r.stderr


@in_tree_binaries
def test_a_failed_node_build_is_stamped_and_not_retried(tmp_path):
    """A COMPILE error in csrc/torch_node.cpp (as opposed to: no compiler): the first process tries the build once, stamps the
    failure with the source hash it belongs to and falls back to the Python glue with an ERROR-level log line; the next
    process finds the stamp and does not build again (every rank of a torchrun job would otherwise compile the same error);
    GD3D_HOST=cpp still raises."""
    code = f'''
import logging, os, sys
sys.path.insert(0, {ROOT!r})
logging.basicConfig(level=logging.INFO, format='%(levelname)s %(message)s')
from mmdet3d_gaussian_amd import _lib
b = _lib._build
b.NODE_PATH = os.path.join({str(tmp_path)!r}, '_gd3d_node.so'); b.NODE_HASH_PATH = b.NODE_PATH + '.srchash'
calls = []
def broken(force=False, verbose=False):
    calls.append(1)
    raise RuntimeError('compiling torch_node.cpp failed: error: use of undeclared identifier')
b.build_node = broken
node = _lib.load_node()
print('RESULT', node.IMPLEMENTATION if hasattr(node, 'IMPLEMENTATION') else 'cpp', len(calls), os.path.isfile(b.NODE_PATH + '.buildfailed'))
'''
    runs = []
    for _ in range(2):
        r = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, timeout=300,
                           env={k: v for k, v in os.environ.items() if k != 'GD3D_HOST'})
        assert r.returncode == 0, r.stderr[-2000:]
        runs.append((r.stdout.strip().splitlines()[-1], r.stderr))
    assert runs[0][0] == 'RESULT python 1 True' and 'ERROR' in runs[0][1] and 'undeclared identifier' in runs[0][1]
    assert runs[1][0] == 'RESULT python 0 True' and 'ERROR' in runs[1][1] and 'not retried' in runs[1][1]
    r = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, timeout=300, env=dict(os.environ, GD3D_HOST='cpp'))
    assert r.returncode != 0 and 'not retried' in r.stderr
