"""The fused kernel's per-pair DEVICE math (csrc/gd3d_device.h), compiled for the host by g++ through the stand-in header
tests/hostmath/hip/hip_runtime.h, against the golden vectors generated from the real reference.  CPU-only coverage of
the arithmetic the GPU runs: closed forms, hand-derived gradients, clamp / tie / NaN / inf rules and the launch dispatch on
(loss type, fun, flag).  The hardware's 1-ulp v_rcp / v_rsq / v_sqrt / v_log / v_exp are IEEE-exact here, everything
else is the same source.  Test infrastructure only: nothing in the product can reach this build."""
import ctypes
import os
import shutil
import subprocess

import numpy as np
import pytest

from gd_golden import (NONFINITE_CASES, check_close, check_golden, check_nonfinite, check_nonfinite_grad_rows, families, grad_bound, index,
                       loss_bound, nonfinite, pair_case_names, pairs)

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


@pytest.fixture(scope='module')
def hostmath():
    # clang (the ROCm toolchain's own host compiler): gd3d_device.h uses clang's vector extensions for packed fp32 math
    cxx = next((c for c in ('/opt/rocm/lib/llvm/bin/clang++', shutil.which('clang++') or '') if c and os.path.exists(c)), None)
    if cxx is None:
        pytest.skip('clang++ not available')
    out_dir = os.path.join(HERE, 'hostmath', '_build')
    os.makedirs(out_dir, exist_ok=True)
    so = os.path.join(out_dir, f'libpairmath.{os.getpid()}.so')
    cmd = [cxx, '-O1', '-std=c++17', '-shared', '-fPIC', '-I', os.path.join(HERE, 'hostmath'), '-I', ROOT,
           os.path.join(HERE, 'hostmath', 'pair_math.cpp'), '-o', so]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    lib = ctypes.CDLL(so)
    lib.hostmath_pairs.restype = ctypes.c_int
    yield lib
    try:
        os.remove(so)
    except OSError:
        pass


def _run(lib, loss_type, kw, pred, target, scale=1.0):
    import mmdet3d_gaussian_amd as amd          # host-side parameter mapping of the product (no GPU needed for it)
    kw = dict(kw)
    prm = amd.make_params(loss_type, kw.pop('fun', 'log1p'), kw.pop('tau', 1.0), kw.pop('alpha', 1.0),
                          tuple(kw.pop('center_offset', (0, 0, 0.5))), kw)
    pred = np.ascontiguousarray(pred, np.float32); target = np.ascontiguousarray(target, np.float32)
    n = pred.shape[0]
    loss = np.empty(n, np.float32); gp = np.empty((n, 7), np.float32); gt = np.empty((n, 7), np.float32)
    vp = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    with np.errstate(all='ignore'):
        rc = lib.hostmath_pairs(ctypes.byref(prm), vp(pred), vp(target), ctypes.c_long(n), ctypes.c_float(scale), vp(loss),
                                vp(gp), vp(gt))
    assert rc == 0
    return loss, gp, gt


@pytest.mark.parametrize('case', pair_case_names())
def test_device_math_on_host_against_reference_golden(hostmath, case):
    """Same data, same tolerance policy as tests/test_gpu_gd_loss.py::test_pairs_against_reference_golden."""
    c = index()['pairs']['cases'][case]
    g = pairs()
    for fam in families():
        kw = {k: (tuple(v) if isinstance(v, list) else v) for k, v in c['kwargs'].items()}
        loss, gp, gt = _run(hostmath, c['loss_type'], kw, g[f'in.{fam}.pred'], g[f'in.{fam}.target'])
        key = f'{case}.{fam}'
        l64, l32 = g[key + '.loss64'], g[key + '.loss32']
        if fam == 'ident':
            check_close(key + '.loss', loss, l64, np.maximum(loss_bound(l64, l32), 2e-3), noisy=True)
            continue
        check_golden(key + '.loss', loss, l64, l32, False)
        check_golden(key + '.gp', gp, g[key + '.gp64'], g[key + '.gp32'], True)
        check_golden(key + '.gt', gt, g[key + '.gt64'], g[key + '.gt32'], True)


def test_device_math_on_host_nonfinite_and_degenerate_rows(hostmath):
    """NaN exactly where the reference has it; +inf centres / heights and an overflowing centre give the reference's 1.0
    (sqrt(inf) and log1p(inf) are inf, not inf * 0); a yaw of 1e6 goes through the two-angle path of gwd3d."""
    gold = nonfinite()
    for lt, kw in NONFINITE_CASES:
        loss, gp, gt = _run(hostmath, lt, kw, gold['pred'], gold['target'])
        check_nonfinite(lt, loss, gold[f'{lt}.loss32'], gold[f'{lt}.loss64'])
        check_nonfinite_grad_rows(lt + '.gp', gp, gold[f'{lt}.gp_nanrow32'], gold[f'{lt}.gp_nanrow64'])
        check_nonfinite_grad_rows(lt + '.gt', gt, gold[f'{lt}.gt_nanrow32'], gold[f'{lt}.gt_nanrow64'])


def test_device_math_large_yaws_keep_the_reference_accuracy(hostmath):
    """gwd3d takes sin / cos of the yaw DIFFERENCE; beyond |yaw| = 16 that difference is formed from the two angles'
    own sines and cosines, so a common offset of 2 pi k (exact multiples are not representable: use the oracle on the
    SAME rounded inputs) does not cost accuracy: kernel math vs fp64 oracle on yaws around 1e3 .. 1e5."""
    import oracle
    rng = np.random.default_rng(3)
    n = 256
    t = np.stack([rng.uniform(0, 70, n), rng.uniform(-40, 40, n), rng.uniform(-3, 1, n), rng.uniform(0.5, 2.5, n),
                  rng.uniform(0.5, 4.5, n), rng.uniform(0.5, 2, n), rng.uniform(-3.14, 3.14, n)], -1)
    p = t + rng.normal(0, 1, (n, 7)) * np.array([0.3, 0.3, 0.1, 0.1, 0.1, 0.1, 0.3])
    off = rng.choice([1e3, -1e3, 1e4, 1e5, 17.0], n)
    p[:, 6] += off
    t[:, 6] += off + rng.choice([0.0, 2 * np.pi, -4 * np.pi], n)
    p32, t32 = p.astype(np.float32), t.astype(np.float32)
    loss, gp, _ = _run(hostmath, 'gwd3d', dict(fun='log1p', tau=1.0), p32, t32)
    prm = oracle.make_params('gwd3d', fun='log1p', tau=1.0)
    r64 = oracle.gd_loss(p32, t32, prm, scale=1.0, dtype=np.float64)
    assert np.max(np.abs(loss - r64['loss']) / (1 + np.abs(r64['loss']))) <= 1e-5
    sc = 1 + np.abs(r64['grad_pred']).max(-1, keepdims=True)
    assert np.max(np.abs(gp - r64['grad_pred']) / sc) <= 3e-5


# ------------------------------------------------------------------------------------------------ rotated-box geometry
@pytest.fixture(scope='module')
def rboxmath():
    # clang (the ROCm toolchain's own host compiler): gd3d_device.h uses clang's vector extensions for packed fp32 math
    cxx = next((c for c in ('/opt/rocm/lib/llvm/bin/clang++', shutil.which('clang++') or '') if c and os.path.exists(c)), None)
    if cxx is None:
        pytest.skip('clang++ not available')
    out_dir = os.path.join(HERE, 'hostmath', '_build')
    os.makedirs(out_dir, exist_ok=True)
    so = os.path.join(out_dir, f'librboxmath.{os.getpid()}.so')
    # -ffp-contract=off as for csrc/rbox.hip: every step one IEEE fp32 operation, so the NMS part is bit-reproducible
    cmd = [cxx, '-O1', '-std=c++17', '-ffp-contract=off', '-shared', '-fPIC', '-I', os.path.join(HERE, 'hostmath'), '-I', ROOT,
           os.path.join(HERE, 'hostmath', 'rbox_math.cpp'), '-o', so]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    lib = ctypes.CDLL(so)
    lib.hostmath_iou_xyxyr.restype = None
    lib.hostmath_eval_iou.restype = None
    yield lib
    try:
        os.remove(so)
    except OSError:
        pass


def _vp(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def _host_iou(lib, a, b):
    a = np.ascontiguousarray(a, np.float32); b = np.ascontiguousarray(b, np.float32)
    out = np.empty((len(a), len(b)), np.float32)
    lib.hostmath_iou_xyxyr(_vp(a), ctypes.c_long(len(a)), _vp(b), ctypes.c_long(len(b)), _vp(out))
    return out


@pytest.mark.parametrize('clutter', [True, False])
def test_nms_overlap_math_on_host_is_bit_identical_to_the_oracle(rboxmath, clutter):
    """The per-pair IoU the NMS mask kernels evaluate (rbox_device.h part 1), compiled for the host, against the CPU
    restatement of mmdet3d's overlap: same bits on clustered and scattered boxes, degenerate ones included."""
    import oracle
    from rbox_inputs import nms_boxes
    a, _ = nms_boxes(300, seed=5, clutter=clutter)
    b, _ = nms_boxes(260, seed=6, clutter=clutter)
    b[:40] = a[:40]                                          # identical pairs
    b[40:50, 2] = b[40:50, 0]                                # zero width
    b[50:60, 4] = a[50:60, 4] + np.float32(np.pi / 2)        # quarter turn of a near-copy
    b[50:60, :4] = a[50:60, :4]
    got = _host_iou(rboxmath, a, b)
    want = oracle.iou_bev_xyxyr(a, b)
    assert np.array_equal(got.view(np.int32), np.asarray(want, np.float32).view(np.int32))
    far = np.array([[498.5, 19.0, 501.5, 21.0, 0.0]], np.float32)     # in-box margin below half an ulp: IoU with itself = 0
    assert _host_iou(rboxmath, far, far)[0, 0] == 0.0 and _host_iou(rboxmath, far - np.float32([400, 0, 400, 0, 0]), far - np.float32([400, 0, 400, 0, 0]))[0, 0] == 1.0


@pytest.mark.parametrize('fam', ['shift', 'dense', 'degen', 'ragged'])
def test_eval_iou_math_on_host_against_reference_golden(rboxmath, fam):
    """rbox_device.h part 2 + eval_iou on the host vs tests/golden/riou_eval.npz (from the reference's own affinity.cpp),
    same tolerance as the GPU test."""
    g = np.load(os.path.join(HERE, 'golden', 'riou_eval.npz'))
    det = np.ascontiguousarray(g[fam + '.det'], np.float32); gt = np.ascontiguousarray(g[fam + '.gt'], np.float32)
    for key, is3d, zo in (('iou_bev', 0, 0.5), ('iou_3d', 1, 0.5), ('iou_3d_z0', 1, 0.0)):
        out = np.empty((len(det), len(gt)), np.float32)
        rboxmath.hostmath_eval_iou(_vp(det), ctypes.c_long(len(det)), _vp(gt), ctypes.c_long(len(gt)), ctypes.c_int(is3d),
                                   ctypes.c_float(zo), _vp(out))
        np.testing.assert_allclose(out, g[f'{fam}.{key}'], atol=1e-5, rtol=0)


@pytest.mark.parametrize('kind', __import__('gd_stress').FAMILIES)
def test_device_math_on_host_stress_families(hostmath, kind):
    """tests/gd_stress.py families through the host build of the kernel math vs the fp64 oracle (same policy as the GPU
    test test_stress_families_against_fp64_oracle)."""
    import oracle
    from gd_golden import stress_bounds
    from gd_stress import stress_pairs
    p, t = stress_pairs(1024, kind, seed=1)
    for lt in ('gwd3d', 'kld3d', 'bd3d', 'jd3d', 'kld3d_symmax', 'kld3d_symmin', 'kfiou3d'):
        for fun, tau in ((('none', 0.0), ('expm1', 0.0)) if lt == 'kfiou3d' else (('log1p', 1.0), ('none', 0.0))):
            prm = oracle.make_params(lt, fun=fun, tau=tau)
            with np.errstate(all='ignore'):
                ref = oracle.gd_loss(p, t, prm, scale=1.0)
                lb, gb = stress_bounds(kind, p, t, prm, ref, 1.0)
            loss, gp, _ = _run(hostmath, lt, dict(fun=fun, tau=tau), p, t)
            check_close(f'{kind}.{lt}.{fun}.loss', loss, ref['loss'], lb)
            check_close(f'{kind}.{lt}.{fun}.gp', gp, ref['grad_pred'], gb)
