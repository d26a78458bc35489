#!/usr/bin/env python3
"""Accuracy of the fused kernel's per-pair math, evaluated ON THE CPU (csrc/gd3d_device.h compiled by g++ through
tests/hostmath; the hardware's 1-ulp v_rcp / v_rsq / v_log / v_exp are exact there), against the golden vectors of the
real reference: per (case, family, quantity) max |ours - ref64| next to the reference's own |ref32 - ref64|, relative to
1 + scale as in tests/gd_golden.py.  Development aid for changes to the closed forms: no GPU needed.
usage: tools/host_accuracy.py [substring of case names ...]"""
import ctypes
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np  # noqa: E402

from gd_golden import _relerr, families, index, pair_case_names, pairs  # noqa: E402


def build():
    so = os.path.join(tempfile.mkdtemp(prefix='hostmath'), 'libpairmath.so')
    cmd = ['/opt/rocm/lib/llvm/bin/clang++', '-O1', '-std=c++17', '-shared', '-fPIC', '-I', os.path.join(ROOT, 'tests', 'hostmath'), '-I', ROOT,
           os.path.join(ROOT, 'tests', 'hostmath', 'pair_math.cpp'), '-o', so]
    subprocess.run(cmd, check=True)
    lib = ctypes.CDLL(so)
    lib.hostmath_pairs.restype = ctypes.c_int
    return lib


def run(lib, loss_type, kw, pred, target):
    import mmdet3d_gaussian_amd as amd
    kw = dict(kw)
    prm = amd.make_params(loss_type, kw.pop('fun', 'log1p'), kw.pop('tau', 1.0), kw.pop('alpha', 1.0),
                          tuple(kw.pop('center_offset', (0, 0, 0.5))), kw)
    pred = np.ascontiguousarray(pred, np.float32)
    target = np.ascontiguousarray(target, np.float32)
    n = pred.shape[0]
    loss = np.empty(n, np.float32)
    gp = np.empty((n, 7), np.float32)
    gt = np.empty((n, 7), np.float32)
    vp = lambda a: a.ctypes.data_as(ctypes.c_void_p)  # noqa: E731
    with np.errstate(all='ignore'):
        rc = lib.hostmath_pairs(ctypes.byref(prm), vp(pred), vp(target), ctypes.c_long(n), ctypes.c_float(1.0), vp(loss),
                                vp(gp), vp(gt))
    assert rc == 0
    return loss, gp, gt


def main():
    subs = sys.argv[1:]
    lib = build()
    g = pairs()
    worse = total = 0
    print(f'{"case.family.quantity":40s} {"ours-ref64":>10s} {"ref32-ref64":>12s}  ours<=ref32  flat_1e-5')
    for case in pair_case_names():
        if subs and not any(s in case for s in subs):
            continue
        c = index()['pairs']['cases'][case]
        for fam in families(with_ident=False):
            kw = {k: (tuple(v) if isinstance(v, list) else v) for k, v in c['kwargs'].items()}
            loss, gp, gt = run(lib, c['loss_type'], kw, g[f'in.{fam}.pred'], g[f'in.{fam}.target'])
            key = f'{case}.{fam}'
            for q, ours, rw in (('loss', loss, False), ('gp', gp, True), ('gt', gt, True)):
                r64, r32 = g[f'{key}.{q}64'], g[f'{key}.{q}32']
                e, e32, er = _relerr(ours, r64, r64, rw), _relerr(ours, r32, r64, rw), _relerr(r32, r64, r64, rw)
                total += 1
                worse += e > er
                print(f'{key + "." + q:40s} {e:10.2e} {er:12.2e}  {"yes" if e <= er else "NO":>11s}  '
                      f'{"yes" if (e <= 1e-5 and e32 <= 1e-5) else "no"}')
    print(f'# {worse} of {total} comparisons are LESS accurate than the reference\'s own fp32')


if __name__ == '__main__':
    main()
