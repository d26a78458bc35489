#!/usr/bin/env python3
"""Greedy NMS keep lists derived from THE REFERENCE'S OWN rotated-IoU arithmetic (build container only).

    make -C oracle ref && python3 -B tests/golden/make_golden_nms_ref_iou.py

The NMS op itself (`mmdet3d.ops.iou3d.nms_gpu`, call site gd_centerpoint_head.py:336-345) is third party and absent.  What
the reference DOES hold is one rotated-IoU implementation: ops/eval/affinity.cpp:51-81 (`iou_bev`) over
rbox_utils.hpp:280-302, compiled unchanged into oracle/_ref.  This script feeds the box sets of the NMS call sites through
it — score-sorted, converted to its (x, y, ., w, h, ., yaw) rows — and stores, per set, the sparse (i < j) IoU matrix and the
greedy keep list that matrix implies.  tests/nms_ref.py documents the layout; tests/test_nms_ref_iou.py (CPU restatement)
and tests/test_gpu_rbox.py (HIP path) compare against it.  Only data is written (tests/golden/nms_ref_iou.npz) plus a
plain-text summary (profiles/r04_nms_ref_crosscheck.txt).
"""
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(HERE))
import oracle  # noqa: E402
from nms_ref import BAND, exact_iou_xyxyr, greedy, to_eval7, uncertain  # noqa: E402
from rbox_inputs import nms_boxes  # noqa: E402


def origin_boxes(n, seed):
    """Dense overlaps within +-8 m of the origin (coordinates carry <= 1e-6 m of rounding): the set on which two fp32 IoU
    evaluations are expected to agree to 1e-5."""
    rng = np.random.default_rng(seed)
    c = rng.uniform(-8, 8, (n, 2)); wl = rng.uniform(0.8, 4.5, (n, 2))
    b = np.concatenate([c - wl / 2, c + wl / 2, rng.uniform(-np.pi, np.pi, (n, 1))], 1).astype(np.float32)
    return b, rng.uniform(0, 1, n).astype(np.float32)


#        name       boxes, scores                                           thr   pre   post
SETS = [('waymo0', lambda: nms_boxes(4096, seed=200), 0.25, 4096, 500),     # BASELINE configs[4]: 3 classes x 4096, thr 0.25
        ('waymo1', lambda: nms_boxes(4096, seed=201), 0.25, 4096, 500),
        ('waymo2', lambda: nms_boxes(4096, seed=202), 0.25, 4096, 500),
        ('nuscenes', lambda: nms_boxes(1000, seed=77, extent=51.2), 0.2, 1000, 83),    # centerpoint nus test_cfg: pre 1000, post 83, thr 0.2
        ('pvrcnn', lambda: nms_boxes(1500, seed=310, extent=40.0), 0.7, 1024, 100),   # PV-RCNN rpn test_cfg shape: nms_pre 1024, nms_post 100, thr 0.7
        ('rpn9000', lambda: nms_boxes(9000, seed=311, extent=70.4), 0.8, 9000, 512),  # PV-RCNN rpn train_cfg shape: 9000 / 512 / 0.8
        ('origin', lambda: origin_boxes(256, 5), 0.5, 256, 256),
        # PointPillars KITTI test_cfg (configs/_base_/models/hv_pointpillars_secfpn_kitti.py:92-96): thr 0.01 — boxes that barely touch
        ('kitti', lambda: nms_boxes(4096, seed=400, extent=40.0), 0.01, 4096, 100)]


def sparse_ref_iou(ev, sorted_boxes, chunk=256):
    d = np.ascontiguousarray(to_eval7(sorted_boxes))
    m = d.shape[0]
    ri, rj, rv = [], [], []
    for s in range(0, m, chunk):
        blk = ev.iou_bev(np.ascontiguousarray(d[s:s + chunk]), d)          # (chunk, m): rows = earlier box, affinity.cpp:51-81
        i, j = np.nonzero(blk > 0)
        sel = j > i + s
        ri.append((i[sel] + s).astype(np.int32)); rj.append(j[sel].astype(np.int32)); rv.append(blk[i[sel], j[sel]])
    return np.concatenate(ri), np.concatenate(rj), np.concatenate(rv).astype(np.float32)


def main():
    ev = oracle.load_ref_eval()
    assert ev is not None, 'build oracle/_ref first (make -C oracle ref)'
    out, report = {}, []
    report.append('NMS keep lists: this repo (CPU restatement of mmdet3d iou3d, oracle/rbox_oracle.c part 1; the HIP kernels are\n'
                  'bit-identical to it) against greedy lists derived from the reference\'s own compiled iou_bev\n'
                  '(ops/eval/affinity.cpp:51-81 + rbox_utils.hpp, oracle/_ref).  Written by tests/golden/make_golden_nms_ref_iou.py.\n')
    for name, make, thr, pre, post in SETS:
        b, s = make()
        order = np.argsort(-s, kind='stable')[:pre]
        bs = np.ascontiguousarray(b[order])
        m = len(order)
        t0 = time.time()
        ni, nj, nv = sparse_ref_iou(ev, bs)
        t_ref = time.time() - t0
        keep = np.flatnonzero(greedy(m, ni, nj, nv, thr))
        ck, cd, unc = uncertain(m, ni, nj, nv, thr, BAND)
        near_sel = np.flatnonzero(np.abs(nv.astype(np.float64) - np.float64(np.float32(thr))) < BAND)
        near_exact = np.array([exact_iou_xyxyr(bs[ni[t]], bs[nj[t]]) for t in near_sel], np.float64)
        out[f'{name}.near_i'], out[f'{name}.near_j'], out[f'{name}.near_exact'] = ni[near_sel], nj[near_sel], near_exact
        for k, v in (('boxes', b), ('scores', s), ('thr', np.float32(thr)), ('pre', np.int64(pre)), ('post', np.int64(post)),
                     ('order', order.astype(np.int64)), ('nz_i', ni), ('nz_j', nj), ('nz_iou', nv), ('keep_ref', keep.astype(np.int64))):
            out[f'{name}.{k}'] = v
        # how this repo's restatement compares, at generation time
        own_keep = oracle.nms_bev(bs, thr)
        own_iou = oracle.iou_bev_xyxyr(bs, bs)[ni, nj] if m <= 4096 else None
        if own_iou is None:   # 9000^2 fp32 = 324 MB: by row blocks
            own_iou = np.empty(len(ni), np.float32)
            st = np.searchsorted(ni, np.arange(0, m + 512, 512))
            for k in range(len(st) - 1):
                a, e = st[k], st[k + 1]
                if e > a:
                    blk = oracle.iou_bev_xyxyr(bs[k * 512:(k + 1) * 512], bs)
                    own_iou[a:e] = blk[ni[a:e] - k * 512, nj[a:e]]
        diff = np.abs(own_iou.astype(np.float64) - nv.astype(np.float64))
        near = np.abs(nv.astype(np.float64) - np.float64(np.float32(thr)))
        same = np.array_equal(own_keep, keep)
        flips = int(((own_iou > np.float32(thr)) != (nv > np.float32(thr))).sum())
        report.append(f'## {name}: {m} boxes (of {len(s)}), thr {thr}, post {post}\n'
                      f'reference-derived keep list: {len(keep)} kept; pairs with reference IoU > 0: {len(nv)} ({t_ref:.1f} s of affinity.cpp)\n'
                      f'|IoU(restatement) - IoU(reference)| over those pairs: max {diff.max():.3e}, mean {diff.mean():.3e}, '
                      f'pairs above 1e-5: {int((diff > 1e-5).sum())}, above 1e-4: {int((diff > 1e-4).sum())}\n'
                      f'pairs with |IoU(reference) - thr| < 1e-5: {int((near < 1e-5).sum())}, < 1e-4: {int((near < 1e-4).sum())}, '
                      f'< 1e-3: {int((near < 1e-3).sum())}; smallest {near.min():.3e}\n'
                      f'suppress decisions (IoU > thr) that differ between the two IoU evaluations, over all stored pairs: {flips}\n'
                      f'boxes whose state is undecidable within +-{BAND:g} of the threshold (tests/nms_ref.py uncertain()): {int(unc.sum())}\n'
                      f'keep list of the restatement vs reference-derived: {"IDENTICAL" if same else "DIFFERENT"}'
                      + ('' if same else f' (symmetric difference {len(set(own_keep.tolist()) ^ set(keep.tolist()))})') + '\n')
        worst = np.argsort(-diff)[:3]
        lines = ['largest IoU differences, with the fp64 clipping of the same fp32 boxes as arbiter (position pair in score order):']
        for t in worst:
            lines.append(f'  ({ni[t]}, {nj[t]}): reference {nv[t]:.9g}  restatement {own_iou[t]:.9g}  fp64 {exact_iou_xyxyr(bs[ni[t]], bs[nj[t]]):.9g}')
        if len(near_sel):
            lines.append(f'pairs with |IoU(reference) - thr| < {BAND:g}: which side of thr = {np.float32(thr):.9g} each evaluation puts them')
            for t, ex in zip(near_sel, near_exact):
                side = lambda v: '>' if v > np.float32(thr) else '<='   # noqa: E731
                lines.append(f'  ({ni[t]}, {nj[t]}): reference {nv[t]:.9g} ({side(nv[t])})  restatement {own_iou[t]:.9g} ({side(own_iou[t])})  '
                             f'fp64 {ex:.9g} ({side(ex)})')
        report[-1] += '\n'.join(lines) + '\n'
        print(report[-1])
    path = os.path.join(HERE, 'nms_ref_iou.npz')
    np.savez_compressed(path, **out)
    report.append(f'fixture: tests/golden/nms_ref_iou.npz, {os.path.getsize(path)} bytes\n')
    with open(os.path.join(ROOT, 'profiles', 'r04_nms_ref_crosscheck.txt'), 'w') as f:
        f.write('\n'.join(report))
    print(report[-1])


if __name__ == '__main__':
    main()
