#!/usr/bin/env python3
"""Generate golden rotated-IoU matrices FROM THE REAL REFERENCE C++ (build container only).

    make -C oracle ref && python3 -B tests/golden/make_golden_riou.py

Uses oracle/_ref/ref_eval*.so = /root/reference/mmdet3d_gaussian/ops/eval/affinity.cpp
(+ rbox_utils.hpp) compiled where they lie (oracle/Makefile target `ref`).  Stores inputs and
iou_bev / iou_3d outputs in tests/golden/riou_eval.npz.  Only data is written.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import oracle  # noqa: E402


def boxes(rng, n, spread):
    return np.stack([rng.uniform(0, spread, n), rng.uniform(0, spread, n), rng.uniform(-3, 1, n),
                     rng.uniform(0.5, 2.5, n), rng.uniform(0.5, 4.5, n), rng.uniform(0.5, 2.0, n),
                     rng.uniform(-np.pi, np.pi, n)], -1).astype(np.float32)


def main():
    ev = oracle.load_ref_eval()
    assert ev is not None, 'build oracle/_ref first (make -C oracle ref)'
    rng = np.random.default_rng(0)
    out = {}
    # 5 random boxes shifted by 0.3 m -> iou_3d = iou_bev diag
    d = boxes(rng, 5, 70)
    g = d.copy(); g[:, 0] += 0.3
    out['shift.det'], out['shift.gt'] = d, g
    # dense overlaps: 64 x 64 in a 12 m patch
    d = boxes(rng, 64, 12); g = boxes(rng, 64, 12)
    out['dense.det'], out['dense.gt'] = d, g
    # near-duplicates / shared edges / axis-aligned / contained: the degenerate geometry cases
    d = boxes(rng, 32, 8)
    g = d.copy()
    g[:8, :2] += rng.normal(0, 1e-3, (8, 2)).astype(np.float32)          # near-identical
    g[8:16, 6] += np.float32(np.pi / 2)                                   # crossed
    d[16:24, 6] = 0; g[16:24, 6] = 0; g[16:24, 0] += d[16:24, 3]          # axis-aligned, shared edge
    g[24:, 3:5] *= 0.5                                                    # contained
    out['degen.det'], out['degen.gt'] = d, g
    # ragged
    out['ragged.det'], out['ragged.gt'] = boxes(rng, 7, 6), boxes(rng, 19, 6)
    for k in ('shift', 'dense', 'degen', 'ragged'):
        dd = np.ascontiguousarray(out[k + '.det']); gg = np.ascontiguousarray(out[k + '.gt'])
        out[k + '.iou_bev'] = ev.iou_bev(dd, gg)
        out[k + '.iou_3d'] = ev.iou_3d(dd, gg, 0.5)
        out[k + '.iou_3d_z0'] = ev.iou_3d(dd, gg, 0.0)
    np.savez_compressed(os.path.join(HERE, 'riou_eval.npz'), **out)
    print('shift diag iou_3d', np.diag(out['shift.iou_3d']))
    print('riou_eval.npz', os.path.getsize(os.path.join(HERE, 'riou_eval.npz')), 'bytes')


if __name__ == '__main__':
    main()
