#!/usr/bin/env python3
"""Generate golden vectors for the eval matcher / centre-distance affinity FROM THE REAL REFERENCE C++.

    make -C oracle ref && python3 -B tests/golden/make_golden_match.py

Uses oracle/_ref/ref_eval*.so = /root/reference/mmdet3d_gaussian/ops/eval/{affinity.cpp, matcher.cpp} compiled where
they lie (oracle/Makefile target `ref`).  Stores inputs and the reference's `match_coco` / `trans_bev` outputs in
tests/golden/match_coco.npz.  Only data is written.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import oracle  # noqa: E402
from make_golden_riou import boxes  # noqa: E402


def main():
    ev = oracle.load_ref_eval()
    assert ev is not None and hasattr(ev, 'match_coco'), 'build oracle/_ref first (make -C oracle ref)'
    rng = np.random.default_rng(7)
    out = {}
    cases = []
    # (name, D, G, T, tie rounding, nan count, p_ignore, p_crowd)
    spec = [('iou_like', 200, 40, 10, None, 0, 0.2, 0.0), ('ties', 150, 25, 5, 1, 0, 0.4, 0.3),
            ('crowd_only', 60, 6, 3, 1, 0, 0.0, 1.0), ('ignore_only', 60, 9, 3, 2, 0, 1.0, 0.0),
            ('nan', 80, 20, 4, None, 25, 0.3, 0.1), ('no_gt', 12, 0, 3, None, 0, 0, 0), ('no_det', 0, 9, 3, None, 0, 0.5, 0.5),
            ('wide', 30, 700, 4, 2, 0, 0.3, 0.1), ('dist_like', 120, 30, 4, None, 0, 0.1, 0.0)]
    for name, D, G, T, rnd, nnan, pi, pc in spec:
        if name == 'dist_like':                       # LARGER_CLOSER = False: costs are distances, thresholds positive
            cost = rng.uniform(0, 6, (D, G)).astype(np.float32)
            thrs = np.array([0.5, 1.0, 2.0, 4.0], np.float32)
        else:                                         # negated IoU (BaseMatcher.__call__, matcher.py:20-24)
            cost = -rng.uniform(0, 1, (D, G)).astype(np.float32)
            thrs = -np.linspace(0.5, 0.95, T).astype(np.float32)
        if rnd is not None:
            cost = np.round(cost, rnd).astype(np.float32)
        if nnan:
            cost.flat[rng.integers(0, cost.size, nnan)] = np.nan
        ign = rng.uniform(0, 1, G) < pi
        crowd = rng.uniform(0, 1, G) < pc
        out[name + '.cost'], out[name + '.thrs'], out[name + '.ignore'], out[name + '.crowd'] = cost, thrs, ign, crowd
        out[name + '.matched'] = ev.match_coco(np.ascontiguousarray(cost), thrs, ign, crowd)
        cases.append(name)
    d, g = boxes(rng, 97, 60), boxes(rng, 33, 60)
    g[:5, :2] = d[:5, :2]                              # zero distances
    out['trans.det'], out['trans.gt'], out['trans.dist'] = d, g, ev.trans_bev(d, g)
    out['cases'] = np.array(cases)
    path = os.path.join(HERE, 'match_coco.npz')
    np.savez_compressed(path, **out)
    print('match_coco.npz', os.path.getsize(path), 'bytes;', {c: int((out[c + '.matched'] >= 0).sum()) for c in cases})


if __name__ == '__main__':
    main()
