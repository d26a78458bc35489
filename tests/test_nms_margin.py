"""Decision margins of the rotated NMS and its sensitivity to an ulp of sin / cos, on the CPU
restatement of the upstream op (parity unpinned: mmdet3d is absent; oracle/rbox_oracle.c part 1).  The GPU kernels are
bit-identical to that restatement (tests/test_gpu_rbox.py), so what holds here holds for them.  The report these
numbers go into is tests/perf/nms_margin_report.py -> profiles/r03_nms_margin.txt."""
import os
import sys

import numpy as np
import pytest

import oracle
from rbox_inputs import nms_boxes



@pytest.fixture(autouse=True)
def _nudge_off():
    oracle.set_trig_nudge(0)
    yield
    oracle.set_trig_nudge(0)


def test_margin_statistics_are_those_of_the_greedy_scan():
    """nms_margin walks exactly the pairs nms_bev evaluates: same kept count; a threshold placed just under / over the
    closest IoU flips exactly the decisions its own histogram announces."""
    b, s = nms_boxes(600, seed=3)
    m = oracle.nms_margin(b, s, 0.3)
    keep = oracle.nms_gpu_oracle(b, s, 0.3)
    assert m['kept'] == len(keep) and 0 < m['overlapping'] <= m['pairs']
    assert m['within'][1e-1] >= m['within'][1e-2] >= m['within'][1e-3]
    assert m['min_margin'] >= 0 and abs(abs(m['iou_at_min'] - np.float32(0.3)) - m['min_margin']) < 1e-12
    # moving the threshold onto the far side of that closest pair changes the outcome of at least that decision
    thr2 = m['iou_at_min'] + (1e-6 if m['iou_at_min'] <= np.float32(0.3) else -1e-6)
    m2 = oracle.nms_margin(b, s, float(thr2))
    assert m2['min_margin'] < 2e-6


@pytest.mark.parametrize('name,n,seed,extent,clutter,thr', [
    ('waymo0', 4096, 200, 74.88, True, 0.25), ('waymo1', 4096, 201, 74.88, True, 0.25), ('waymo2', 4096, 202, 74.88, True, 0.25),
    ('nuscenes', 1000, 77, 51.2, True, 0.2), ('sparse', 4096, 5, 74.88, False, 0.25)])
def test_keep_list_survives_an_ulp_of_sin_and_cos(name, n, seed, extent, clutter, thr):
    """Every box's sin and cos nudged by +1, -1 and pseudo-random {-1, 0, +1} ulp (what a different but correct fp32
    sin / cos could do): the keep list of the BASELINE workloads does not move.  Where a decision sits closer to the
    threshold than 2e-7 the test would report the moved entries instead of failing blindly."""
    b, s = nms_boxes(n, seed=seed, extent=extent, clutter=clutter)
    base = oracle.nms_gpu_oracle(b, s, thr, pre_max_size=n)
    m = oracle.nms_margin(b, s, thr, pre_max_size=n)
    moved = {}
    for mode, sd in [(1, 0), (2, 0)] + [(3, k) for k in range(1, 9)]:
        oracle.set_trig_nudge(mode, sd)
        k = oracle.nms_gpu_oracle(b, s, thr, pre_max_size=n)
        oracle.set_trig_nudge(0)
        if not np.array_equal(k, base):
            moved[(mode, sd)] = len(set(k.tolist()) ^ set(base.tolist()))
    if m['min_margin'] > 2e-7:
        assert not moved, (name, m['min_margin'], moved)
    else:   # a decision within the reach of an ulp: say so, with the damage
        pytest.xfail(f'{name}: closest decision {m["min_margin"]:.2e} from the threshold; keep entries moved: {moved}')


def test_the_nudge_knob_really_moves_sin_and_cos():
    """Sanity of the knob itself: with a threshold placed ON an IoU value, one ulp does flip that decision."""
    b, s = nms_boxes(64, seed=11)
    order = np.argsort(-s, kind='stable')
    iou = oracle.iou_bev_xyxyr(b[order], b[order])
    flips = 0
    vals = np.unique(iou[np.triu_indices(64, 1)])
    vals = vals[(vals > 0.05) & (vals < 0.9)][:40]
    for v in vals:
        base = oracle.nms_gpu_oracle(b, s, float(v))
        for mode in (1, 2):
            oracle.set_trig_nudge(mode, 0)
            k = oracle.nms_gpu_oracle(b, s, float(v))
            oracle.set_trig_nudge(0)
            flips += not np.array_equal(k, base)
    assert flips > 0
