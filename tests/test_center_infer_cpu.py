"""The restated CenterPoint inference slice (oracle/center_infer_torch.py) against vectors of the REAL reference coders
(tests/golden/center_infer.npz, written by tests/golden/make_golden_center_infer.py).  CPU only."""
import os

import numpy as np
import pytest
import torch

from oracle import center_infer_torch as cit
from oracle import coder_torch

GOLD = os.path.join(os.path.dirname(__file__), 'golden', 'center_infer.npz')
CASES = ('rev_c1', 'rev_c3', 'yaw_c2', 'yaw_c1_k500', 'yaw_ties')


def load(name):
    z = np.load(GOLD)
    d = {k.split('.', 1)[1]: torch.from_numpy(z[k]) for k in z.files if k.startswith(name + '.')}
    cfg = dict(pc_range=z['cfg_pc_range'].tolist(), voxel_size=z['cfg_voxel_size'].tolist(),
               out_size_factor=int(z['cfg_out_size_factor']), norm_bbox=True)
    return d, cfg


@pytest.mark.parametrize('name', CASES)
def test_select_best_and_decode_equal_the_reference_coders(name):
    d, cfg = load(name)
    kind = 'rev' if name.startswith('rev') else 'yaw'
    pd = {k: d[k] for k in ('reg', 'height', 'dim', 'rot', 'yaw', 'dir', 'vel') if k in d}
    K = int(d['K'])
    scores, clses, locs, preds = cit.select_best(d['heat'].sigmoid(), cit.reconstruct(pd, kind), K)
    assert torch.equal(scores, d['scores'])            # the value multiset is unique even under ties
    if name.endswith('ties'):
        # which of several equal scores is taken is torch.topk's choice: compare order-free facts only
        assert torch.equal(scores.sort(descending=True)[0], scores)
    else:
        assert torch.equal(clses, d['clses']) and torch.equal(locs, d['locs']) and torch.equal(preds, d['preds'])
    dec = (cit.decode_rev(d['locs'], d['preds'], cfg['pc_range'], cfg['out_size_factor'], cfg['voxel_size']) if kind == 'rev' else
           coder_torch.center_decode(d['locs'], d['preds'], cfg['pc_range'], cfg['out_size_factor'], cfg['voxel_size']))
    assert torch.equal(dec, d['boxes'])


def test_limit_range_mask_keeps_the_reference_quirk():
    s = torch.tensor([[0.5, 0.05, 0.9]])
    b = torch.tensor([[[0.0, 0.0, 0.0], [0.0, 0.0, 0.0], [-70.0, 0.0, 0.0]]])
    # `.ge(lo).le(hi)`: the BOOLEAN (x >= lo) is compared with hi — with hi >= 1 every box passes, whatever its centre
    m = cit.center_mask(s, b, 0.1, [-61.2, -61.2, -10.0, 61.2, 61.2, 10.0])
    assert m.tolist() == [[True, False, True]]
    # hi in [0, 1): passes iff the centre is BELOW lo; hi < 0: nothing passes
    assert cit.center_mask(s, b, 0.1, [-61.2, -61.2, -10.0, 0.5, 61.2, 10.0]).tolist() == [[False, False, True]]
    assert cit.center_mask(s, b, 0.1, [-61.2, -61.2, -10.0, 61.2, -1.0, 10.0]).tolist() == [[False, False, False]]


def test_get_bboxes_restatement_runs_and_is_consistent():
    d, cfg = load('yaw_c2')
    pd = {k: d[k] for k in ('reg', 'height', 'dim', 'yaw', 'dir', 'vel')}
    pd['heatmap'] = d['heat']
    test_cfg = dict(max_per_img=64, score_threshold=0.8, post_center_limit_range=[-61.2, -61.2, -10.0, 61.2, 61.2, 10.0],
                    nms_type='rotate', nms_thr=0.2, pre_max_size=1000, post_max_size=83)
    stage = {}
    out = cit.get_bboxes([pd, pd], 'yaw', cfg, test_cfg, [2, 2], stage=stage)
    assert len(out) == 2
    for b, (boxes, scores, labels) in enumerate(out):
        n0 = int(stage[0]['mask'][b].sum())
        assert 0 < boxes.shape[0] <= 2 * n0 and boxes.shape[1] == 9 and labels.dtype == torch.int32
        half = boxes.shape[0] // 2               # the two tasks see the same maps: same detections, labels shifted by 2
        assert torch.equal(boxes[:half], boxes[half:]) and torch.equal(labels[:half] + 2, labels[half:])
        assert bool((scores[:half][:-1] >= scores[:half][1:]).all()) and float(scores.min()) >= 0.8
