"""The `_cpu` twins of the rotated-box entry points (csrc/rbox_cpu.cpp: the kernels' geometry source compiled for the host) through
the Python surface on CPU tensors.  Pinned like the GPU path: ops/eval IoU against the fixtures written by the reference's own
compiled affinity.cpp (tests/golden/riou_eval.npz), NMS against the keep lists derived from the reference's iou_bev
(tests/golden/nms_ref_iou.npz) and, bit for bit, against the oracle's restatement of the same operation sequence."""
import ctypes
import os

import numpy as np
import pytest
import torch

import oracle
import nms_ref
from rbox_inputs import eval_boxes, nms_boxes

import mmdet3d_gaussian_amd as amd

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'riou_eval.npz')


@pytest.mark.parametrize('case', ['shift', 'dense', 'degen', 'ragged'])
def test_eval_iou_against_the_compiled_reference_fixtures(case):
    g = np.load(GOLD)
    d, t = torch.from_numpy(g[case + '.det']), torch.from_numpy(g[case + '.gt'])
    # same bound as the GPU test (tests/test_gpu_rbox.py): the reference evaluates sin / cos in double and rounds; 2e-6 covers it
    np.testing.assert_allclose(amd.iou_bev(d, t).numpy(), g[case + '.iou_bev'], rtol=0, atol=2e-6)
    np.testing.assert_allclose(amd.iou_3d(d, t, 0.5).numpy(), g[case + '.iou_3d'], rtol=0, atol=2e-6)
    np.testing.assert_allclose(amd.iou_3d(d, t, 0.0).numpy(), g[case + '.iou_3d_z0'], rtol=0, atol=2e-6)


def test_twins_equal_the_oracle_restatement_bit_for_bit():
    """Same single-operation fp32 sequence in both: IoU matrices, centre distances and keep lists are identical bits."""
    d, g = eval_boxes(80, 3), eval_boxes(61, 4)
    assert np.array_equal(amd.iou_bev(torch.from_numpy(d), torch.from_numpy(g)).numpy(), oracle.eval_iou_bev(d, g))
    assert np.array_equal(amd.iou_3d(torch.from_numpy(d), torch.from_numpy(g), 0.25).numpy(), oracle.eval_iou_3d(d, g, 0.25))
    assert np.array_equal(amd.trans_bev(torch.from_numpy(d), torch.from_numpy(g)).numpy(), oracle.eval_trans_bev(d, g))
    b, s = nms_boxes(700, seed=9)
    assert np.array_equal(amd.boxes_iou_bev(torch.from_numpy(b[:200]), torch.from_numpy(b)).numpy(), oracle.iou_bev_xyxyr(b[:200], b))
    for thr, pre, post in ((0.25, None, None), (0.6, 500, 40), (0.01, None, 7)):
        got = amd.nms_gpu(torch.from_numpy(b), torch.from_numpy(s), thr, pre_max_size=pre, post_max_size=post)
        assert got.dtype == torch.int64 and np.array_equal(got.numpy(), oracle.nms_gpu_oracle(b, s, thr, pre, post))
    got = amd.nms_normal_gpu(torch.from_numpy(b), torch.from_numpy(s), 0.3)
    assert np.array_equal(got.numpy(), oracle.nms_gpu_oracle(b, s, 0.3, normal=True))


@pytest.mark.parametrize('name', ['waymo0', 'waymo2', 'nuscenes', 'pvrcnn', 'origin', 'kitti'])
def test_nms_on_cpu_tensors_against_the_reference_derived_keep_lists(name):
    from test_nms_ref_iou import MAX_UNCERTAIN
    g = nms_ref.load(name)
    keep = amd.nms_gpu(torch.from_numpy(g['boxes']), torch.from_numpy(g['scores']), g['thr'], pre_max_size=g['pre']).numpy()
    n_unc, bad, total = nms_ref.compare_keep(g, keep)
    assert bad == 0 and total <= n_unc <= MAX_UNCERTAIN[name]


def test_padded_form_empty_inputs_nan_scores_and_thread_count():
    b, s = nms_boxes(300, seed=4)
    s[::37] = np.nan; s[5] = np.inf; s[9] = -np.inf
    bt, st = torch.from_numpy(b), torch.from_numpy(s)
    order = torch.sort(st, descending=True, stable=True)[1].numpy()
    want = order[oracle.nms_bev(b[order], 0.3)]
    assert np.array_equal(amd.nms_gpu(bt, st, 0.3).numpy(), want)                # NaN first, as torch.sort orders them
    keep, num = amd.nms_gpu(bt, st, 0.3, post_max_size=50, padded=True)
    assert keep.shape == (50,) and int(num) == min(50, len(want)) and np.array_equal(keep[:int(num)].numpy(), want[:50])
    assert amd.nms_gpu(torch.zeros(0, 5), torch.zeros(0), 0.5).shape == (0,)
    assert amd.iou_bev(torch.zeros(0, 7), torch.zeros(3, 7)).shape == (0, 3)
    d, g = eval_boxes(400, 5), eval_boxes(300, 6)
    keep_threads = torch.get_num_threads()
    outs = []
    try:
        for nt in (1, 5):
            torch.set_num_threads(nt)
            outs.append(amd.iou_3d(torch.from_numpy(d), torch.from_numpy(g)))
    finally:
        torch.set_num_threads(keep_threads)
    assert torch.equal(outs[0], outs[1])
    with pytest.raises(RuntimeError, match='different devices|no CPU path'):
        amd.nms_gpu_batched(bt, st.reshape(1, -1), 0.3)


def test_c_abi_argument_validation_of_the_cpu_entries():
    lib = amd.load_library()
    a = eval_boxes(4, 1); out = np.zeros((4, 4), np.float32)
    vp = lambda x: x.ctypes.data   # noqa: E731
    assert lib.riou_eval_bev_cpu(vp(a), 4, vp(a), 4, vp(out), 1) == 0 and np.allclose(np.diag(out), 1.0, atol=1e-5)
    assert lib.riou_eval_bev_cpu(None, 4, vp(a), 4, vp(out), 1) == 10001 and lib.riou_eval_3d_cpu(vp(a), -1, vp(a), 4, 0.5, vp(out), 1) == 10001
    assert lib.riou_eval_bev_cpu(None, 0, vp(a), 4, None, 1) == 0
    assert lib.riou_eval_trans_bev_cpu(vp(a), 4, 1, vp(a), 4, 7, vp(out), 1) == 10001      # fewer than two columns
    keep = np.zeros(4, np.int64); num = np.full(1, 9, np.int64)
    b, _ = nms_boxes(4, seed=1)
    assert lib.rnms_bev_cpu(vp(b), 4, 0.5, vp(keep), vp(num)) == 0 and 1 <= num[0] <= 4
    assert lib.rnms_bev_cpu(None, 0, 0.5, None, vp(num)) == 0 and num[0] == 0
    assert lib.rnms_bev_cpu(vp(b), 4, 0.5, vp(keep), None) == 10001 and lib.rnms_normal_bev_cpu(None, 4, 0.5, vp(keep), vp(num)) == 10001


def test_match_coco_on_cpu_against_the_compiled_reference_fixtures():
    """The matcher's `_cpu` twin (eval_match_coco_cpu) on numpy / CPU-tensor inputs vs the vectors of the reference's own compiled
    matcher.cpp (tests/golden/match_coco.npz): the reference's matcher is CPU code, so an evaluation without a GPU must work."""
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'match_coco.npz'))
    for name in g['cases']:
        args = [torch.from_numpy(np.ascontiguousarray(g[f'{name}.{k}'])) for k in ('cost', 'thrs', 'ignore', 'crowd')]
        got = amd.match_coco(*args)
        assert got.dtype == torch.int32 and got.device.type == 'cpu' and np.array_equal(got.numpy(), g[f'{name}.matched']), name
        if not torch.cuda.is_available():   # numpy in (what the reference's evaluation passes): the same path on a GPU-less machine
            assert np.array_equal(amd.match_coco(*[g[f'{name}.{k}'] for k in ('cost', 'thrs', 'ignore', 'crowd')]).numpy(), g[f'{name}.matched'])


@pytest.mark.parametrize('D,G,T', [(3000, 50, 10), (200, 5000, 3), (1, 1, 1), (64, 64, 4), (500, 65, 2), (7, 0, 3), (0, 5, 2)])
def test_match_coco_cpu_twin_vs_oracle_ties_signed_zeros_empty_sides_and_threads(D, G, T):
    rng = np.random.default_rng(D + G)
    cost = np.round(-rng.uniform(0, 1, (D, G)), 2).astype(np.float32)        # rounded: plenty of exact ties
    cost[rng.uniform(0, 1, (D, G)) < 0.01] = -0.0                             # signed zeros compare equal to +0
    if D * G > 100:
        cost[rng.uniform(0, 1, (D, G)) < 0.005] = np.nan                      # NaN never matches
    thrs = -np.linspace(0.0, 0.9, T).astype(np.float32)
    ign = rng.uniform(0, 1, G) < 0.3
    crowd = rng.uniform(0, 1, G) < 0.1
    want = oracle.match_coco(cost, thrs, ign, crowd) if D and G else np.full((T, D), -1, np.int32)
    t = [torch.from_numpy(x) for x in (cost, thrs, ign, crowd)]
    got = amd.match_coco(*t)
    assert got.shape == (T, D) and np.array_equal(got.numpy(), want)
    lib = amd.load_library()
    out = np.empty((T, D), np.int32)
    c8, i8 = np.ascontiguousarray(crowd.astype(np.uint8)), np.ascontiguousarray(ign.astype(np.uint8))
    for threads in (1, 3):   # thresholds are independent: the team size cannot show in the result
        out.fill(7)
        assert lib.eval_match_coco_cpu(cost.ctypes.data, thrs.ctypes.data, i8.ctypes.data, c8.ctypes.data, D, G, T, out.ctypes.data, threads) == 0
        assert np.array_equal(out, want) or D == 0


def test_match_coco_cpu_entry_argument_rules():
    lib = amd.load_library()
    one = np.zeros(4, np.float32)
    flags = np.zeros(4, np.uint8)
    out = np.zeros(4, np.int32)
    call = lambda *a: lib.eval_match_coco_cpu(*a)
    assert call(one.ctypes.data, one.ctypes.data, flags.ctypes.data, flags.ctypes.data, -1, 1, 1, out.ctypes.data, 1) == 10001
    assert call(None, one.ctypes.data, flags.ctypes.data, flags.ctypes.data, 1, 1, 1, out.ctypes.data, 1) == 10001
    assert call(one.ctypes.data, one.ctypes.data, flags.ctypes.data, flags.ctypes.data, 1, 1, 1, None, 1) == 10001
    assert call(one.ctypes.data, one.ctypes.data, flags.ctypes.data, flags.ctypes.data, 1, (1 << 20) + 1, 1, out.ctypes.data, 1) == 10002
    assert call(None, one.ctypes.data, None, None, 0, 0, 0, None, 1) == 0      # nothing to do
