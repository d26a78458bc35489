"""GPU parity tests for the rotated NMS / IoU kernels through the C ABI wrappers.
NMS keep indices: BIT-EXACT against the CPU oracle (same fp32 operation sequence).
ops/eval IoU: against golden matrices from the compiled reference (tests/golden/riou_eval.npz)."""
import os

import numpy as np
import pytest
import torch

import oracle
from rbox_inputs import eval_boxes, nms_boxes

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'riou_eval.npz')


@pytest.fixture(scope='module')
def amd():
    import mmdet3d_gaussian_amd as m
    assert torch.cuda.is_available()
    m.load_library()
    return m


@pytest.mark.parametrize('n,thr,clutter', [(1, 0.25, True), (2, 0.25, True), (63, 0.25, True), (64, 0.01, True),
                                           (65, 0.25, True), (129, 0.5, False), (1000, 0.2, True), (4096, 0.25, True),
                                           (4096, 0.01, False), (9000, 0.7, True), (20000, 0.5, True)])
def test_nms_keep_indices_bit_exact(amd, host_glue, n, thr, clutter):
    boxes, scores = nms_boxes(n, seed=n + int(thr * 100), clutter=clutter)
    want = oracle.nms_gpu_oracle(boxes, scores, thr)
    got = amd.nms_gpu(torch.from_numpy(boxes).cuda(), torch.from_numpy(scores).cuda(), thr)
    assert got.dtype == torch.int64
    np.testing.assert_array_equal(got.cpu().numpy(), want)


@pytest.mark.parametrize('n,thr', [(8448, 0.6), (8449, 0.6), (12288, 0.3), (16384, 0.7)])
def test_nms_two_level_scan_boundaries_bit_exact(amd, host_glue, n, thr):
    """n > 8448 takes the two-level scan (super-blocks of 4096 boxes resolved in turn, kept rows spread to the right by a
    chip-wide kernel): the last single-level size, the first two-level size, an exact multiple of the super-block and
    the rank-path maximum, against the CPU restatement."""
    boxes, scores = nms_boxes(n, seed=n, clutter=True)
    want = oracle.nms_gpu_oracle(boxes, scores, thr)
    got = amd.nms_gpu(torch.from_numpy(boxes).cuda(), torch.from_numpy(scores).cuda(), thr)
    np.testing.assert_array_equal(got.cpu().numpy(), want)


@pytest.mark.parametrize('mode', ['rot', 'normal'])
def test_batched_nms_ragged_groups_across_super_blocks(amd, mode):
    """Three groups over one 11 000-box array with 11 000 / 4 500 / 8 700 valid boxes: the group sizes are only known on
    the device, so every group runs the two-level launch sequence and leaves it where ITS boxes end; each keep list
    equals the single call on the compacted group (which takes the one- or two-level scan by its own size)."""
    n = 11000
    boxes, _ = nms_boxes(n, seed=33, clutter=True)
    rng = np.random.default_rng(9)
    scores = rng.uniform(0, 1, (3, n)).astype(np.float32)
    valid = np.ones((3, n), bool)
    valid[1, rng.permutation(n)[:n - 4500]] = False
    valid[2, rng.permutation(n)[:n - 8700]] = False
    b = torch.from_numpy(boxes).cuda(); s = torch.from_numpy(scores).cuda(); v = torch.from_numpy(valid).cuda()
    normal = mode == 'normal'
    got = amd.nms_gpu_batched(b, s, [0.6, 0.3, 0.7], v, normal=normal)
    single = amd.nms_normal_gpu if normal else (lambda bb, ss, t: amd.nms_gpu(bb, ss, t))
    for g, thr in enumerate([0.6, 0.3, 0.7]):
        one = single(b[v[g]], s[g][v[g]], thr)
        assert torch.equal(got[g], v[g].nonzero().reshape(-1)[one]), g
    idx = np.nonzero(valid[1])[0]
    want = idx[oracle.nms_gpu_oracle(boxes[idx], scores[1, idx], 0.3, normal=normal)]
    assert np.array_equal(got[1].cpu().numpy(), want)


@pytest.mark.parametrize('mode', ['rot', 'normal'])
def test_batched_nms_empty_group_beyond_the_single_level_size(amd, mode):
    """with more than 8448 boxes per group the two-level scan runs, and a group WITHOUT any valid box never
    reaches a resolver wave — its count must still come out 0 (the count buffer is allocated uninitialised).  Poison
    the caching allocator first so that a stale count would be visible."""
    n = 9500
    boxes, _ = nms_boxes(n, seed=71, clutter=True)
    rng = np.random.default_rng(3)
    scores = rng.uniform(0, 1, (4, n)).astype(np.float32)
    valid = np.ones((4, n), bool)
    valid[1] = False                                   # empty group between two full ones
    valid[3, rng.permutation(n)[:n - 100]] = False     # a small group, far below one super-block
    b = torch.from_numpy(boxes).cuda(); s = torch.from_numpy(scores).cuda(); v = torch.from_numpy(valid).cuda()
    normal = mode == 'normal'
    single = amd.nms_normal_gpu if normal else (lambda bb, ss, t: amd.nms_gpu(bb, ss, t))
    for _ in range(2):
        junk = [torch.full((k,), 0x7f7f7f7f7f7f7f7f, dtype=torch.int64, device='cuda') for k in (4, 8, 64, 512, 4096)]
        del junk
        got = amd.nms_gpu_batched(b, s, 0.5, v, normal=normal)
        assert got[1].numel() == 0
        for g in (0, 2, 3):
            assert torch.equal(got[g], v[g].nonzero().reshape(-1)[single(b[v[g]], s[g][v[g]], 0.5)]), g
    # the same through the list form: an empty entry among entries above the single-level size
    bl = [b, b[:0], b[:9000]]
    sl = [s[0], s[0][:0], s[2][:9000]]
    for _ in range(2):
        junk = [torch.full((k,), 0x7f7f7f7f7f7f7f7f, dtype=torch.int64, device='cuda') for k in (4, 8, 64, 512, 4096)]
        del junk
        multi = amd.nms_gpu_multi(bl, sl, 0.5, normal=normal)
        assert multi[1].numel() == 0
        assert torch.equal(multi[0], single(bl[0], sl[0], 0.5))
        assert torch.equal(multi[2], single(bl[2], sl[2], 0.5))


def test_nms_randomised_sweep_bit_exact(host_glue, amd):
    """40 seeded random problems (size, threshold, clutter, extent, pre/post cuts drawn at random): keep indices equal to
    the CPU oracle in every one — the mask compaction, the register clipping path, the score ranking and the scan are
    all exercised at sizes that are not multiples of anything."""
    rng = np.random.default_rng(2024)
    for case in range(40):
        n = int(rng.integers(1, 3200))
        thr = float(rng.choice([0.01, 0.1, 0.2, 0.25, 0.5, 0.7, 0.9]))
        clutter = bool(rng.integers(0, 2))
        extent = float(rng.choice([5.0, 30.0, 74.88]))
        boxes, scores = nms_boxes(n, seed=1000 + case, extent=extent, clutter=clutter)
        if rng.random() < 0.3:
            scores = np.round(scores, 1)                       # heavy ties
        pre = None if rng.random() < 0.5 else int(rng.integers(1, n + 1))
        post = None if rng.random() < 0.5 else int(rng.integers(1, 600))
        want = oracle.nms_gpu_oracle(boxes, scores, thr, pre, post)
        got = amd.nms_gpu(torch.from_numpy(boxes).cuda(), torch.from_numpy(scores).cuda(), thr, pre_max_size=pre,
                          post_max_size=post)
        assert np.array_equal(got.cpu().numpy(), want), (case, n, thr, clutter, extent, pre, post)


def test_nms_pre_post_cuts_and_alias(host_glue, amd):
    boxes, scores = nms_boxes(3000, seed=1)
    b, s = torch.from_numpy(boxes).cuda(), torch.from_numpy(scores).cuda()
    want = oracle.nms_gpu_oracle(boxes, scores, 0.2, pre_max_size=1000, post_max_size=83)   # nuScenes test_cfg
    got = amd.nms_gpu(b, s, thresh=0.2, pre_max_size=1000, post_max_size=83)
    np.testing.assert_array_equal(got.cpu().numpy(), want)
    got2 = amd.nms_gpu(b, s, 0.2, pre_maxsize=1000, post_max_size=83)
    np.testing.assert_array_equal(got2.cpu().numpy(), want)


def test_nms_empty_and_identical_boxes(host_glue, amd):
    e = amd.nms_gpu(torch.zeros(0, 5).cuda(), torch.zeros(0).cuda(), 0.5)
    assert e.shape == (0,) and e.dtype == torch.int64
    b = torch.tensor([[0, 0, 2, 1, 0.3]]).repeat(200, 1).cuda()
    s = torch.linspace(0, 1, 200).cuda()
    k = amd.nms_gpu(b, s, 0.5)
    assert k.tolist() == [199]


def test_nms_mask_words_bit_exact(amd):
    """Not only the keep list: every 64-bit suppression word the scan can read equals the oracle's."""
    import ctypes
    lib = amd.load_library()
    n, thr = 700, 0.25
    boxes, scores = nms_boxes(n, seed=77)
    order = np.argsort(-scores, kind='stable')
    bs = np.ascontiguousarray(boxes[order])
    want = oracle.nms_mask(bs, thr)
    d = torch.from_numpy(bs).cuda()
    keep = torch.empty(n, dtype=torch.int64, device='cuda'); num = torch.zeros(1, dtype=torch.int64, device='cuda')
    wsb = lib.rnms_workspace_bytes(n)
    ws = torch.zeros(wsb, dtype=torch.uint8, device='cuda')
    vp = lambda t: ctypes.c_void_p(t.data_ptr())
    assert lib.rnms_bev(vp(d), n, thr, vp(keep), vp(num), vp(ws), None) == 0
    torch.cuda.synchronize()
    cb = (n + 63) // 64
    off = (n * 64 + 255) // 256 * 256
    mask = ws[off:off + n * cb * 8].cpu().numpy().view(np.uint64).reshape(n, cb)
    rows = np.arange(n)[:, None] // 64
    cols = np.arange(cb)[None, :]
    upper = cols >= rows                       # the kernel only produces the blocks the scan reads
    np.testing.assert_array_equal(mask[upper], want[upper])


def _mask_words(amd, bs, thr):
    import ctypes
    lib = amd.load_library()
    n = bs.shape[0]
    d = torch.from_numpy(bs).cuda()
    keep = torch.empty(n, dtype=torch.int64, device='cuda'); num = torch.zeros(1, dtype=torch.int64, device='cuda')
    ws = torch.zeros(lib.rnms_workspace_bytes(n), dtype=torch.uint8, device='cuda')
    vp = lambda t: ctypes.c_void_p(t.data_ptr())
    assert lib.rnms_bev(vp(d), n, thr, vp(keep), vp(num), vp(ws), None) == 0
    torch.cuda.synchronize()
    cb = (n + 63) // 64
    off = (n * 64 + 255) // 256 * 256
    mask = ws[off:off + n * cb * 8].cpu().numpy().view(np.uint64).reshape(n, cb)
    upper = np.arange(cb)[None, :] >= np.arange(n)[:, None] // 64
    return mask, upper, keep[:int(num.item())].cpu().numpy()


@pytest.mark.parametrize('n,thr,kind', [
    (2900, 0.25, 'clutter'),    # 64 rows per wave (>= 1024 block pairs), several 16-row chunks per wave
    (1500, 0.3, 'pile'),        # hundreds of boxes on one spot: chunks overflow 64 candidates, remainders are carried
    (200, 0.5, 'same'),         # every pair is a candidate and a hit: full queue in every chunk
    (450, -0.5, 'clutter'),     # negative threshold: IoU 0 of far-apart pairs is a hit too (all-pairs mode)
    (450, float('nan'), 'clutter'),  # NaN threshold: nothing is ever suppressed
    # from 768 boxes on the QUEUED form (circle-test kernel -> candidate queue in HBM -> clipping kernel, csrc/rbox.hip):
    (1200, 0.4, 'same'),        # every pair a candidate: 719 400 > 128 per box -> most block pairs take the overflow list
    (3000, 0.25, 'pile'),       # 1500 boxes on one spot: queue filled to capacity AND overflowed block pairs in one call
    (1500, -0.5, 'clutter'),    # negative threshold on the queued form: every block pair is routed to the all-pairs path
    (1500, float('nan'), 'clutter'),
    (4096, 0.25, 'clutter'),    # BASELINE configs[4] size: nothing overflows
    (833, 0.6, 'clutter'),      # ragged last block, just above the switch-over
    (767, 0.6, 'clutter'),      # ... and the largest set the one-kernel form still takes
])
def test_nms_compacted_mask_paths_bit_exact(amd, host_glue, n, thr, kind):
    """The rotated mask kernels pack the pairs that survive the bounding-circle test densely before clipping them — inside one
    wave below 768 boxes (nms_mask_compact_kernel), through a queue in HBM between two kernels above: queue overflow / carry,
    8-64 rows per wave, the overflow list, and the thresholds for which a far-apart pair is NOT a non-hit must all give the
    oracle's words."""
    boxes, scores = nms_boxes(n, seed=n, clutter=(kind != 'same'))
    if kind == 'pile':
        boxes[: n // 2] = boxes[0] + np.random.default_rng(5).normal(0, 0.05, (n // 2, 5)).astype(np.float32)
    if kind == 'same':
        boxes[:] = boxes[0]
    order = np.argsort(-scores, kind='stable')
    bs = np.ascontiguousarray(boxes[order])
    want = oracle.nms_mask(bs, thr)
    mask, upper, keep = _mask_words(amd, bs, thr)
    np.testing.assert_array_equal(mask[upper], want[upper])
    want_keep = oracle.nms_gpu_oracle(boxes, scores, thr)
    got = amd.nms_gpu(torch.from_numpy(boxes).cuda(), torch.from_numpy(scores).cuda(), thr)
    np.testing.assert_array_equal(got.cpu().numpy(), want_keep)


@pytest.mark.parametrize('n,pre,normal', [(1, None, False), (63, None, False), (64, 10, False), (65, None, True), (1000, 500, False),
                                          (2049, None, False), (4096, None, False), (4096, 1000, True), (3000, 0, False),
                                          (9000, 4096, False), (16384, 2000, False), (16385, 2000, False)])
def test_nms_fused_score_sort_equals_torch_sort_path(amd, host_glue, n, pre, normal):
    """Up to 16384 candidates the library orders the scores itself (rank by counting, prep scattered to the rank): the kept
    indices must equal those of the torch.sort(descending, stable) + rnms_*_ordered path, for ties, +-0, infinities and
    NaN scores too; one box more takes the torch.sort path inside nms_gpu."""
    import ctypes
    lib = amd.load_library()
    boxes, scores = nms_boxes(n, seed=3 * n + 1)
    rng = np.random.default_rng(n)
    scores = np.round(scores, 2)                               # many exact ties
    if n >= 64:
        scores[rng.integers(0, n, 5)] = 0.0
        scores[rng.integers(0, n, 5)] = -0.0
        scores[rng.integers(0, n, 3)] = np.inf
        scores[rng.integers(0, n, 3)] = -np.inf
        scores[rng.integers(0, n, 4)] = np.nan
    b, s = torch.from_numpy(boxes).cuda(), torch.from_numpy(scores).cuda()
    fn = amd.nms_normal_gpu if normal else amd.nms_gpu
    got = fn(b, s, 0.3) if normal else fn(b, s, 0.3, pre_max_size=pre)
    # reference path: torch's stable descending sort, then the ordered entry point
    order = s.sort(dim=0, descending=True, stable=True)[1]
    if pre is not None and not normal:
        order = order[:pre]
    order = order.contiguous()
    m = order.shape[0]
    if m == 0:
        assert got.numel() == 0
        return
    keep = torch.empty(m, dtype=torch.int64, device='cuda'); num = torch.zeros(1, dtype=torch.int64, device='cuda')
    ws = torch.empty(lib.rnms_workspace_bytes(m), dtype=torch.uint8, device='cuda')
    vp = lambda t: ctypes.c_void_p(t.data_ptr())
    f2 = lib.rnms_normal_bev_ordered if normal else lib.rnms_bev_ordered
    assert f2(vp(b), vp(order), m, 0.3, vp(keep), vp(num), vp(ws), None) == 0
    want = keep[:int(num.item())]
    assert torch.equal(got, want)


def test_nms_float64_scores_keep_torch_sort(amd):
    boxes, scores = nms_boxes(500, seed=9)
    s64 = torch.from_numpy(scores.astype(np.float64)).cuda() + 1e-12 * torch.arange(500, device='cuda', dtype=torch.float64)
    got = amd.nms_gpu(torch.from_numpy(boxes).cuda(), s64, 0.25)
    want = oracle.nms_gpu_oracle(boxes, s64.cpu().numpy(), 0.25)
    np.testing.assert_array_equal(got.cpu().numpy(), want)


def test_nms_normal(host_glue, amd):
    boxes, scores = nms_boxes(2000, seed=4)
    want = oracle.nms_gpu_oracle(boxes, scores, 0.3, normal=True)
    got = amd.nms_normal_gpu(torch.from_numpy(boxes).cuda(), torch.from_numpy(scores).cuda(), 0.3)
    np.testing.assert_array_equal(got.cpu().numpy(), want)


def test_boxes_iou_bev_matches_oracle(amd):
    a, _ = nms_boxes(300, seed=8, extent=10)
    b, _ = nms_boxes(200, seed=9, extent=10)
    got = amd.boxes_iou_bev(torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()).cpu().numpy()
    np.testing.assert_array_equal(got, oracle.iou_bev_xyxyr(a, b))


@pytest.mark.parametrize('fam', ['shift', 'dense', 'degen', 'ragged'])
def test_eval_iou_matches_reference_golden(amd, fam):
    """(D,7)x(G,7) IoU matrices vs the reference's own C++ (compiled in the build container).
    fp32 tolerance 1e-5 abs on IoU (the only non-identical arithmetic is the device's fp64 cos/sin)."""
    g = np.load(GOLD)
    det, gt = torch.from_numpy(g[fam + '.det']).cuda(), torch.from_numpy(g[fam + '.gt']).cuda()
    np.testing.assert_allclose(amd.iou_bev(det, gt).cpu().numpy(), g[fam + '.iou_bev'], atol=1e-5, rtol=0)
    np.testing.assert_allclose(amd.iou_3d(det, gt, 0.5).cpu().numpy(), g[fam + '.iou_3d'], atol=1e-5, rtol=0)
    np.testing.assert_allclose(amd.iou_3d(det, gt, 0.0).cpu().numpy(), g[fam + '.iou_3d_z0'], atol=1e-5, rtol=0)


def test_eval_iou_large_matches_oracle(amd):
    d, g = eval_boxes(700, 1, spread=30), eval_boxes(500, 2, spread=30)
    got = amd.iou_3d(torch.from_numpy(d).cuda(), torch.from_numpy(g).cuda()).cpu().numpy()
    np.testing.assert_allclose(got, oracle.eval_iou_3d(d, g), atol=1e-5, rtol=0)
    assert amd.iou_bev(torch.zeros(0, 7).cuda(), torch.from_numpy(g).cuda()).shape == (0, 500)


def test_nms_degenerate_geometry_bit_exact(host_glue, amd):
    """Zero-area, identical, shared-edge, nested, axis-aligned, huge and NaN boxes: no hang, no fault, and the same keep
    list as the oracle (the fp32 operation sequence is the same on both sides, also for garbage)."""
    rng = np.random.default_rng(3)
    n = 640
    boxes, scores = nms_boxes(n, seed=12)
    boxes[0:40, 2] = boxes[0:40, 0]                       # zero width
    boxes[40:80] = boxes[40]                              # identical copies
    boxes[80:120, 4] = 0.0                                # axis aligned
    boxes[120:160] = boxes[80:120]; boxes[120:160, 0] += (boxes[80:120, 2] - boxes[80:120, 0]); boxes[120:160, 2] += (boxes[80:120, 2] - boxes[80:120, 0])  # shared edge
    c = (boxes[160:200, :2] + boxes[160:200, 2:4]) / 2
    boxes[200:240, :2] = c - 0.1; boxes[200:240, 2:4] = c + 0.1; boxes[200:240, 4] = boxes[160:200, 4]            # nested
    boxes[240:250, :4] *= 1e4                             # far away / huge
    boxes[250:255, 1] = np.nan                            # NaN coordinate
    boxes[255:260, 4] = np.inf                            # inf angle
    scores = rng.uniform(0, 1, n).astype(np.float32)
    for thr in (0.0, 0.25, 0.9):
        want = oracle.nms_gpu_oracle(boxes, scores, thr)
        got = amd.nms_gpu(torch.from_numpy(boxes).cuda(), torch.from_numpy(scores).cuda(), thr)
        np.testing.assert_array_equal(got.cpu().numpy(), want)


@pytest.mark.parametrize('normal', [False, True])
def test_batched_nms_equals_per_group_calls_and_oracle(amd, normal):
    """G groups (classes) over one box array, group sizes read on the device: each group's keep list must equal both the
    single-call nms_gpu on the compacted group and the CPU oracle, bit for bit — incl. an empty group, a full group,
    a 1-box group and per-group thresholds."""
    n, G = 1500, 5
    boxes, _ = nms_boxes(n, seed=21)
    rng = np.random.default_rng(5)
    scores = rng.uniform(0, 1, (G, n)).astype(np.float32)
    valid = scores >= np.array([0.3, 2.0, 0.0, 0.9, 0.5], np.float32)[:, None]      # group 1 empty, group 2 full
    valid[3] = False; valid[3, 77] = True                                          # group 3: a single box
    thr = [0.25, 0.5, 0.1, 0.7, 0.01]
    b = torch.from_numpy(boxes).cuda(); s = torch.from_numpy(scores).cuda(); v = torch.from_numpy(valid).cuda()
    got = amd.nms_gpu_batched(b, s, thr, v, normal=normal)
    assert len(got) == G
    single = amd.nms_normal_gpu if normal else amd.nms_gpu
    for g in range(G):
        idx = np.nonzero(valid[g])[0]
        want = idx[oracle.nms_gpu_oracle(boxes[idx], scores[g, idx], thr[g], normal=normal)] if len(idx) else np.zeros(0, np.int64)
        assert np.array_equal(got[g].cpu().numpy(), want), g
        if len(idx):
            one = single(b[v[g]], s[g][v[g]], thr[g])
            assert torch.equal(got[g], v[g].nonzero().reshape(-1)[one])
    # pre / post cuts apply per group; no mask = every box takes part
    cut = amd.nms_gpu_batched(b, s, 0.25, None, pre_max_size=700, post_max_size=50, normal=normal)
    for g in range(G):
        want = oracle.nms_gpu_oracle(boxes, scores[g], 0.25, 700, 50, normal=normal)
        assert np.array_equal(cut[g].cpu().numpy(), want)


@pytest.mark.parametrize('mode', ['rot', 'normal', 'circle'])
def test_library_score_order_equals_torch_sort_fallback(amd, mode):
    """nms_gpu / nms_gpu_batched take the score order inside the library up to 16384 boxes and fall back to torch.sort
    above: both routes must give the same kept indices (ties, invalid boxes, pre / post cuts, per-group thresholds)."""
    import sys
    iou3d = sys.modules[amd.nms_gpu.__module__]
    n, G = 2100, 4
    boxes, _ = nms_boxes(n, seed=33)
    rng = np.random.default_rng(8)
    scores = np.round(rng.uniform(0, 1, (G, n)), 2).astype(np.float32)             # many ties
    valid = scores >= np.array([0.2, 0.0, 1.5, 0.6], np.float32)[:, None]           # group 2 empty
    b = torch.from_numpy(boxes[:, :2].copy() if mode == 'circle' else boxes).cuda()
    s, v = torch.from_numpy(scores).cuda(), torch.from_numpy(valid).cuda()
    thr = [1.0, 4.0, 0.5, 2.0] if mode == 'circle' else [0.25, 0.5, 0.1, 0.7]
    kw = dict(normal=(mode == 'normal'), circle=(mode == 'circle'))

    def run():
        a = amd.nms_gpu_batched(b, s, thr, v, pre_max_size=900, post_max_size=120, **kw)
        c = amd.nms_gpu_batched(b, s, thr, None, **kw)
        d = None if mode == 'circle' else (amd.nms_normal_gpu(b, s[0], 0.3) if mode == 'normal'
                                           else amd.nms_gpu(b, s[0], 0.3, pre_max_size=1000, post_max_size=200))
        return a, c, d
    lib = amd.load_library()
    assert iou3d._scored_max(lib) == 16384
    fast = run()
    saved = iou3d._SCORED_MAX
    iou3d._SCORED_MAX = 0                         # force the torch.sort route
    try:
        slow = run()
    finally:
        iou3d._SCORED_MAX = saved
    for x, y in zip(fast[0] + fast[1], slow[0] + slow[1]):
        assert torch.equal(x, y)
    if fast[2] is not None:
        assert torch.equal(fast[2], slow[2])
    assert fast[0][2].numel() == 0 and all(k.numel() <= 120 for k in fast[0])


def test_nms_gpu_multi_equals_the_per_entry_loop(amd):
    """B samples x T tasks of CenterHeadRev.get_bboxes (gd_centerpoint_head.py:233-345) as one batched call: every entry's
    keep list equals nms_gpu on that entry alone — incl. empty entries and per-entry thresholds."""
    rng = np.random.default_rng(17)
    sizes = [500, 0, 37, 500, 1, 264, 64, 129, 0, 480, 333, 12]          # 2 samples x 6 tasks
    bl, sl = [], []
    for k, n in enumerate(sizes):
        b, s = nms_boxes(max(n, 1), seed=400 + k, extent=20.0)
        bl.append(torch.from_numpy(b[:n]).cuda()); sl.append(torch.from_numpy(np.round(s[:n], 2)).cuda())
    thr = [float(rng.choice([0.1, 0.2, 0.5])) for _ in sizes]
    got = amd.nms_gpu_multi(bl, sl, thr, pre_max_size=400, post_max_size=83)
    assert len(got) == len(sizes)
    for k, n in enumerate(sizes):
        want = amd.nms_gpu(bl[k], sl[k], thr[k], pre_max_size=400, post_max_size=83) if n else torch.zeros(0, dtype=torch.int64).cuda()
        assert torch.equal(got[k], want), k
    assert amd.nms_gpu_multi([], [], 0.2) == []
    # a negative pre_max_size is a slice bound (order[:pre]): n_g + pre boxes survive per entry, in every path
    neg = amd.nms_gpu_multi(bl, sl, thr, pre_max_size=-10)
    for k, n in enumerate(sizes):
        want = amd.nms_gpu(bl[k], sl[k], thr[k], pre_max_size=-10) if n else torch.zeros(0, dtype=torch.int64).cuda()
        assert torch.equal(neg[k], want), k
    allb = torch.cat(bl); N = allb.shape[0]
    alls = torch.zeros(2, N).cuda(); allv = torch.zeros(2, N, dtype=torch.bool).cuda()
    alls[0, :500] = sl[0]; allv[0, :500] = True
    alls[1, 537:1037] = sl[3]; allv[1, 537:1037] = True
    bat = amd.nms_gpu_batched(allb, alls, [thr[0], thr[3]], allv, pre_max_size=-10)
    assert torch.equal(bat[0], amd.nms_gpu(bl[0], sl[0], thr[0], pre_max_size=-10))
    assert torch.equal(bat[1] - 537, amd.nms_gpu(bl[3], sl[3], thr[3], pre_max_size=-10))


def test_nms_gpu_multi_many_entries_stay_segmented(amd):
    """48 entries x 500 boxes (8 samples x 6 tasks): each entry ranks only its own slice; results equal the loop."""
    bl, sl = [], []
    for k in range(48):
        b, s = nms_boxes(500 - 7 * (k % 5), seed=900 + k, extent=25.0)
        bl.append(torch.from_numpy(b).cuda()); sl.append(torch.from_numpy(s).cuda())
    got = amd.nms_gpu_multi(bl, sl, 0.2, pre_max_size=1000, post_max_size=83)
    for k in range(48):
        assert torch.equal(got[k], amd.nms_gpu(bl[k], sl[k], 0.2, pre_max_size=1000, post_max_size=83)), k


def test_multi_class_nms_matches_reference_loop(amd):
    """pvrcnn_bbox_head.py:438-480 restated as the loop it is (score mask -> nonzero -> nms -> original_idxs[selected]
    -> cat) on the oracle vs the one-shot batched call."""
    n, C = 2000, 3
    boxes, _ = nms_boxes(n, seed=8)
    rng = np.random.default_rng(9)
    probs = rng.uniform(0, 1, (n, C)).astype(np.float32)
    score_thr, nms_thr = [0.6, 0.1, 0.99], [0.1, 0.3, 0.5]
    want = []
    for k in range(C):
        m = probs[:, k] >= np.float32(score_thr[k])
        if m.sum() > 0:
            idx = np.nonzero(m)[0]
            sel = oracle.nms_gpu_oracle(boxes[idx], probs[idx, k], nms_thr[k])
            if len(sel):
                want.append(idx[sel])
    want = np.concatenate(want)
    got = amd.multi_class_nms(torch.from_numpy(probs).cuda(), torch.from_numpy(boxes).cuda(), score_thr, nms_thr)
    assert np.array_equal(got.cpu().numpy(), want)
    none = amd.multi_class_nms(torch.from_numpy(probs).cuda(), torch.from_numpy(boxes).cuda(), 2.0, 0.1)
    assert isinstance(none, list) and none == []


def test_box3d_multiclass_nms_matches_the_restated_loop(amd):
    """mmdet3d's box3d_multiclass_nms (what the inherited Anchor3DHead.get_bboxes_single runs; third party, restated) as the loop it
    is — strict score threshold, nms per class, concat, max_num cut by score — on the CPU oracle vs the one batched call.
    BASELINE configs[4]'s shape: 4096 candidates, 3 classes + padding column, thr 0.25 / score_thr 0.1 / max_num 500"""
    n, C = 4096, 3
    boxes5, _ = nms_boxes(n, seed=41)
    rng = np.random.default_rng(42)
    scores = np.concatenate([rng.uniform(0, 1, (n, C)).astype(np.float32) ** 3, np.zeros((n, 1), np.float32)], 1)
    box7 = rng.uniform(-1, 1, (n, 7)).astype(np.float32)
    dirs = rng.integers(0, 2, n)
    for score_thr, nms_thr, max_num in ((0.1, 0.25, 500), (0.5, 0.01, 100), (0.999999, 0.25, 500)):
        wb, ws, wl, wd = [], [], [], []
        for i in range(C):
            m = scores[:, i] > np.float32(score_thr)
            if not m.any():
                continue
            idx = np.nonzero(m)[0]
            sel = idx[oracle.nms_gpu_oracle(boxes5[idx], scores[idx, i], nms_thr)]
            wb.append(box7[sel]); ws.append(scores[sel, i]); wl.append(np.full(len(sel), i)); wd.append(dirs[sel])
        if wb:
            wb, ws, wl, wd = np.concatenate(wb), np.concatenate(ws), np.concatenate(wl), np.concatenate(wd)
            if wb.shape[0] > max_num:
                inds = np.argsort(-ws, kind='stable')[:max_num]
                wb, ws, wl, wd = wb[inds], ws[inds], wl[inds], wd[inds]
        else:
            wb, ws, wl, wd = np.zeros((0, 7), np.float32), np.zeros(0, np.float32), np.zeros(0, np.int64), np.zeros(0, np.int64)
        cfg = dict(use_rotate_nms=True, nms_thr=nms_thr)
        gb, gs, gl, gd = amd.box3d_multiclass_nms(torch.from_numpy(box7).cuda(), torch.from_numpy(boxes5).cuda(), torch.from_numpy(scores).cuda(),
                                                  score_thr, max_num, cfg, torch.from_numpy(dirs).cuda())
        assert np.array_equal(gb.cpu().numpy(), wb) and np.array_equal(gs.cpu().numpy(), ws)
        assert gl.dtype == torch.int64 and np.array_equal(gl.cpu().numpy(), wl) and np.array_equal(gd.cpu().numpy(), wd)


def test_nms_gpu_padded_is_sync_free_and_replays_as_a_hipgraph(host_glue, amd):
    """nms_gpu(..., padded=True): kept indices padded to the candidate count + a device count, no read-back: equal to the plain
    call, also from inside a captured graph on new boxes in the same buffers"""
    b0, s0 = nms_boxes(1500, seed=31)
    b1, s1 = nms_boxes(1500, seed=32)
    boxes, scores = torch.from_numpy(b0).cuda(), torch.from_numpy(s0).cuda()
    for pre, post in ((None, None), (1000, 83), (4000, None)):
        keep, num = amd.nms_gpu(boxes, scores, 0.2, pre_max_size=pre, post_max_size=post, padded=True)
        want = amd.nms_gpu(boxes, scores, 0.2, pre_max_size=pre, post_max_size=post)
        assert num.shape == (1,) and int(num) == want.shape[0] and torch.equal(keep[:int(num)], want)
    amd.nms_gpu(boxes, scores, 0.2, pre_max_size=1000, post_max_size=83, padded=True)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        keep, num = amd.nms_gpu(boxes, scores, 0.2, pre_max_size=1000, post_max_size=83, padded=True)
    boxes.copy_(torch.from_numpy(b1).cuda())
    scores.copy_(torch.from_numpy(s1).cuda())
    graph.replay()
    torch.cuda.synchronize()
    want = amd.nms_gpu(boxes, scores, 0.2, pre_max_size=1000, post_max_size=83)
    assert int(num) == want.shape[0] and torch.equal(keep[:int(num)], want)
    e, n = amd.nms_gpu(boxes[:0], scores[:0], 0.2, padded=True)
    assert e.shape == (0,) and int(n) == 0


def test_multi_class_nms_batch_equals_the_per_sample_loop(amd):
    """pvrcnn_bbox_head.py:393-405: `multi_class_nms(class_pred[b], boxes[roi_batch_id == b], ...)` per sample, against the one
    batched call over all (sample, class) groups; rois of the samples interleaved, one sample with nothing above threshold"""
    B, C = 4, 3
    n = 600
    boxes, _ = nms_boxes(n, seed=18)
    rng = np.random.default_rng(19)
    probs = rng.uniform(0, 1, (n, C)).astype(np.float32)
    bid = rng.integers(0, B, n)
    probs[bid == 2] *= 0.05                                        # sample 2: nothing reaches the thresholds
    score_thr, nms_thr = [0.5, 0.2, 0.8], [0.1, 0.3, 0.5]
    gp, gb, gi = torch.from_numpy(probs).cuda(), torch.from_numpy(boxes).cuda(), torch.from_numpy(bid).cuda()
    got = amd.multi_class_nms_batch(gp, gb, gi, B, score_thr, nms_thr)
    assert len(got) == B
    for b in range(B):
        m = gi == b
        want = amd.multi_class_nms(gp[m], gb[m], score_thr, nms_thr)
        if isinstance(want, list):
            assert isinstance(got[b], list) and got[b] == [] and b == 2
        else:
            assert torch.equal(got[b], want), b
    one = amd.multi_class_nms_batch(gp, gb, torch.zeros_like(gi), 1, 0.3, 0.2, use_rotate_nms=False)
    assert torch.equal(one[0], amd.multi_class_nms(gp, gb, 0.3, 0.2, use_rotate_nms=False))


@pytest.mark.parametrize('n,thr', [(1, 1.0), (500, 4.0), (3000, 0.85), (3000, 0.175), (5000, 12.0)])
def test_circle_nms_bit_exact(amd, n, thr):
    """mmdet3d circle_nms restated (oracle) vs the device version: same kept indices, same order, post_max_size cut."""
    rng = np.random.default_rng(n)
    nc = max(1, n // 6)
    c = rng.uniform(-50, 50, (nc, 2))
    xy = c[rng.integers(0, nc, n)] + rng.normal(0, 0.6, (n, 2))
    dets = np.concatenate([xy, rng.uniform(0, 1, (n, 1))], 1).astype(np.float32)
    dets[::7, :2] = dets[0, :2]                       # exact duplicates: distance 0 <= thresh
    want = oracle.circle_nms(dets, thr, post_max_size=83)
    got = amd.circle_nms(torch.from_numpy(dets).cuda(), thr, post_max_size=83)
    assert np.array_equal(got.cpu().numpy(), want)
    full = amd.circle_nms(torch.from_numpy(dets).cuda(), thr, post_max_size=None)
    assert np.array_equal(full.cpu().numpy(), oracle.circle_nms(dets, thr, post_max_size=None))
    assert amd.circle_nms(torch.zeros(0, 3).cuda(), thr).numel() == 0


def test_match_coco_and_trans_bev_vs_reference_golden(amd):
    """Device matcher / centre-distance vs vectors produced by the reference's own compiled matcher.cpp / affinity.cpp."""
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'match_coco.npz'))
    for name in g['cases']:
        got = amd.match_coco(g[f'{name}.cost'], g[f'{name}.thrs'], g[f'{name}.ignore'], g[f'{name}.crowd'])
        assert got.dtype == torch.int32 and np.array_equal(got.cpu().numpy(), g[f'{name}.matched']), name
    assert np.array_equal(amd.trans_bev(g['trans.det'], g['trans.gt']).cpu().numpy(), g['trans.dist'])
    # column stride other than 7
    assert np.array_equal(amd.trans_bev(g['trans.det'][:, :2], g['trans.gt'][:, :3]).cpu().numpy(), g['trans.dist'])


@pytest.mark.parametrize('D,G,T', [(3000, 50, 10), (200, 5000, 3), (1, 1, 1), (64, 64, 4), (500, 65, 2)])
def test_match_coco_large_vs_oracle(amd, D, G, T):
    rng = np.random.default_rng(D + G)
    cost = np.round(-rng.uniform(0, 1, (D, G)), 2).astype(np.float32)        # rounded: plenty of exact ties
    cost[rng.uniform(0, 1, (D, G)) < 0.01] = -0.0                             # signed zeros compare equal to +0
    thrs = -np.linspace(0.0, 0.9, T).astype(np.float32)
    ign = rng.uniform(0, 1, G) < 0.3; crowd = rng.uniform(0, 1, G) < 0.1
    want = oracle.match_coco(cost, thrs, ign, crowd)
    got = amd.match_coco(torch.from_numpy(cost).cuda(), thrs, torch.from_numpy(ign).cuda(), crowd)
    assert np.array_equal(got.cpu().numpy(), want)
    twin = amd.match_coco(*[torch.from_numpy(x) for x in (cost, thrs, ign, crowd)])     # CPU tensors: eval_match_coco_cpu
    assert twin.device.type == 'cpu' and np.array_equal(twin.numpy(), want)


def test_iou_to_matches_pipeline_on_the_device(amd):
    """The evaluation flow on the device: iou_3d affinity -> match_coco on negated costs (what the reference's matcher does for
    IoU-like scores, core/evaluation/matcher.py:20-24) vs the same flow on the CPU oracle (both pinned to the compiled
    reference); numpy inputs are accepted and the affinity stays in HBM between the two calls."""
    det = eval_boxes(300, seed=1, spread=25.0); gt = eval_boxes(40, seed=2, spread=25.0)
    gt[:20, :] = det[:20, :] + np.float32(0.05)
    aff = amd.iou_3d(torch.from_numpy(det).cuda(), torch.from_numpy(gt).cuda(), 0.5)
    thrs = np.array([0.3, 0.5, 0.7], np.float32)
    ign = np.zeros(40, bool); ign[::4] = True
    got = amd.match_coco(-aff, -thrs, ign, np.zeros(40, bool))
    aff_o = oracle.eval_iou_3d(det, gt, 0.5)
    assert np.array_equal(aff.cpu().numpy(), aff_o)
    want = oracle.match_coco(-aff_o, -thrs, ign, np.zeros(40, bool))
    assert got.is_cuda and np.array_equal(got.cpu().numpy(), want)
    assert np.array_equal(amd.trans_bev(det, gt).cpu().numpy(), oracle.eval_trans_bev(det, gt))


def test_pairwise_matrix_offsets_beyond_2_31(amd):
    """50 000 x 45 000 centre distances = 2.25e9 outputs: the flat pair index and the output offset exceed int32."""
    D, G = 50_000, 45_000
    free, _ = torch.cuda.mem_get_info()
    if free < (24 << 30):
        pytest.skip('needs 24 GB of free HBM')
    g = torch.Generator(device='cuda').manual_seed(2)
    det = torch.rand(D, 7, generator=g, device='cuda') * 100
    gt = torch.rand(G, 9, generator=g, device='cuda') * 100
    out = amd.trans_bev(det, gt)
    assert out.numel() > 2 ** 31
    for rows in (slice(0, 64), slice(D - 64, D)):           # first rows and the rows past offset 2^31
        dx = det[rows, None, 0] - gt[None, :, 0]; dy = det[rows, None, 1] - gt[None, :, 1]
        assert torch.equal(out[rows], torch.sqrt(dx * dx + dy * dy))
    del out
    torch.cuda.empty_cache()


def test_nms_at_the_maximum_size_closed_form(host_glue, amd):
    """n = 65 536 (RNMS_MAX_N): 32 768 separated sites on a 256 x 128 unit grid with two identical axis-aligned boxes each
    (all coordinates exact in fp32 and below 256).  Whatever the order, exactly the higher-scored box of every site
    survives, in descending score order.  One box more is refused, not truncated.
    Beyond |coordinate| >= 256 the published overlap algorithm's in-box margin of 1e-5 is less than half an fp32 ulp,
    `x1 - MARGIN` rounds back to x1 and a corner ON the boundary no longer counts as inside: identical boxes then have
    IoU 0 and do not suppress each other (oracle: iou_bev_xyxyr of a box at (500, 20) with itself = 0).  That behaviour of
    the algorithm is pinned against the CPU restatement on 8192 far-away duplicates (bit-exact), not 'fixed'."""
    n = 65536
    g = torch.Generator().manual_seed(11)
    site = torch.arange(n // 2)

    def problem(cx, cy, hx, hy, rot):
        one = torch.stack([cx - hx, cy - hy, cx + hx, cy + hy, rot], 1)           # xyxyr
        boxes = one.repeat_interleave(2, 0)
        scores = torch.rand(n, generator=g)
        perm = torch.randperm(n, generator=g)
        return boxes[perm].contiguous(), scores[perm].contiguous(), perm // 2

    boxes, scores, site_of = problem((site % 256).float(), (site // 256).float(), 0.25, 0.125, torch.zeros(n // 2))
    keep = amd.nms_gpu(boxes.cuda(), scores.cuda(), 0.5).cpu()
    order = torch.argsort(scores, descending=True, stable=True)
    seen = torch.zeros(n // 2, dtype=torch.bool)
    want = []
    for i in order.tolist():                       # closed form: the first box of a site in score order
        s = int(site_of[i])
        if not seen[s]:
            seen[s] = True; want.append(i)
    assert keep.tolist() == want
    with pytest.raises(RuntimeError):
        amd.nms_gpu(torch.cat([boxes, boxes[:1]]).cuda(), torch.cat([scores, scores[:1]]).cuda(), 0.5)
    # rotated duplicates up to 2500 m from the origin: whatever the algorithm decides there, GPU == CPU restatement
    rb, rs, _ = problem((site % 256).float() * 10.0, (site // 256).float() * 10.0, 1.5, 0.8,
                        torch.rand(n // 2, generator=g) * 6.0 - 3.0)
    top = torch.argsort(rs, descending=True, stable=True)[:8192]
    rb, rs = rb[top].contiguous(), rs[top].contiguous()
    want_r = oracle.nms_gpu_oracle(rb.numpy(), rs.numpy(), 0.5)
    got_r = amd.nms_gpu(rb.cuda(), rs.cuda(), 0.5).cpu().numpy()
    assert np.array_equal(got_r, want_r)
    far = np.array([[498.5, 19.0, 501.5, 21.0, 0.0]], np.float32)
    assert oracle.iou_bev_xyxyr(far, far)[0, 0] == 0.0 and float(amd.boxes_iou_bev(torch.from_numpy(far).cuda(), torch.from_numpy(far).cuda())[0, 0]) == 0.0


@pytest.mark.parametrize('n', [300, 5000, 20000])
def test_nms_nan_and_infinite_scores_order_as_torch_sort(amd, host_glue, n):
    """mmdet3d's nms_gpu orders with `scores.sort(0, descending=True)`: NaN first, then +inf ... -inf.  The library's own
    score ordering (n <= 16 384) and the torch.sort path above it do the same (stable among equals)."""
    b, s = nms_boxes(n, seed=4)
    s[::37] = np.nan; s[5] = np.inf; s[9] = -np.inf
    bt, st = torch.from_numpy(b).cuda(), torch.from_numpy(s).cuda()
    order = torch.sort(st, descending=True, stable=True)[1].cpu().numpy()
    want = order[oracle.nms_bev(b[order], 0.3)]
    got = amd.nms_gpu(bt, st, 0.3).cpu().numpy()
    assert np.array_equal(got, want)


@pytest.mark.parametrize('n,seed,extent,thr', [(4096, 200, 74.88, 0.25), (4096, 202, 74.88, 0.25), (1000, 77, 51.2, 0.2)])
def test_nms_keep_list_survives_an_ulp_of_yaw_on_the_gpu(amd, n, seed, extent, thr):
    """VERDICT r02 item 5, product side: every box's yaw moved to the next / previous fp32 value (sin and cos then move by
    up to an ulp, differently per box) on the BASELINE workloads — the GPU keep list equals the oracle's on the SAME nudged
    input bit for bit, and equals the un-nudged keep list (profiles/r03_nms_margin.txt: the closest decision of these
    workloads sits 9e-7 from the threshold, an ulp of sin / cos moves an IoU by ~1e-7)."""
    boxes, scores = nms_boxes(n, seed=seed, extent=extent)
    s = torch.from_numpy(scores).cuda()
    base = amd.nms_gpu(torch.from_numpy(boxes).cuda(), s, thr, pre_max_size=n)
    assert np.array_equal(base.cpu().numpy(), oracle.nms_gpu_oracle(boxes, scores, thr, pre_max_size=n))
    rng = np.random.default_rng(seed)
    for direction in (np.float32(np.inf), np.float32(-np.inf), None):
        b = boxes.copy()
        if direction is None:   # a random third up, a third down, a third untouched
            pick = rng.integers(0, 3, n)
            b[:, 4] = np.where(pick == 0, np.nextafter(b[:, 4], np.float32(np.inf)),
                               np.where(pick == 1, np.nextafter(b[:, 4], np.float32(-np.inf)), b[:, 4]))
        else:
            b[:, 4] = np.nextafter(b[:, 4], direction)
        assert (b[:, 4] != boxes[:, 4]).sum() >= n // 2
        got = amd.nms_gpu(torch.from_numpy(b).cuda(), s, thr, pre_max_size=n)
        assert np.array_equal(got.cpu().numpy(), oracle.nms_gpu_oracle(b, scores, thr, pre_max_size=n))
        assert torch.equal(got, base)


@pytest.mark.parametrize('mode', [0, 1])
def test_scored_nms_over_box_sets_equals_the_per_set_calls(amd, mode):
    """rnms_batched_scored_sets (the class problems of every sample of a batch in one set of launches: group g works on box set
    g // groups_per_set) against nms_gpu_batched on each set alone: same kept indices (offset by set * n), same counts."""
    import ctypes
    from mmdet3d_gaussian_amd import _lib
    lib = _lib.load()
    sets, gps, n = 3, 4, 700
    dev = torch.device('cuda:0')
    bx, sc = [], []
    rng = np.random.default_rng(4)
    for k in range(sets):
        b, _ = nms_boxes(n, seed=20 + k)
        bx.append(torch.from_numpy(b))
        sc.append(torch.from_numpy(rng.random((gps, n)).astype(np.float32)))
    boxes = torch.stack(bx).to(dev).contiguous()                 # (sets, n, 5)
    scores = torch.stack(sc).to(dev).contiguous()                # (sets, gps, n)
    valid = (scores > 0.3).contiguous()
    valid[1, 2] = False                                          # an empty group
    th = torch.tensor([0.1, 0.25, 0.5, 0.7] * sets, dtype=torch.float32, device=dev)
    keep = torch.full((sets * gps, n), -1, dtype=torch.int64, device=dev)
    num = torch.empty(sets * gps, dtype=torch.int64, device=dev)
    ws = torch.empty(lib.rnms_batched_scored_workspace_bytes(sets * gps, n, n), dtype=torch.uint8, device=dev)
    rc = lib.rnms_batched_scored_sets(mode, boxes.data_ptr(), scores.data_ptr(), valid.data_ptr(), sets, gps, n, -1, th.data_ptr(), keep.data_ptr(),
                                      num.data_ptr(), ws.data_ptr(), torch.cuda.current_stream().cuda_stream)
    assert rc == 0
    for k in range(sets):
        ref = amd.nms_gpu_batched(boxes[k], scores[k], [0.1, 0.25, 0.5, 0.7], valid[k], normal=bool(mode))
        for g in range(gps):
            m = int(num[k * gps + g])
            assert m == ref[g].shape[0]
            assert torch.equal(keep[k * gps + g, :m], ref[g] + k * n)
    assert int(num[1 * gps + 2]) == 0 and int(num.sum()) > 100
    assert lib.rnms_batched_scored_sets(mode, boxes.data_ptr(), scores.data_ptr(), valid.data_ptr(), sets, 0, n, -1, th.data_ptr(), keep.data_ptr(),
                                        num.data_ptr(), ws.data_ptr(), None) == 10001



# ---- NMS against keep lists derived from the REFERENCE's own rotated-IoU arithmetic (tests/golden/nms_ref_iou.npz) ----
import nms_ref  # noqa: E402


@pytest.mark.parametrize('name', nms_ref.SETS)
def test_nms_keep_list_against_the_reference_derived_one(amd, host_glue, name):
    """nms_gpu against the greedy list that the reference's own compiled iou_bev implies (ops/eval/affinity.cpp:51-81 over
    rbox_utils.hpp:280-302; call site gd_centerpoint_head.py:336-345) on the box sets of the call sites: configs[4] 3 x 4096
    thr 0.25, nuScenes 1000 / 83 thr 0.2, PV-RCNN shapes thr 0.7 / 0.8.  Boxes whose decision (or a decision upstream of
    theirs) sits within 1e-4 of the threshold in the reference's IoU are excluded — at most 4 of 9000, 0-2 of 4096; everything
    else must agree exactly, and with nothing excluded the post-cut list is the same list."""
    from test_nms_ref_iou import MAX_UNCERTAIN
    g = nms_ref.load(name)
    b, s = torch.from_numpy(g['boxes']).cuda(), torch.from_numpy(g['scores']).cuda()
    keep = amd.nms_gpu(b, s, g['thr'], pre_max_size=g['pre']).cpu().numpy()
    n_unc, bad, total = nms_ref.compare_keep(g, keep)
    assert bad == 0, (name, bad)
    assert n_unc == MAX_UNCERTAIN[name]
    assert nms_ref.disagreements(g, keep) == nms_ref.RESIDUE.get(name, [])    # the residue by identity: one box, in waymo2
    if n_unc == 0:
        cut = amd.nms_gpu(b, s, g['thr'], pre_max_size=g['pre'], post_max_size=g['post']).cpu().numpy()
        assert np.array_equal(cut, g['order'][g['keep_ref']][:g['post']])


@pytest.mark.parametrize('name', nms_ref.SETS)
def test_riou_bev_xyxyr_against_the_reference_iou(amd, name):
    """riou_bev_xyxyr (boxes_iou_bev) on every pair the reference gives a positive IoU: to 1e-5 near the origin; to 1e-4 at
    scene scale (both sides evaluate in absolute fp32 coordinates and scatter ~3e-5 around fp64 at 70 m), except for at most
    two pairs per set where the reference's code loses a sliver intersection and the fp64 clipping sides with this kernel."""
    g = nms_ref.load(name)
    bs = g['boxes'][g['order']]
    ni, nj, nv = g['nz_i'].astype(np.int64), g['nz_j'].astype(np.int64), g['nz_iou']
    bt = torch.from_numpy(np.ascontiguousarray(bs)).cuda()
    own = np.empty(len(ni), np.float32)
    for k in range(0, len(bs), 1024):
        a, e = np.searchsorted(ni, [k, k + 1024])
        if e > a:
            blk = amd.boxes_iou_bev(bt[k:k + 1024].contiguous(), bt)
            own[a:e] = blk[torch.from_numpy(ni[a:e] - k).cuda(), torch.from_numpy(nj[a:e]).cuda()].cpu().numpy()
    diff = np.abs(own.astype(np.float64) - nv)
    if name == 'origin':
        assert diff.max() <= 1e-5
        return
    out = np.flatnonzero(diff > 1e-4)
    assert len(out) <= 2, (name, len(out))
    for t in out:
        ex = nms_ref.exact_iou_xyxyr(bs[ni[t]], bs[nj[t]])
        assert abs(own[t] - ex) <= 1e-4 < abs(nv[t] - ex)


def test_cpu_twins_equal_the_hip_kernels(amd):
    """csrc/rbox_cpu.cpp compiles the kernels' own geometry source for the host with the same -ffp-contract=off: pairwise IoU
    (both box formats), centre distances and NMS keep lists are the same BITS on CPU tensors and on GPU tensors."""
    d, g = eval_boxes(200, 7), eval_boxes(150, 8)
    dt, gt = torch.from_numpy(d), torch.from_numpy(g)
    assert torch.equal(amd.iou_bev(dt, gt), amd.iou_bev(dt.cuda(), gt.cuda()).cpu())
    assert torch.equal(amd.iou_3d(dt, gt, 0.3), amd.iou_3d(dt.cuda(), gt.cuda(), 0.3).cpu())
    assert torch.equal(amd.trans_bev(dt, gt), amd.trans_bev(dt.cuda(), gt.cuda()).cpu())
    b, s = nms_boxes(3000, seed=12)
    bt, st = torch.from_numpy(b), torch.from_numpy(s)
    assert torch.equal(amd.boxes_iou_bev(bt[:500], bt), amd.boxes_iou_bev(bt[:500].cuda(), bt.cuda()).cpu())
    for thr, pre, post in ((0.25, None, None), (0.7, 2048, 300)):
        assert torch.equal(amd.nms_gpu(bt, st, thr, pre_max_size=pre, post_max_size=post),
                           amd.nms_gpu(bt.cuda(), st.cuda(), thr, pre_max_size=pre, post_max_size=post).cpu())
    assert torch.equal(amd.nms_normal_gpu(bt, st, 0.4), amd.nms_normal_gpu(bt.cuda(), st.cuda(), 0.4).cpu())


@pytest.mark.parametrize('n', [300, 1000])
def test_batched_scored_sets_replay_in_a_hipgraph_with_a_persistent_workspace(amd, n):
    """rnms_batched_scored_sets (2 box sets x 3 class problems: the shape of the anchor heads' inference NMS) captured ONCE into a
    hipGraph and replayed five times on new boxes / scores / valid masks, workspace and outputs persistent across replays: every
    replay equals an eager call on a zeroed workspace.  Regression test of round 4: the per-group counts used to be cleared by a
    hipMemsetAsync in front of the launches, and as a graph MEMSET NODE that clear was not reliably ordered against the kernels
    around it on this ROCm — later replays saw counts of 0 or of twice the boxes (groups keeping nothing, or everything).  n = 300
    takes the compacted mask kernel, n = 1000 the queued form (counters and queue in the persistent workspace)."""
    lib = amd.load_library()
    sets, gps = 2, 3
    G = sets * gps
    dev = torch.device('cuda:0')

    def data(seed):
        rng = np.random.default_rng(seed)
        bx = torch.stack([torch.from_numpy(nms_boxes(n, seed=seed * 10 + k)[0]) for k in range(sets)]).to(dev).contiguous()
        sc = torch.from_numpy(rng.random((G, n)).astype(np.float32)).to(dev)
        va = torch.from_numpy((rng.random((G, n)) < float(rng.uniform(0.5, 1.0))).astype(np.uint8)).to(dev)
        return bx, sc, va

    thr = torch.full((G,), 0.1, device=dev)
    wsb = lib.rnms_batched_scored_workspace_bytes(G, n, n)

    def run(bx, sc, va, ws, keep, num):
        assert lib.rnms_batched_scored_sets(0, bx.data_ptr(), sc.data_ptr(), va.data_ptr(), sets, gps, n, -1, thr.data_ptr(), keep.data_ptr(),
                                            num.data_ptr(), ws.data_ptr(), torch.cuda.current_stream().cuda_stream) == 0

    sb, ss, sv = data(0)
    ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
    keep = torch.empty(G, n, dtype=torch.int64, device=dev)
    num = torch.empty(G, dtype=torch.int64, device=dev)
    run(sb, ss, sv, ws, keep, num)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        run(sb, ss, sv, ws, keep, num)
    for seed in (1, 2, 1, 1, 3):
        bx, sc, va = data(seed)
        sb.copy_(bx); ss.copy_(sc); sv.copy_(va)
        graph.replay()
        torch.cuda.synchronize()
        ws2 = torch.zeros(wsb, dtype=torch.uint8, device=dev)
        k2, n2 = torch.empty_like(keep), torch.empty_like(num)
        run(bx, sc, va, ws2, k2, n2)
        torch.cuda.synchronize()
        for g in range(G):
            assert int(num[g]) == int(n2[g]) > 0 and torch.equal(keep[g, :int(num[g])], k2[g, :int(n2[g])]), (seed, g, num.tolist(), n2.tolist())


# ---- the list scan (victim lists + one state byte per box in LDS; one group, n <= 16384) and its device-side fallback ----
@pytest.mark.parametrize('n,thr', [(768, 0.5), (1000, 0.7), (1000, 0.2), (4096, 0.5), (4096, 0.25), (4096, 0.1), (8448, 0.6), (9000, 0.7),
                                   (9000, 0.8), (9000, 0.3), (12288, 0.55), (16384, 0.7), (16384, 0.3), (16321, 0.45)])
def test_list_scan_keep_indices_bit_exact(amd, n, thr):
    """Single-group calls of 768..16384 boxes take nms_list_or_scan_kernel: same greedy keep list as the CPU oracle, through nms_gpu
    (score ranking inside the library) and through the pre-sorted C ABI entry."""
    boxes, scores = nms_boxes(n, seed=n + 5, clutter=True)
    b, s = torch.from_numpy(boxes).cuda(), torch.from_numpy(scores).cuda()
    want = oracle.nms_gpu_oracle(boxes, scores, thr)
    assert np.array_equal(amd.nms_gpu(b, s, thr).cpu().numpy(), want)
    lib = amd.load_library()
    order = torch.sort(s, dim=0, descending=True, stable=True)[1]
    sb = b[order].contiguous()
    keep = torch.empty(n, dtype=torch.int64, device='cuda')
    num = torch.zeros(1, dtype=torch.int64, device='cuda')
    nbytes = lib.rnms_workspace_bytes(n)
    ws = torch.full((nbytes,), 0xff, dtype=torch.uint8, device='cuda')   # garbage: nothing in the workspace may need initialising
    for _ in range(2):   # the second call reuses the workspace (lists, counts and the failure word are reset by the call itself)
        assert lib.rnms_bev(sb.data_ptr(), n, thr, keep.data_ptr(), num.data_ptr(), ws.data_ptr(), None) == 0
        k = int(num.item())
        assert np.array_equal(order[keep[:k]].cpu().numpy(), want)
    if thr >= 0.25:   # the list scan itself ran (no fallback); at lower thresholds these clustered scenes may overflow a victim list
        assert int(ws[nbytes - 256:nbytes - 252].view(torch.int32).item()) == 0


def test_list_scan_sparse_scene_bit_exact(amd):
    """No clutter: nearly every box is kept (dense keeps were the classic scan's worst case)."""
    n = 4096
    boxes, scores = nms_boxes(n, seed=4096, clutter=False)
    b, s = torch.from_numpy(boxes).cuda(), torch.from_numpy(scores).cuda()
    for thr in (0.25, 0.01):
        assert np.array_equal(amd.nms_gpu(b, s, thr).cpu().numpy(), oracle.nms_gpu_oracle(boxes, scores, thr))


def _ranked(boxes, scores):
    order = np.argsort(-scores, kind='stable')
    return order


def test_list_scan_box_with_many_near_victims(amd):
    """Forty near-duplicates at consecutive ranks 60..99: the first of them has > 16 victims inside the next blocks — more than a
    near list holds: the failure word is set and the same launch runs the classic scan; the result is still the oracle's."""
    rng = np.random.default_rng(5)
    n = 3000
    boxes, scores = nms_boxes(n, seed=91, clutter=False)
    order = _ranked(boxes, scores)
    dup = order[60:100]
    boxes[dup] = boxes[dup[0]] + rng.normal(0, 1e-3, (40, 5)).astype(np.float32)
    b, s = torch.from_numpy(boxes).cuda(), torch.from_numpy(scores).cuda()
    lib = amd.load_library()
    for thr in (0.5, 0.2):
        want = oracle.nms_gpu_oracle(boxes, scores, thr)
        assert np.array_equal(amd.nms_gpu(b, s, thr).cpu().numpy(), want)
    o = torch.sort(s, dim=0, descending=True, stable=True)[1]
    sb = b[o].contiguous()
    keep = torch.empty(n, dtype=torch.int64, device='cuda')
    num = torch.zeros(1, dtype=torch.int64, device='cuda')
    nbytes = lib.rnms_workspace_bytes(n)
    ws = torch.empty(nbytes, dtype=torch.uint8, device='cuda')
    assert lib.rnms_bev(sb.data_ptr(), n, 0.5, keep.data_ptr(), num.data_ptr(), ws.data_ptr(), None) == 0
    torch.cuda.synchronize()
    assert int(ws[nbytes - 256:nbytes - 252].view(torch.int32).item()) == 1          # fallback: a near list is full (39 > 16)
    assert np.array_equal(o[keep[:int(num.item())]].cpu().numpy(), oracle.nms_gpu_oracle(boxes, scores, 0.5))


@pytest.mark.parametrize('first,count', [(60, 13), (55, 20), (120, 24)])
def test_list_scan_nine_to_sixteen_near_victims(amd, first, count):
    """Near-duplicates at consecutive ranks across a block boundary: the best of them has 9..16 victims inside the next blocks — the
    third and fourth 4-entry chunks of a ring slot, which the resolver reads from the block's own slot — and no list overflows."""
    rng = np.random.default_rng(first)
    n = 3000
    boxes, scores = nms_boxes(n, seed=93, clutter=False)
    order = _ranked(boxes, scores)
    dup = order[first:first + count]
    boxes[dup] = boxes[dup[0]] + rng.normal(0, 1e-3, (count, 5)).astype(np.float32)
    b, s = torch.from_numpy(boxes).cuda(), torch.from_numpy(scores).cuda()
    lib = amd.load_library()
    want = oracle.nms_gpu_oracle(boxes, scores, 0.5)
    assert np.array_equal(amd.nms_gpu(b, s, 0.5).cpu().numpy(), want)
    o = torch.sort(s, dim=0, descending=True, stable=True)[1]
    sb = b[o].contiguous()
    keep = torch.empty(n, dtype=torch.int64, device='cuda')
    num = torch.zeros(1, dtype=torch.int64, device='cuda')
    nbytes = lib.rnms_workspace_bytes(n)
    ws = torch.empty(nbytes, dtype=torch.uint8, device='cuda')
    assert lib.rnms_bev(sb.data_ptr(), n, 0.5, keep.data_ptr(), num.data_ptr(), ws.data_ptr(), None) == 0
    torch.cuda.synchronize()
    nearest = min(64 - first % 64, count) - 1                       # victims of the first duplicate inside its own block: colm, not a list
    assert 8 < count - 1 - nearest <= 16
    assert int(ws[nbytes - 256:nbytes - 252].view(torch.int32).item()) == 0          # the list scan itself ran
    assert np.array_equal(o[keep[:int(num.item())]].cpu().numpy(), want)


def test_list_scan_long_far_lists(amd):
    """Fifty near-duplicates spread over the whole ranking: the best of them has ~49 FAR victims (several uint4 of far list per
    box, the rolled far loop runs more than once) and no list overflows."""
    rng = np.random.default_rng(6)
    n = 6000
    boxes, scores = nms_boxes(n, seed=92, clutter=False)
    order = _ranked(boxes, scores)
    dup = order[np.arange(50) * 117 + 5]          # one per ~2 blocks: at most 5 of them inside any window of 8 blocks
    boxes[dup] = boxes[dup[0]] + rng.normal(0, 1e-3, (50, 5)).astype(np.float32)
    b, s = torch.from_numpy(boxes).cuda(), torch.from_numpy(scores).cuda()
    lib = amd.load_library()
    want = oracle.nms_gpu_oracle(boxes, scores, 0.5)
    assert np.array_equal(amd.nms_gpu(b, s, 0.5).cpu().numpy(), want)
    o = torch.sort(s, dim=0, descending=True, stable=True)[1]
    sb = b[o].contiguous()
    keep = torch.empty(n, dtype=torch.int64, device='cuda')
    num = torch.zeros(1, dtype=torch.int64, device='cuda')
    nbytes = lib.rnms_workspace_bytes(n)
    ws = torch.empty(nbytes, dtype=torch.uint8, device='cuda')
    assert lib.rnms_bev(sb.data_ptr(), n, 0.5, keep.data_ptr(), num.data_ptr(), ws.data_ptr(), None) == 0
    torch.cuda.synchronize()
    assert int(ws[nbytes - 256:nbytes - 252].view(torch.int32).item()) == 0          # the list scan itself ran
    assert np.array_equal(o[keep[:int(num.item())]].cpu().numpy(), want)


def test_list_scan_structured_rank_distances(amd):
    """Disjoint little gadgets on a grid, placed at chosen RANK distances around the near / far boundary of the list scan (8 blocks):
    pairs (A suppresses B) and chains (A suppresses B, B would suppress C, A and C do not touch: C stays because B is not kept).  A
    suppressed box must not mark anybody, and a victim must be marked in time whether the resolver wave (near) or a helper wave
    (far) does it."""
    n = 6144
    dists = [1, 63, 64, 65, 127, 8 * 64 - 1, 8 * 64, 8 * 64 + 1, 9 * 64 - 1, 9 * 64, 9 * 64 + 1, 10 * 64 + 7, 20 * 64 + 3, 40 * 64 + 11]
    rng = np.random.default_rng(11)
    gx, gy = np.meshgrid(np.arange(80) * 10.0, np.arange(80) * 10.0)
    cells = np.stack([gx.ravel(), gy.ravel()], -1)[:n]          # one box per cell by default: nothing overlaps
    ctr = cells.copy()
    size = np.tile(np.array([[3.0, 1.5]]), (n, 1))
    yaw = rng.uniform(-np.pi, np.pi, n)
    used = set()                                                # boxes are laid out IN RANK ORDER (scores descend with the index)
    cursor = 100
    for rep in range(3):
        for d in dists:
            for d2 in (None, 1, 64 * 8, 64 * 9 + 5):
                offs = (0, d) if d2 is None else (0, d, d + d2)
                a_ = cursor
                while any(a_ + o in used for o in offs):
                    a_ += 1
                if a_ + offs[-1] >= n:
                    continue
                used.update(a_ + o for o in offs)
                cursor += 2
                b_ = a_ + d
                if d2 is None:
                    ctr[b_] = ctr[a_] + 0.05; yaw[b_] = yaw[a_]                # B on top of A
                else:
                    c_ = b_ + d2
                    ctr[b_] = ctr[a_] + np.array([1.6, 0.0]); yaw[a_] = yaw[b_] = 0.0   # A-B overlap 47 %: IoU 0.30
                    ctr[c_] = ctr[b_] + np.array([1.6, 0.0]); yaw[c_] = 0.0             # B-C the same, A-C apart
    boxes = np.concatenate([ctr - size / 2, ctr + size / 2, yaw[:, None]], -1).astype(np.float32)
    scores = np.linspace(1.0, 0.01, n).astype(np.float32)
    b, s = torch.from_numpy(boxes).cuda(), torch.from_numpy(scores).cuda()
    for thr in (0.25, 0.6):
        want = oracle.nms_gpu_oracle(boxes, scores, thr)
        assert 0 < n - len(want) < n // 2
        assert np.array_equal(amd.nms_gpu(b, s, thr).cpu().numpy(), want)
    lib = amd.load_library()
    keep = torch.empty(n, dtype=torch.int64, device='cuda')
    num = torch.zeros(1, dtype=torch.int64, device='cuda')
    nbytes = lib.rnms_workspace_bytes(n)
    ws = torch.empty(nbytes, dtype=torch.uint8, device='cuda')
    assert lib.rnms_bev(b.data_ptr(), n, 0.25, keep.data_ptr(), num.data_ptr(), ws.data_ptr(), None) == 0   # (already in rank order)
    torch.cuda.synchronize()
    assert int(ws[nbytes - 256:nbytes - 252].view(torch.int32).item()) == 0          # the list scan itself ran
    assert np.array_equal(keep[:int(num.item())].cpu().numpy(), oracle.nms_gpu_oracle(boxes, scores, 0.25))


def test_list_scan_falls_back_when_a_victim_list_overflows(amd):
    """A hundred near-duplicates of one box spread over many 64-blocks: the best of them has > 64 later-block victims, the clip kernel
    sets the failure word, and the same launch runs the classic scan instead."""
    rng = np.random.default_rng(3)
    n = 3000
    boxes, scores = nms_boxes(n, seed=77, clutter=False)
    dup = rng.choice(n, 100, replace=False)
    boxes[dup] = boxes[dup[0]] + rng.normal(0, 1e-3, (100, 5)).astype(np.float32)
    b, s = torch.from_numpy(boxes).cuda(), torch.from_numpy(scores).cuda()
    for thr in (0.5, 0.9):
        want = oracle.nms_gpu_oracle(boxes, scores, thr)
        assert np.array_equal(amd.nms_gpu(b, s, thr).cpu().numpy(), want)
    lib = amd.load_library()
    order = torch.sort(s, dim=0, descending=True, stable=True)[1]
    sb = b[order].contiguous()
    keep = torch.empty(n, dtype=torch.int64, device='cuda')
    num = torch.zeros(1, dtype=torch.int64, device='cuda')
    nbytes = lib.rnms_workspace_bytes(n)
    ws = torch.empty(nbytes, dtype=torch.uint8, device='cuda')
    assert lib.rnms_bev(sb.data_ptr(), n, 0.5, keep.data_ptr(), num.data_ptr(), ws.data_ptr(), None) == 0
    torch.cuda.synchronize()
    assert int(ws[nbytes - 256:nbytes - 252].view(torch.int32).item()) == 1          # the failure word (last 256 bytes of the workspace)
    assert np.array_equal(order[keep[:int(num.item())]].cpu().numpy(), oracle.nms_gpu_oracle(boxes, scores, 0.5))
    # and a call on the same workspace whose lists fit clears it again
    b2, s2 = nms_boxes(n, seed=78, clutter=False)
    o2 = torch.sort(torch.from_numpy(s2).cuda(), dim=0, descending=True, stable=True)[1]
    sb2 = torch.from_numpy(b2).cuda()[o2].contiguous()
    assert lib.rnms_bev(sb2.data_ptr(), n, 0.5, keep.data_ptr(), num.data_ptr(), ws.data_ptr(), None) == 0
    torch.cuda.synchronize()
    assert int(ws[nbytes - 256:nbytes - 252].view(torch.int32).item()) == 0
    assert np.array_equal(o2[keep[:int(num.item())]].cpu().numpy(), oracle.nms_gpu_oracle(b2, s2, 0.5))


@pytest.mark.parametrize('n,thr', [(768, 0.3), (4096, 0.25), (4096, 0.6), (9000, 0.5), (16384, 0.4)])
def test_list_scan_axis_aligned_boxes(amd, n, thr):
    """nms_normal_gpu from 768 boxes on: the mask kernel also fills the victim lists (wave-aggregated appends) and the list scan
    runs; keep lists equal the CPU oracle's, through the scored entry (rank_place clears the counters) and the pre-sorted one (a
    fill kernel does); the failure word stays clear."""
    boxes, scores = nms_boxes(n, seed=n + 9, clutter=True)
    b, s = torch.from_numpy(boxes).cuda(), torch.from_numpy(scores).cuda()
    want = oracle.nms_gpu_oracle(boxes, scores, thr, normal=True)
    assert np.array_equal(amd.nms_normal_gpu(b, s, thr).cpu().numpy(), want)
    lib = amd.load_library()
    order = torch.sort(s, dim=0, descending=True, stable=True)[1]
    sb = b[order].contiguous()
    keep = torch.empty(n, dtype=torch.int64, device='cuda')
    num = torch.zeros(1, dtype=torch.int64, device='cuda')
    nbytes = lib.rnms_workspace_bytes(n)
    ws = torch.full((nbytes,), 0xff, dtype=torch.uint8, device='cuda')     # garbage counters: the call must clear them itself
    for _ in range(2):
        assert lib.rnms_normal_bev(sb.data_ptr(), n, thr, keep.data_ptr(), num.data_ptr(), ws.data_ptr(), None) == 0
        assert np.array_equal(order[keep[:int(num.item())]].cpu().numpy(), want)
    if thr >= 0.25:
        assert int(ws[nbytes - 256:nbytes - 252].view(torch.int32).item()) == 0


@pytest.mark.parametrize('n,thr', [(300, 0.3), (4096, 0.25), (9000, 0.7), (20000, 0.5)])
def test_num_keep_is_written_once_and_may_live_in_pinned_host_memory(amd, n, thr):
    """include/gd3d.h: every rnms_* entry writes each num_keep word exactly once, with the kernel that resolves the group's last
    box — so the word may be pinned host memory that the caller polls (nms_gpu's count mailbox, round 6).  Sizes: the compacted
    one-launch form, the list scan, the list scan at 9000 boxes, and the two-level classic scan (20 000 boxes: five scan launches
    that used to keep their running count in num_keep).  The host polls WHILE the kernels run and records every value it sees."""
    import ctypes
    lib = amd.load_library()
    b, s = nms_boxes(n, seed=n + 3)
    order = np.argsort(-s, kind='stable')
    sb = torch.from_numpy(b[order]).cuda()
    vp = lambda t: ctypes.c_void_p(t.data_ptr())
    keep = torch.empty(n, dtype=torch.int64, device='cuda')
    ws = torch.empty(lib.rnms_workspace_bytes(n), dtype=torch.uint8, device='cuda')
    num_dev = torch.empty(1, dtype=torch.int64, device='cuda')
    assert lib.rnms_bev(vp(sb), n, thr, vp(keep), vp(num_dev), vp(ws), None) == 0
    want_k = int(num_dev.item())
    want = keep[:want_k].clone()
    box = torch.empty(8, dtype=torch.int64).pin_memory()
    words = box.numpy()
    pending = -(1 << 62)
    for _ in range(5):
        words[0] = pending
        keep.fill_(-7)
        torch.cuda.synchronize()
        assert lib.rnms_bev(vp(sb), n, thr, vp(keep), ctypes.c_void_p(box.data_ptr()), vp(ws), None) == 0
        seen = []
        for _spin in range(20_000_000):
            v = int(words[0])
            if v != pending:
                if not seen or seen[-1] != v:
                    seen.append(v)
                if len(seen) > 1 or _spin > 0 and torch.cuda.current_stream().query():
                    break
        torch.cuda.synchronize()
        v = int(words[0])
        if not seen or seen[-1] != v:
            seen.append(v)
        assert seen == [want_k], (seen, want_k)
        assert torch.equal(keep[:want_k], want)
    # and the module surface (both glues poll a mailbox of their own) returns the same list
    got = amd.nms_gpu(torch.from_numpy(b).cuda(), torch.from_numpy(s).cuda(), thr)
    assert np.array_equal(got.cpu().numpy(), order[want.cpu().numpy()])


def test_classic_scan_bit_exact_with_the_list_scan_switched_off(amd):
    """The classic scan (mask rows, two-level above 8448 boxes) is the list scan's fallback and the path above 16384 boxes: with
    RNMS_LIST_MIN_THR=2 (read once per process: a child process) every call takes it; same keep lists as the CPU oracle."""
    import subprocess
    import sys
    code = (
        "import sys, numpy as np, torch\n"
        "sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
        "import mmdet3d_gaussian_amd as amd, oracle\n"
        "from rbox_inputs import nms_boxes\n"
        "for n, thr, clutter in ((1000, 0.2, True), (4096, 0.25, True), (4096, 0.25, False), (9000, 0.7, True)):\n"
        "    b, s = nms_boxes(n, seed=n, clutter=clutter)\n"
        "    got = amd.nms_gpu(torch.from_numpy(b).cuda(), torch.from_numpy(s).cuda(), thr).cpu().numpy()\n"
        "    assert np.array_equal(got, oracle.nms_gpu_oracle(b, s, thr)), (n, thr)\n"
        "print('classic ok')\n" % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.path.dirname(os.path.abspath(__file__))))
    env = dict(os.environ, RNMS_LIST_MIN_THR='2')
    r = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0 and 'classic ok' in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
