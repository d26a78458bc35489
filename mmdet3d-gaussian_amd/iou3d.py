"""Rotated BEV NMS / IoU — host-side mirror of the mmdet3d ``iou3d`` surface the reference calls.

``nms_gpu(boxes, scores, thresh, pre_max_size=None, post_max_size=None)`` is what the reference
imports from third-party mmdet3d and calls at
/root/reference/mmdet3d_gaussian/models/dense_heads/gd_centerpoint_head.py:9,340-345 and
models/roi_heads/bbox_heads/pvrcnn_bbox_head.py:12,463-464 (and, through upstream
``box3d_multiclass_nms``, for the PointPillars heads).  Same argument names and meaning, same
return (LongTensor of kept indices into the INPUT order, by descending score).  Sorting stays in
torch; the suppression mask AND the greedy scan run on the device (csrc/rbox.hip), so only the
final count crosses to the host (the returned tensor has a data-dependent length).

``iou_bev`` / ``iou_3d`` are GPU counterparts of the reference's CPU eval helpers
(/root/reference/mmdet3d_gaussian/ops/eval/affinity.cpp:8-81), ``(D,7) x (G,7) -> (D,G)``.

Device follows the tensors: CPU tensors take the library's `_cpu` twins (csrc/rbox_cpu.cpp: the kernels' own geometry source
compiled for the host, bit-identical results) in ``nms_gpu`` / ``nms_normal_gpu`` / ``boxes_iou_bev`` / ``iou_bev`` / ``iou_3d``;
the batched, segmented and scored forms are GPU-only.
"""
import ctypes

import torch

from . import _lib, _pynode


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr())


def _stream(dev):
    return ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)


def _check_boxes(boxes, cols, name, cpu_ok=False):
    if not boxes.is_cuda and not cpu_ok:
        raise RuntimeError(f'{name}: the MI355X implementation has no CPU path; tensors must be on the GPU')
    if boxes.dim() != 2 or boxes.shape[1] != cols:
        raise RuntimeError(f'{name}: expected (N,{cols}) boxes, got {tuple(boxes.shape)}')
    return boxes.to(torch.float32).contiguous()


def _same_device(a, b, name):
    if a.device != b.device:
        raise RuntimeError(f'{name}: operands live on different devices ({a.device}, {b.device})')


def _nms_cpu(lib, boxes, scores, thresh, n, post_max_size, normal, padded):
    """CPU tensors: torch's stable descending sort (the order mmdet3d's nms_gpu takes), then the `_cpu` twin of the greedy scan
    (rnms_bev_cpu: the mask kernel's own predicate on one thread, same decisions bit for bit)."""
    order = torch.sort(scores.reshape(-1).float() if scores.dtype != torch.float64 else scores.reshape(-1),
                       descending=True, stable=True)[1][:n]
    sb = boxes[order].contiguous()
    keep = torch.empty(n, dtype=torch.int64)
    num = torch.zeros(1, dtype=torch.int64)
    fn = lib.rnms_normal_bev_cpu if normal else lib.rnms_bev_cpu
    _lib.check(fn(sb.data_ptr(), n, float(thresh), keep.data_ptr(), num.data_ptr()), 'rnms_bev_cpu')
    k = int(num[0])
    if padded:
        out = torch.zeros(n if post_max_size is None else min(n, post_max_size), dtype=torch.int64)
        kk = min(k, out.numel())
        out[:kk] = order[keep[:kk]]
        return out, torch.tensor([kk], dtype=torch.int64)
    keep = order[keep[:k]]
    return keep if post_max_size is None else keep[:post_max_size]


def _nms(boxes, scores, thresh, pre_max_size, post_max_size, normal, padded=False):
    name = 'nms_normal_gpu' if normal else 'nms_gpu'
    boxes = _check_boxes(boxes, 5, name, cpu_ok=True)
    if scores.shape[0] != boxes.shape[0]:
        raise RuntimeError(f'{name}: {boxes.shape[0]} boxes but {scores.shape[0]} scores')
    lib = _lib.load()
    dev = boxes.device
    n_all = boxes.shape[0]
    if scores.device != dev:
        raise RuntimeError(f'{name}: boxes and scores live on different devices')
    if pre_max_size is not None and pre_max_size < 0:   # a negative slice bound, as `order[:pre_max_size]` would take it
        pre_max_size = max(n_all + pre_max_size, 0)
    n = n_all if pre_max_size is None else min(n_all, pre_max_size)
    if n == 0:
        empty = torch.zeros((0,), dtype=torch.int64, device=dev)
        return (empty, torch.zeros(1, dtype=torch.int64, device=dev)) if padded else empty
    if not boxes.is_cuda:
        return _nms_cpu(lib, boxes, scores, thresh, n, post_max_size, normal, padded)
    # up to 16384 candidates (the heads cut to nms_pre first) the library orders the scores itself (rank by counting, prep
    # scattered to the rank: no torch.sort); float64 scores keep torch.sort (their order may differ after rounding to fp32)
    fused_sort = n_all <= _scored_max(lib) and scores.dim() == 1 and scores.dtype in (torch.float32, torch.float16,
                                                                                     torch.bfloat16)
    if fused_sort and scores.dtype == torch.float32 and scores.is_contiguous():
        # the usual call: allocations, launch, count read-back and cut in the host glue (`nms_scored` of _pynode.py or of its
        # C++ twin csrc/torch_node.cpp: _lib.load_node())
        post = int(post_max_size) if (post_max_size is not None and post_max_size >= 0 and not padded) else -1
        keep, num = _lib.load_node().nms_scored(boxes, scores, float(thresh), n, bool(normal), bool(padded), post)
        if padded:
            if post_max_size is not None:
                keep, num = keep[:post_max_size], num.clamp(max=max(int(post_max_size), 0))
            return keep, num
        return keep if (post_max_size is None or post_max_size >= 0) else keep[:post_max_size]   # a negative bound: Python slicing
    # raw device / stream accessors and a memoised workspace size: the call is a handful of launches (30-70 us of device time at
    # inference sizes) and the Python around it was a third of nms_gpu's end-to-end time
    prev = _get_device()
    if prev != dev.index:
        _set_device(dev.index)
    try:
        keep = torch.empty(n, dtype=torch.int64, device=dev)
        num = torch.empty(1, dtype=torch.int64, device=dev)
        stream = _raw_stream(dev.index)
        if fused_sort:
            sc = scores if scores.dtype == torch.float32 else scores.float()
            sc = sc if sc.is_contiguous() else sc.contiguous()
            ws = torch.empty(_ws_bytes(lib, True, n_all, n), dtype=torch.uint8, device=dev)
            rc = lib.rnms_scored(int(normal), boxes.data_ptr(), sc.data_ptr(), n_all, n, float(thresh), keep.data_ptr(),
                                 num.data_ptr(), ws.data_ptr(), stream)
        else:
            order = scores.sort(dim=0, descending=True, stable=True)[1]   # ties: lower index first
            order = order[:n].contiguous()
            ws = torch.empty(_ws_bytes(lib, False, n, n), dtype=torch.uint8, device=dev)
            # the kernels read boxes[order[i]] themselves and emit kept indices in the caller's numbering
            fn = lib.rnms_normal_bev_ordered if normal else lib.rnms_bev_ordered
            rc = fn(boxes.data_ptr(), order.data_ptr(), n, float(thresh), keep.data_ptr(), num.data_ptr(), ws.data_ptr(), stream)
    finally:
        if prev != dev.index:
            _set_device(prev)
    if rc != 0:
        _lib.check(rc, name)
    if padded:   # nothing read back: (kept indices padded to n rows, count on the device); the post_max_size cut applies to both
        if post_max_size is not None:
            keep, num = keep[:post_max_size], num.clamp(max=max(int(post_max_size), 0))
        return keep, num
    k = _kept_count(int(num.item()), 'nms_gpu')  # the one unavoidable sync: the result length is data dependent
    keep = keep[:k]
    if post_max_size is not None:
        keep = keep[:post_max_size]
    return keep


_SCORED_MAX = None
_WS_BYTES = {}
_raw_stream = torch._C._cuda_getCurrentRawStream
_get_device = torch._C._cuda_getDevice
_set_device = torch._C._cuda_setDevice


def _ws_bytes(lib, scored, n_all, n):
    """rnms_scored_workspace_bytes(n_all, n) / rnms_workspace_bytes(n), memoised (one ctypes call less per NMS)."""
    key = (scored, n_all, n)
    b = _WS_BYTES.get(key)
    if b is None:
        if len(_WS_BYTES) > 4096:
            _WS_BYTES.clear()
        b = _WS_BYTES[key] = int(lib.rnms_scored_workspace_bytes(n_all, n) if scored else lib.rnms_workspace_bytes(n))
    return b


_BATCHED_WS = {}


def _batched_ws_bytes(lib, G, N, cap):
    key = (G, N, cap)
    b = _BATCHED_WS.get(key)
    if b is None:
        if len(_BATCHED_WS) > 4096:
            _BATCHED_WS.clear()
        b = _BATCHED_WS[key] = int(lib.rnms_batched_scored_workspace_bytes(G, N, cap))
    return b


def _scored_max(lib):
    global _SCORED_MAX
    if _SCORED_MAX is None:
        _SCORED_MAX = int(lib.rnms_scored_max_n())
    return _SCORED_MAX


def nms_gpu(boxes, scores, thresh, pre_max_size=None, post_max_size=None, pre_maxsize=None, padded=False):
    """Rotated BEV NMS.  boxes (N,5) [x1,y1,x2,y2,ry]; returns kept indices (LongTensor).
    `pre_maxsize` is the spelling of older mmdet3d releases and is accepted as an alias.
    padded=True (not in mmdet3d): no host sync — returns (keep, num): keep (min(N, pre_max_size[, post_max_size]),) int64 whose
    first num[0] entries are the kept indices (the rest undefined) and num (1,) int64 on the device; the call can then sit
    inside a captured hipGraph."""
    if pre_max_size is None:
        pre_max_size = pre_maxsize
    return _nms(boxes, scores, thresh, pre_max_size, post_max_size, normal=False, padded=padded)


def nms_normal_gpu(boxes, scores, thresh):
    """Axis-aligned BEV NMS (angle ignored), mmdet3d `nms_normal_gpu`."""
    return _nms(boxes, scores, thresh, None, None, normal=True)


def _kept_count(k, name):
    """A negative count is the scan kernel's failure mark (a wave of the list scan stopped making progress and its bounded polling
    loop gave up: never observed; a bug must surface as an error, not as a hang or a wrong list)."""
    if k < 0:
        raise RuntimeError(f'{name}: the device-side NMS scan gave up (num_keep = {k}); the result is void')
    return k


_THRESH_CACHE = {}


def _thresh_tensor(thresh, groups, dev):
    vals = tuple(float(t) for t in thresh) if isinstance(thresh, (list, tuple)) else (float(thresh),) * groups
    if len(vals) != groups:
        raise RuntimeError(f'{len(vals)} thresholds for {groups} groups')
    key = (vals, dev)
    t = _THRESH_CACHE.get(key)
    if t is None:
        if len(_THRESH_CACHE) > 64:
            _THRESH_CACHE.clear()
        t = _THRESH_CACHE[key] = torch.tensor(vals, dtype=torch.float32, device=dev)
    return t


def nms_gpu_batched(boxes, scores, thresh, valid=None, pre_max_size=None, post_max_size=None, normal=False,
                    circle=False):
    """G independent NMS problems over ONE box array in one set of launches and one host sync.

    boxes (N,5) [x1,y1,x2,y2,ry] (circle=True: (N,2) centres); scores (G,N); valid (G,N) bool or None — which boxes take
    part in group g (e.g. `box_probs[:, k] >= score_thr[k]`, pvrcnn_bbox_head.py:454); thresh: float or G floats.
    Returns a list of G LongTensors: kept indices into `boxes`, by descending score — each equal to
    `nms_gpu(boxes[valid[g]], scores[g][valid[g]], thresh[g], pre_max_size, post_max_size)` mapped back through
    `valid[g].nonzero()`.  Group sizes stay on the device (the kernels read them), so the loop-over-classes of the
    reference, with two host syncs per class, becomes three launches and one sync."""
    name = 'nms_gpu_batched'
    boxes = _check_boxes(boxes, 2 if circle else 5, name)
    if scores.dim() != 2 or scores.shape[1] != boxes.shape[0]:
        raise RuntimeError(f'{name}: scores must be (G,{boxes.shape[0]}), got {tuple(scores.shape)}')
    if valid is not None and valid.shape != scores.shape:
        raise RuntimeError(f'{name}: valid {tuple(valid.shape)} vs scores {tuple(scores.shape)}')
    if scores.device != boxes.device or (valid is not None and valid.device != boxes.device):
        raise RuntimeError(f'{name}: operands live on different devices')
    G, N = scores.shape
    dev = boxes.device
    if G == 0:
        return []
    if N == 0:
        return [torch.zeros((0,), dtype=torch.int64, device=dev) for _ in range(G)]
    if pre_max_size is not None and pre_max_size < 0:
        # `order[:pre_max_size]` with a negative bound keeps (group size + bound) boxes: a different cut per group, so the
        # promise "equal to nms_gpu per group" is kept by making exactly those calls
        ths = list(thresh) if isinstance(thresh, (list, tuple)) else [thresh] * G
        out = []
        for g in range(G):
            idx = torch.arange(N, device=dev) if valid is None else valid[g].nonzero(as_tuple=False).reshape(-1)
            sub = boxes[idx]
            if circle:
                raise RuntimeError('nms_gpu_batched: a negative pre_max_size is not defined for circle NMS')
            k = _nms(sub, scores[g][idx], ths[g], pre_max_size, post_max_size, normal)
            out.append(idx[k])
        return out
    lib = _lib.load()
    mode = 2 if circle else (1 if normal else 0)
    if N <= _scored_max(lib) and G <= 65535 and scores.dtype in (torch.float32, torch.float16, torch.bfloat16):
        # the library takes the score order itself (rank by counting per group): no masked_fill / sum / sort passes here
        cap = N if pre_max_size is None else max(min(N, int(pre_max_size)), 0)
        if cap == 0:
            return [torch.zeros((0,), dtype=torch.int64, device=dev) for _ in range(G)]
        with torch.cuda.device(dev):
            sc = (scores if scores.dtype == torch.float32 else scores.float()).contiguous()
            vb = None if valid is None else valid.to(torch.bool).contiguous()   # 1 byte per flag
            th = _thresh_tensor(thresh, G, dev)
            keep = torch.empty((G, cap), dtype=torch.int64, device=dev)
            ws = torch.empty(_batched_ws_bytes(lib, G, N, cap), dtype=torch.uint8, device=dev)
            # the G data-dependent result lengths arrive in pinned host memory, written by the scan kernels themselves, and are
            # polled there: no copy call, no stream synchronisation (_pynode.count_mailbox)
            box, words = _pynode.count_mailbox(G)
            words[:G] = _pynode._PENDING
            _lib.check(lib.rnms_batched_scored(mode, boxes.data_ptr(), sc.data_ptr(), None if vb is None else vb.data_ptr(), G, N,
                                               cap, th.data_ptr(), keep.data_ptr(), box.data_ptr(), ws.data_ptr(),
                                               _raw_stream(dev.index)), name)
        nums = [_kept_count(k, name) for k in _pynode.wait_counts(words, G, dev)]  # the one wait
        out = []
        for g in range(G):
            k = keep[g, :nums[g]]
            out.append(k if post_max_size is None else k[:post_max_size])
        return out
    with torch.cuda.device(dev):
        if valid is None:
            key = scores
            counts = torch.full((G,), N, dtype=torch.int32, device=dev)
        else:
            valid = valid.to(torch.bool)
            key = scores.masked_fill(~valid, float('-inf'))
            counts = valid.sum(1, dtype=torch.int32)
        order = key.sort(dim=1, descending=True, stable=True)[1]
        cap = N
        if pre_max_size is not None:
            cap = max(min(N, int(pre_max_size)), 0)
            order = order[:, :cap]
            counts = counts.clamp(max=cap)
        if cap == 0:
            return [torch.zeros((0,), dtype=torch.int64, device=dev) for _ in range(G)]
        order = order.contiguous()
        th = _thresh_tensor(thresh, G, dev)
        keep = torch.empty((G, cap), dtype=torch.int64, device=dev)
        num = torch.empty(G, dtype=torch.int64, device=dev)
        ws = torch.empty(lib.rnms_batched_workspace_bytes(G, cap), dtype=torch.uint8, device=dev)
        _lib.check(lib.rnms_batched(mode, boxes.data_ptr(), order.data_ptr(), counts.data_ptr(), G, cap, th.data_ptr(),
                                    keep.data_ptr(), num.data_ptr(), ws.data_ptr(),
                                    torch.cuda.current_stream().cuda_stream), name)
    nums = [_kept_count(k, 'nms (batched)') for k in num.tolist()]  # the one sync: G data-dependent result lengths
    out = []
    for g in range(G):
        k = keep[g, :nums[g]]
        out.append(k if post_max_size is None else k[:post_max_size])
    return out


def nms_gpu_multi(boxes_list, scores_list, thresh, pre_max_size=None, post_max_size=None, normal=False):
    """`[nms_gpu(b, s, thresh, pre_max_size, post_max_size) for b, s in zip(boxes_list, scores_list)]` in ONE set of
    launches and ONE host sync: the per-sample loop of CenterHeadRev.get_task_detections (gd_centerpoint_head.py:329-345)
    and the per-task loop around it (:233-282) — B samples x T tasks small NMS problems per inference step, each with its
    own sort, launches and `.item()`.  The entries are concatenated and every entry ranks and suppresses only ITS OWN
    run of boxes (rnms_segmented_scored: work and workspace grow with sum n_g^2, not with (sum n_g)^2).
    boxes_list[g] (n_g,5) [x1,y1,x2,y2,ry], scores_list[g] (n_g,); thresh: float or one per entry.
    Returns a list of LongTensors: kept indices LOCAL to each entry, by descending score.
    Entries larger than the library's rank limit, float64 scores, or a negative pre_max_size (a slice bound that means a
    different cut per entry) take the per-entry calls."""
    G = len(boxes_list)
    if len(scores_list) != G:
        raise RuntimeError(f'nms_gpu_multi: {G} box sets but {len(scores_list)} score sets')
    if G == 0:
        return []
    sizes = [int(b.shape[0]) for b in boxes_list]
    for b, s2, n in zip(boxes_list, scores_list, sizes):
        if s2.shape[0] != n:
            raise RuntimeError(f'nms_gpu_multi: {n} boxes but {s2.shape[0]} scores')
    dev = boxes_list[0].device
    total, nmax = sum(sizes), max(sizes)
    if total == 0:
        return [torch.zeros((0,), dtype=torch.int64, device=dev) for _ in range(G)]
    lib = _lib.load()
    ths = list(thresh) if isinstance(thresh, (list, tuple)) else [thresh] * G
    if len(ths) != G:
        raise RuntimeError(f'{len(ths)} thresholds for {G} groups')
    per_entry = (nmax > _scored_max(lib) or G > 65535 or (pre_max_size is not None and pre_max_size < 0) or
                 any(s2.dtype not in (torch.float32, torch.float16, torch.bfloat16) for s2 in scores_list))
    if per_entry:
        return [_nms(b, s2, t, pre_max_size, post_max_size, normal) for b, s2, t in zip(boxes_list, scores_list, ths)]
    cap = nmax if pre_max_size is None else min(nmax, int(pre_max_size))
    if cap == 0:
        return [torch.zeros((0,), dtype=torch.int64, device=dev) for _ in range(G)]
    offs = [0]
    for n in sizes:
        offs.append(offs[-1] + n)
    with torch.cuda.device(dev):
        boxes = _check_boxes(torch.cat([b.reshape(-1, 5) for b in boxes_list], dim=0), 5, 'nms_gpu_multi')
        flat = torch.cat([s2.reshape(-1) for s2 in scores_list], dim=0).to(torch.float32).contiguous()
        seg = torch.tensor(offs, dtype=torch.int32, device=dev)
        th = _thresh_tensor(ths, G, dev)
        keep = torch.empty((G, cap), dtype=torch.int64, device=dev)
        num = torch.empty(G, dtype=torch.int64, device=dev)
        ws = torch.empty(lib.rnms_batched_scored_workspace_bytes(G, nmax, cap), dtype=torch.uint8, device=dev)
        _lib.check(lib.rnms_segmented_scored(1 if normal else 0, boxes.data_ptr(), flat.data_ptr(), seg.data_ptr(), G, nmax, cap,
                                             th.data_ptr(), keep.data_ptr(), num.data_ptr(), ws.data_ptr(),
                                             torch.cuda.current_stream().cuda_stream), 'nms_gpu_multi')
        keep = keep - seg[:G].to(torch.int64).unsqueeze(1)          # indices local to each entry
    nums = [_kept_count(k, 'nms (batched)') for k in num.tolist()]  # the one sync: G data-dependent result lengths
    out = []
    for g in range(G):
        k = keep[g, :nums[g]]
        out.append(k if post_max_size is None else k[:post_max_size])
    return out


def multi_class_nms(box_probs, boxes_for_nms, score_thr, nms_thr, use_rotate_nms=True):
    """PVRCNNBboxHead.multi_class_nms (pvrcnn_bbox_head.py:438-480) with the class loop inside one batched NMS.
    box_probs (N,C); boxes_for_nms (N,5) [x1,y1,x2,y2,ry]; score_thr / nms_thr: float or list of C.
    Returns `selected` as the reference does: the kept indices of class 0, then class 1, ... (`[]` when none)."""
    N, C = box_probs.shape
    st = score_thr if isinstance(score_thr, (list, tuple)) else [score_thr] * C
    sc = box_probs.t().contiguous().to(torch.float32)
    valid = sc >= _thresh_tensor(st, C, sc.device).unsqueeze(1)
    per_class = nms_gpu_batched(boxes_for_nms, sc, nms_thr, valid, normal=not use_rotate_nms)
    sel = [k for k in per_class if k.shape[0] > 0]
    return torch.cat(sel, dim=0) if sel else []


def multi_class_nms_batch(box_probs, boxes_for_nms, roi_batch_id, batch_size, score_thr, nms_thr, use_rotate_nms=True):
    """The per-sample loop of PVRCNNBboxHead.get_bboxes around multi_class_nms (pvrcnn_bbox_head.py:393-405) in ONE batched
    NMS and one host sync for the whole batch: group (sample b, class k) takes the rois of sample b with probability
    >= score_thr[k].
    box_probs (R, C) / boxes_for_nms (R, 5) [x1,y1,x2,y2,ry] for ALL rois of the batch; roi_batch_id (R,) sample of every roi
    (`rois[..., 0]`); batch_size: number of samples (a host value: the reference reads it back with `.item()`, :378).
    Returns per sample what `multi_class_nms(class_pred[b], boxes[roi_batch_id == b], ...)` returns: indices LOCAL to the
    sample's own rois (its rois in their original order), classes in turn; `[]` when a sample keeps nothing."""
    R, C = box_probs.shape
    st = score_thr if isinstance(score_thr, (list, tuple)) else [score_thr] * C
    nt = list(nms_thr) if isinstance(nms_thr, (list, tuple)) else [nms_thr] * C
    dev = box_probs.device
    bid = roi_batch_id.to(device=dev, dtype=torch.int64).reshape(-1)
    sc = box_probs.t().contiguous().to(torch.float32)                                   # (C, R)
    above = sc >= _thresh_tensor(st, C, dev).unsqueeze(1)                              # (C, R)
    member = bid.unsqueeze(0) == torch.arange(batch_size, device=dev).unsqueeze(1)      # (B, R)
    valid = (member.unsqueeze(1) & above.unsqueeze(0)).reshape(batch_size * C, R)       # group b * C + k
    scores = sc.unsqueeze(0).expand(batch_size, C, R).reshape(batch_size * C, R)
    kept = nms_gpu_batched(boxes_for_nms, scores.contiguous(), nt * batch_size, valid, normal=not use_rotate_nms)
    # position of every roi among the rois of its own sample
    local = (torch.cumsum(member.to(torch.int64), dim=1) - 1).gather(0, bid.unsqueeze(0)).reshape(-1)
    out = []
    for b in range(batch_size):
        sel = [local[k] for k in kept[b * C:(b + 1) * C] if k.shape[0] > 0]
        out.append(torch.cat(sel, dim=0) if sel else [])
    return out


def box3d_multiclass_nms(mlvl_bboxes, mlvl_bboxes_for_nms, mlvl_scores, score_thr, max_num, cfg, mlvl_dir_scores=None):
    """mmdet3d `core/post_processing/box3d_nms.py::box3d_multiclass_nms` (third party, absent: restated from the published 0.x
    text) — what `Anchor3DHead.get_bboxes_single`, which the reference's GDAnchor3DHead inherits, runs on the `nms_pre` best
    anchors of a sample — with the class loop inside ONE batched NMS and one host sync instead of two per class:
      per class i < C-1 (the last score column is the padding / background column): candidates = scores[:, i] > score_thr
      (strictly), `nms_gpu` / `nms_normal_gpu` (cfg.use_rotate_nms) at cfg.nms_thr, kept boxes / scores / labels / direction
      scores concatenated class after class; more than max_num detections: the max_num best by score.
    mlvl_bboxes (N, box_dim), mlvl_bboxes_for_nms (N, 5) [x1,y1,x2,y2,ry], mlvl_scores (N, C), cfg: `.use_rotate_nms`, `.nms_thr`
    (attributes or keys).  Returns (bboxes, scores, labels int64, dir_scores | None)."""
    get = (lambda k: cfg[k]) if isinstance(cfg, dict) else (lambda k: getattr(cfg, k))
    N, C1 = mlvl_scores.shape
    C = C1 - 1
    dev = mlvl_scores.device
    if C <= 0 or N == 0:
        keep_flat = torch.zeros(0, dtype=torch.int64, device=dev)
        labels = keep_flat
    else:
        sc = mlvl_scores[:, :C].t().contiguous().to(torch.float32)                  # (C, N)
        valid = sc > float(score_thr)
        kept = nms_gpu_batched(mlvl_bboxes_for_nms, sc, float(get('nms_thr')), valid, normal=not get('use_rotate_nms'))
        keep_flat = torch.cat(kept, dim=0)
        labels = torch.cat([torch.full((k.shape[0],), i, dtype=torch.int64, device=dev) for i, k in enumerate(kept)], dim=0)
    bboxes = mlvl_bboxes[keep_flat]
    scores = mlvl_scores[keep_flat, labels] if keep_flat.numel() else mlvl_scores.new_zeros((0,))
    dir_scores = None if mlvl_dir_scores is None else mlvl_dir_scores[keep_flat]
    if bboxes.shape[0] > max_num:
        inds = scores.sort(descending=True, stable=True)[1][:max_num]
        bboxes, scores, labels = bboxes[inds], scores[inds], labels[inds]
        if dir_scores is not None:
            dir_scores = dir_scores[inds]
    return bboxes, scores, labels, dir_scores


def circle_nms(dets, thresh, post_max_size=83):
    """mmdet3d `circle_nms` on the device (the reference copies the detections to the host for the numba version,
    gd_centerpoint_head.py:256-272).  dets (N,3) [x, y, score]; a detection is suppressed by a kept, higher-scored one
    iff (dx^2 + dy^2) <= thresh.  Returns kept indices (LongTensor on dets.device), at most post_max_size."""
    dets = _check_boxes(dets, 3, 'circle_nms')
    n = dets.shape[0]
    dev = dets.device
    if n == 0:
        return torch.zeros((0,), dtype=torch.int64, device=dev)
    lib = _lib.load()
    with torch.cuda.device(dev):
        order = dets[:, 2].sort(dim=0, descending=True, stable=True)[1].contiguous()
        xy = dets[:, :2].contiguous()
        keep = torch.empty(n, dtype=torch.int64, device=dev)
        num = torch.empty(1, dtype=torch.int64, device=dev)
        ws = torch.empty(lib.rnms_workspace_bytes(n), dtype=torch.uint8, device=dev)
        _lib.check(lib.rnms_circle_ordered(xy.data_ptr(), order.data_ptr(), n, float(thresh), keep.data_ptr(),
                                           num.data_ptr(), ws.data_ptr(), torch.cuda.current_stream().cuda_stream),
                   'circle_nms')
    keep = keep[:_kept_count(int(num.item()), 'circle_nms')]
    return keep if post_max_size is None else keep[:post_max_size]


def boxes_iou_bev(boxes_a, boxes_b):
    """Pairwise rotated BEV IoU of [x1,y1,x2,y2,ry] boxes: (M,5),(N,5) -> (M,N) (mmdet3d `boxes_iou_bev`)."""
    a = _check_boxes(boxes_a, 5, 'boxes_iou_bev', cpu_ok=True)
    b = _check_boxes(boxes_b, 5, 'boxes_iou_bev', cpu_ok=True)
    lib = _lib.load()
    _same_device(a, b, 'boxes_iou_bev')
    out = torch.empty((a.shape[0], b.shape[0]), dtype=torch.float32, device=a.device)
    if not a.is_cuda:
        _lib.check(lib.riou_bev_xyxyr_cpu(a.data_ptr(), a.shape[0], b.data_ptr(), b.shape[0], out.data_ptr(), torch.get_num_threads()),
                   'riou_bev_xyxyr_cpu')
        return out
    with torch.cuda.device(a.device):
        _lib.check(lib.riou_bev_xyxyr(_ptr(a), a.shape[0], _ptr(b), b.shape[0], _ptr(out), _stream(a.device)),
                   'riou_bev_xyxyr')
    return out


def iou_bev(det, gt):
    """(D,7),(G,7) [x,y,z,w,h,l,yaw] -> (D,G) BEV IoU; GPU counterpart of ops/eval `iou_bev`."""
    d = _check_boxes(det, 7, 'iou_bev', cpu_ok=True)
    g = _check_boxes(gt, 7, 'iou_bev', cpu_ok=True)
    lib = _lib.load()
    _same_device(d, g, 'iou_bev')
    out = torch.empty((d.shape[0], g.shape[0]), dtype=torch.float32, device=d.device)
    if not d.is_cuda:   # the reference's own helper is CPU code (affinity.cpp:51-81): the `_cpu` twin
        _lib.check(lib.riou_eval_bev_cpu(d.data_ptr(), d.shape[0], g.data_ptr(), g.shape[0], out.data_ptr(), torch.get_num_threads()),
                   'riou_eval_bev_cpu')
        return out
    with torch.cuda.device(d.device):
        _lib.check(lib.riou_eval_bev(_ptr(d), d.shape[0], _ptr(g), g.shape[0], _ptr(out), _stream(d.device)),
                   'riou_eval_bev')
    return out


def iou_3d(det, gt, z_offset=0.5):
    """(D,7),(G,7) -> (D,G) 3D IoU with the reference's `z_offset` convention (affinity.cpp:26-29)."""
    d = _check_boxes(det, 7, 'iou_3d', cpu_ok=True)
    g = _check_boxes(gt, 7, 'iou_3d', cpu_ok=True)
    lib = _lib.load()
    _same_device(d, g, 'iou_3d')
    out = torch.empty((d.shape[0], g.shape[0]), dtype=torch.float32, device=d.device)
    if not d.is_cuda:
        _lib.check(lib.riou_eval_3d_cpu(d.data_ptr(), d.shape[0], g.data_ptr(), g.shape[0], float(z_offset), out.data_ptr(),
                                        torch.get_num_threads()), 'riou_eval_3d_cpu')
        return out
    with torch.cuda.device(d.device):
        _lib.check(lib.riou_eval_3d(_ptr(d), d.shape[0], _ptr(g), g.shape[0], float(z_offset), _ptr(out),
                                    _stream(d.device)), 'riou_eval_3d')
    return out


def xywhr2xyxyr(boxes_xywhr):
    """(N,5) [cx,cy,w,h,r] -> [x1,y1,x2,y2,r] (mmdet3d helper used at gd_centerpoint_head.py:336-337)."""
    out = torch.zeros_like(boxes_xywhr)
    half_w = boxes_xywhr[:, 2] / 2
    half_h = boxes_xywhr[:, 3] / 2
    out[:, 0] = boxes_xywhr[:, 0] - half_w
    out[:, 1] = boxes_xywhr[:, 1] - half_h
    out[:, 2] = boxes_xywhr[:, 0] + half_w
    out[:, 3] = boxes_xywhr[:, 1] + half_h
    out[:, 4] = boxes_xywhr[:, 4]
    return out
