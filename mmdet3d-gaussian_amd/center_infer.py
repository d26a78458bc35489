"""CenterPoint inference slice on the device: head maps of all tasks -> detections per sample, in three kernels families and
ONE host read-back (csrc/center_infer.hip + the batched NMS of csrc/rbox.hip).

Call surface of the reference's
  CenterHeadRev.get_bboxes / get_task_detections   /root/reference/mmdet3d_gaussian/models/dense_heads/gd_centerpoint_head.py:218-361
  CenterGDHead (yaw + dir maps)                    gd_centerpoint_head.py:372-387
  bbox_coder.select_best / decode                  core/bbox/coders/centerpoint_bbox_coders.py:23-58, :87-112;
                                                   centerpoint_bbox_yaw_coders.py:18-56
with the same test_cfg keys (max_per_img, score_threshold, post_center_limit_range, nms_type 'rotate' | 'circle', nms_thr,
min_radius, pre_max_size, post_max_size).  GPU tensors only: there is no CPU path.
"""
import ctypes

import torch

from . import _lib

_REV_ORDER = ('reg', 'height', 'dim', 'rot', 'vel')
_YAW_ORDER = ('reg', 'height', 'dim', 'yaw', 'dir', 'vel')
_WIDTH = dict(reg=2, height=1, dim=3, rot=2, yaw=1, dir=2, vel=2)
_DESC_CACHE = {}


def _task_dict(entry):
    """The reference hands `preds_dicts` as a tuple of one-element lists of dicts (multi_apply over one feature level)."""
    return entry[0] if isinstance(entry, (list, tuple)) else entry


def _f32c(t):
    if t.dtype != torch.float32:
        t = t.float()
    return t if t.is_contiguous() else t.contiguous()


def _fill_geometry(desc, coder):
    desc.norm_bbox = int(bool(coder.norm_bbox))
    desc.out_size_factor = float(coder.out_size_factor)
    desc.voxel_size = (ctypes.c_float * 2)(float(coder.voxel_size[0]), float(coder.voxel_size[1]))
    desc.pc_range = (ctypes.c_float * 2)(float(coder.pc_range[0]), float(coder.pc_range[1]))


def select_best(scores, preds, topk):
    """`bbox_coder.select_best(scores, preds, topk)` (centerpoint_bbox_coders.py:51-58): scores (B,C,H,W) — already passed
    through the sigmoid —, preds (B,N,H,W) -> scores (B,K), classes (B,K) int64, locs (B,K,2) int64 (x, y), preds (B,K,N):
    the K best cells over all classes by descending score (equal scores: ascending class, y, x), one launch."""
    if not scores.is_cuda:
        raise RuntimeError('select_best: the MI355X implementation has no CPU path')
    lib = _lib.load_extras()
    B, C, H, W = scores.shape
    N = preds.shape[1]
    if preds.shape[0] != B or tuple(preds.shape[2:]) != (H, W):
        raise RuntimeError(f'select_best: preds {tuple(preds.shape)} do not match scores {tuple(scores.shape)}')
    if N > 16:
        raise RuntimeError(f'select_best: {N} channels per box (the kernel gathers at most 16)')
    K = int(topk)
    if K > H * W:
        raise RuntimeError('selected index k out of range')     # what torch.topk raises in the reference
    sc, pr = _f32c(scores), _f32c(preds)
    dev = sc.device
    task = _lib.CenterInferTask()
    task.heatmap = sc.data_ptr()
    for j in range(N):
        task.channel[j] = pr.data_ptr() + 4 * j * H * W
        task.sample_stride[j] = N * H * W
    task.classes = C
    desc = _lib.CenterInferDesc()
    desc.num_tasks, desc.batch, desc.height, desc.width = 1, B, H, W
    desc.max_per_img, desc.num_channels, desc.decode, desc.heat_is_logit = K, N, 0, 0
    desc.tasks = ctypes.pointer(task)
    out_s = torch.empty((B, K), dtype=torch.float32, device=dev)
    out_c = torch.empty((B, K), dtype=torch.int64, device=dev)
    out_xy = torch.empty((B, K, 2), dtype=torch.int64, device=dev)
    out_p = torch.empty((B, K, N), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        ws = torch.empty(lib.center_infer_select_workspace_bytes(ctypes.byref(desc)), dtype=torch.uint8, device=dev)
        _lib.check(lib.center_infer_select(ctypes.byref(desc), ws.data_ptr(), out_s.data_ptr(), out_c.data_ptr(), out_xy.data_ptr(),
                                           out_p.data_ptr(), torch.cuda.current_stream().cuda_stream), 'center_infer_select')
    if scores.dtype != torch.float32:
        out_s = out_s.to(scores.dtype)
    if preds.dtype != torch.float32:
        out_p = out_p.to(preds.dtype)
    return out_s, out_c, out_xy, out_p


def _coder_kind(coder, first):
    kind = getattr(coder, 'infer_kind', None)
    if kind is None:
        kind = 'yaw' if 'yaw' in first else 'rev'
    need = _YAW_ORDER if kind == 'yaw' else _REV_ORDER
    for k in need:
        if k not in first and k not in ('reg', 'vel'):
            raise RuntimeError(f'center_head_get_bboxes: head map {k!r} is missing (maps: {sorted(first)})')
    return kind, [k for k in need if k in first or k == 'reg']


def center_head_get_bboxes(preds_dicts, bbox_coder, test_cfg, num_classes, img_metas=None, return_candidates=False,
                           padded=False):
    """`CenterHeadRev.get_bboxes(preds_dicts, img_metas)` (gd_centerpoint_head.py:218-303) for heads with rotate or circle NMS.

    preds_dicts : per task a dict (or the reference's one-element list of it) with 'heatmap' (B,C,H,W) LOGITS and the head
                  maps 'reg' (optional), 'height', 'dim', then 'rot' (CenterHeadRev) or 'yaw' + 'dir' (CenterGDHead), 'vel'
                  (optional) — the SEPARATE maps as the head produced them;
    bbox_coder  : a CenterPointBBoxCoderRev / CenterPointBBoxYawCoder of this package (pc_range, out_size_factor, voxel_size,
                  norm_bbox);
    test_cfg    : the head's test_cfg (dict-like);  num_classes: classes per task (label offsets, :293-297).
    Returns, per sample, [bboxes (n, 7 + vel) with z at the box bottom, scores (n,), labels (n,) int32]; with `img_metas`
    given the boxes are wrapped by img_metas[i]['box_type_3d'](bboxes, bbox_coder.code_size) as the reference does.
    padded=True: no read-back at all — returns dict(bboxes (B, R, co), scores (B, R), labels (B, R) int32, counts (B,) int64) on
    the device, rows beyond counts[b] undefined (counts[b] = -1: a device-side NMS scan gave up, the sample's rows are void — the
    caller of the padded form checks it where it next reads the counts): the whole slice is then stream-ordered and can sit inside a captured hipGraph
    (tests/test_gpu_center_infer.py::test_get_bboxes_replays_as_a_hipgraph) or feed a tracker without a host round trip.
    return_candidates: also return what went INTO the NMS, per task a dict(boxes (B,K,co), scores (B,K), labels (B,K) int32,
    counts (B,) int32): the survivors of the score / range mask in score order (rows beyond counts[b] are undefined)."""
    tasks = [_task_dict(e) for e in preds_dicts]
    if not tasks:
        return []
    first = tasks[0]
    heat0 = first['heatmap']
    if not heat0.is_cuda:
        raise RuntimeError('center_head_get_bboxes: the MI355X implementation has no CPU path')
    if len(num_classes) != len(tasks):
        raise RuntimeError(f'{len(tasks)} tasks but {len(num_classes)} class counts')
    lib = _lib.load_extras()
    dev = heat0.device
    B, _, H, W = heat0.shape
    kind, names = _coder_kind(bbox_coder, first)
    nchan = sum(_WIDTH[k] for k in names)
    nms_type = test_cfg['nms_type']
    if nms_type not in ('circle', 'rotate'):
        raise AssertionError(nms_type)
    K = int(test_cfg.get('max_per_img', 128))
    if K > H * W:
        raise RuntimeError('selected index k out of range')
    if K > lib.center_infer_max_k():
        raise RuntimeError(f'max_per_img {K} > {lib.center_infer_max_k()} (the selection kernel sorts in LDS)')
    # first pass: validate, make the maps fp32-contiguous, collect their addresses.  The descriptor (some 150 ctypes fields) is
    # rebuilt only when an address, a shape or a setting changed: an inference loop whose head writes into the same buffers
    # (the caching allocator hands them back step after step) pays for it once.
    keep_alive, ptrs, ncls = [], [], []
    for t, pd in enumerate(tasks):
        heat = _f32c(pd['heatmap'])
        if heat.dim() != 4 or heat.shape[0] != B or tuple(heat.shape[2:]) != (H, W):
            raise RuntimeError(f'task {t}: heatmap {tuple(heat.shape)} vs (B={B}, C, {H}, {W})')
        keep_alive.append(heat)
        ptrs.append(heat.data_ptr())
        ncls.append(heat.shape[1])
        for k in names:
            m = pd.get(k) if hasattr(pd, 'get') else (pd[k] if k in pd else None)
            if m is None:
                if k != 'reg':
                    raise RuntimeError(f'task {t}: head map {k!r} is missing')
                ptrs.append(0)               # no 'reg' head: the constant 0.5 (:206-208)
                continue
            m = _f32c(m)
            if tuple(m.shape) != (B, _WIDTH[k], H, W):
                raise RuntimeError(f'task {t}: {k} is {tuple(m.shape)}, expected {(B, _WIDTH[k], H, W)}')
            keep_alive.append(m)
            ptrs.append(m.data_ptr())
    rng = test_cfg.get('post_center_limit_range', None)
    pre = None if nms_type == 'circle' else test_cfg.get('pre_max_size', None)
    post = test_cfg.get('post_max_size', None)
    if (pre is not None and pre < 0) or (post is not None and post < 0):
        raise RuntimeError('center_head_get_bboxes: negative pre_max_size / post_max_size (a slice bound in the reference) is not supported')
    thr = [float(test_cfg['min_radius'][t] if nms_type == 'circle' else test_cfg['nms_thr']) for t in range(len(tasks))]
    key = (dev.index, tuple(ptrs), tuple(ncls), tuple(names), B, H, W, K, kind, nms_type, tuple(thr), tuple(int(c) for c in num_classes),
           float(test_cfg.get('score_threshold', 0.1)), None if rng is None else tuple(float(v) for v in rng), pre, post,
           bool(bbox_coder.norm_bbox), float(bbox_coder.out_size_factor), tuple(float(v) for v in bbox_coder.voxel_size[:2]),
           tuple(float(v) for v in bbox_coder.pc_range[:2]))
    hit = _DESC_CACHE.get(key)
    if hit is None:
        arr = (_lib.CenterInferTask * len(tasks))()
        flag, pi = 0, 0
        for t, nc in enumerate(num_classes):
            arr[t].heatmap = ptrs[pi]
            pi += 1
            arr[t].classes = ncls[t]
            j = 0
            for k in names:
                w = _WIDTH[k]
                base = ptrs[pi]
                pi += 1
                for q in range(w):
                    arr[t].channel[j + q] = (base + 4 * q * H * W) if base else None
                    arr[t].sample_stride[j + q] = w * H * W if base else 0
                j += w
            arr[t].label_offset = flag
            flag += int(nc)
            arr[t].nms_thresh = thr[t]
        desc = _lib.CenterInferDesc()
        desc.num_tasks, desc.batch, desc.height, desc.width = len(tasks), B, H, W
        desc.max_per_img, desc.num_channels = K, nchan
        desc.decode = 2 if kind == 'yaw' else 1
        desc.heat_is_logit = 1
        _fill_geometry(desc, bbox_coder)
        desc.use_score_threshold = 1
        desc.score_threshold = float(test_cfg.get('score_threshold', 0.1))
        desc.use_limit_range = int(rng is not None)
        if rng is not None:
            desc.limit_range = (ctypes.c_float * 6)(*[float(v) for v in rng])
        desc.nms_type = 2 if nms_type == 'circle' else 0
        # `if pre_max_size is not None` / `if post_max_size is not None` in nms_gpu; circle_nms always cuts to post_max_size
        desc.pre_max_size = -1 if pre is None else int(pre)
        desc.post_max_size = -1 if post is None else int(post)
        desc.tasks = arr
        if len(_DESC_CACHE) >= 16:
            _DESC_CACHE.clear()
        hit = _DESC_CACHE[key] = (desc, arr, int(lib.center_infer_rows_per_task(ctypes.byref(desc))),
                                  int(lib.center_infer_workspace_bytes(ctypes.byref(desc))))
    desc, _, rows, ws_bytes = hit
    co = nchan - (2 if kind == 'yaw' else 1)
    T = len(tasks)
    with torch.cuda.device(dev):
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
        boxes = torch.empty((B, T * rows, co), dtype=torch.float32, device=dev)
        scores = torch.empty((B, T * rows), dtype=torch.float32, device=dev)
        labels = torch.empty((B, T * rows), dtype=torch.int32, device=dev)
        count = torch.empty(B, dtype=torch.int64, device=dev)
        _lib.check(lib.center_infer_bboxes(ctypes.byref(desc), ws.data_ptr(), boxes.data_ptr(), scores.data_ptr(),
                                           labels.data_ptr(), count.data_ptr(), torch.cuda.current_stream().cuda_stream),
                   'center_infer_bboxes')
    if padded:
        return dict(bboxes=boxes, scores=scores, labels=labels, counts=count)
    ns = count.tolist()          # the one sync: B data-dependent detection counts
    if min(ns, default=0) < 0:   # a device-side NMS scan gave up (include/gd3d.h: num_keep = -1): the result is void
        raise RuntimeError('center_infer: a device-side NMS scan gave up (count -1); the result is void')
    out = []
    for i in range(B):
        bx = boxes[i, :ns[i]]
        if img_metas is not None:
            bx = img_metas[i]['box_type_3d'](bx, bbox_coder.code_size)
        out.append([bx, scores[i, :ns[i]], labels[i, :ns[i]]])
    if not return_candidates:
        return out
    offs = (ctypes.c_int64 * 4)()
    _lib.check(lib.center_infer_candidates(ctypes.byref(desc), offs), 'center_infer_candidates')
    G = T * B
    cb = ws[offs[0]:offs[0] + 4 * G * K * co].view(torch.float32).view(T, B, K, co)
    cs = ws[offs[1]:offs[1] + 4 * G * K].view(torch.float32).view(T, B, K)
    cl = ws[offs[2]:offs[2] + 4 * G * K].view(torch.int32).view(T, B, K)
    cn = ws[offs[3]:offs[3] + 4 * G].view(torch.int32).view(T, B)
    return out, [dict(boxes=cb[t], scores=cs[t], labels=cl[t], counts=cn[t]) for t in range(T)]
