"""oracle — TEST INFRASTRUCTURE, NOT PRODUCT CODE.

ctypes front-end of the CPU oracles (oracle/gd_oracle.c, oracle/rbox_oracle.c) and of the
compiled reference helpers (oracle/_ref, built from /root/reference sources where they lie).

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this package, and only as the checker / the reported CPU baseline.  The product package
(``mmdet3d-gaussian_amd/``) never imports it; the product's own CPU path (the _cpu twins) is compiled from its kernel source, not from here.
"""
import ctypes
import importlib.util
import glob
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(HERE, '_build', 'libgd3d_oracle.so')

LOSS_TYPES = {'gwd3d': 0, 'kld3d': 1, 'bd3d': 2, 'jd3d': 3, 'kld3d_symmax': 4,
              'kld3d_symmin': 5, 'kfiou3d': 6}
FUNS = {'none': 0, 'log1p': 1, 'expm1': 2, 'nlog': 3}


class Params(ctypes.Structure):
    """Mirror of gd3d_params (include/gd3d.h)."""
    _fields_ = [('loss_type', ctypes.c_int32), ('fun', ctypes.c_int32),
                ('tau', ctypes.c_float), ('alpha', ctypes.c_float),
                ('center_offset', ctypes.c_float * 3), ('flag', ctypes.c_int32)]


def build(force=False):
    """Compile the oracle library (and oracle/_ref when the reference tree is present)."""
    srcs = [os.path.join(HERE, f) for f in ('gd_oracle.c', 'gd_oracle_body.inc', 'rbox_oracle.c')]
    stale = (not os.path.isfile(_LIB_PATH) or
             any(os.path.getmtime(s) > os.path.getmtime(_LIB_PATH) for s in srcs))
    if force or stale:
        subprocess.run(['make', '-C', HERE, '_build/libgd3d_oracle.so'] + (['-B'] if force else []),
                       check=True, capture_output=True)
    if os.path.isdir('/root/reference') and (force or not ref_eval_path()):
        subprocess.run(['make', '-C', HERE, 'ref'], check=True, capture_output=True)


_lib = None


def lib():
    global _lib
    if _lib is None:
        override = os.environ.get('GD3D_ORACLE_LIB')   # an instrumented build of the same sources (tools/sanitize.sh)
        if not override:
            build()
        L = ctypes.CDLL(override or _LIB_PATH)
        i64, f32, vp = ctypes.c_int64, ctypes.c_float, ctypes.c_void_p
        for sfx, real in (('_f32', ctypes.c_float), ('_f64', ctypes.c_double)):
            fn = getattr(L, 'gd_oracle_loss' + sfx)
            fn.restype = ctypes.c_int
            fn.argtypes = [ctypes.POINTER(Params), vp, vp, vp, i64, real, vp, vp, vp, vp, ctypes.c_int]
        for sfx, real in (('_f32', ctypes.c_float), ('_f64', ctypes.c_double)):
            fn = getattr(L, 'gd_oracle_loss_decoded' + sfx)
            fn.restype = ctypes.c_int
            fn.argtypes = [ctypes.POINTER(Params), ctypes.c_int, ctypes.c_int, vp, real, real, real, real, real,
                           vp, vp, vp, i64, real, vp, vp, vp, vp]
        for name in ('rbox_oracle_nms_bev', 'rbox_oracle_nms_normal'):
            fn = getattr(L, name)
            fn.restype = i64
            fn.argtypes = [vp, i64, f32, vp]
        L.rbox_oracle_nms_mask.restype = None
        L.rbox_oracle_nms_mask.argtypes = [vp, i64, f32, vp]
        L.rbox_oracle_iou_bev_xyxyr.restype = None
        L.rbox_oracle_iou_bev_xyxyr.argtypes = [vp, i64, vp, i64, vp]
        L.rbox_oracle_eval_iou_bev.restype = None
        L.rbox_oracle_eval_iou_bev.argtypes = [vp, i64, vp, i64, vp]
        L.rbox_oracle_eval_iou_3d.restype = None
        L.rbox_oracle_eval_iou_3d.argtypes = [vp, i64, vp, i64, f32, vp]
        L.rbox_oracle_trans_bev.restype = None
        L.rbox_oracle_trans_bev.argtypes = [vp, i64, i64, vp, i64, i64, vp]
        L.rbox_oracle_match_coco.restype = None
        L.rbox_oracle_match_coco.argtypes = [vp, vp, vp, vp, i64, i64, i64, vp]
        _lib = L
    return _lib


def make_params(loss_type, fun='log1p', tau=1.0, alpha=1.0, center_offset=(0, 0, 0.5), **kwargs):
    """GDLoss ctor kwargs -> Params.  `normalize` (gwd3d) / `sqrt` (others) map to `flag`;
    defaults follow the reference signatures (gaussian_distance_loss.py:42,109,144,189,201,214,227)."""
    p = Params()
    p.loss_type = LOSS_TYPES[loss_type]
    p.fun = FUNS[fun]
    p.tau = float(tau)
    p.alpha = float(alpha)
    p.center_offset = (ctypes.c_float * 3)(*[float(c) for c in center_offset])
    if loss_type == 'gwd3d':
        flag = kwargs.pop('normalize', True)
    elif loss_type == 'kfiou3d':
        flag = kwargs.pop('sqrt', False)
    else:
        flag = kwargs.pop('sqrt', True)
    if kwargs:
        raise TypeError(f'unexpected kwargs {sorted(kwargs)}')
    p.flag = int(bool(flag))
    return p


def _ptr(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def gd_loss(pred, target, params, row_weight=None, scale=1.0, dtype=np.float64,
            want_grad_target=True, nthreads=1):
    """Per-pair loss, its sum and gradients on the CPU.

    Returns dict(loss (N,), loss_sum float, grad_pred (N,7), grad_target (N,7) or None) with
        loss[i] = scale * w_i * L_i, grads scaled likewise (contract of gd3d_loss_fused).
    """
    dtype = np.dtype(dtype)
    sfx = '_f32' if dtype == np.float32 else '_f64'
    pred = np.ascontiguousarray(np.asarray(pred, dtype=dtype).reshape(-1, 7))
    target = np.ascontiguousarray(np.asarray(target, dtype=dtype).reshape(-1, 7))
    n = pred.shape[0]
    assert target.shape[0] == n
    w = None if row_weight is None else np.ascontiguousarray(np.asarray(row_weight, dtype=dtype).reshape(n))
    loss = np.empty(n, dtype)
    gp = np.empty((n, 7), dtype)
    gt = np.empty((n, 7), dtype) if want_grad_target else None
    total = ctypes.c_double(0.0)
    rc = getattr(lib(), 'gd_oracle_loss' + sfx)(
        ctypes.byref(params), _ptr(pred), _ptr(target), _ptr(w), n, float(scale), _ptr(loss),
        ctypes.cast(ctypes.byref(total), ctypes.c_void_p), _ptr(gp), _ptr(gt), int(nthreads))
    if rc != 0:
        raise RuntimeError(f'gd_oracle_loss{sfx} -> {rc}')
    return dict(loss=loss, loss_sum=total.value, grad_pred=gp, grad_target=gt)


PRO_ANCHOR_DELTA, PRO_CENTER = 1, 2


def gd_loss_decoded(pred_enc, target, params, kind, aux, row_weight=None, scale=1.0, dtype=np.float64,
                    norm_bbox=True, out_size_factor=1.0, voxel_size=(1.0, 1.0), pc_range=(0.0, 0.0)):
    """Head-level oracle: bbox-coder decode (kind = PRO_ANCHOR_DELTA on pred and target with aux = anchors (N,7);
    kind = PRO_CENTER on pred only with aux = locs (N,2)) followed by the GD loss, gradients chained back to the
    ENCODED rows.  Same contract as gd3d_loss_fused_decoded."""
    dtype = np.dtype(dtype)
    sfx = '_f32' if dtype == np.float32 else '_f64'
    pred = np.ascontiguousarray(np.asarray(pred_enc, dtype=dtype).reshape(-1, 7))
    target = np.ascontiguousarray(np.asarray(target, dtype=dtype).reshape(-1, 7))
    n = pred.shape[0]
    aux = np.ascontiguousarray(np.asarray(aux, dtype=dtype).reshape(n, 7 if kind == PRO_ANCHOR_DELTA else 2))
    w = None if row_weight is None else np.ascontiguousarray(np.asarray(row_weight, dtype=dtype).reshape(n))
    loss = np.empty(n, dtype); gp = np.empty((n, 7), dtype); gt = np.empty((n, 7), dtype)
    total = ctypes.c_double(0.0)
    rc = getattr(lib(), 'gd_oracle_loss_decoded' + sfx)(
        ctypes.byref(params), int(kind), int(bool(norm_bbox)), _ptr(aux), float(out_size_factor), float(voxel_size[0]),
        float(voxel_size[1]), float(pc_range[0]), float(pc_range[1]), _ptr(pred), _ptr(target), _ptr(w), n,
        float(scale), _ptr(loss), ctypes.cast(ctypes.byref(total), ctypes.c_void_p), _ptr(gp), _ptr(gt))
    if rc != 0:
        raise RuntimeError(f'gd_oracle_loss_decoded{sfx} -> {rc}')
    return dict(loss=loss, loss_sum=total.value, grad_pred=gp, grad_target=gt)


def gd_loss_timed(pred32, target32, params, scale, out_loss, out_gp, nthreads):
    """fp32 port with caller-provided outputs (for bench.py's cpu_baseline leg)."""
    n = pred32.shape[0]
    total = ctypes.c_double(0.0)
    rc = lib().gd_oracle_loss_f32(ctypes.byref(params), _ptr(pred32), _ptr(target32), None, n,
                                  float(scale), _ptr(out_loss),
                                  ctypes.cast(ctypes.byref(total), ctypes.c_void_p), _ptr(out_gp), None,
                                  int(nthreads))
    if rc != 0:
        raise RuntimeError(f'gd_oracle_loss_f32 -> {rc}')
    return total.value


def _boxes(a, cols):
    a = np.ascontiguousarray(np.asarray(a, dtype=np.float32).reshape(-1, cols))
    return a


def nms_bev(boxes_sorted, thresh, normal=False):
    """Greedy rotated (or axis-aligned) BEV NMS on score-sorted [x1,y1,x2,y2,ry] boxes -> int64 keep."""
    b = _boxes(boxes_sorted, 5)
    keep = np.empty(b.shape[0], np.int64)
    fn = lib().rbox_oracle_nms_normal if normal else lib().rbox_oracle_nms_bev
    k = fn(_ptr(b), b.shape[0], float(thresh), _ptr(keep))
    return keep[:k].copy()


def set_trig_nudge(mode=0, seed=0):
    """Sensitivity knob of the NMS restatement: move every box's sin / cos by one ulp (1: up, 2: down, 3: pseudo-random
    per box in {-1, 0, +1}); 0 switches it off.  Process-global: always reset to 0 (see tests/test_nms_margin.py)."""
    lib().rbox_oracle_set_trig_nudge(int(mode), ctypes.c_uint32(int(seed) & 0xffffffff))


MARGIN_EDGES = (1e-7, 1e-6, 1e-5, 1e-4, 1e-3, 1e-2, 1e-1)


def nms_margin(boxes, scores, thresh, pre_max_size=None):
    """Decision margins |IoU - thresh| over the pairs the greedy scan evaluates (kept box vs later box still alive).
    Returns dict(kept, pairs, overlapping, min_margin, iou_at_min, within={edge: count})."""
    scores = np.asarray(scores, dtype=np.float32)
    order = np.argsort(-scores, kind='stable')
    if pre_max_size is not None:
        order = order[:pre_max_size]
    b = _boxes(np.asarray(boxes, np.float32)[order], 5)
    edges = np.asarray(MARGIN_EDGES, np.float64)
    counts = np.zeros(len(edges), np.int64)
    stats = np.zeros(4, np.float64)
    fn = lib().rbox_oracle_nms_margin
    fn.restype = ctypes.c_int64
    kept = fn(_ptr(b), ctypes.c_int64(b.shape[0]), ctypes.c_float(float(thresh)), _ptr(edges), len(edges), _ptr(counts),
              _ptr(stats))
    return dict(kept=int(kept), pairs=int(stats[0]), overlapping=int(stats[1]), min_margin=float(stats[2]),
                iou_at_min=float(stats[3]), within={e: int(c) for e, c in zip(MARGIN_EDGES, counts)})


def nms_mask(boxes_sorted, thresh):
    b = _boxes(boxes_sorted, 5)
    n = b.shape[0]
    mask = np.zeros((n, (n + 63) // 64), np.uint64)
    lib().rbox_oracle_nms_mask(_ptr(b), n, float(thresh), _ptr(mask))
    return mask


def nms_gpu_oracle(boxes, scores, thresh, pre_max_size=None, post_max_size=None, normal=False):
    """CPU restatement of mmdet3d `nms_gpu` (sort, pre cut, greedy, map back, post cut)."""
    scores = np.asarray(scores, dtype=np.float32)
    order = np.argsort(-scores, kind='stable')
    if pre_max_size is not None:
        order = order[:pre_max_size]
    keep = order[nms_bev(np.asarray(boxes, np.float32)[order], thresh, normal=normal)]
    if post_max_size is not None:
        keep = keep[:post_max_size]
    return keep.astype(np.int64)


def circle_nms(dets, thresh, post_max_size=83):
    """Restatement of mmdet3d's numba `circle_nms(dets, thresh, post_max_size)` (third party, absent, unpinned —
    PARITY UNPINNED), mmdet3d/core/utils/gaussian.py: order by descending score; a detection is suppressed by an
    earlier kept one iff (dx^2 + dy^2) <= thresh (float32 distance compared with the Python-float threshold).
    Ties in the score keep the lower index first (numpy's reversed argsort leaves them unspecified)."""
    dets = np.ascontiguousarray(dets, dtype=np.float32)
    x, y = dets[:, 0], dets[:, 1]
    order = np.argsort(-dets[:, 2], kind='stable')
    n = dets.shape[0]
    suppressed = np.zeros(n, bool)
    keep = []
    for _i in range(n):
        i = order[_i]
        if suppressed[i]:
            continue
        keep.append(i)
        rest = order[_i + 1:]
        dx, dy = x[i] - x[rest], y[i] - y[rest]
        dist = (dx * dx + dy * dy).astype(np.float32)          # fp32 products and sum, no fused multiply-add
        suppressed[rest[dist.astype(np.float64) <= float(thresh)]] = True
    keep = np.asarray(keep, dtype=np.int64)
    return keep if post_max_size is None else keep[:post_max_size]


def iou_bev_xyxyr(a, b):
    a, b = _boxes(a, 5), _boxes(b, 5)
    out = np.empty((a.shape[0], b.shape[0]), np.float32)
    lib().rbox_oracle_iou_bev_xyxyr(_ptr(a), a.shape[0], _ptr(b), b.shape[0], _ptr(out))
    return out


def eval_iou_bev(det, gt):
    det, gt = _boxes(det, 7), _boxes(gt, 7)
    out = np.empty((det.shape[0], gt.shape[0]), np.float32)
    lib().rbox_oracle_eval_iou_bev(_ptr(det), det.shape[0], _ptr(gt), gt.shape[0], _ptr(out))
    return out


def eval_iou_3d(det, gt, z_offset=0.5):
    det, gt = _boxes(det, 7), _boxes(gt, 7)
    out = np.empty((det.shape[0], gt.shape[0]), np.float32)
    lib().rbox_oracle_eval_iou_3d(_ptr(det), det.shape[0], _ptr(gt), gt.shape[0], float(z_offset), _ptr(out))
    return out


def eval_trans_bev(det, gt):
    """affinity.cpp:83-105: (D,cols>=2),(G,cols>=2) -> (D,G) BEV centre distance."""
    det = np.ascontiguousarray(det, np.float32); gt = np.ascontiguousarray(gt, np.float32)
    out = np.empty((det.shape[0], gt.shape[0]), np.float32)
    lib().rbox_oracle_trans_bev(_ptr(det), det.shape[0], det.shape[1], _ptr(gt), gt.shape[0], gt.shape[1], _ptr(out))
    return out


def match_coco(cost_mat, cost_thrs, is_ignore, is_crowd):
    """matcher.cpp:8-74: (D,G) costs, (T) thresholds, (G) bools -> (T,D) int32 matched gt index or -1."""
    cost = np.ascontiguousarray(cost_mat, np.float32)
    thrs = np.ascontiguousarray(cost_thrs, np.float32)
    ign = np.ascontiguousarray(is_ignore, np.uint8); crowd = np.ascontiguousarray(is_crowd, np.uint8)
    nd, ng = cost.shape
    out = np.empty((thrs.shape[0], nd), np.int32)
    lib().rbox_oracle_match_coco(_ptr(cost), _ptr(thrs), _ptr(ign), _ptr(crowd), nd, ng, thrs.shape[0], _ptr(out))
    return out


def ref_eval_path():
    hits = glob.glob(os.path.join(HERE, '_ref', 'ref_eval*.so'))
    return hits[0] if hits else None


def load_ref_eval():
    """The reference's own affinity.cpp compiled (oracle/_ref); None if it was never built."""
    path = ref_eval_path()
    if not path:
        return None
    spec = importlib.util.spec_from_file_location('ref_eval', path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod
