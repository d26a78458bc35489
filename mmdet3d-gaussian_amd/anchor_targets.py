"""Target assignment of the anchor heads on the device: ground truth of a batch -> labels, weights, regression targets and
direction bins of every anchor, in two launches (csrc/anchor_targets.hip).

Call surface of what `GDAnchor3DHead.loss` obtains from `self.anchor_target_3d(...)`
(/root/reference/mmdet3d_gaussian/models/dense_heads/gd_anchor3d_head.py:206-222; mmdet3d's AnchorTrainMixin with mmdet's
MaxIoUAssigner + PseudoSampler, third party) for one feature level: the six per-anchor target arrays in the (h, w, a) order that
`loss_single` reshapes them to, and the two counts.  GPU tensors only: there is no CPU path.
"""
import ctypes

import torch

from . import _lib


def _get(cfg, key, default=None):
    if isinstance(cfg, dict):
        return cfg.get(key, default)
    return getattr(cfg, key, default)


def _assigner(cfg):
    kind = _get(cfg, 'type', type(cfg).__name__)
    if kind != 'MaxIoUAssigner':
        raise RuntimeError(f'anchor_head_get_targets: assigner {kind!r}; the reference configures MaxIoUAssigner')
    calc = _get(cfg, 'iou_calculator', dict(type='BboxOverlapsNearest3D'))
    ckind = _get(calc, 'type', type(calc).__name__)
    if ckind != 'BboxOverlapsNearest3D':
        raise RuntimeError(f'anchor_head_get_targets: iou_calculator {ckind!r}; the reference configures BboxOverlapsNearest3D')
    neg = _get(cfg, 'neg_iou_thr')
    if isinstance(neg, bool) or not isinstance(neg, float):
        # mmdet's MaxIoUAssigner applies the negative rule only `if isinstance(self.neg_iou_thr, float)`: with an int (e.g. 0) those
        # anchors stay at -1 (ignored) there, while the kernel would apply it as a threshold; the tuple form is not implemented
        raise RuntimeError('anchor_head_get_targets: neg_iou_thr must be a Python float (mmdet skips the negative rule for an int: '
                           'write 0.0, not 0; the tuple form is not implemented)')
    if _get(cfg, 'ignore_iof_thr', -1) > 0:
        raise RuntimeError('anchor_head_get_targets: ignore_iof_thr > 0 (ignore boxes) is not implemented; the reference configures -1')
    return (float(_get(cfg, 'pos_iou_thr')), float(neg), float(_get(cfg, 'min_pos_iou', 0.0)),
            bool(_get(cfg, 'match_low_quality', True)), bool(_get(cfg, 'gt_max_assign_all', True)))


def anchor_head_get_targets(anchors, gt_bboxes, gt_labels, assigner, num_classes, assign_per_class=True, pos_weight=-1, dir_offset=0.0,
                            num_dir_bins=2, sampling=False, padded=False):
    """anchors    : one level's grid as the generator returns it with reshape_out=False: (H, W, S, R, 7), a leading 1 allowed
                 ((1, H, W, S, R, 7)); S size classes, R rotations; with a single assigner also the flat (N, 7) list of reshape_out=True;
    gt_bboxes  : per sample a box object with `.tensor` (G, 7+) or a plain (G, 7+) tensor [x, y, z, dx, dy, dz, yaw] (bottom centre,
                 as LiDARInstance3DBoxes stores it);  gt_labels: per sample (G,) integer labels;
    assigner   : train_cfg.assigner — a list of S MaxIoUAssigner configs (dicts or objects with pos_iou_thr, neg_iou_thr, min_pos_iou;
                 iou_calculator BboxOverlapsNearest3D), one per size class, or a single one for all anchors;
    assign_per_class : with the list form, assigner q sees only the boxes labelled q (the head's `assign_per_class`);
    pos_weight : train_cfg.pos_weight;  dir_offset: the head's;  sampling: must be False (PseudoSampler, as with a sigmoid focal loss).
    Returns (labels, label_weights, bbox_targets, bbox_weights, dir_targets, dir_weights, num_total_pos, num_total_neg): the first six
    (B, N) / (B, N, 7) with N = H*W*S*R in the (h, w, a) order of the head's maps (what `images_to_levels` yields for the one level);
    the counts as the mixin computes them, `sum(max(count_b, 1))` — one read-back of 2 B integers.
    padded=True: no read-back; the seventh value is the (B, 2) int32 device tensor of per-sample (positives, negatives)."""
    if sampling:
        raise RuntimeError('anchor_head_get_targets: sampling=True (a random sampler) is not implemented; the reference heads use PseudoSampler')
    B = len(gt_bboxes)
    if B == 0 or len(gt_labels) != B:
        raise RuntimeError(f'anchor_head_get_targets: {B} box sets and {len(gt_labels)} label sets')
    if anchors.dim() == 6 and anchors.shape[0] == 1:
        anchors = anchors[0]
    single = not isinstance(assigner, (list, tuple))
    if anchors.dim() == 2 and anchors.shape[-1] == 7 and single:
        # reshape_out=True (the car-only configs): a flat (N, 7) list in the head's order; one assigner sees all of it, so the grid's shape
        # does not matter
        anchors = anchors.reshape(-1, 1, 1, 1, 7)
    if anchors.dim() != 5 or anchors.shape[-1] != 7:
        raise RuntimeError(f'anchor_head_get_targets: anchors {tuple(anchors.shape)} are not (H, W, sizes, rotations, 7)')
    if not anchors.is_cuda:
        raise RuntimeError('anchor_head_get_targets: the MI355X implementation has no CPU path')
    H, W, S, R, _ = anchors.shape
    cfgs = [_assigner(assigner)] if single else [_assigner(c) for c in assigner]
    if not single and len(cfgs) != S:
        raise RuntimeError(f'anchor_head_get_targets: {len(cfgs)} assigners for {S} anchor sizes')
    if S > 16 or B > 64:
        raise RuntimeError('anchor_head_get_targets: at most 16 anchor sizes and 64 samples')
    if any(c[3:] != cfgs[0][3:] for c in cfgs):
        raise RuntimeError('anchor_head_get_targets: match_low_quality / gt_max_assign_all must agree across the assigners')
    lib = _lib.load_extras()
    dev = anchors.device
    rows = [(b.tensor if (hasattr(b, 'tensor') and not isinstance(b, torch.Tensor)) else b) for b in gt_bboxes]
    sizes = [int(r.shape[0]) for r in rows]
    for r, l in zip(rows, gt_labels):
        if r.dim() != 2 or r.shape[1] < 7 or l.shape[0] != r.shape[0]:
            raise RuntimeError('anchor_head_get_targets: boxes must be (G, 7+) with one label each')
    if max(sizes) > lib.anchor_targets_max_gt():
        raise RuntimeError(f'anchor_head_get_targets: {max(sizes)} boxes in a sample (at most {lib.anchor_targets_max_gt()})')
    d = _lib.AnchorTargetsDesc()
    d.batch, d.cells, d.num_sizes, d.num_rots, d.num_classes = B, H * W, S, R, int(num_classes)
    d.num_assigners = len(cfgs)
    d.assign_per_class = int(bool(assign_per_class) and not single)
    d.match_low_quality, d.gt_max_assign_all = int(cfgs[0][3]), int(cfgs[0][4])
    d.num_dir_bins = int(num_dir_bins)
    off = 0
    for b, n in enumerate(sizes):
        d.gt_start[b] = off
        off += n
    d.gt_start[B] = off
    for q, c in enumerate(cfgs):
        d.pos_iou_thr[q], d.neg_iou_thr[q], d.min_pos_iou[q] = c[0], c[1], c[2]
    d.pos_weight, d.dir_offset = float(pos_weight), float(dir_offset)
    N = H * W * S * R
    with torch.cuda.device(dev):
        an = anchors.detach()
        an = an if (an.dtype == torch.float32 and an.is_contiguous()) else an.float().contiguous()
        if off:
            boxes = torch.cat([r.detach()[:, :7].to(dev) for r in rows], dim=0).float().contiguous()
            labs = torch.cat([l.reshape(-1).to(dev) for l in gt_labels], dim=0).to(torch.int64).contiguous()
        else:
            boxes, labs = None, None
        labels = torch.empty((B, N), dtype=torch.int64, device=dev)
        label_weights = torch.empty((B, N), dtype=torch.float32, device=dev)
        bbox_targets = torch.empty((B, N, 7), dtype=torch.float32, device=dev)
        bbox_weights = torch.empty((B, N, 7), dtype=torch.float32, device=dev)
        dir_targets = torch.empty((B, N), dtype=torch.int64, device=dev)
        dir_weights = torch.empty((B, N), dtype=torch.float32, device=dev)
        counts = torch.empty((B, 2), dtype=torch.int32, device=dev)
        ws = torch.empty(lib.anchor_targets_workspace_bytes(len(cfgs), off), dtype=torch.uint8, device=dev)
        _lib.check(lib.anchor_targets_build(ctypes.byref(d), an.data_ptr(), None if boxes is None else boxes.data_ptr(),
                                            None if labs is None else labs.data_ptr(), ws.data_ptr(), labels.data_ptr(),
                                            label_weights.data_ptr(), bbox_targets.data_ptr(), bbox_weights.data_ptr(), dir_targets.data_ptr(),
                                            dir_weights.data_ptr(), counts.data_ptr(), torch.cuda.current_stream().cuda_stream),
                   'anchor_targets_build')
    if padded:
        return labels, label_weights, bbox_targets, bbox_weights, dir_targets, dir_weights, counts
    c = counts.tolist()          # the one sync
    return (labels, label_weights, bbox_targets, bbox_weights, dir_targets, dir_weights,
            sum(max(p, 1) for p, _ in c), sum(max(n, 1) for _, n in c))
