"""CenterPoint target assignment on the device: ground-truth boxes of a batch -> heat maps, anno_boxes, pos_inds of every task,
in two launches and one read-back (csrc/center_targets.hip).

Call surface of the reference's `CenterHeadRev.get_targets(gt_bboxes_3d, gt_labels_3d)`
(/root/reference/mmdet3d_gaussian/models/dense_heads/gd_centerpoint_head.py:65-81, get_targets_single :83-156): same return
value — per task the stacked heat maps (B, C_t, H, W), the valid boxes (n_t, 9) as `cat(gravity_center, tensor[:, 3:])` rows
and their [batch, x, y] positions (n_t, 3) int64 — which is what `center_head_losses` / the head's `loss` consume.
GPU tensors only: there is no CPU path.
"""
import ctypes

import torch

from . import _lib


def center_head_get_targets(gt_bboxes_3d, gt_labels_3d, class_names, train_cfg, padded=False):
    """gt_bboxes_3d : per sample either a box object with `.tensor` (N, 7+) in bottom-centre form (LiDARInstance3DBoxes: the
                   gravity centre z + h/2 is taken here, as `.gravity_center` does, :85-87) or a plain (N, 7+) tensor whose rows
                   already are `cat(gravity_center, tensor[:, 3:])`;
    gt_labels_3d : per sample (N,) integer labels (global class index; task t owns the next len(class_names[t]) of them;
                   any other value, e.g. -1, is ignored);
    class_names  : per task the list of its class names (only the lengths are used), or the lengths themselves;
    train_cfg    : 'grid_size', 'point_cloud_range', 'voxel_size', 'out_size_factor', 'gaussian_overlap', 'min_radius'.
    Returns (heatmaps, anno_boxes, batch_pos_inds), lists over tasks.
    padded=True: no read-back — returns (heatmaps, anno (N, C), pos (N, 3) int64, task_start (T+1,) int64 on the device): task t
    owns the rows [task_start[t], task_start[t+1]) of the two shared arrays, N = all boxes of the batch (rows past
    task_start[T] are undefined).  `center_head_losses(..., rows=task_start)` consumes exactly that."""
    B = len(gt_bboxes_3d)
    if B == 0 or len(gt_labels_3d) != B:
        raise RuntimeError(f'center_head_get_targets: {B} box sets and {len(gt_labels_3d)} label sets')
    counts = [c if isinstance(c, int) else len(c) for c in class_names]
    T = len(counts)
    objs = [hasattr(b, 'tensor') and not isinstance(b, torch.Tensor) for b in gt_bboxes_3d]
    if any(objs) != all(objs):
        raise RuntimeError('center_head_get_targets: box objects and plain tensors mixed in one batch')
    rows = [b.tensor if o else b for b, o in zip(gt_bboxes_3d, objs)]
    if not rows[0].is_cuda:
        raise RuntimeError('center_head_get_targets: the MI355X implementation has no CPU path')
    lib = _lib.load_extras()
    dev = rows[0].device
    cols = rows[0].shape[1]
    sizes = [int(r.shape[0]) for r in rows]
    for r, l in zip(rows, gt_labels_3d):
        if r.dim() != 2 or r.shape[1] != cols or l.shape[0] != r.shape[0]:
            raise RuntimeError('center_head_get_targets: boxes must be (N, C) with one label each, the same C in every sample')
    total = sum(sizes)
    if total > lib.center_targets_max_boxes():
        raise RuntimeError(f'center_head_get_targets: {total} boxes in the batch (the kernel sorts at most {lib.center_targets_max_boxes()})')
    if B > 64 or T > 40:
        raise RuntimeError('center_head_get_targets: at most 64 samples and 40 tasks')
    osf = train_cfg['out_size_factor']
    grid = train_cfg['grid_size']
    H, W = int(grid[0]) // int(osf), int(grid[1]) // int(osf)     # feature_map_size[0] / [1] as the reference uses them (:94, :115)
    d = _lib.CenterTargetsDesc()
    d.num_tasks, d.batch, d.height, d.width = T, B, H, W
    d.total, d.box_cols, d.bottom_center = total, cols, int(all(objs))
    d.min_radius = int(train_cfg['min_radius'])
    for t, c in enumerate(counts):
        d.classes[t] = c
    off = 0
    for b, n in enumerate(sizes):
        d.sample_start[b] = off
        off += n
    d.sample_start[B] = off
    pc, vs = train_cfg['point_cloud_range'], train_cfg['voxel_size']
    d.pc_range = (ctypes.c_float * 2)(float(pc[0]), float(pc[1]))
    d.voxel_size = (ctypes.c_float * 2)(float(vs[0]), float(vs[1]))
    d.out_size_factor = float(osf)
    d.gaussian_overlap = float(train_cfg['gaussian_overlap'])
    with torch.cuda.device(dev):
        boxes = torch.cat([r.detach() for r in rows], dim=0).float().contiguous() if total else torch.zeros((0, cols), device=dev)
        labels = torch.cat([l.reshape(-1) for l in gt_labels_3d], dim=0).to(torch.int64).contiguous() if total else \
            torch.zeros(0, dtype=torch.int64, device=dev)
        heat = torch.zeros(B * sum(counts) * H * W, dtype=torch.float32, device=dev)       # ONE fill for every task's maps
        anno = torch.empty((total, cols), dtype=torch.float32, device=dev)
        pos = torch.empty((total, 3), dtype=torch.int64, device=dev)
        start = torch.empty(T + 1, dtype=torch.int64, device=dev)
        ws = torch.empty(lib.center_targets_workspace_bytes(total), dtype=torch.uint8, device=dev)
        _lib.check(lib.center_targets_build(ctypes.byref(d), boxes.data_ptr(), labels.data_ptr(), ws.data_ptr(), heat.data_ptr(),
                                            anno.data_ptr(), pos.data_ptr(), start.data_ptr(),
                                            torch.cuda.current_stream().cuda_stream), 'center_targets_build')
    if padded:
        heatmaps, ho = [], 0
        for c in counts:
            heatmaps.append(heat[ho:ho + B * c * H * W].view(B, c, H, W))
            ho += B * c * H * W
        return heatmaps, anno, pos, start
    st = start.tolist()          # the one sync: T + 1 data-dependent row offsets
    heatmaps, anno_boxes, pos_inds, ho = [], [], [], 0
    for t, c in enumerate(counts):
        heatmaps.append(heat[ho:ho + B * c * H * W].view(B, c, H, W))
        ho += B * c * H * W
        anno_boxes.append(anno[st[t]:st[t + 1]])
        pos_inds.append(pos[st[t]:st[t + 1]])
    return heatmaps, anno_boxes, pos_inds
