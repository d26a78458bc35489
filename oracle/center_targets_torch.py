"""oracle/center_targets_torch.py — TEST INFRASTRUCTURE, NOT PRODUCT CODE.

CPU torch / numpy restatement of the CenterPoint target assignment that produces the inputs of the head's loss slice
(heatmaps, anno_boxes, pos_inds):
  get_targets          /root/reference/mmdet3d_gaussian/models/dense_heads/gd_centerpoint_head.py:65-81
  get_targets_single   gd_centerpoint_head.py:83-156
with the two helpers it imports from mmdet3d (third party, absent here, version not pinned by the reference: restated from the
published mmdet3d 0.x `core/utils/gaussian.py` — PARITY UNPINNED):
  gaussian_radius(det_size, min_overlap)     the CornerNet three-case radius, evaluated on 0-dim float32 tensors
  gaussian_2d / draw_heatmap_gaussian        float64 numpy Gaussian of sigma = diameter / 6, values below eps * max zeroed,
                                             cast to float32, element-wise max into the heat map window
The head class needs mmdet3d / mmcv to import, so this follows the text of :83-156 statement by statement (boxes arrive as the
(N, 9) rows `cat(gravity_center, tensor[:, 3:])` of :85-87; a task's classes are the label range [flag, flag + len(class_names))).
Never imported by the product package."""
import numpy as np
import torch


def gaussian_radius(det_size, min_overlap=0.5):
    height, width = det_size
    a1 = 1
    b1 = (height + width)
    c1 = width * height * (1 - min_overlap) / (1 + min_overlap)
    sq1 = torch.sqrt(b1 ** 2 - 4 * a1 * c1)
    r1 = (b1 + sq1) / 2
    a2 = 4
    b2 = 2 * (height + width)
    c2 = (1 - min_overlap) * width * height
    sq2 = torch.sqrt(b2 ** 2 - 4 * a2 * c2)
    r2 = (b2 + sq2) / 2
    a3 = 4 * min_overlap
    b3 = -2 * min_overlap * (height + width)
    c3 = (min_overlap - 1) * width * height
    sq3 = torch.sqrt(b3 ** 2 - 4 * a3 * c3)
    r3 = (b3 + sq3) / 2
    return min(r1, r2, r3)


def gaussian_2d(shape, sigma=1):
    m, n = [(ss - 1.) / 2. for ss in shape]
    y, x = np.ogrid[-m:m + 1, -n:n + 1]
    h = np.exp(-(x * x + y * y) / (2 * sigma * sigma))
    h[h < np.finfo(h.dtype).eps * h.max()] = 0
    return h


def draw_heatmap_gaussian(heatmap, center, radius, k=1):
    diameter = 2 * radius + 1
    gaussian = gaussian_2d((diameter, diameter), sigma=diameter / 6)
    x, y = int(center[0]), int(center[1])
    height, width = heatmap.shape[0:2]
    left, right = min(x, radius), min(width - x, radius + 1)
    top, bottom = min(y, radius), min(height - y, radius + 1)
    masked_heatmap = heatmap[y - top:y + bottom, x - left:x + right]
    masked_gaussian = torch.from_numpy(gaussian[radius - top:radius + bottom, radius - left:radius + right]).to(torch.float32)
    if min(masked_gaussian.shape) > 0 and min(masked_heatmap.shape) > 0:
        torch.max(masked_heatmap, masked_gaussian * k, out=masked_heatmap)
    return heatmap


def get_targets_single(gt9, gt_labels, class_counts, train_cfg):
    """:83-156 for one sample.  gt9 (N, 9) = cat(gravity_center, tensor[:, 3:]); class_counts[t] = len(class_names[t])."""
    grid_size = torch.tensor(train_cfg['grid_size'])
    pc_range = torch.tensor(train_cfg['point_cloud_range'])
    voxel_size = torch.tensor(train_cfg['voxel_size'])
    osf = train_cfg['out_size_factor']
    feature_map_size = grid_size[:2] // osf
    task_masks, flag = [], 0
    for n in class_counts:
        task_masks.append([torch.where(gt_labels == i + flag) for i in range(n)])
        flag += n
    task_boxes, task_classes, flag2 = [], [], 0
    for mask in task_masks:
        task_boxes.append(torch.cat([gt9[m] for m in mask], dim=0))
        task_classes.append(torch.cat([gt_labels[m] + 1 - flag2 for m in mask]).long())
        flag2 += len(mask)
    heatmaps, anno_boxes, pos_inds = [], [], []
    for idx, n in enumerate(class_counts):
        heatmap = gt9.new_zeros((n, int(feature_map_size[0]), int(feature_map_size[1])))
        width = task_boxes[idx][:, 3] / voxel_size[0] / osf
        length = task_boxes[idx][:, 4] / voxel_size[1] / osf
        x_ind = ((task_boxes[idx][:, 0] - pc_range[0]) / voxel_size[0] / osf).long()
        y_ind = ((task_boxes[idx][:, 1] - pc_range[1]) / voxel_size[1] / osf).long()
        valid = width.gt(0) * length.gt(0)
        valid = valid * (x_ind.ge(0) * x_ind.lt(feature_map_size[1]))
        valid = valid * (y_ind.ge(0) * y_ind.lt(feature_map_size[0]))
        center_xy_int = torch.stack((x_ind, y_ind), dim=-1)
        for k in valid.nonzero(as_tuple=True)[0]:
            cls_id = task_classes[idx][k] - 1
            radius = gaussian_radius((length[k], width[k]), min_overlap=train_cfg['gaussian_overlap'])
            radius = max(train_cfg['min_radius'], int(radius))
            draw_heatmap_gaussian(heatmap[cls_id], center_xy_int[k], radius)
        heatmaps.append(heatmap)
        anno_boxes.append(task_boxes[idx][valid])
        pos_inds.append(center_xy_int[valid])
    return heatmaps, anno_boxes, pos_inds


def get_targets(gt9_list, gt_labels_list, class_counts, train_cfg):
    """:65-81: per task the stacked heat maps (B, C_t, H, W), the concatenated boxes and [batch, x, y] positions."""
    per_sample = [get_targets_single(b, l, class_counts, train_cfg) for b, l in zip(gt9_list, gt_labels_list)]
    heatmaps = [torch.stack(h) for h in zip(*[p[0] for p in per_sample])]
    anno_boxes = [torch.cat(a, dim=0) for a in zip(*[p[1] for p in per_sample])]
    batch_pos_inds = []
    for pos_ind in zip(*[p[2] for p in per_sample]):
        prefix = torch.cat([ind.new_full((ind.size(0), 1), b) for b, ind in enumerate(pos_ind)], dim=0)
        batch_pos_inds.append(torch.cat((prefix, torch.cat(pos_ind, dim=0)), dim=-1))
    return heatmaps, anno_boxes, batch_pos_inds
