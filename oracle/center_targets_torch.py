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
The head class needs mmdet3d / mmcv to import, so this restates :83-156 step by step in the same fp32 operation order (boxes arrive as the
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
    """:83-156 for one sample.  gt9 (N, 9) = cat(gravity_center, tensor[:, 3:]); class_counts[t] = len(class_names[t]).
    Per task: boxes are regrouped class by class (index order inside a class, :97-113), mapped to cells by float division and
    truncation (:118-126), filtered, and every survivor stamps a Gaussian of its radius into its class plane (:131-141)."""
    pc_range = torch.tensor(train_cfg['point_cloud_range'])
    voxel = torch.tensor(train_cfg['voxel_size'])
    osf = train_cfg['out_size_factor']
    fmap = torch.tensor(train_cfg['grid_size'])[:2] // osf              # rows, columns of a plane as the reference uses them
    out_heat, out_boxes, out_cells = [], [], []
    first_label = 0
    for n_cls in class_counts:
        members = [torch.where(gt_labels == first_label + c)[0] for c in range(n_cls)]
        boxes = torch.cat([gt9[m] for m in members], dim=0)
        cls_of = torch.cat([torch.full((m.numel(),), c, dtype=torch.long) for c, m in enumerate(members)])
        first_label += n_cls
        plane = gt9.new_zeros((n_cls, int(fmap[0]), int(fmap[1])))
        wide = boxes[:, 3] / voxel[0] / osf
        long_ = boxes[:, 4] / voxel[1] / osf
        col = ((boxes[:, 0] - pc_range[0]) / voxel[0] / osf).long()
        row = ((boxes[:, 1] - pc_range[1]) / voxel[1] / osf).long()
        keep = wide.gt(0) & long_.gt(0) & col.ge(0) & col.lt(fmap[1]) & row.ge(0) & row.lt(fmap[0])
        cells = torch.stack((col, row), dim=-1)
        for k in keep.nonzero(as_tuple=True)[0]:
            r = gaussian_radius((long_[k], wide[k]), min_overlap=train_cfg['gaussian_overlap'])
            draw_heatmap_gaussian(plane[cls_of[k]], cells[k], max(train_cfg['min_radius'], int(r)))
        out_heat.append(plane)
        out_boxes.append(boxes[keep])
        out_cells.append(cells[keep])
    return out_heat, out_boxes, out_cells


def get_targets(gt9_list, gt_labels_list, class_counts, train_cfg):
    """:65-81: per task the stacked heat maps (B, C_t, H, W), the concatenated boxes and [batch, x, y] positions."""
    per_sample = [get_targets_single(b, l, class_counts, train_cfg) for b, l in zip(gt9_list, gt_labels_list)]
    T = len(class_counts)
    heatmaps = [torch.stack([s[0][t] for s in per_sample]) for t in range(T)]
    anno_boxes = [torch.cat([s[1][t] for s in per_sample], dim=0) for t in range(T)]
    pos_inds = []
    for t in range(T):
        rows = [torch.cat((torch.full((s[2][t].shape[0], 1), b, dtype=s[2][t].dtype), s[2][t]), dim=-1) for b, s in enumerate(per_sample)]
        pos_inds.append(torch.cat(rows, dim=0))
    return heatmaps, anno_boxes, pos_inds
