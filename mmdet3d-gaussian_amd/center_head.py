"""`CenterGDHead.loss` end to end on the device: target assignment, heat-map loss and regression losses of all tasks.

The reference's method (/root/reference/mmdet3d_gaussian/models/dense_heads/gd_centerpoint_head.py:390-441) composed from this
package's pieces — `center_head_get_targets` (:400-401), `center_head_heatmap_loss` (:403-411), `center_head_losses`
(:413-434) — with TWO host read-backs for the whole batch (the per-task row offsets of the targets and the per-task num_pos)
where the reference loops over samples, tasks and boxes in Python and syncs per box and per task.
"""
from .center_targets import center_head_get_targets
from .head_loss import center_head_losses
from .heat_loss import center_head_heatmap_loss


def center_gd_head_loss(loss_cls, loss_bbox, loss_gd, bbox_coder, class_names, train_cfg, gt_bboxes_3d, gt_labels_3d, preds_dicts,
                        static=False):
    """loss_cls / loss_bbox / loss_gd : the head's loss modules (or config dicts for the first two): GaussianFocalLoss, L1Loss,
                                      this package's GDLoss;
    bbox_coder  : CenterPointBBoxYawCoder;  class_names: per task its class names;  train_cfg: the head's train_cfg
                  (grid_size, point_cloud_range, voxel_size, out_size_factor, gaussian_overlap, min_radius, code_weights);
    gt_bboxes_3d, gt_labels_3d : the batch's ground truth (see center_head_get_targets);
    preds_dicts : per task the head outputs (a dict, or the reference's one-element list of it) with 'heatmap' LOGITS and the
                  regression maps.  Unlike the reference (:405) the heat maps are NOT replaced by their clipped sigmoid.
    static=True : NO read-back at all — the targets' row offsets and num_pos stay on the device and the regression kernels read
                  them there; with ground truth padded to a fixed number of rows (label -1 for the padding) the whole method is
                  one stream-ordered sequence that `GraphedStep` can capture (forward and backward) and replay.
    Returns the reference's loss_dict: 'task{t}.loss_heatmap', 'task{t}.loss_l1', 'task{t}.loss_gd'."""
    pds = [p[0] if isinstance(p, (list, tuple)) else p for p in preds_dicts]
    if static:
        heatmaps, anno, pos, rows = center_head_get_targets(gt_bboxes_3d, gt_labels_3d, class_names, train_cfg, padded=True)
        hm_losses, num_pos = center_head_heatmap_loss(loss_cls, [p['heatmap'] for p in pds], heatmaps)
        reg = center_head_losses(loss_gd, loss_bbox, bbox_coder, pds, pos, anno, num_pos, train_cfg['code_weights'], rows=rows)
    else:
        heatmaps, anno_boxes, pos_inds = center_head_get_targets(gt_bboxes_3d, gt_labels_3d, class_names, train_cfg)
        hm_losses, num_pos = center_head_heatmap_loss(loss_cls, [p['heatmap'] for p in pds], heatmaps)
        npos = num_pos.tolist()                      # avg_factor of the regression losses (:408, :431-434)
        # code_weights as configured: one weight per L1 column (sin, cos(, vx, vy)), :426-428
        reg = center_head_losses(loss_gd, loss_bbox, bbox_coder, pds, pos_inds, anno_boxes, npos, train_cfg['code_weights'])
    hm = hm_losses.unbind(0)
    loss_dict = {}
    for t, (l1, gd) in enumerate(reg):
        loss_dict[f'task{t}.loss_heatmap'] = hm[t]
        loss_dict[f'task{t}.loss_l1'] = l1
        loss_dict[f'task{t}.loss_gd'] = gd
    return loss_dict
