"""Extras OUTSIDE SURVEY.md §8 — frozen since round 3, parity unpinned (DESIGN_EXTRAS.md): the head-level slices built around the
hot path inside the reference's own head files (CenterPoint / anchor-head / PV-RCNN inference around the NMS, target assignment,
focal / heat-map losses).  An import shim: the package's top level exports the §8 surface only; these names live here.

  center_head_get_bboxes, select_best   CenterPoint inference slice that ends in rotated NMS (gd_centerpoint_head.py:218-361)
  center_head_get_targets               CenterPoint target assignment (gd_centerpoint_head.py:65-156)
  center_head_heatmap_loss              clip_sigmoid + GaussianFocalLoss of all tasks in one pass (:403-411)
  center_gd_head_loss                   CenterGDHead.loss end to end (:390-441)
  anchor3d_range_anchors                the anchor heads' grid (mmdet3d's Anchor3DRangeGenerator, one level)
  anchor_head_get_targets               the anchor heads' target assignment (mmdet3d's anchor_target_3d, gd_anchor3d_head.py:206-214)
  anchor_head_cls_dir_loss              focal classification + direction losses in one pass (gd_anchor3d_head.py:84-92, :143-149)
  anchor_head_get_bboxes                the anchor heads' inference slice around the NMS (mmdet3d's, inherited by GDAnchor3DHead)
  gd_anchor_head_loss[_single]          GDAnchor3DHead.loss / loss_single end to end (gd_anchor3d_head.py:62-240)
  pvrcnn_head_get_bboxes                PVRCNNBboxHead.get_bboxes around its NMS (pvrcnn_bbox_head.py:352-480)
"""
from .anchor_cls import anchor_head_cls_dir_loss
from .anchor_head import gd_anchor_head_loss, gd_anchor_head_loss_single
from .anchor_infer import anchor_head_get_bboxes
from .anchor_targets import anchor_head_get_targets
from .anchors import anchor3d_range_anchors
from .center_head import center_gd_head_loss
from .center_infer import center_head_get_bboxes, select_best
from .center_targets import center_head_get_targets
from .heat_loss import center_head_heatmap_loss
from .pvrcnn_infer import pvrcnn_head_get_bboxes

__all__ = ['anchor_head_cls_dir_loss', 'gd_anchor_head_loss', 'gd_anchor_head_loss_single', 'anchor_head_get_bboxes', 'anchor_head_get_targets',
           'anchor3d_range_anchors', 'center_gd_head_loss', 'center_head_get_bboxes', 'select_best', 'center_head_get_targets',
           'center_head_heatmap_loss', 'pvrcnn_head_get_bboxes']
