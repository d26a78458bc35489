"""mmdet3d-gaussian hot path for AMD MI355X (gfx950).

Exposes the reference's call surface for the one path this package accelerates:
  * ``GDLoss``                 — /root/reference/mmdet3d_gaussian/models/losses/gaussian_distance_loss.py:251
  * ``LOSSES`` / ``build_loss``— the registry verbs heads use to build it from config dicts
  * ``nms_gpu`` & friends      — the mmdet3d iou3d ops the reference imports (gd_centerpoint_head.py:9)
  * ``iou_bev`` / ``iou_3d``   — GPU counterparts of ops/eval/affinity.cpp
  * ``sharded``                — pair-sharded multi-GPU evaluation (one process per GPU, RCCL)
  * ``coders`` / ``head_loss`` — the bbox coders either side of the loss and the two head-level loss slices with the
                                 decode fused into the kernel (gd_anchor3d_head.py:95-141, gd_centerpoint_head.py:413-434)
  * ``GraphedStep``            — a launch-bound loss slice, forward and backward, captured once as a hipGraph and replayed
  * ``center_gd_head_loss``    — CenterGDHead.loss end to end (targets + heat-map loss + regression losses, :390-441)
  * ``center_head_heatmap_loss`` — clip_sigmoid + GaussianFocalLoss of all tasks in one pass (:403-411)
  * ``center_head_get_targets``— CenterPoint target assignment (heat maps, anno_boxes, pos_inds: gd_centerpoint_head.py:65-156)
  * ``gd_anchor_head_loss``    — GDAnchor3DHead.loss end to end: target assignment + the three losses (gd_anchor3d_head.py:167-240)
  * ``gd_anchor_head_loss_single`` — GDAnchor3DHead.loss_single end to end (gd_anchor3d_head.py:62-161)
  * ``pvrcnn_head_get_bboxes`` — PVRCNNBboxHead.get_bboxes around its NMS (pvrcnn_bbox_head.py:352-480): decode launch + one batched class NMS
  * ``anchor3d_range_anchors`` — the anchor heads' grid (mmdet3d's [Aligned]Anchor3DRangeGenerator, one level, reshape_out=False): static data
  * ``anchor_head_get_targets`` — the anchor heads' target assignment (mmdet3d's anchor_target_3d, called at gd_anchor3d_head.py:206-214)
  * ``anchor_head_cls_dir_loss`` — the anchor heads' focal classification + direction losses in one pass (gd_anchor3d_head.py:84-92, :143-149)
  * ``anchor_head_get_bboxes`` — the anchor heads' inference slice (mmdet3d's, inherited by GDAnchor3DHead) around the NMS
  * ``center_head_get_bboxes`` — the CenterPoint inference slice that ends in rotated NMS (gd_centerpoint_head.py:218-361)
All arithmetic runs in hand-written HIP kernels reached through the C ABI of include/gd3d.h
(libgd3d.so, built in-tree by ``build.py``).  There is no CPU fallback.
"""
from . import build as _build_mod
from ._lib import load as load_library, lib_path
from .gd_loss import GDLoss, make_params
from .iou3d import (box3d_multiclass_nms, boxes_iou_bev, circle_nms, iou_3d, iou_bev, multi_class_nms, multi_class_nms_batch, nms_gpu, nms_gpu_batched,
                    nms_gpu_multi, nms_normal_gpu, xywhr2xyxyr)
from .registry import LOSSES, Registry, build_loss
from . import sharded
from .coders import CenterPointBBoxCoderRev, CenterPointBBoxYawCoder, DeltaXYZWLHRBBoxCoder, PointBBoxYawCoder
from .center_infer import center_head_get_bboxes, select_best
from .graphed import GraphedStep
from .anchor_infer import anchor_head_get_bboxes
from .anchor_cls import anchor_head_cls_dir_loss
from .anchor_targets import anchor_head_get_targets
from .anchors import anchor3d_range_anchors
from .pvrcnn_infer import pvrcnn_head_get_bboxes
from .anchor_head import gd_anchor_head_loss, gd_anchor_head_loss_single
from .center_targets import center_head_get_targets
from .heat_loss import center_head_heatmap_loss
from .center_head import center_gd_head_loss
from .evaluation import match_coco, trans_bev
from .scatter import Scatter, scatter_index, scatter_reduce
from .head_loss import (anchor_decoded_gd_loss, anchor_head_bbox_loss, anchor_head_decoded_loss,
                        anchor_head_decoded_loss_fused, center_head_gd_loss, center_head_losses)


def build(force=False, verbose=False):
    """Compile csrc/*.hip for gfx950 into mmdet3d-gaussian_amd/libgd3d.so, and csrc/torch_node.cpp (GDLoss's autograd node,
    host C++ above the C ABI) into mmdet3d-gaussian_amd/_gd3d_node.so.  Returns the library's path."""
    path = _build_mod.build(force=force, verbose=verbose)
    _build_mod.build_node(force=force, verbose=verbose)
    return path


__all__ = ['GDLoss', 'LOSSES', 'Registry', 'build_loss', 'make_params', 'nms_gpu', 'nms_normal_gpu', 'nms_gpu_batched', 'nms_gpu_multi', 'multi_class_nms', 'multi_class_nms_batch', 'box3d_multiclass_nms', 'circle_nms',
           'boxes_iou_bev', 'iou_bev', 'iou_3d', 'xywhr2xyxyr', 'sharded', 'build', 'load_library', 'lib_path',
           'CenterPointBBoxCoderRev', 'CenterPointBBoxYawCoder', 'DeltaXYZWLHRBBoxCoder', 'PointBBoxYawCoder', 'center_head_get_bboxes', 'anchor_head_get_bboxes', 'anchor_head_cls_dir_loss', 'anchor_head_get_targets', 'anchor3d_range_anchors', 'pvrcnn_head_get_bboxes', 'gd_anchor_head_loss_single', 'gd_anchor_head_loss', 'select_best', 'GraphedStep', 'center_head_get_targets', 'center_head_heatmap_loss', 'center_gd_head_loss', 'anchor_decoded_gd_loss', 'anchor_head_decoded_loss',
           'anchor_head_decoded_loss_fused', 'anchor_head_bbox_loss', 'center_head_gd_loss', 'center_head_losses', 'Scatter', 'scatter_index', 'scatter_reduce',
           'trans_bev', 'match_coco']
