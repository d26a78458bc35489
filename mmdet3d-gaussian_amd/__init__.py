"""mmdet3d-gaussian hot path for AMD MI355X (gfx950) — the rows of SURVEY.md §8 and nothing else at the top level.

  * ``GDLoss``                 — §8 a1-a9: /root/reference/mmdet3d_gaussian/models/losses/gaussian_distance_loss.py:251
  * ``LOSSES`` / ``build_loss``— §8 b: the registry verbs heads use to build it from config dicts
  * ``nms_gpu`` & friends      — §8 aN: the mmdet3d iou3d ops the reference imports (gd_centerpoint_head.py:9, pvrcnn_bbox_head.py:12)
                                 and the multi-class loops around them at the call sites
  * ``iou_bev`` / ``iou_3d`` / ``trans_bev`` / ``boxes_iou_bev`` / ``match_coco`` — §8 aI, f3: ops/eval/affinity.cpp, matcher.cpp
  * ``sharded``                — §8 e: pair-sharded multi-GPU evaluation (one process per GPU, RCCL)
  * coders + ``anchor_*_loss`` / ``center_head_*`` loss slices — §8 f1, f2: the bbox coders either side of the loss and the two
                                 head-level regression slices with the decode fused into the kernel
                                 (gd_anchor3d_head.py:95-161, gd_centerpoint_head.py:413-434)
  * ``Scatter`` / ``scatter_index`` / ``scatter_reduce`` — §8 f4: ops/voxel/scatter.py
  * ``GraphedStep``            — a launch-bound loss slice, forward and backward, captured once as a hipGraph and replayed
Everything outside §8 that earlier rounds built (frozen, DESIGN_EXTRAS.md) lives in ``mmdet3d_gaussian_amd.extras``.

All arithmetic runs in hand-written HIP kernels reached through the C ABI of include/gd3d.h (libgd3d.so, built in-tree by
``build.py``); CPU tensors take the library's own `_cpu` twins (the kernels' per-pair source compiled for the host).  Nothing
substitutes for the library: a GPU tensor never takes a CPU path.  The host layer above the C ABI is Python
(torch.autograd.Function + ctypes, ``_pynode.py``); ``csrc/torch_node.cpp`` is an optional C++ accelerator with the same surface
(``GD3D_HOST=python|cpp``, ``_lib.host_glue()``).
"""
from . import build as _build_mod
from ._lib import host_glue, load as load_library, lib_path, set_host_glue
from .gd_loss import GDLoss, make_params
from .iou3d import (box3d_multiclass_nms, boxes_iou_bev, circle_nms, iou_3d, iou_bev, multi_class_nms, multi_class_nms_batch, nms_gpu, nms_gpu_batched,
                    nms_gpu_multi, nms_normal_gpu, xywhr2xyxyr)
from .registry import LOSSES, Registry, build_loss
from . import sharded
from .coders import CenterPointBBoxCoderRev, CenterPointBBoxYawCoder, DeltaXYZWLHRBBoxCoder, PointBBoxYawCoder
from .graphed import GraphedStep
from .evaluation import match_coco, trans_bev
from .scatter import Scatter, scatter_index, scatter_reduce
from .head_loss import (anchor_decoded_gd_loss, anchor_head_bbox_loss, anchor_head_decoded_loss,
                        anchor_head_decoded_loss_fused, center_head_gd_loss, center_head_losses)
from . import extras


def build(force=False, verbose=False):
    """Compile csrc/*.hip for gfx950 into mmdet3d-gaussian_amd/libgd3d.so, and csrc/torch_node.cpp (GDLoss's autograd node,
    host C++ above the C ABI) into mmdet3d-gaussian_amd/_gd3d_node.so.  Returns the library's path."""
    path = _build_mod.build(force=force, verbose=verbose)
    _build_mod.build_node(force=force, verbose=verbose)
    return path


__all__ = ['GDLoss', 'LOSSES', 'Registry', 'build_loss', 'make_params', 'nms_gpu', 'nms_normal_gpu', 'nms_gpu_batched', 'nms_gpu_multi',
           'multi_class_nms', 'multi_class_nms_batch', 'box3d_multiclass_nms', 'circle_nms', 'boxes_iou_bev', 'iou_bev', 'iou_3d', 'xywhr2xyxyr',
           'trans_bev', 'match_coco', 'sharded', 'build', 'load_library', 'lib_path', 'host_glue', 'set_host_glue',
           'CenterPointBBoxCoderRev', 'CenterPointBBoxYawCoder', 'DeltaXYZWLHRBBoxCoder', 'PointBBoxYawCoder', 'GraphedStep',
           'anchor_decoded_gd_loss', 'anchor_head_decoded_loss', 'anchor_head_decoded_loss_fused', 'anchor_head_bbox_loss', 'center_head_gd_loss',
           'center_head_losses', 'Scatter', 'scatter_index', 'scatter_reduce', 'extras']
