"""ctypes binding of libgd3d.so — the only way the Python host code reaches the HIP kernels — and the choice of host glue
above it (`load_node`: Python autograd.Function + ctypes, or the optional C++ node).

Nothing substitutes for the library: if it cannot be loaded (or built) every op raises; a GPU tensor never takes a CPU
path (CPU tensors run the library's own `_cpu` twins).  The signatures mirror include/gd3d.h one to one.
"""
import ctypes
import os

from . import build as _build

_lib = None


class Params(ctypes.Structure):
    """gd3d_params (include/gd3d.h)."""
    _fields_ = [('loss_type', ctypes.c_int32), ('fun', ctypes.c_int32),
                ('tau', ctypes.c_float), ('alpha', ctypes.c_float),
                ('center_offset', ctypes.c_float * 3), ('flag', ctypes.c_int32)]


class Prologue(ctypes.Structure):
    """gd3d_prologue (include/gd3d.h)."""
    _fields_ = [('kind', ctypes.c_int32), ('norm_bbox', ctypes.c_int32), ('aux', ctypes.c_void_p),
                ('out_size_factor', ctypes.c_float), ('voxel_size', ctypes.c_float * 2),
                ('pc_range', ctypes.c_float * 2), ('reserved', ctypes.c_float)]


class SmoothL1(ctypes.Structure):
    """gd3d_smooth_l1 (include/gd3d.h)."""
    _fields_ = [('beta', ctypes.c_float), ('scale', ctypes.c_float), ('diff_rad_by_sin', ctypes.c_int32),
                ('has_code_weight', ctypes.c_int32), ('code_weight', ctypes.c_float * 7), ('reserved', ctypes.c_float)]


class CenterTask(ctypes.Structure):
    """gd3d_center_task (include/gd3d.h)."""
    _fields_ = [('maps', ctypes.c_void_p * 6), ('grads', ctypes.c_void_p * 6), ('pos_ind', ctypes.c_void_p),
                ('anno', ctypes.c_void_p), ('cell_count', ctypes.c_void_p), ('n', ctypes.c_int64), ('B', ctypes.c_int32),
                ('H', ctypes.c_int32),
                ('W', ctypes.c_int32), ('anno_cols', ctypes.c_int32), ('gd_scale', ctypes.c_float),
                ('l1_scale', ctypes.c_float), ('rows_dev', ctypes.c_void_p), ('avg_dev', ctypes.c_void_p),
                ('gd_weight', ctypes.c_double), ('l1_weight', ctypes.c_double)]


class CenterInferTask(ctypes.Structure):
    """center_infer_task (include/gd3d.h)."""
    _fields_ = [('heatmap', ctypes.c_void_p), ('channel', ctypes.c_void_p * 16), ('sample_stride', ctypes.c_int64 * 16),
                ('classes', ctypes.c_int32), ('label_offset', ctypes.c_int32), ('nms_thresh', ctypes.c_float),
                ('reserved', ctypes.c_int32)]


class CenterInferDesc(ctypes.Structure):
    """center_infer_desc (include/gd3d.h)."""
    _fields_ = [('num_tasks', ctypes.c_int32), ('batch', ctypes.c_int32), ('height', ctypes.c_int32), ('width', ctypes.c_int32),
                ('max_per_img', ctypes.c_int32), ('num_channels', ctypes.c_int32), ('decode', ctypes.c_int32),
                ('heat_is_logit', ctypes.c_int32), ('norm_bbox', ctypes.c_int32), ('use_score_threshold', ctypes.c_int32),
                ('use_limit_range', ctypes.c_int32), ('nms_type', ctypes.c_int32), ('pre_max_size', ctypes.c_int32),
                ('post_max_size', ctypes.c_int32), ('out_size_factor', ctypes.c_float), ('voxel_size', ctypes.c_float * 2),
                ('pc_range', ctypes.c_float * 2), ('score_threshold', ctypes.c_float), ('limit_range', ctypes.c_float * 6),
                ('tasks', ctypes.POINTER(CenterInferTask))]


class CenterTargetsDesc(ctypes.Structure):
    """center_targets_desc (include/gd3d.h)."""
    _fields_ = [('num_tasks', ctypes.c_int32), ('batch', ctypes.c_int32), ('height', ctypes.c_int32), ('width', ctypes.c_int32),
                ('total', ctypes.c_int32), ('box_cols', ctypes.c_int32), ('bottom_center', ctypes.c_int32),
                ('min_radius', ctypes.c_int32), ('classes', ctypes.c_int32 * 40), ('sample_start', ctypes.c_int32 * 65),
                ('reserved', ctypes.c_int32), ('pc_range', ctypes.c_float * 2), ('voxel_size', ctypes.c_float * 2),
                ('out_size_factor', ctypes.c_float), ('reserved2', ctypes.c_float), ('gaussian_overlap', ctypes.c_double)]


class HeatFocalTask(ctypes.Structure):
    """gd3d_heat_focal_task (include/gd3d.h)."""
    _fields_ = [('logits', ctypes.c_void_p), ('target', ctypes.c_void_p), ('grad', ctypes.c_void_p), ('n', ctypes.c_int64)]


class AnchorInferLevel(ctypes.Structure):
    """anchor_infer_level (include/gd3d.h)."""
    _fields_ = [('cls_score', ctypes.c_void_p), ('bbox_pred', ctypes.c_void_p), ('dir_cls_pred', ctypes.c_void_p),
                ('anchors', ctypes.c_void_p), ('height', ctypes.c_int32), ('width', ctypes.c_int32)]


class AnchorInferDesc(ctypes.Structure):
    """anchor_infer_desc (include/gd3d.h)."""
    _fields_ = [('num_levels', ctypes.c_int32), ('batch', ctypes.c_int32), ('num_anchors', ctypes.c_int32), ('num_classes', ctypes.c_int32),
                ('nms_pre', ctypes.c_int32), ('max_num', ctypes.c_int32), ('use_rotate_nms', ctypes.c_int32), ('reserved', ctypes.c_int32),
                ('score_thr', ctypes.c_float), ('nms_thr', ctypes.c_float), ('dir_offset', ctypes.c_float),
                ('dir_limit_offset', ctypes.c_float), ('levels', ctypes.POINTER(AnchorInferLevel))]


class AnchorTargetsDesc(ctypes.Structure):
    """anchor_targets_desc (include/gd3d.h)."""
    _fields_ = [('batch', ctypes.c_int32), ('cells', ctypes.c_int32), ('num_sizes', ctypes.c_int32), ('num_rots', ctypes.c_int32),
                ('num_classes', ctypes.c_int32), ('num_assigners', ctypes.c_int32), ('assign_per_class', ctypes.c_int32),
                ('match_low_quality', ctypes.c_int32), ('gt_max_assign_all', ctypes.c_int32), ('num_dir_bins', ctypes.c_int32),
                ('gt_start', ctypes.c_int32 * 65), ('pos_iou_thr', ctypes.c_float * 16), ('neg_iou_thr', ctypes.c_float * 16),
                ('min_pos_iou', ctypes.c_float * 16), ('pos_weight', ctypes.c_float), ('dir_offset', ctypes.c_float)]


# every symbol include/gd3d.h declares: name -> (restype, argtypes)
_vp, _i64, _f32, _int, _sz = ctypes.c_void_p, ctypes.c_int64, ctypes.c_float, ctypes.c_int, ctypes.c_size_t
SYMBOLS = {
    'gd3d_loss_workspace_bytes': (_sz, [_i64]),
    'gd3d_loss_fused': (_int, [ctypes.POINTER(Params), _vp, _vp, _vp, _i64, _f32, _vp, _vp, _vp, _vp, _vp, _vp]),
    'gd3d_loss_fused_w7': (_int, [ctypes.POINTER(Params), _vp, _vp, _vp, _vp, _i64, _f32, _vp, _vp, _vp, _vp, _vp, _vp]),
    'gd3d_loss_fused_decoded': (_int, [ctypes.POINTER(Params), ctypes.POINTER(Prologue), _vp, _vp, _vp, _vp, _i64, _f32,
                                       _vp, _vp, _vp, _vp, _vp, _vp]),
    'gd3d_loss_fused_timed': (_int, [ctypes.POINTER(Params), ctypes.POINTER(Prologue), _vp, _vp, _vp, _vp, _i64, _f32,
                                     _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    'gd3d_loss_fused_select': (_int, [ctypes.POINTER(Params), ctypes.POINTER(Prologue), _vp, _vp, _vp, _i64, _f32, _vp, _vp,
                                      _vp, _vp, _vp, _vp, _vp, _vp]),
    'gd3d_one_launch_max_n': (_i64, []),
    'gd3d_loss_fused_one_launch': (_int, [ctypes.POINTER(Params), ctypes.POINTER(Prologue), _vp, _vp, _vp, _vp, _i64, _f32, _vp, _vp,
                                          _vp, _vp, _vp, _vp, _vp]),
    'gd3d_loss_fused_cpu': (_int, [ctypes.POINTER(Params), _vp, _vp, _vp, _vp, _i64, _f32, _vp, _vp, _vp, _vp, _vp, ctypes.c_int32]),
    'gd3d_loss_reduce_cpu': (_int, [_vp, _i64, _vp]),
    'gd3d_scale_rows_cpu': (_int, [_vp, _vp, _int, _i64, ctypes.c_int32]),
    'riou_bev_xyxyr_cpu': (_int, [_vp, _i64, _vp, _i64, _vp, ctypes.c_int32]),
    'riou_eval_bev_cpu': (_int, [_vp, _i64, _vp, _i64, _vp, ctypes.c_int32]),
    'riou_eval_3d_cpu': (_int, [_vp, _i64, _vp, _i64, _f32, _vp, ctypes.c_int32]),
    'riou_eval_trans_bev_cpu': (_int, [_vp, _i64, ctypes.c_int32, _vp, _i64, ctypes.c_int32, _vp, ctypes.c_int32]),
    'rnms_bev_cpu': (_int, [_vp, _i64, _f32, _vp, _vp]),
    'rnms_normal_bev_cpu': (_int, [_vp, _i64, _f32, _vp, _vp]),
    'gd3d_grad_finish': (_int, [_vp, _vp, _vp, _i64, _vp, _vp, _vp, ctypes.POINTER(Prologue), _vp]),
    'gd3d_probe_stream': (_int, [_vp, _vp, _vp, _i64, _vp, _vp, _vp]),
    'gd3d_prof_event_create': (_int, [ctypes.POINTER(_vp)]),
    'gd3d_prof_event_destroy': (_int, [_vp]),
    'gd3d_prof_event_elapsed_ms': (_int, [_vp, _vp, ctypes.POINTER(ctypes.c_float)]),
    'gd3d_anchor_head_loss': (_int, [ctypes.POINTER(Params), _vp, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32,
                                     ctypes.c_int32, _vp, _vp, ctypes.POINTER(ctypes.c_float), _vp, _vp, _i64, _f32, _vp,
                                     _vp, _vp, _vp]),
    'gd3d_anchor_head_loss_dense': (_int, [ctypes.POINTER(Params), _vp, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32,
                                           ctypes.c_int32, _vp, _vp, ctypes.POINTER(ctypes.c_float), _vp, _vp,
                                           ctypes.c_int32, _f32, _vp, _vp, _vp, _vp]),
    'gd3d_anchor_head_bbox_loss': (_int, [ctypes.POINTER(Params), ctypes.POINTER(SmoothL1), _vp, ctypes.c_int32,
                                          ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, _vp, _vp,
                                          ctypes.POINTER(ctypes.c_float), _vp, _vp, _i64, _vp, ctypes.c_int32, _f32, _vp,
                                          _vp, _vp, _vp]),
    'gd3d_anchor_head_bbox_loss_dyn': (_int, [ctypes.POINTER(Params), ctypes.POINTER(SmoothL1), _vp, ctypes.c_int32, ctypes.c_int32,
                                              ctypes.c_int32, ctypes.c_int32, _vp, _vp, ctypes.POINTER(ctypes.c_float), _vp, _vp, ctypes.c_int32,
                                              ctypes.c_double, ctypes.c_double, _vp, _vp, _vp, _vp, _vp]),
    'gd3d_center_head_workspace_bytes': (_sz, [ctypes.c_int32, _i64]),
    'gd3d_center_head_loss': (_int, [ctypes.POINTER(Params), ctypes.POINTER(Prologue), ctypes.POINTER(CenterTask), ctypes.c_int32,
                                     ctypes.POINTER(ctypes.c_float), ctypes.c_int32, _vp, _vp, _vp]),
    'gd3d_center_head_scale': (_int, [ctypes.POINTER(CenterTask), ctypes.c_int32, _vp, _vp]),
    'gd3d_center_head_stage': (_int, [ctypes.POINTER(Params), ctypes.POINTER(Prologue), ctypes.POINTER(CenterTask), ctypes.c_int32,
                                      ctypes.POINTER(ctypes.c_float), ctypes.c_int32, _vp, _vp, _vp]),
    'gd3d_center_head_keys': (_int, [ctypes.c_int32, _i64, ctypes.POINTER(_i64), ctypes.POINTER(_i64)]),
    'gd3d_center_head_finish': (_int, [ctypes.POINTER(Params), ctypes.POINTER(Prologue), ctypes.POINTER(CenterTask), ctypes.c_int32,
                                       ctypes.POINTER(ctypes.c_float), ctypes.c_int32, _vp, _vp, _vp, _vp]),
    'coder_center_decode': (_int, [ctypes.POINTER(Prologue), _vp, _vp, _i64, ctypes.c_int32, ctypes.c_int32, _vp, _vp, _vp]),
    'coder_center_decode_backward': (_int, [ctypes.POINTER(Prologue), _vp, _vp, _vp, _i64, ctypes.c_int32, _vp, _vp]),
    'coder_center_encode': (_int, [_vp, _i64, ctypes.c_int32, _vp, _vp]),
    'coder_point_decode': (_int, [_vp, _vp, _i64, ctypes.c_int32, ctypes.c_int32, _vp, _vp, _vp]),
    'coder_roi_decode': (_int, [_vp, ctypes.c_int32, ctypes.c_int32, _vp, _i64, ctypes.c_int32, _vp, _vp, _vp]),
    'coder_point_decode_backward': (_int, [_vp, _vp, _vp, _vp, _i64, ctypes.c_int32, _vp, _vp]),
    'gd3d_loss_reduce': (_int, [_vp, _i64, _vp, _vp]),
    'gd3d_scale_rows': (_int, [_vp, _vp, _int, _i64, _vp]),
    'rnms_workspace_bytes': (_sz, [_i64]),
    'rnms_bev': (_int, [_vp, _i64, _f32, _vp, _vp, _vp, _vp]),
    'rnms_normal_bev': (_int, [_vp, _i64, _f32, _vp, _vp, _vp, _vp]),
    'rnms_bev_ordered': (_int, [_vp, _vp, _i64, _f32, _vp, _vp, _vp, _vp]),
    'rnms_normal_bev_ordered': (_int, [_vp, _vp, _i64, _f32, _vp, _vp, _vp, _vp]),
    'rnms_scored_max_n': (_int, []),
    'rnms_scored_workspace_bytes': (_sz, [_i64, _i64]),
    'rnms_scored': (_int, [ctypes.c_int32, _vp, _vp, _i64, _i64, _f32, _vp, _vp, _vp, _vp]),
    'rnms_batched_workspace_bytes': (_sz, [ctypes.c_int32, _i64]),
    'rnms_batched': (_int, [ctypes.c_int32, _vp, _vp, _vp, ctypes.c_int32, _i64, _vp, _vp, _vp, _vp, _vp]),
    'rnms_batched_prepared': (_int, [_vp, _vp, _vp, ctypes.c_int32, _i64, _vp, _vp, _vp, _vp, _vp]),
    'rnms_batched_scored_workspace_bytes': (_sz, [ctypes.c_int32, _i64, _i64]),
    'rnms_batched_scored': (_int, [ctypes.c_int32, _vp, _vp, _vp, ctypes.c_int32, _i64, _i64, _vp, _vp, _vp, _vp, _vp]),
    'rnms_batched_scored_sets': (_int, [ctypes.c_int32, _vp, _vp, _vp, ctypes.c_int32, ctypes.c_int32, _i64, _i64, _vp, _vp, _vp, _vp, _vp]),
    'rnms_segmented_scored': (_int, [ctypes.c_int32, _vp, _vp, _vp, ctypes.c_int32, _i64, _i64, _vp, _vp, _vp, _vp, _vp]),
    'rnms_circle_ordered': (_int, [_vp, _vp, _i64, ctypes.c_double, _vp, _vp, _vp, _vp]),
    'riou_bev_xyxyr': (_int, [_vp, _i64, _vp, _i64, _vp, _vp]),
    'riou_eval_bev': (_int, [_vp, _i64, _vp, _i64, _vp, _vp]),
    'riou_eval_3d': (_int, [_vp, _i64, _vp, _i64, _f32, _vp, _vp]),
    'riou_eval_trans_bev': (_int, [_vp, _i64, ctypes.c_int32, _vp, _i64, ctypes.c_int32, _vp, _vp]),
    'eval_match_coco': (_int, [_vp, _vp, _vp, _vp, _i64, _i64, _i64, _vp, _vp]),
    'eval_match_coco_cpu': (_int, [_vp, _vp, _vp, _vp, _i64, _i64, _i64, _vp, ctypes.c_int32]),
    'vox_scatter_reduce': (_int, [_vp, _vp, _vp, _i64, ctypes.c_int32, _i64, _int, _vp, _vp, _vp]),
    'vox_scatter_backward': (_int, [_vp, _vp, _vp, _vp, _i64, ctypes.c_int32, _i64, _int, _vp, _vp]),
    'vox_index_workspace_bytes': (_sz, [_i64, ctypes.c_int32]),
    'vox_index_build': (_int, [_vp, _i64, ctypes.c_int32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    'vox_scatter_backward_grouped': (_int, [_vp, _vp, _vp, _vp, _i64, ctypes.c_int32, _i64, _int, _vp, _vp]),
    'center_infer_max_k': (_int, []),
    'center_infer_rows_per_task': (_i64, [ctypes.POINTER(CenterInferDesc)]),
    'center_infer_workspace_bytes': (_sz, [ctypes.POINTER(CenterInferDesc)]),
    'center_infer_candidates': (_int, [ctypes.POINTER(CenterInferDesc), ctypes.POINTER(_i64)]),
    'center_infer_debug_clocks': (_int, [_vp]),
    'center_infer_debug_clock_probe': (_int, [_vp, ctypes.c_int32, ctypes.c_int32, _vp]),
    'center_infer_select_workspace_bytes': (_sz, [ctypes.POINTER(CenterInferDesc)]),
    'center_infer_select': (_int, [ctypes.POINTER(CenterInferDesc), _vp, _vp, _vp, _vp, _vp, _vp]),
    'center_infer_bboxes': (_int, [ctypes.POINTER(CenterInferDesc), _vp, _vp, _vp, _vp, _vp, _vp]),
    'center_targets_max_boxes': (_int, []),
    'center_targets_workspace_bytes': (_sz, [_i64]),
    'center_targets_build': (_int, [ctypes.POINTER(CenterTargetsDesc), _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    'gd3d_heat_focal_workspace_bytes': (_sz, [ctypes.POINTER(HeatFocalTask), ctypes.c_int32]),
    'gd3d_heat_focal_loss': (_int, [ctypes.POINTER(HeatFocalTask), ctypes.c_int32, _f32, _f32, _f32, _f32, _f32, _vp, _vp, _vp, _vp, _vp]),
    'gd3d_heat_focal_scale': (_int, [ctypes.POINTER(HeatFocalTask), ctypes.c_int32, _vp, _vp, _vp]),
    'anchor_infer_workspace_bytes': (_sz, [ctypes.POINTER(AnchorInferDesc)]),
    'anchor_infer_candidates': (_i64, [ctypes.POINTER(AnchorInferDesc), ctypes.POINTER(_i64)]),
    'anchor_infer_bboxes': (_int, [ctypes.POINTER(AnchorInferDesc), _vp, _vp, _vp, _vp, _vp, _vp]),
    'anchor_targets_max_gt': (ctypes.c_int32, []),
    'anchor_targets_workspace_bytes': (_sz, [ctypes.c_int32, ctypes.c_int32]),
    'anchor_targets_build': (_int, [ctypes.POINTER(AnchorTargetsDesc), _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    'gd3d_anchor_cls_dir_workspace_bytes': (_sz, [ctypes.c_int32, ctypes.c_int32, ctypes.c_int32]),
    'gd3d_anchor_cls_dir_loss': (_int, [_vp, _vp, _vp, _vp, _vp, _vp, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32,
                                        ctypes.c_int32, _f32, _f32, _f32, _f32, _vp, _vp, _vp, _vp, _vp]),
    'gd3d_anchor_cls_dir_loss_dyn': (_int, [_vp, _vp, _vp, _vp, _vp, _vp, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32,
                                            ctypes.c_int32, _f32, _f32, ctypes.c_double, ctypes.c_double, _vp, _vp, _vp, _vp, _vp, _vp]),
    'gd3d_abi_version': (_int, [ctypes.POINTER(ctypes.c_char_p)]),
}


# the frozen extras' entry points (include/gd3d_extras.h) live in libgd3d_extras.so since round 6: out of the §8 library
EXTRA_SYMBOLS = {name: SYMBOLS.pop(name) for name in (
    'center_infer_max_k',
    'center_infer_rows_per_task',
    'center_infer_workspace_bytes',
    'center_infer_candidates',
    'center_infer_debug_clocks',
    'center_infer_debug_clock_probe',
    'center_infer_select_workspace_bytes',
    'center_infer_select',
    'center_infer_bboxes',
    'center_targets_max_boxes',
    'center_targets_workspace_bytes',
    'center_targets_build',
    'gd3d_heat_focal_workspace_bytes',
    'gd3d_heat_focal_loss',
    'gd3d_heat_focal_scale',
    'anchor_infer_workspace_bytes',
    'anchor_infer_candidates',
    'anchor_infer_bboxes',
    'anchor_targets_max_gt',
    'anchor_targets_workspace_bytes',
    'anchor_targets_build',
    'gd3d_anchor_cls_dir_workspace_bytes',
    'gd3d_anchor_cls_dir_loss',
    'gd3d_anchor_cls_dir_loss_dyn',
)}


def lib_path():
    return _build.LIB_PATH


def extras_path():
    return _build.EXTRAS_PATH


ABI_VERSION = 6


def host_simd_ok():
    """The `_cpu` twins are compiled for x86-64-v3 (AVX2 + FMA: the per-pair math is written in explicit fmaf(), build.py); on a
    host CPU without them a twin would die of SIGILL.  Read once from /proc/cpuinfo; unknown platforms are given the benefit of
    the doubt (the HIP entry points need none of this)."""
    global _simd_ok
    if _simd_ok is None:
        _simd_ok = True
        try:
            with open('/proc/cpuinfo') as f:
                for line in f:
                    if line.startswith('flags'):
                        flags = set(line.split(':', 1)[1].split())
                        _simd_ok = {'avx2', 'fma'} <= flags
                        break
        except OSError:
            pass
    return _simd_ok


_simd_ok = None


def _unsupported_cpu_twin(name):
    def raiser(*_a, **_k):
        raise RuntimeError(f'{name}: the CPU twins of libgd3d.so are built for x86-64-v3 (AVX2 + FMA) and this host CPU lacks them; '
                           'move the tensors to the GPU (the HIP entry points are unaffected)')
    return raiser


def _bind(path):
    L = ctypes.CDLL(path)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(L, name)  # AttributeError if the .so does not export what the header declares
        fn.restype = res
        fn.argtypes = args
    if not host_simd_ok():   # a clear error instead of SIGILL (the C++ node checks for itself: it resolves its own pointers)
        for name in SYMBOLS:
            if name.endswith('_cpu'):
                setattr(L, name, _unsupported_cpu_twin(name))
    arch = ctypes.c_char_p()
    ver = L.gd3d_abi_version(ctypes.byref(arch))
    if ver != ABI_VERSION:
        raise RuntimeError(f'{path}: ABI version {ver} != {ABI_VERSION}')
    return L


_extras = None


def load_extras():
    """libgd3d_extras.so — the frozen round-3 extras outside SURVEY.md §8 (include/gd3d_extras.h, DESIGN_EXTRAS.md): loaded on
    first use by mmdet3d_gaussian_amd.extras' modules only.  It resolves the NMS entry points it calls from the libgd3d.so next
    to it; `load()` runs first so that a stale pair is rebuilt together."""
    global _extras
    if _extras is None:
        load()
        L = ctypes.CDLL(_build.EXTRAS_PATH)
        for name, (res, args) in EXTRA_SYMBOLS.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _extras = L
    return _extras


def load():
    """Load libgd3d.so, building it first when the in-tree binary is missing or does not match its sources (content hash).
    A binary that does not match is NEVER loaded: it is rebuilt, or — on a machine without hipcc — refused (a library built
    from other sources may disagree with this binding about arguments: the ABI version only covers the header)."""
    global _lib
    if _lib is not None:
        return _lib
    override = os.environ.get('GD3D_LIB')  # A/B runs of experimental builds (tools/build_variants.py)
    if override:
        _lib = _bind(override)
        return _lib
    if _build.is_stale():
        if not os.path.exists(_build.hipcc_path()):
            what = 'does not match its sources' if os.path.isfile(_build.LIB_PATH) else 'is missing'
            raise RuntimeError(f'libgd3d.so (HIP kernels for gfx950) {what} (source hash {_build.source_hash()[:12]}) and hipcc is '
                               'not available to build it; a GPU tensor has no other path in this package')
        _build.build()                  # compile or link errors propagate
    _lib = _bind(_build.LIB_PATH)
    return _lib


# ---- host glue above the C ABI: Python (torch.autograd.Function + ctypes, _pynode.py) or C++ (csrc/torch_node.cpp) -------------
# Both call the same extern "C" entry points with the same arguments: a choice of glue, not of arithmetic.
#   GD3D_HOST=python  the Python glue (needs nothing beyond libgd3d.so)
#   GD3D_HOST=cpp     the C++ node; a failure to build or load it is an error
#   unset / auto      the C++ node when its in-tree binary matches its sources or can be rebuilt; the Python glue otherwise,
#                     with ONE loud log line saying why
HOST_GLUE_MODES = ('python', 'cpp')
_node = None
_glue = None        # the resolved mode of `_node`
_forced = None      # set_host_glue()
_unit_grads = {}    # device index (-1: CPU) -> address of gd_loss.unit_grad's constant; pushed into whichever glue is loaded


def register_unit_grad(device_index, address):
    _unit_grads[int(device_index)] = int(address)
    if _node is not None:
        _node.set_unit_grad(int(device_index), int(address))


class NodeBuildFailed(RuntimeError):
    """csrc/torch_node.cpp did not compile (as opposed to: no compiler, no binary) — a defect worth an ERROR-level log line."""


def _load_cpp_node():
    """_gd3d_node.so, bound to the loaded libgd3d.so.  Raises when it is missing or stale and cannot be rebuilt; never loads a
    binary whose hash (sources + flags + torch version) differs: it would have been compiled against another libtorch."""
    node_path = os.environ.get('GD3D_NODE_LIB')   # an instrumented build of the node (tools/sanitize.sh), as GD3D_LIB is for the library
    if not node_path and _build.node_is_stale():
        if _build.host_cxx_path() is None:
            what = 'does not match its sources / this torch' if os.path.isfile(_build.NODE_PATH) else 'is missing'
            raise RuntimeError(f'_gd3d_node.so {what} and the ROCm clang++ is not available to build it')
        # a build of exactly these sources + flags + torch that already FAILED is not tried again by every new process (each rank
        # of a torchrun job would spend ~25 s compiling the same error): the failure is stamped with the hash it belongs to
        want = _build.node_source_hash()
        stamp = _build.NODE_PATH + '.buildfailed'
        try:
            with open(stamp) as f:
                failed_hash, _, failed_why = f.read().partition('\n')
        except OSError:
            failed_hash, failed_why = '', ''
        if failed_hash.strip() == want:
            raise NodeBuildFailed(f'an earlier build of csrc/torch_node.cpp at this source hash failed and is not retried '
                                  f'(delete {stamp} to retry): {failed_why.strip()[:300]}')
        try:
            _build.build_node()
        except Exception as e:
            try:
                with open(stamp, 'w') as f:
                    f.write(want + '\n' + str(e)[-2000:])
            except OSError:
                pass
            raise NodeBuildFailed(str(e)) from e
    import importlib.util
    import torch  # noqa: F401  (libtorch must be loaded before the node is)
    spec = importlib.util.spec_from_file_location('_gd3d_node', node_path or _build.NODE_PATH)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    ver = mod.bind(os.environ.get('GD3D_LIB') or _build.LIB_PATH)
    if ver != ABI_VERSION:
        raise RuntimeError(f'_gd3d_node.so bound a library of ABI version {ver} != {ABI_VERSION}')
    return mod


def _requested_glue():
    mode = _forced or os.environ.get('GD3D_HOST', '').strip().lower() or 'auto'
    if mode not in HOST_GLUE_MODES + ('auto',):
        raise RuntimeError(f"GD3D_HOST={mode!r}: expected 'python', 'cpp' or 'auto'")
    return mode


def load_node():
    """The host-glue module GDLoss (reduced forms), nms_gpu (scored path), the anchor-head slice and scatter_reduce call:
    `reduced`, `nms_scored`, `anchor_head`, `scatter_reduce`, `set_unit_grad`, `finish_calls`, `bind` — from _pynode.py or from
    _gd3d_node.so (see the table above)."""
    global _node, _glue
    if _node is not None:
        return _node
    load()
    mode = _requested_glue()
    if mode == 'python':
        from . import _pynode
        _node, _glue = _pynode, 'python'
    elif mode == 'cpp':
        _node, _glue = _load_cpp_node(), 'cpp'
    else:
        try:
            _node, _glue = _load_cpp_node(), 'cpp'
        except Exception as e:   # missing compiler, torch headers this compiler rejects, an unloadable binary ...
            import logging
            log = logging.getLogger('mmdet3d_gaussian_amd')
            # a COMPILE error in the node's source is a defect, not an environment property: error level (and stamped, so that
            # later processes neither retry the build nor hide it)
            (log.error if isinstance(e, NodeBuildFailed) else log.warning)(
                'host glue: the optional C++ autograd node is unavailable (%s); using the Python glue (torch.autograd.Function '
                '+ ctypes over the same C ABI and kernels: identical results, 7-30 us more host time per training-size call). '
                'Set GD3D_HOST=python to silence this, GD3D_HOST=cpp to make it an error.', str(e).splitlines()[0][:300])
            from . import _pynode
            _node, _glue = _pynode, 'python'
    for idx, address in _unit_grads.items():
        _node.set_unit_grad(idx, address)
    return _node


def host_glue():
    """'python' or 'cpp': which glue load_node() resolved to (resolves it on first use)."""
    load_node()
    return _glue


def set_host_glue(mode):
    """Select the glue for the calls that follow ('python' | 'cpp' | None = GD3D_HOST / auto again).  Nodes already attached to
    live graphs keep working: each is self-contained."""
    global _node, _glue, _forced
    if mode is not None and mode not in HOST_GLUE_MODES:
        raise RuntimeError(f"set_host_glue({mode!r}): expected 'python', 'cpp' or None")
    _forced = mode
    _node = _glue = None


def check(rc, what):
    if rc != 0:
        raise RuntimeError(f'{what} failed with code {rc}' + {10001: ' (bad argument)', 10002: ' (too large)',
                                                               10003: ' (a host worker thread failed: out of memory?)'}.get(rc, ''))
