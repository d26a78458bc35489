/* gd3d_extras.h — C ABI of the FROZEN extras of libgd3d.so: callers around the hot path that SURVEY.md §8 does not list
 * (CenterPoint / anchor-head inference slices, target assignment, focal / heat-map losses; DESIGN_EXTRAS.md).  Parity unpinned
 * (every oracle behind them restates absent third-party code); no work since round 3.  The hot path's boundary is gd3d.h, which
 * this header includes for the shared types and error codes; same conventions (extern "C", caller-allocated memory,
 * stream-ordered, integer return codes, never exit()). */
#ifndef GD3D_EXTRAS_H_
#define GD3D_EXTRAS_H_

#include "gd3d.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------------------------
 * CenterPoint inference slice: from the head maps of all tasks to the detections of every sample, stream-ordered, no
 * host round trip inside (ABI 4).  Replaces, per task,
 *   heatmap.sigmoid() -> _reconstruct_bbox (a cat of the head maps) -> bbox_coder.select_best (two torch.topk, gathers, a
 *   per-sample Python loop) -> bbox_coder.decode -> score / centre-range mask -> boolean indexing per sample ->
 *   per sample xywhr2xyxyr + nms_gpu (or circle_nms on the host) -> the merge of all tasks per sample
 *   /root/reference/mmdet3d_gaussian/models/dense_heads/gd_centerpoint_head.py:218-303 (get_bboxes), :305-361
 *   (get_task_detections), :202-216 and :372-387 (_reconstruct_bbox),
 *   /root/reference/mmdet3d_gaussian/core/bbox/coders/centerpoint_bbox_coders.py:23-58 (select_best / _topk), :87-112
 *   (decode), centerpoint_bbox_yaw_coders.py:18-56 (decode with correct_yaw)
 * by: one selection kernel (one workgroup per task x sample: the max_per_img best cells over all classes straight from the
 * logits — sampled threshold + one filtering pass, exact radix select when that pass cannot prove itself —, LDS bitonic
 * sort, gather of the head channels at those cells from the SEPARATE maps, decode, mask, ordered compaction, and the
 * oriented-box records of the survivors), the batched NMS over all task x sample groups (rnms_batched_prepared; circle:
 * rnms_batched), one merge kernel.  Maps above 131072 cells per group: threshold and filtering pass as two chip-wide launches.
 *
 * Selection rule: descending score; equal scores in ascending flat index (class * H*W + y * W + x).  The reference's
 * per-class-then-global torch.topk picks the same set whenever the scores are distinct; the order among EQUAL scores is
 * unspecified there.  With heat_is_logit the selection runs on the logits (sigmoid is monotonic) and only the selected
 * scores are passed through 1 / (1 + exp(-x)).  NaN scores rank first (torch.topk) and fail the score threshold.
 * The centre-range test is the reference's expression `x.ge(lo).le(hi)`: the BOOLEAN (x >= lo), as 0 / 1, is compared with
 * hi (:248-250) — kept as written.
 *
 * center_infer_task (HOST array of num_tasks entries):
 *   heatmap       (batch, classes, H, W) fp32, contiguous;
 *   channel[j]    plane of gathered channel j of sample 0 ((H, W) fp32, contiguous); NULL = the constant 0.5 (head without
 *                 'reg', :206-208).  Channel order = _reconstruct_bbox: reg x2, height, dim x3, then rot(sin, cos) [decode 1]
 *                 or yaw, dir(sin, cos) [decode 2], then vel x2 if present;  sample_stride[j] floats from sample b to b+1;
 *   label_offset  added to the class index in the merged labels (:293-297);  nms_thresh: nms_thr, or min_radius[task] for
 *                 circle NMS.
 * center_infer_desc:
 *   decode 0: none (center_infer_select only), 1: CenterPointBBoxCoderRev.decode, 2: CenterPointBBoxYawCoder.decode with
 *   correct_yaw;  max_per_img <= center_infer_max_k() and <= H*W (torch.topk's own limit);  num_channels <= 16;
 *   H*W*classes < 2^31;  nms_type 0 rotate / 2 circle;  pre_max_size, post_max_size < 0: none;  num_tasks: any number for
 *   center_infer_select (10 tasks travel per launch), at most 40 for center_infer_bboxes.
 * center_infer_select: the selection alone (the coder's select_best): sel_scores (G, K) fp32, sel_cls (G, K) int64,
 *   sel_xy (G, K, 2) int64 (x, y), sel_preds (G, K, num_channels) fp32 raw channels; group G = task * batch + sample;
 *   workspace: center_infer_select_workspace_bytes(desc), 256-byte aligned (only maps above 131072 cells per group use it:
 *   their threshold and filtering pass run as chip-wide launches of their own and hand the candidates over in it).
 * center_infer_bboxes: out_boxes (batch, num_tasks * P, co) fp32 with z moved to the box bottom (:289), out_scores
 *   (batch, num_tasks * P), out_labels (batch, num_tasks * P) int32, out_count (batch) int64 on the DEVICE — the one thing the
 *   host reads back;  P = center_infer_rows_per_task(desc) = min(max_per_img, pre_max_size, post_max_size);  co = num_channels
 *   - 1 (decode 1) or - 2 (decode 2).  workspace: center_infer_workspace_bytes(desc), 256-byte aligned.
 * ---------------------------------------------------------------------------------- */
#define CENTER_INFER_MAX_CHANNELS 16
#define CENTER_INFER_MAX_TASKS 10

typedef struct center_infer_task {
  const float* heatmap;
  const float* channel[CENTER_INFER_MAX_CHANNELS];
  int64_t sample_stride[CENTER_INFER_MAX_CHANNELS];
  int32_t classes;
  int32_t label_offset;
  float nms_thresh;
  int32_t reserved;
} center_infer_task;

typedef struct center_infer_desc {
  int32_t num_tasks, batch, height, width;
  int32_t max_per_img, num_channels, decode, heat_is_logit;
  int32_t norm_bbox, use_score_threshold, use_limit_range, nms_type;
  int32_t pre_max_size, post_max_size;
  float out_size_factor;
  float voxel_size[2];
  float pc_range[2];
  float score_threshold;
  float limit_range[6];
  const center_infer_task* tasks;
} center_infer_desc;

int center_infer_max_k(void);
int64_t center_infer_rows_per_task(const center_infer_desc* desc);
size_t center_infer_workspace_bytes(const center_infer_desc* desc);
/* Where center_infer_bboxes leaves the candidates of the NMS in `workspace` (for inspection and stage-wise tests): byte offsets
 * of boxes (G, max_per_img, co) fp32, scores (G, max_per_img) fp32, class indices (G, max_per_img) int32 and counts (G) int32 —
 * per group the survivors of the score / range mask, compacted, in score order. */
int center_infer_candidates(const center_infer_desc* desc, int64_t* byte_offsets);
/* Profiling aid: while device_buffer != NULL every later selection launch records, per group, 8 int64: s_memrealtime stamps
 * (100 MHz) at kernel start / after the threshold sample / after the filtering pass / after the exact radix select (when it
 * had to run) / after the ordering / at the end, then the number of candidates, then the stamp taken when the sample had been
 * loaded.  (groups, 8) int64 on the device. */
int center_infer_debug_clocks(int64_t* device_buffer);
/* Clock probe: `iters` dependent FMAs per thread in `blocks` workgroups of 256; device_out[0] = wall time of workgroup 0 in
 * 10 ns ticks (device_out: 2 int64).  One workgroup vs a chip-filling launch shows the clock an almost idle chip is granted. */
int center_infer_debug_clock_probe(int64_t* device_out, int32_t blocks, int32_t iters, void* stream);
size_t center_infer_select_workspace_bytes(const center_infer_desc* desc);
int center_infer_select(const center_infer_desc* desc, void* workspace, float* sel_scores, int64_t* sel_cls,
                        int64_t* sel_xy, float* sel_preds, void* stream);
int center_infer_bboxes(const center_infer_desc* desc, void* workspace, float* out_boxes, float* out_scores,
                        int32_t* out_labels, int64_t* out_count, void* stream);

/* ------------------------------------------------------------------------------------
 * CenterPoint target assignment on the device (ABI 4): heat maps, anno_boxes and pos_inds of ALL samples and tasks in two
 * launches — what produces the inputs of gd3d_center_head_loss.  Replaces the per-sample / per-task / per-box Python loops of
 *   /root/reference/mmdet3d_gaussian/models/dense_heads/gd_centerpoint_head.py:65-81 (get_targets), :83-156
 *   (get_targets_single) with mmdet3d's gaussian_radius / draw_heatmap_gaussian inside them (third party, absent: restated).
 *   boxes (total, box_cols) fp32: the samples' ground-truth rows one after the other, sample b = rows
 *   [sample_start[b], sample_start[b+1]); columns x, y, z, w, l, h, yaw, ... ; bottom_center != 0: z is the box bottom
 *   (LiDARInstance3DBoxes.tensor) and the gravity centre z + h/2 is written to anno_boxes (:85-87), else rows are copied;
 *   labels (total) int64: task t owns the labels [sum classes[<t], + classes[t]); others (e.g. -1) are ignored;
 *   height / width = feature_map_size[0] / [1] as the reference uses them (rows, columns of a heat-map plane, :115-126).
 * Outputs: heatmaps = ONE flat fp32 buffer, ZEROED BY THE CALLER, task t's (batch, classes[t], height, width) block after the
 *   blocks of the tasks before it;  anno_boxes (total, box_cols) and pos_inds (total, 3) int64 [batch, x, y]: the valid boxes in
 *   the reference's order (tasks; samples in batch order; the task's classes in turn; index order), task t = rows
 *   [task_start[t], task_start[t+1]);  task_start (num_tasks + 1) int64 on the DEVICE (the one thing the host reads back).
 *   A box is valid iff w, l > 0 and its cell lies on the map; cell = trunc((x - pc_range[0]) / voxel_size[0] / out_size_factor)
 *   (`.long()` truncates toward zero).  total <= center_targets_max_boxes() (8192).
 *   workspace: center_targets_workspace_bytes(total), 256-byte aligned.
 * ---------------------------------------------------------------------------------- */
#define CENTER_TARGETS_MAX_TASKS 40
#define CENTER_TARGETS_MAX_BATCH 64

typedef struct center_targets_desc {
  int32_t num_tasks, batch, height, width;
  int32_t total, box_cols, bottom_center, min_radius;
  int32_t classes[CENTER_TARGETS_MAX_TASKS];
  int32_t sample_start[CENTER_TARGETS_MAX_BATCH + 1];
  int32_t reserved;
  float pc_range[2];
  float voxel_size[2];
  float out_size_factor;
  float reserved2;
  double gaussian_overlap;
} center_targets_desc;

int center_targets_max_boxes(void);
size_t center_targets_workspace_bytes(int64_t total);
int center_targets_build(const center_targets_desc* desc, const float* boxes, const int64_t* labels, void* workspace,
                         float* heatmaps, float* anno_boxes, int64_t* pos_inds, int64_t* task_start, void* stream);

/* ------------------------------------------------------------------------------------
 * Heat-map classification loss of the CenterPoint heads for all tasks (ABI 4): clip_sigmoid + GaussianFocalLoss + the
 * num_pos normaliser, forward and gradient in one pass over the maps (12 bytes per cell), no host sync.  Replaces, per task,
 *   /root/reference/mmdet3d_gaussian/models/dense_heads/gd_centerpoint_head.py:403-411
 *   (mmdet3d `clip_sigmoid`, mmdet `GaussianFocalLoss`: third party, absent, restated from the published text):
 *     p = clamp(sigmoid(x), clip_eps, 1 - clip_eps);  pos = -log(p + log_eps) (1 - p)^alpha [t == 1];
 *     neg = -log(1 - p + log_eps) p^alpha (1 - t)^gamma;  num_pos = #(t == 1);
 *     loss = loss_weight * sum(pos + neg) / max(num_pos, 1).
 * gd3d_heat_focal_task (HOST array): logits, target (n) fp32; grad (n) fp32 or NULL: receives d sum(pos + neg) / d logit (the
 *   raw gradient; gd3d_heat_focal_scale turns it into the gradient of the caller's scalar in place).
 * gd3d_heat_focal_loss: losses, factor = loss_weight / max(num_pos, 1), num_pos: (num_tasks) fp32 each, on the device;
 *   workspace: gd3d_heat_focal_workspace_bytes(tasks, num_tasks).  Sums are taken in a fixed order in fp64: deterministic.
 * gd3d_heat_focal_scale: grad[i] *= factor[t] * upstream[t] for every task (upstream (num_tasks) fp32 on the device).
 * ---------------------------------------------------------------------------------- */
#define GD3D_HEAT_FOCAL_MAX_TASKS 16

typedef struct gd3d_heat_focal_task {
  const float* logits;
  const float* target;
  float* grad;
  int64_t n;
} gd3d_heat_focal_task;

size_t gd3d_heat_focal_workspace_bytes(const gd3d_heat_focal_task* tasks, int32_t num_tasks);
int gd3d_heat_focal_loss(const gd3d_heat_focal_task* tasks, int32_t num_tasks, float alpha, float gamma,
                         float clip_eps, float log_eps, float loss_weight, float* losses, float* factor,
                         float* num_pos, void* workspace, void* stream);
int gd3d_heat_focal_scale(const gd3d_heat_focal_task* tasks, int32_t num_tasks, const float* factor,
                          const float* upstream, void* stream);

/* ------------------------------------------------------------------------------------
 * Anchor-head inference slice (ABI 4): from the head maps of a batch to its detections, no host sync inside.
 * The reference's GDAnchor3DHead inherits inference from mmdet3d unchanged (gd_anchor3d_head.py:10): what is replaced is
 * mmdet3d's Anchor3DHead.get_bboxes_single + box3d_multiclass_nms + DeltaXYZWLHRBBoxCoder.decode + limit_period (third party,
 * absent: restated from the published 0.x text), i.e. per sample: sigmoid of all class maps, max over classes, topk(nms_pre),
 * four gathers, ~15 decode ops, and per class a mask, two host syncs and an nms_gpu call, then concat / sort / yaw correction.
 *   anchor_infer_level (HOST array): cls_score (batch, A*C, H, W), bbox_pred (batch, A*7, H, W), dir_cls_pred (batch, A*2, H, W)
 *     fp32 contiguous, the head's raw outputs; anchors (H*W*A, 7) in the head's anchor order (h, w, a).
 *   anchor_infer_desc: nms_pre <= 0: every anchor enters the NMS (then H*W*A must be <= 4096 per level); the candidates of all
 *     levels together must not exceed rnms_scored_max_n();  score_thr is compared STRICTLY (score > score_thr), as mmdet3d does;
 *     box code size 7 only.
 *   out_boxes (batch, max_num, 7) with the direction-bin correction applied, out_scores (batch, max_num), out_labels
 *   (batch, max_num) int64, out_count (batch) int64 on the DEVICE.  Order: classes in turn, score order inside a class; beyond
 *   max_num detections the max_num best by score (equal scores keep that order).
 *   workspace: anchor_infer_workspace_bytes(desc), 256-byte aligned.
 * anchor_infer_candidates: byte offsets in the workspace of what entered the NMS — boxes (batch, K, 7) fp32, scores
 *   (batch, C, K) fp32, direction bins (batch, K) int32 — and K (return value; -1 on a bad descriptor), for stage-wise tests.
 * ---------------------------------------------------------------------------------- */
#define ANCHOR_INFER_MAX_LEVELS 4
#define ANCHOR_INFER_MAX_CLASSES 16

typedef struct anchor_infer_level {
  const float* cls_score;
  const float* bbox_pred;
  const float* dir_cls_pred;
  const float* anchors;
  int32_t height, width;
} anchor_infer_level;

typedef struct anchor_infer_desc {
  int32_t num_levels, batch, num_anchors, num_classes;
  int32_t nms_pre, max_num, use_rotate_nms, reserved;
  float score_thr, nms_thr, dir_offset, dir_limit_offset;
  const anchor_infer_level* levels;
} anchor_infer_desc;

size_t anchor_infer_workspace_bytes(const anchor_infer_desc* desc);
int64_t anchor_infer_candidates(const anchor_infer_desc* desc, int64_t* byte_offsets);
int anchor_infer_bboxes(const anchor_infer_desc* desc, void* workspace, float* out_boxes, float* out_scores,
                        int64_t* out_labels, int64_t* out_count, void* stream);

/* ------------------------------------------------------------------------------------
 * Classification (sigmoid focal) and direction (2-way cross entropy) terms of the anchor heads' loss for a whole batch,
 * forward + gradient in one pass over the head's NCHW maps (ABI 4).  Replaces, in GDAnchor3DHead.loss_single,
 *   /root/reference/mmdet3d_gaussian/models/dense_heads/gd_anchor3d_head.py:84-92 (permuted copy of the class maps, mmdet FocalLoss)
 *   and :143-149 (gather of the positives' direction logits, mmdet CrossEntropyLoss) — both loss modules third party, restated:
 *     t = [label == c], p = sigmoid(x):  BCEwithLogits(x, t) (alpha t + (1-alpha)(1-t)) ((1-p) t + p (1-t))^gamma label_weight;
 *     direction, positives only (label in [0, num_classes)): (logsumexp(d) - d[dir_target]) dir_weight.
 *   cls_score (batch, A*C, H, W), dir_cls_preds (batch, A*2, H, W) or NULL: the head's raw outputs;  labels / dir_targets
 *   (batch, H*W*A) int64 and label_weights / dir_weights (batch, H*W*A) fp32 in the head's anchor order (h, w, a);
 *   cls_scale = loss_weight / avg_factor of loss_cls, dir_scale likewise (avg_factor = num_total_samples, a host value).
 * Outputs: losses (2) fp32 on the device [loss_cls, loss_dir]; grad_cls / grad_dir (nullable): d loss / d logit in the maps'
 *   own layout (every entry written: no zero fill needed).  workspace: gd3d_anchor_cls_dir_workspace_bytes(batch, H, W).
 *   Sums in a fixed order in fp64: deterministic.  No host sync.  num_anchors <= 32, batch <= 65535 (GD3D_E_TOOLARGE).
 * ---------------------------------------------------------------------------------- */
size_t gd3d_anchor_cls_dir_workspace_bytes(int32_t batch, int32_t height, int32_t width);
int gd3d_anchor_cls_dir_loss(const float* cls_score, const float* dir_cls_preds, const int64_t* labels,
                             const float* label_weights, const int64_t* dir_targets, const float* dir_weights,
                             int32_t batch, int32_t num_anchors, int32_t num_classes, int32_t height, int32_t width,
                             float gamma, float alpha, float cls_scale, float dir_scale, float* grad_cls,
                             float* grad_dir, float* losses, void* workspace, void* stream);
/* The same with the normaliser ON THE DEVICE (no read-back of the positives' count): avg_dev -> one fp32 (num_total_samples, e.g.
 * sum_b max(positives_b, 1) of anchor_targets_build's counts); the scales are cls_weight / *avg_dev and dir_weight / *avg_dev, divided in
 * double and rounded once as the host form does. */
int gd3d_anchor_cls_dir_loss_dyn(const float* cls_score, const float* dir_cls_preds, const int64_t* labels,
                                 const float* label_weights, const int64_t* dir_targets, const float* dir_weights,
                                 int32_t batch, int32_t num_anchors, int32_t num_classes, int32_t height, int32_t width,
                                 float gamma, float alpha, double cls_weight, double dir_weight, const float* avg_dev,
                                 float* grad_cls, float* grad_dir, float* losses, void* workspace, void* stream);

/* ------------------------------------------------------------------------------------
 * Target assignment of the anchor heads for a whole batch, one feature level (ABI 4).  Replaces what GDAnchor3DHead.loss calls at
 *   /root/reference/mmdet3d_gaussian/models/dense_heads/gd_anchor3d_head.py:206-214 (`self.anchor_target_3d(...)`), inherited from
 *   mmdet3d's AnchorTrainMixin with mmdet's MaxIoUAssigner / PseudoSampler, mmdet3d's BboxOverlapsNearest3D,
 *   DeltaXYZWLHRBBoxCoder.encode and get_direction_target (third party, absent: restated).
 *   anchors (cells, num_sizes, num_rots, 7) fp32: one level's grid, reshape_out=False order; gt_boxes (gt_start[batch], 7) fp32
 *   [x, y, z, dx, dy, dz, yaw] and gt_labels int64, sample b owning rows [gt_start[b], gt_start[b+1]) (at most
 *   anchor_targets_max_gt() each).  num_assigners == num_sizes: assigner q matches the anchors of size q, with assign_per_class
 *   only against the boxes labelled q; num_assigners == 1: one assigner for all anchors and boxes.  Thresholds per assigner.
 * Outputs, anchor order (cell, size, rotation) = the (h, w, a) order of the head's maps: labels (batch, N) int64 (num_classes =
 *   background / ignored), label_weights (0 for the anchors between the thresholds), bbox_targets / bbox_weights (batch, N, 7),
 *   dir_targets int64 / dir_weights (batch, N); counts (batch, 2) int32 = positives, negatives per sample.  Every entry is
 *   written.  workspace: anchor_targets_workspace_bytes(num_assigners, gt_start[batch]).  Two launches, integer atomics only.
 * ---------------------------------------------------------------------------------- */
#define ANCHOR_TARGETS_MAX_SIZES 16
#define ANCHOR_TARGETS_MAX_BATCH 64
#define ANCHOR_TARGETS_MAX_GT 1024
typedef struct {
  int32_t batch, cells, num_sizes, num_rots, num_classes;
  int32_t num_assigners;       /* num_sizes, or 1 */
  int32_t assign_per_class;
  int32_t match_low_quality;   /* MaxIoUAssigner: default 1 */
  int32_t gt_max_assign_all;   /* MaxIoUAssigner: default 1 */
  int32_t num_dir_bins;        /* get_direction_target: 2 */
  int32_t gt_start[ANCHOR_TARGETS_MAX_BATCH + 1];
  float pos_iou_thr[ANCHOR_TARGETS_MAX_SIZES];
  float neg_iou_thr[ANCHOR_TARGETS_MAX_SIZES];
  float min_pos_iou[ANCHOR_TARGETS_MAX_SIZES];
  float pos_weight;            /* train_cfg.pos_weight: <= 0 means 1 */
  float dir_offset;
} anchor_targets_desc;
int32_t anchor_targets_max_gt(void);
size_t anchor_targets_workspace_bytes(int32_t num_assigners, int32_t gt_total);
int anchor_targets_build(const anchor_targets_desc* desc, const float* anchors, const float* gt_boxes,
                         const int64_t* gt_labels, void* workspace, int64_t* labels, float* label_weights,
                         float* bbox_targets, float* bbox_weights, int64_t* dir_targets, float* dir_weights,
                         int32_t* counts, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* GD3D_EXTRAS_H_ */
