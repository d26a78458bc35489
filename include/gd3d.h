/*
 * gd3d.h — C ABI of libgd3d.so: the MI355X (gfx950) hot path of mmdet3d-gaussian.
 *
 * Drop-in boundary.  The reference has NO native entry point on
 * this path: the losses are ~110-145 ATen ops per call in
 *   /root/reference/mmdet3d_gaussian/models/losses/gaussian_distance_loss.py:8-310
 * and the NMS is the third-party mmdet3d `iou3d_cuda.nms_gpu` pybind call reached from
 *   /root/reference/mmdet3d_gaussian/models/dense_heads/gd_centerpoint_head.py:9,340-345
 *   /root/reference/mmdet3d_gaussian/models/roi_heads/bbox_heads/pvrcnn_bbox_head.py:12,463-464
 * Each entry point below names the reference interface it replaces.
 *
 * Conventions
 *   - plain pointers + sizes; no torch types; every pointer is DEVICE memory of the
 *     current HIP device unless the name says `_host`;
 *   - the caller allocates everything, including the workspace (size from the
 *     *_workspace_bytes query); the library keeps no global state and frees nothing;
 *   - every call is asynchronous and stream-ordered on `stream` (a hipStream_t passed
 *     as void*; NULL = the null stream); no hipDeviceSynchronize / hipMalloc inside,
 *     so calls may be captured in a hipGraph;
 *   - return value: 0 on success, otherwise a hipError_t value or one of the
 *     GD3D_E_* codes below; nothing throws, nothing calls exit().
 *   - float arrays must be 4-byte aligned; 16-byte aligned (N,7) arrays take the fast
 *     LDS-DMA path, others a slower scalar path with identical results.
 */
#ifndef GD3D_H_
#define GD3D_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GD3D_ABI_VERSION 6

/* error codes outside the hipError_t range */
#define GD3D_E_BADARG 10001   /* null pointer / negative size / unknown enum */
#define GD3D_E_TOOLARGE 10002 /* n exceeds what the launch geometry supports */
#define GD3D_E_HOST 10003     /* `_cpu` twins only: a worker thread failed (out of host memory); outputs are not to be used */

/* loss_type: keys of GDLoss.BAG_GD_LOSS (gaussian_distance_loss.py:253-259) */
enum {
  GD3D_GWD3D = 0,        /* gwd3d_loss        :42-106  */
  GD3D_KLD3D = 1,        /* kld3d_loss        :109-141 */
  GD3D_BD3D = 2,         /* bd3d_loss         :144-186 */
  GD3D_JD3D = 3,         /* jd3d_loss         :189-198 */
  GD3D_KLD3D_SYMMAX = 4, /* kld3d_symmax_loss :201-211 */
  GD3D_KLD3D_SYMMIN = 5, /* kld3d_symmin_loss :214-224 */
  GD3D_KFIOU3D = 6,      /* kfiou3d_loss      :227-248 */
  GD3D_NUM_LOSS_TYPES = 7
};

/* fun: postprocess non-linearity (gaussian_distance_loss.py:24-39) */
enum { GD3D_FUN_NONE = 0, GD3D_FUN_LOG1P = 1, GD3D_FUN_EXPM1 = 2, GD3D_FUN_NLOG = 3 };

/* GDLoss hyper-parameters that reach the arithmetic (gaussian_distance_loss.py:261-278). */
typedef struct gd3d_params {
  int32_t loss_type;      /* GD3D_* above */
  int32_t fun;            /* GD3D_FUN_* */
  float tau;              /* tau >= 1 -> 1 - tau/(tau+d); else identity (:36-39) */
  float alpha;            /* xyz vs whlr balance */
  float center_offset[3]; /* gravity-centre offset, default (0,0,0.5) (:12) */
  int32_t flag;           /* gwd3d: `normalize`; all others: `sqrt` (kwargs at :42,:109,...) */
} gd3d_params;

/* ------------------------------------------------------------------------------------
 * Gaussian-distance loss, fused forward + gradient.
 * Replaces: preprocess x2 + one BAG_GD_LOSS function + postprocess + mmdet
 * weight_reduce_loss + `* loss_weight`, and the autograd backward of all of it
 * (gaussian_distance_loss.py:8-39, :42-248, :280-310).
 *
 * For pair i (rows pred[i,:], target[i,:] = (x,y,z,w,h,l,r), fp32, row-major, stride 7):
 *     L_i            = BAG_GD_LOSS[loss_type](preprocess(pred_i), preprocess(target_i))
 *     loss[i]        = scale * w_i * L_i                       (if loss != NULL)
 *     *loss_sum      = scale * sum_i w_i * L_i                 (if loss_sum != NULL; fp32,
 *                      deterministic: fixed-order two-stage reduction, no float atomics)
 *     grad_pred[i,:] = scale * w_i * dL_i/dpred_i              (if grad_pred != NULL)
 *     grad_target[i,:] = scale * w_i * dL_i/dtarget_i          (if grad_target != NULL)
 * with w_i = row_weight[i] (1 if row_weight == NULL).  `scale` carries
 * loss_weight / (avg_factor | N | 1) so that the kernel writes final gradients.
 * n == 0 is legal (loss_sum = 0).  workspace: gd3d_loss_workspace_bytes(n) bytes,
 * 16-byte aligned; required when loss_sum != NULL; whenever it is given the kernel leaves one
 * fp32 partial sum per 256-pair tile at its start.
 * ---------------------------------------------------------------------------------- */
size_t gd3d_loss_workspace_bytes(int64_t n);

int gd3d_loss_fused(const gd3d_params* params, const float* pred, const float* target,
                    const float* row_weight, int64_t n, float scale, float* loss,
                    float* loss_sum, float* grad_pred, float* grad_target, void* workspace,
                    void* stream);

/* Same, with the per-row weight given as an (n,7) fp32 array whose row mean is taken inside the kernel:
 *   w_i = mean(weight7[i,:])   (GDLoss.forward: `if weight.shape == pred.shape: weight = weight.mean(dim=-1)`,
 * gaussian_distance_loss.py:295-296; the heads pass decode_weight of shape (P,7), gd_anchor3d_head.py:129-141).
 * Saves the separate mean pass.  row_weight and weight7 are mutually exclusive (GD3D_E_BADARG if both). */
int gd3d_loss_fused_w7(const gd3d_params* params, const float* pred, const float* target,
                       const float* row_weight, const float* weight7, int64_t n, float scale,
                       float* loss, float* loss_sum, float* grad_pred, float* grad_target,
                       void* workspace, void* stream);

/* ------------------------------------------------------------------------------------
 * Head-level fusion: the bbox-coder decode that the dense heads run
 * immediately before GDLoss is applied inside the kernel prologue, and the gradient is chained back
 * to the ENCODED prediction, so `pred`/`grad_pred` are the raw head outputs of the positives.
 *   kind = GD3D_PRO_ANCHOR_DELTA: mmdet3d DeltaXYZWLHRBBoxCoder.decode(anchors, deltas) applied to BOTH
 *          pred and target (gd_anchor3d_head.py:133-136).  aux = anchors (n,7) [xa,ya,za,wa,la,ha,ra]:
 *            diag = sqrt(la^2+wa^2); x = xt*diag+xa; y = yt*diag+ya; w = exp(wt)*wa; l = exp(lt)*la;
 *            h = exp(ht)*ha; z = zt*ha + (za + ha/2) - h/2; r = rt + ra
 *   kind = GD3D_PRO_CENTER: CenterPointBBoxYawCoder.decode(locs, pred, correct_yaw=False)[..., :7] applied to
 *          pred only (core/bbox/coders/centerpoint_bbox_yaw_coders.py:32-42, gd_centerpoint_head.py:422-423);
 *          target rows are the annotated boxes as they are.  aux = locs (n,2) fp32 grid coordinates:
 *            x = (p0 + loc0) * out_size_factor * voxel_size[0] + pc_range[0]; y likewise; z = p2;
 *            dims = exp(p3..p5) if norm_bbox else p3..p5; yaw = p6
 * ---------------------------------------------------------------------------------- */
enum { GD3D_PRO_NONE = 0, GD3D_PRO_ANCHOR_DELTA = 1, GD3D_PRO_CENTER = 2 };

typedef struct gd3d_prologue {
  int32_t kind;           /* GD3D_PRO_* */
  int32_t norm_bbox;      /* GD3D_PRO_CENTER: dims are predicted as logs */
  const float* aux;       /* anchors (n,7) or locs (n,2), device memory */
  float out_size_factor;  /* GD3D_PRO_CENTER */
  float voxel_size[2];
  float pc_range[2];
  float reserved;
} gd3d_prologue;

int gd3d_loss_fused_decoded(const gd3d_params* params, const gd3d_prologue* prologue,
                            const float* pred, const float* target, const float* row_weight,
                            const float* weight7, int64_t n, float scale, float* loss,
                            float* loss_sum, float* grad_pred, float* grad_target,
                            void* workspace, void* stream);

/* Profiling form of the same call (bench.py): the two hipEvent_t (as void*, from gd3d_prof_event_create; either may be
 * NULL) are bound to the begin / end timestamps of the fused kernel's OWN dispatch (hipExtLaunchKernel), so
 * gd3d_prof_event_elapsed_ms(start, stop) is that kernel's execution time — what rocprofv3 --kernel-trace reports for
 * it — and no marker packets are added to the stream (events recorded around a launch cost two barrier packets, ~3 us
 * per bracket, and would be charged to the kernel).  The reduce stage, when loss_sum != NULL, follows as usual and is
 * outside the bracket.  With both events NULL this IS gd3d_loss_fused_decoded.  The elapsed-time query has
 * hipEventElapsedTime semantics: it returns hipErrorNotReady (600) while the dispatch is in flight — synchronise first.
 * (No reference counterpart: the reference has no native entry point on this path) */
int gd3d_loss_fused_timed(const gd3d_params* params, const gd3d_prologue* prologue,
                          const float* pred, const float* target, const float* row_weight,
                          const float* weight7, int64_t n, float scale, float* loss,
                          float* loss_sum, float* grad_pred, float* grad_target,
                          void* workspace, void* stream, void* start_event, void* stop_event);
int gd3d_prof_event_create(void** event);
int gd3d_prof_event_destroy(void* event);
int gd3d_prof_event_elapsed_ms(void* start_event, void* stop_event, float* ms);

/* The same slice with the GATHER of the positives fused in as well (one thread per positive, gd_anchor3d_head.py:95-141):
 *   bbox_pred (B, A*7, H, W) raw head output (NCHW, read in place: no permute copy, no index kernels);
 *   bbox_targets / bbox_weights (M,7) with M = B*H*W*A and row m = ((b*H + h)*W + w)*A + a (bbox_weights nullable);
 *   decode_weight: HOST array of 7 floats or NULL (train_cfg['decode_weight']); w_i = mean(bbox_weights[m,:] * decode_weight);
 *   anchors (H*W*A, 7): anchors of ONE sample (the reference repeats them over the batch, :110-112);
 *   pos_inds (P) int64 positive rows m; scale = loss_weight / avg_factor;
 *   *loss_sum = scale * sum_i w_i L_i ;  grad_bbox_pred (B, A*7, H, W) must be ZERO-FILLED by the caller and receives
 *   d loss / d bbox_pred at the positives' 7 channels (nullable).  workspace: gd3d_loss_workspace_bytes(P). */
int gd3d_anchor_head_loss(const gd3d_params* params, const float* bbox_pred, int32_t B, int32_t A,
                          int32_t H, int32_t W, const float* bbox_targets, const float* bbox_weights,
                          const float* decode_weight, const float* anchors, const int64_t* pos_inds,
                          int64_t P, float scale, float* loss_sum, float* grad_bbox_pred,
                          void* workspace, void* stream);

/* Dense form of the same slice: no positive list at all.  One thread per ANCHOR m tests labels[m] (int64 (M), positive iff
 * 0 <= label < num_classes, gd_anchor3d_head.py:101-105) and leaves unless positive — torch.nonzero() (a host sync), the
 * compaction and every index kernel disappear from the training step.  workspace: gd3d_loss_workspace_bytes(M). */
int gd3d_anchor_head_loss_dense(const gd3d_params* params, const float* bbox_pred, int32_t B, int32_t A,
                                int32_t H, int32_t W, const float* bbox_targets, const float* bbox_weights,
                                const float* decode_weight, const float* anchors, const int64_t* labels,
                                int32_t num_classes, float scale, float* loss_sum, float* grad_bbox_pred,
                                void* workspace, void* stream);

/* The whole regression loss of GDAnchor3DHead.loss_single (gd_anchor3d_head.py:95-159) in one launch:
 *   loss_bbox = loss_decoded_bbox(decode(anchors, pred), decode(anchors, target), decode_weight, avg_factor)   (:133-141)
 *             + loss_bbox(pred', target', code_weight, avg_factor)                                             (:152-159)
 * where the second term is mmdet's SmoothL1Loss (third party) on the ENCODED rows, after mmdet3d's
 * add_sin_difference when diff_rad_by_sin (pred'[6] = sin p6 cos t6, target'[6] = cos p6 sin t6).
 *   smooth_l1 == NULL         : only the Gaussian-distance term (same as the two functions above);
 *   smooth_l1->beta           : SmoothL1Loss.beta; 0 selects L1Loss;
 *   smooth_l1->scale          : loss_weight / avg_factor of that term;
 *   smooth_l1->has_code_weight: element weight = bbox_weights[m,k] * code_weight[k] (bbox_weights required), else 1
 *                               (the reference passes weight=None when train_cfg['code_weight'] is falsy, :124-127).
 * The positives are named by EXACTLY ONE of pos_inds (P rows) or labels (dense form, P ignored).
 * With smooth_l1 != NULL the Gaussian term is weighted by mean_k(bbox_weights*decode_weight) only when decode_weight
 * != NULL (:128-131); *loss_sum receives the SUM of both terms, grad_bbox_pred (zero-filled by the caller) their
 * combined gradient.  workspace: gd3d_loss_workspace_bytes(P or M). */
typedef struct gd3d_smooth_l1 {
  float beta;
  float scale;
  int32_t diff_rad_by_sin;
  int32_t has_code_weight;
  float code_weight[7];
  float reserved;
} gd3d_smooth_l1;

int gd3d_anchor_head_bbox_loss(const gd3d_params* params, const gd3d_smooth_l1* smooth_l1,
                               const float* bbox_pred, int32_t B, int32_t A, int32_t H, int32_t W,
                               const float* bbox_targets, const float* bbox_weights,
                               const float* decode_weight, const float* anchors,
                               const int64_t* pos_inds, int64_t P, const int64_t* labels,
                               int32_t num_classes, float scale, float* loss_sum,
                               float* grad_bbox_pred, void* workspace, void* stream);
/* gd3d_anchor_head_bbox_loss, dense form, with the normaliser ON THE DEVICE (ABI 4): scale = gd_weight / *avg_dev and the SmoothL1
 * scale = sl1_weight / *avg_dev (smooth_l1->scale is ignored), divided in double and rounded once.  gd_weight / sl1_weight are the two
 * modules' loss_weight. */
int gd3d_anchor_head_bbox_loss_dyn(const gd3d_params* params, const gd3d_smooth_l1* smooth_l1, const float* bbox_pred,
                                   int32_t B, int32_t A, int32_t H, int32_t W, const float* bbox_targets,
                                   const float* bbox_weights, const float* decode_weight, const float* anchors,
                                   const int64_t* labels, int32_t num_classes, double gd_weight, double sl1_weight,
                                   const float* avg_dev, float* loss_sum, float* grad_bbox_pred, void* workspace,
                                   void* stream);

/* CenterGDHead regression losses of up to 8 tasks in ONE launch (gd_centerpoint_head.py:402-441), reading the head
 * outputs where they lie.  Per task (host struct, device pointers):
 *   maps[6]   : reg (B,2,H,W) | height (B,1,H,W) | dim (B,3,H,W) | yaw (B,1,H,W) | dir (B,2,H,W) | vel (B,2,H,W), NCHW
 *               fp32; reg may be NULL (0.5 is used, :377-378), vel NULL when n_l1 == 2, dir NULL when n_l1 == 0;
 *   grads[6]  : gradient maps of the same shapes, ZERO-FILLED by the caller (any may be NULL).  Two objects may share a
 *               cell: contributions are added in ascending object index by one thread per cell — deterministic, no
 *               float atomics; cell_count receives the number of objects per cell (integer atomics);
 *   pos_ind   : (n,3) int64 [batch, x, y] (:72-80);  anno : (n, anno_cols) fp32 boxes [x,y,z,w,l,h,yaw(,vx,vy)];
 *   gd_scale  : loss_weight / avg_factor of loss_gd,  l1_scale: the same for loss_bbox (L1Loss).
 * coder: kind GD3D_PRO_CENTER (norm_bbox, out_size_factor, voxel_size, pc_range; aux ignored).
 * code_weights: HOST array of n_l1 (0, 2 or 4) floats = train_cfg['code_weights'] for [sin, cos(, vx, vy)].
 * losses (num_tasks, 2) fp32 on the device: [loss_l1, loss_gd] per task.
 * workspace: gd3d_center_head_workspace_bytes(num_tasks, max n).
 * gd3d_center_head_scale: autograd backward for upstream gradients != 1: grads of task t are multiplied by
 *   grad_losses[t*2+1] (reg/height/dim/yaw) or grad_losses[t*2] (dir/vel), device scalars, early exit when == 1. */
typedef struct gd3d_center_task {
  const float* maps[6];
  float* grads[6];
  const int64_t* pos_ind;
  const float* anno;
  int32_t* cell_count; /* (B*H*W) int32, ZERO-FILLED by the caller; required when any grads[] is non-NULL */
  int64_t n;
  int32_t B, H, W, anno_cols;
  float gd_scale, l1_scale;
  /* device-resident form (ABI 4), both nullable: rows_dev -> two int64 on the device, the task's rows are
   * [rows_dev[0], rows_dev[1]) of pos_ind / anno (shared arrays; `n` is then the CAPACITY the launch is sized for, rows
   * beyond it are ignored);  avg_dev -> one fp32 on the device, the scales become gd_weight / max(*avg_dev, 1) and
   * l1_weight / max(*avg_dev, 1) (gd_scale / l1_scale ignored).  With them neither the task's size nor its normaliser
   * passes through the host: center_targets_build's task_start and gd3d_heat_focal_loss's num_pos plug in directly and
   * the whole CenterGDHead.loss becomes one stream-ordered, graph-capturable sequence. */
  const int64_t* rows_dev;
  const float* avg_dev;
  double gd_weight, l1_weight;
} gd3d_center_task;

size_t gd3d_center_head_workspace_bytes(int32_t num_tasks, int64_t max_n);

int gd3d_center_head_loss(const gd3d_params* params, const gd3d_prologue* coder,
                          const gd3d_center_task* tasks, int32_t num_tasks,
                          const float* code_weights, int32_t n_l1, float* losses, void* workspace,
                          void* stream);

int gd3d_center_head_scale(const gd3d_center_task* tasks, int32_t num_tasks,
                           const float* grad_losses, void* stream);

/* The same call in its two steps, for callers that can sort (ABI 3).  gd3d_center_head_loss adds the objects of a shared
 * cell by letting the cell's lowest-index object scan the task's keys: O(shared objects x n / 64) wave steps — nothing at
 * detection sizes (n = 4000, a handful of shared cells), quadratic when tens of thousands of objects fall into few cells.
 *   gd3d_center_head_stage : first step only (losses per object, staged gradient rows, per-object cell keys: one int32 row
 *                            of max_n entries per task inside the workspace, -1 = not live);
 *   gd3d_center_head_keys  : where those rows are: task t's row starts byte_offset + t * byte_stride into the workspace;
 *   gd3d_center_head_finish: second step with the SAME arguments plus `order` (num_tasks, max_n) int64: per task the
 *                            positions 0..max_n-1 of its key row sorted by key, STABLE (so that objects of one cell stay in
 *                            ascending index; e.g. torch.sort(rows, dim=1, stable=True).indices).  One thread per run of equal
 *                            keys adds the run in that order: O(n).  order == NULL: the scanning form.
 * stage + finish(order = NULL) == gd3d_center_head_loss; every form is deterministic, none uses float atomics. */
int gd3d_center_head_stage(const gd3d_params* params, const gd3d_prologue* coder,
                           const gd3d_center_task* tasks, int32_t num_tasks, const float* code_weights,
                           int32_t n_l1, float* losses, void* workspace, void* stream);
int gd3d_center_head_keys(int32_t num_tasks, int64_t max_n, int64_t* byte_offset, int64_t* byte_stride);
int gd3d_center_head_finish(const gd3d_params* params, const gd3d_prologue* coder,
                            const gd3d_center_task* tasks, int32_t num_tasks, const float* code_weights,
                            int32_t n_l1, float* losses, void* workspace, const int64_t* order,
                            void* stream);

/* GDLoss.forward with an (n,7) weight and a reduced result, INCLUDING its early-out, without a host sync.  The reference
 * begins with `if not torch.any(weight > 0): return (pred * weight).sum()` (gaussian_distance_loss.py:290-292) — a full
 * pass over the weights and a device-to-host wait on every training call.  Here the fused kernel leaves, per tile, the
 * loss partial, sum(pred * weight7) and "some weight > 0"; the reduce stage adds them and selects on the device:
 *     *any_positive = any(weight7 > 0)                  (int32, device; NaN weights are not > 0)
 *     *loss_sum     = *any_positive ? scale * sum_i mean(weight7[i,:]) L_i : sum(dec(pred) * weight7)
 * grad_pred / grad_target receive the gradient of the FIRST branch; gd3d_grad_finish (below) puts the second branch's
 * gradient there when *any_positive == 0.  `prologue` as in gd3d_loss_fused_decoded (the early-out's `pred` is then the
 * decoded row); start_event / stop_event as in gd3d_loss_fused_timed (NULL: none).  n == 0: *loss_sum = 0, flag 0. */
int gd3d_loss_fused_select(const gd3d_params* params, const gd3d_prologue* prologue, const float* pred,
                           const float* target, const float* weight7, int64_t n, float scale,
                           float* loss_sum, int32_t* any_positive, float* grad_pred, float* grad_target,
                           void* workspace, void* stream, void* start_event, void* stop_event);

/* Training-size form (n <= gd3d_one_launch_max_n() = 16384 pairs): the same result in ONE launch.  The workgroup whose
 * arrival ticket comes last adds the per-tile partials itself (lane order, fp64) and writes *loss_sum (and, when
 * any_positive != NULL, selects as gd3d_loss_fused_select does; weight7 is then required and row_weight must be NULL).
 *   ticket: device int32 that is 0 when the call is launched and is 0 again when it has completed; calls that share a
 *   ticket must be ordered on one stream (allocate one zeroed word per stream once).  hipGraph-replay safe.
 * The two-stage form above stays the general one: in the streaming regime the arrival ticket costs 20 %. */
int64_t gd3d_one_launch_max_n(void);
int gd3d_loss_fused_one_launch(const gd3d_params* params, const gd3d_prologue* prologue, const float* pred,
                               const float* target, const float* row_weight, const float* weight7, int64_t n,
                               float scale, float* loss_sum, int32_t* any_positive, float* grad_pred,
                               float* grad_target, void* workspace, int32_t* ticket, void* stream);

/* Autograd backward of a reduced call; g is the upstream gradient (device scalar).
 *   any_positive == NULL or *any_positive != 0 : grad_pred *= g, grad_target *= g (the grid leaves after two scalar loads
 *       when g == 1: one empty launch, no HBM traffic, no host sync — gd3d_scale_rows for both arrays at once);
 *   *any_positive == 0 : grad_pred = g * d sum(dec(pred) * weight7) / d pred (= g * weight7 without a prologue),
 *       grad_target = 0.  weight7 (and pred / prologue when the forward had a prologue) must be the forward's. */
int gd3d_grad_finish(float* grad_pred, float* grad_target, const float* g, int64_t n,
                     const int32_t* any_positive, const float* weight7, const float* pred,
                     const gd3d_prologue* prologue, void* stream);

/* HBM ceiling probe with the fused kernel's access mix (two streams read, one written): z = x + y over n_floats fp32
 * (multiple of 4, 16-byte aligned arrays), nontemporal 16-byte loads and stores, one vector per thread, 64-thread
 * workgroups (the fastest and most repeatable flat-copy shape measured on MI355X).  The two events
 * (nullable) are bound to the dispatch as in gd3d_loss_fused_timed.  bench.py runs it on the fused kernel's own buffers
 * to report the box's copy ceiling next to roofline.frac.  (Measurement aid; no reference counterpart.) */
int gd3d_probe_stream(const float* x, const float* y, float* z, int64_t n_floats, void* stream,
                      void* start_event, void* stop_event);

/* ------------------------------------------------------------------------------------
 * CenterPointBBoxYawCoder on the device.  Replaces the elementwise torch ops (and their autograd nodes) of
 *   /root/reference/mmdet3d_gaussian/core/bbox/coders/centerpoint_bbox_yaw_coders.py:18-56 (decode), :11-16 (encode);
 * decode is what CenterGDHead.get_bboxes runs at inference (gd_centerpoint_head.py:244, correct_yaw=True).
 *   coder : kind GD3D_PRO_CENTER fields (norm_bbox, out_size_factor, voxel_size, pc_range; aux ignored)
 *   locs (n,2) fp32 grid coordinates, preds (n,c) fp32 rows [dx, dy, z, dim x3, yaw, sin, cos, others...], c >= 7
 *   (c >= 9 when correct_yaw);  out (n, 7 + max(c-9, 0)) = [x, y, z, dim x3, yaw, others...].
 *   correct_yaw == 1: k = floor((atan2(sin, cos) - yaw) / (pi/2) + 0.5), yaw += k pi/2, w <-> l swapped when k is odd.
 *   correct_yaw == 2: CenterPointBBoxCoderRev.decode instead (centerpoint_bbox_coders.py:87-112): preds rows
 *   [dx, dy, z, dim x3, sin, cos, others...], c >= 8, rot = atan2(sin, cos), out (n, c - 1); no backward.
 *   num_rot_parity (n) int32, nullable: receives k & 1 (what the backward needs).
 * coder_center_decode_backward: grad_preds (n,c) from grad_out (n, out cols), `out` of the forward and the parity
 *   flags (NULL = no swap); the sin / cos columns receive 0 (k is computed under no_grad in the reference).
 * coder_center_encode: boxes (n,c) [x,y,z,w,l,h,yaw, others] -> out (n,c+2) [first 7, sin yaw, cos yaw, others].
 * ---------------------------------------------------------------------------------- */
int coder_center_decode(const gd3d_prologue* coder, const float* locs, const float* preds, int64_t n,
                        int32_t c, int32_t correct_yaw, float* out, int32_t* num_rot_parity, void* stream);

int coder_center_decode_backward(const gd3d_prologue* coder, const float* grad_out, const float* out,
                                 const int32_t* num_rot_parity, int64_t n, int32_t c,
                                 float* grad_preds, void* stream);

int coder_center_encode(const float* boxes, int64_t n, int32_t c, float* out, void* stream);

/* PointBBoxYawCoder.decode (/root/reference/mmdet3d_gaussian/core/bbox/coders/point_bbox_yaw_coders.py:19-52) and its backward
 * wrt preds.  priors (n,3) [px, py, scale]; preds (n,c) [dx, dy, z, log dims x3, yaw, sin dir, cos dir, others], c >= 7
 * (>= 9 with correct_yaw = 1); out (n, 7 + max(c - 9, 0)) = [dx scale + px, dy scale + py, z, exp(dims) (first two x scale),
 * yaw, others]; correct_yaw = 1: yaw snapped by whole quarter turns towards atan2(sin, cos), w <-> l on odd turns;
 * num_rot_parity (n) nullable: parity of the turns for the backward.  encode (:12-16) is coder_center_encode. */
int coder_point_decode(const float* priors, const float* preds, int64_t n, int32_t c, int32_t correct_yaw, float* out,
                       int32_t* num_rot_parity, void* stream);
int coder_point_decode_backward(const float* priors, const float* grad_out, const float* out,
                                const int32_t* num_rot_parity, int64_t n, int32_t c, float* grad_preds, void* stream);

/* The decode in front of PVRCNNBboxHead.get_bboxes' NMS (/root/reference/mmdet3d_gaussian/models/roi_heads/bbox_heads/pvrcnn_bbox_head.py:
 * 376-386, with mmdet3d's DeltaXYZWLHRBBoxCoder.decode and rotation_3d_in_axis, third party): rois (n, roi_stride) with the box
 * [x, y, z, dx, dy, dz, yaw] at columns first_col .. first_col + 6 (1 for the head's [batch_id, box] rows), bbox_pred (n, 7) residuals
 * -> boxes (n, 7) and, if not NULL, bev_xyxyr (n, 5) = [x - dx/2, y - dy/2, x + dx/2, y + dy/2, yaw] as multi_class_nms builds it (:447-448).
 * clockwise = 0: the centre is turned counter-clockwise by the roi's yaw (mmdet3d 1.0); 1: the transpose (0.x). */
int coder_roi_decode(const float* rois, int32_t roi_stride, int32_t first_col, const float* bbox_pred, int64_t n,
                     int32_t clockwise, float* boxes, float* bev_xyxyr, void* stream);

/* Second stage of the reduction on its own: *loss_sum = fixed-order fp64 sum of the per-workgroup
 * partials that gd3d_loss_fused(..., workspace != NULL) left in `workspace` for the same n.
 * gd3d_loss_fused calls it itself when loss_sum != NULL; it is exported so that a caller can
 * bracket the fused kernel alone with events (bench.py) or defer the scalar. */
int gd3d_loss_reduce(const void* workspace, int64_t n, float* loss_sum, void* stream);

/* In-place row scaling used by autograd backward when the upstream gradient is not 1:
 *   grad[i,:] *= (per_row ? g[i] : g[0]),  grad is (n,7) fp32, g is DEVICE memory.
 * With per_row == 0 the kernel reads g[0] first and exits without touching grad when it is
 * exactly 1.0f, so the common `loss.backward()` costs one empty launch and no HBM traffic
 * and no host sync.  (Replaces the autograd mul nodes of `* self.loss_weight`, :310.) */
int gd3d_scale_rows(float* grad, const float* g, int per_row, int64_t n, void* stream);

/* ------------------------------------------------------------------------------------
 * `_cpu` twins: the same contracts on HOST memory, no stream, no HIP call — they run on a machine
 * without a GPU.  The reference's GDLoss.forward is device-agnostic (gaussian_distance_loss.py:280-310: the same op chain
 * on CPU tensors, BASELINE configs[0]); these entry points are what a CPU tensor takes in this library.  They evaluate the
 * kernel's own per-pair closed forms and hand-derived gradients (csrc/gd3d_device.h compiled for the host), one pass,
 * loss and final gradients together, over `nthreads` host threads (<= 0: std::thread::hardware_concurrency()).
 *   gd3d_loss_fused_cpu : contract of gd3d_loss_fused_w7 (row_weight / weight7 mutually exclusive, n == 0 legal).
 *       workspace: gd3d_loss_workspace_bytes(n) bytes of host memory, required when loss_sum != NULL; whenever given it
 *       receives one fp32 partial per 256-pair tile (fp64 accumulation in row order, rounded once).  *loss_sum = the
 *       fixed-order fp64 sum of those partials, rounded to fp32: independent of nthreads.
 *   gd3d_loss_reduce_cpu: the second stage alone (twin of gd3d_loss_reduce).
 *   gd3d_scale_rows_cpu : twin of gd3d_scale_rows (g is host memory; per_row == 0 and g[0] == 1 touches nothing).
 * The bbox-coder prologues (gd3d_loss_fused_decoded) have no CPU twin: the head-level fusions are GPU-only.
 * ---------------------------------------------------------------------------------- */
int gd3d_loss_fused_cpu(const gd3d_params* params, const float* pred, const float* target,
                        const float* row_weight, const float* weight7, int64_t n, float scale,
                        float* loss, float* loss_sum, float* grad_pred, float* grad_target,
                        void* workspace, int32_t nthreads);
int gd3d_loss_reduce_cpu(const void* workspace, int64_t n, float* loss_sum);
int gd3d_scale_rows_cpu(float* grad, const float* g, int per_row, int64_t n, int32_t nthreads);

/* `_cpu` twins of the rotated-box entry points (declared further down): csrc/rbox_device.h — the geometry of the GPU kernels,
 * every step one IEEE fp32 operation in a fixed order — compiled for the host with the same -ffp-contract=off, so results
 * are BIT-IDENTICAL to the HIP kernels'.  Host memory, no stream, no workspace.  The reference's own pairwise IoU helpers are
 * CPU code (ops/eval/affinity.cpp:8-105); rnms_*_cpu are the greedy scan on one thread (decisions of rnms_bev /
 * rnms_normal_bev; mmdet3d's nms_gpu itself has no CPU form).  nthreads <= 0: std::thread::hardware_concurrency(). */
int riou_bev_xyxyr_cpu(const float* a, int64_t na, const float* b, int64_t nb, float* iou, int32_t nthreads);
int riou_eval_bev_cpu(const float* det, int64_t nd, const float* gt, int64_t ng, float* iou, int32_t nthreads);
int riou_eval_3d_cpu(const float* det, int64_t nd, const float* gt, int64_t ng, float z_offset, float* iou, int32_t nthreads);
int riou_eval_trans_bev_cpu(const float* det, int64_t nd, int32_t det_cols, const float* gt, int64_t ng, int32_t gt_cols,
                            float* dist, int32_t nthreads);
int rnms_bev_cpu(const float* boxes_sorted, int64_t n, float thresh, int64_t* keep, int64_t* num_keep);
int rnms_normal_bev_cpu(const float* boxes_sorted, int64_t n, float thresh, int64_t* keep, int64_t* num_keep);

/* ------------------------------------------------------------------------------------
 * Rotated BEV NMS.  Replaces mmdet3d `iou3d_cuda.nms_gpu(boxes, keep, thresh, device)`
 * behind `nms_gpu(boxes, scores, thresh, pre_max_size, post_max_size)`
 * (call sites gd_centerpoint_head.py:340-345, pvrcnn_bbox_head.py:463-464).
 *   boxes_sorted : (n,5) fp32 [x1,y1,x2,y2,ry], ALREADY sorted by descending score
 *   keep         : (n) int64, receives indices into boxes_sorted, ascending
 *   num_keep     : (1) int64; -1 (every rnms_* entry, per group): the device-side scan gave
 *                  up — a wave of the list scan stopped making progress and its bounded
 *                  polling loop ended (never observed; the host layer raises on it).
 *                  Every rnms_* entry writes each num_keep word EXACTLY ONCE, with the kernel
 *                  that resolves the group's last box: the pointer may be device-visible PINNED
 *                  HOST memory that the caller pre-sets to a sentinel and polls instead of
 *                  copying the count back (what nms_gpu does: iou3d.py, _pynode.count_mailbox).
 * Greedy: box i is kept iff no kept j < i has IoU_bev(j,i) > thresh.  The suppression
 * bit-mask (n x ceil(n/64) uint64), from 768 boxes on also per-box victim lists, and the
 * greedy scan all stay on the device; nothing in the workspace needs initialising.
 * rnms_normal_bev: same with axis-aligned IoU (mmdet3d nms_normal_gpu; angle ignored).
 * ---------------------------------------------------------------------------------- */
size_t rnms_workspace_bytes(int64_t n);

int rnms_bev(const float* boxes_sorted, int64_t n, float thresh, int64_t* keep,
             int64_t* num_keep, void* workspace, void* stream);

int rnms_normal_bev(const float* boxes_sorted, int64_t n, float thresh, int64_t* keep,
                    int64_t* num_keep, void* workspace, void* stream);

/* The whole `nms_gpu` body in one call for up to rnms_scored_max_n() (= 16384) candidates — the heads cut to nms_pre
 * before they call it.  The score order (descending, ties by ascending index, NaN first: what
 * torch.sort(descending=True, stable=True) yields) is obtained by COUNTING, for every box, the boxes with a larger
 * (score, -index) key — parallel over the whole chip, no sort — and the per-box prep is scattered straight to its rank;
 * the `pre_max` cut (pre_max < 0: none) keeps ranks < pre_max.  Mask and scan follow as in rnms_bev_ordered.
 *   boxes (n_all,5) fp32, scores (n_all) fp32; keep (min(n_all, pre_max)) int64 receives indices into `boxes` by
 *   descending score; num_keep (1) int64; workspace: rnms_scored_workspace_bytes(n_all, min(n_all, pre_max)).
 *   normal != 0: axis-aligned IoU (nms_normal_gpu).  n_all > rnms_scored_max_n(): GD3D_E_TOOLARGE (sort outside and
 *   call rnms_bev_ordered). */
int rnms_scored_max_n(void);
size_t rnms_scored_workspace_bytes(int64_t n_all, int64_t n_keep);
int rnms_scored(int32_t normal, const float* boxes, const float* scores, int64_t n_all, int64_t pre_max, float thresh,
                int64_t* keep, int64_t* num_keep, void* workspace, void* stream);

/* Same, taking the UNSORTED boxes plus the score order (what `scores.sort(descending=True)` returns, already cut to
 * pre_max_size): box i of the NMS is boxes[order[i]], and `keep` receives indices into the caller's original box
 * numbering (= order[kept]) — the gather before and the index mapping after the call disappear.
 *   boxes (any number of rows >= max(order)+1, 5) fp32;  order (n) int64;  keep (n) int64;  num_keep (1) int64. */
int rnms_bev_ordered(const float* boxes, const int64_t* order, int64_t n, float thresh, int64_t* keep,
                     int64_t* num_keep, void* workspace, void* stream);

int rnms_normal_bev_ordered(const float* boxes, const int64_t* order, int64_t n, float thresh,
                            int64_t* keep, int64_t* num_keep, void* workspace, void* stream);

/* G independent NMS problems in ONE set of launches (classes of a multi-class NMS, pvrcnn_bbox_head.py:452-477; the
 * samples / tasks a head loops over, gd_centerpoint_head.py:322-345): all groups index the same box array.
 *   mode     : 0 rotated BEV IoU (nms_gpu), 1 axis-aligned IoU (nms_normal_gpu), 2 circle (below; boxes is (N,2));
 *   order    : (groups, cap) int64, row g = box indices of group g by descending score; only its first counts[g] used;
 *   counts   : (groups) int32 ON THE DEVICE — group sizes are data dependent (score thresholds), reading them in the
 *              kernels removes the per-group host sync the reference loop pays; values are clamped to [0, cap];
 *   thresh   : (groups) fp32 on the device (per-class thresholds);
 *   keep     : (groups, cap) int64, row g receives the kept box indices (caller's numbering) in score order;
 *   num_keep : (groups) int64.   workspace: rnms_batched_workspace_bytes(groups, cap).
 * Each group's result equals the single call on that group (same kernels, blockIdx.y = group). */
size_t rnms_batched_workspace_bytes(int32_t groups, int64_t cap);

int rnms_batched(int32_t mode, const float* boxes, const int64_t* order, const int32_t* counts,
                 int32_t groups, int64_t cap, const float* thresh, int64_t* keep, int64_t* num_keep,
                 void* workspace, void* stream);

/* rnms_batched (mode 0) for a caller INSIDE the library that has already written the oriented boxes: the first counts[g]
 * entries of row g of a (groups, cap) array of 64-byte records (csrc/rbox_device.h `OBox`, made by `obox_make` from the
 * [x1,y1,x2,y2,ry] rows in score order) at the start of `workspace`.  Saves the preparation launch; center_infer.hip's
 * selection kernel writes the records while it compacts its survivors. */
int rnms_batched_prepared(const float* boxes, const int64_t* order, const int32_t* counts, int32_t groups,
                          int64_t cap, const float* thresh, int64_t* keep, int64_t* num_keep,
                          void* workspace, void* stream);

/* rnms_batched with the score order taken inside the library (rank by counting, as rnms_scored), so the caller needs no
 * masked_fill / sum / sort passes:
 *   scores (groups, n) fp32;  valid (groups, n) bytes, nullable — which boxes take part in group g;  pre_max < 0: none.
 *   Group g runs NMS over its valid boxes by descending score (ties by ascending index), cut to cap = min(n, pre_max).
 *   keep (groups, cap) int64 indices into `boxes`; num_keep (groups) int64; thresh (groups) fp32, DEVICE.
 *   n <= rnms_scored_max_n(), else GD3D_E_TOOLARGE.  workspace: rnms_batched_scored_workspace_bytes(groups, n, cap). */
size_t rnms_batched_scored_workspace_bytes(int32_t groups, int64_t n, int64_t cap);
int rnms_batched_scored(int32_t mode, const float* boxes, const float* scores, const uint8_t* valid, int32_t groups,
                        int64_t n, int64_t pre_max, const float* thresh, int64_t* keep, int64_t* num_keep,
                        void* workspace, void* stream);

/* rnms_batched_scored over several box sets in one set of launches (ABI 4): `sets` x groups_per_set groups, group g working on the
 * n boxes of set g / groups_per_set = rows [set n, (set + 1) n) of boxes (sets n, 5) — the per-sample class problems of a whole
 * batch (box3d_multiclass_nms of every sample at once).  scores / valid (sets groups_per_set, n), thresh (sets groups_per_set) on the
 * device; keep holds indices into the flat box array (set n + i).  workspace: rnms_batched_scored_workspace_bytes(sets
 * groups_per_set, n, cap).  Each group's result equals rnms_batched_scored on its own set. */
int rnms_batched_scored_sets(int32_t mode, const float* boxes, const float* scores, const uint8_t* valid, int32_t sets,
                             int32_t groups_per_set, int64_t n, int64_t pre_max, const float* thresh, int64_t* keep,
                             int64_t* num_keep, void* workspace, void* stream);

/* The same for G problems that each own a CONTIGUOUS run of one flat box / score array (the per-sample x per-task NMS
 * calls of CenterHeadRev.get_bboxes, gd_centerpoint_head.py:233-345, concatenated): group g = boxes [seg[g], seg[g+1]).
 * Every group ranks only its own slice — O(sum n_g^2) key compares, where the dense (G, N) form above would compare every
 * key in every group.
 *   scores (N_total) fp32;  seg (groups+1) int32 on the DEVICE, ascending;  max_seg >= every group size (host value:
 *   the caller knows the sizes from its tensor shapes), <= rnms_scored_max_n();  cap = min(max_seg, pre_max).
 *   keep (groups, cap) int64 GLOBAL indices into `boxes`; num_keep (groups) int64.
 *   workspace: rnms_batched_scored_workspace_bytes(groups, max_seg, cap). */
int rnms_segmented_scored(int32_t mode, const float* boxes, const float* scores, const int32_t* seg, int32_t groups,
                          int64_t max_seg, int64_t pre_max, const float* thresh, int64_t* keep, int64_t* num_keep,
                          void* workspace, void* stream);

/* Circle NMS (mmdet3d `circle_nms(dets, thresh, post_max_size)`, numba, CPU; the reference copies the detections
 * D->H for it, gd_centerpoint_head.py:256-272): centres xy (rows, 2) fp32, order (n) by descending score; box j is
 * suppressed by a kept i before it iff (x_i-x_j)^2 + (y_i-y_j)^2 <= thresh (fp32 distance, float64 compare; note
 * the published code compares the SQUARED distance with the radius as given).  keep/num_keep as rnms_bev_ordered;
 * workspace: rnms_workspace_bytes(n). */
int rnms_circle_ordered(const float* xy, const int64_t* order, int64_t n, double thresh,
                        int64_t* keep, int64_t* num_keep, void* workspace, void* stream);

/* Pairwise rotated BEV IoU in the NMS box format (mmdet3d `boxes_iou_bev`):
 *   a (na,5), b (nb,5) [x1,y1,x2,y2,ry] -> iou (na,nb) fp32 row-major. */
int riou_bev_xyxyr(const float* a, int64_t na, const float* b, int64_t nb, float* iou,
                   void* stream);

/* Pairwise rotated IoU of 7-dof boxes (x,y,z,w,h,l,yaw), the reference's CPU eval helpers
 *   /root/reference/mmdet3d_gaussian/ops/eval/affinity.cpp:8-49  (iou_3d, z_offset)
 *   /root/reference/mmdet3d_gaussian/ops/eval/affinity.cpp:51-81 (iou_bev)
 * built on rotated_boxes_intersection (ops/eval/rbox_utils.hpp:280-302).
 *   det (nd,7), gt (ng,7) -> iou (nd,ng) fp32 row-major. */
int riou_eval_bev(const float* det, int64_t nd, const float* gt, int64_t ng, float* iou,
                  void* stream);

int riou_eval_3d(const float* det, int64_t nd, const float* gt, int64_t ng, float z_offset,
                 float* iou, void* stream);

/* BEV centre distance (affinity.cpp:83-105, `LidarCenterTransBEV`): det (nd, det_cols), gt (ng, gt_cols) row-major,
 * columns 0 and 1 are the centre -> dist (nd, ng) fp32 = sqrt(dx^2 + dy^2). */
int riou_eval_trans_bev(const float* det, int64_t nd, int32_t det_cols, const float* gt, int64_t ng,
                        int32_t gt_cols, float* dist, void* stream);

/* COCO-style greedy matcher (matcher.cpp:8-74, `MatcherCoCo`), all on the device:
 *   cost (nd, ng) fp32 row-major (e.g. the negated IoU matrix the kernels above wrote), cost_thrs (nt) fp32,
 *   is_ignore / is_crowd (ng) bytes (0/1)  ->  matched (nt, nd) int32: gt index or -1.
 * For every threshold the detections are visited in row order; a detection takes the cheapest gt with cost <= thr
 * that is still free (or a crowd gt), a non-ignore gt beating any ignore gt, ties going to the later gt.  Bit-exact
 * with the reference (integer output).  ng <= 1048576. */
int eval_match_coco(const float* cost, const float* cost_thrs, const uint8_t* is_ignore,
                    const uint8_t* is_crowd, int64_t nd, int64_t ng, int64_t nt, int32_t* matched,
                    void* stream);
/* `_cpu` twin (ABI 6): the same contract on HOST memory — the reference's matcher IS CPU code (matcher.cpp:8-74, numpy in
 * and out), so an evaluation without a GPU keeps working.  Thresholds are spread over `nthreads` std::threads (<= 0: hardware
 * concurrency).  Same integers as the kernel and as the reference. */
int eval_match_coco_cpu(const float* cost, const float* cost_thrs, const uint8_t* is_ignore,
                        const uint8_t* is_crowd, int64_t nd, int64_t ng, int64_t nt, int32_t* matched,
                        int32_t nthreads);

/* ------------------------------------------------------------------------------------
 * Dynamic point-to-voxel scatter-reduce.  Replaces
 * dynamic_point_to_voxel_scatter_reduce / dynamic_point_to_voxel_backward
 * (/root/reference/mmdet3d_gaussian/ops/voxel/src/voxelization.h:35-79,
 *  src/scatter_points_cuda.cu:183-321) behind ops/voxel/scatter.py:29-72.
 *   feats (n,c) fp32; map (n) int32 point->voxel (-1 = dropped point); count (v) int32 points per voxel;
 *   order (n_valid...) int32: point ids grouped by voxel in ascending point id (stable argsort of map, with the
 *   -1 entries skipped by seg[0]); seg (v+1) int32 segment bounds into `order`.
 *   reduce: GD3D_REDUCE_* = the reference's reduce_t (voxelization.h:4).
 * vox_scatter_reduce : out (v,c) = max | mean | sum over the voxel's points; argmax (v,c) int32 (max only,
 *                      nullable) receives the smallest point id attaining the max.
 * vox_scatter_backward: grad_feats (n,c): sum -> grad_vox[map[i]], mean -> / count, max -> routed to argmax,
 *                      0 elsewhere (the kernel zero-fills).  Deterministic, no float atomics.
 * ---------------------------------------------------------------------------------- */
enum { GD3D_REDUCE_SUM = 0, GD3D_REDUCE_MEAN = 1, GD3D_REDUCE_MAX = 2 };

int vox_scatter_reduce(const float* feats, const int32_t* order, const int32_t* seg, int64_t n,
                       int32_t c, int64_t v, int reduce, float* out, int32_t* argmax, void* stream);

int vox_scatter_backward(const float* grad_vox, const int32_t* map, const int32_t* count,
                         const int32_t* argmax, int64_t n, int32_t c, int64_t v, int reduce,
                         float* grad_feats, void* stream);

/* From point coordinates to the index structures above, in one stream-ordered call (ABI 3; replaces ~20 framework launches:
 * per-column extents, one mixed-radix int64 key per point, a stable radix sort of (key, point id), run heads -> voxel ids).
 *   coors (n, ndim) int32, ndim <= 8; a point with ANY negative coordinate is dropped (scatter_points_cuda.cu:236-246).
 * Outputs, caller-allocated at their UPPER bounds (every point a voxel of its own):
 *   point2voxel_map (n) int32 (-1 = dropped);  order (n) int32: point ids grouped by voxel, ascending inside a voxel, the
 *   dropped points first;  seg (n + 1) int32: seg[v] .. seg[v+1] = voxel v's run in `order` (seg[0] = dropped points);
 *   counts (n) int32;  voxel_coors (n, ndim) int32: the sorted unique rows (lexicographic = the reference's unique_dim order,
 *   samples in batch order);  num (2) int64 on the DEVICE: [number of voxels V, number of dropped points] — the one thing
 *   the host has to read back before it can cut voxel_coors / counts / seg to V rows.
 *   workspace: vox_index_workspace_bytes(n, ndim) bytes, 256-byte aligned (0 is returned if the query itself fails). */
size_t vox_index_workspace_bytes(int64_t n, int32_t ndim);
int vox_index_build(const int32_t* coors, int64_t n, int32_t ndim, void* workspace,
                    int32_t* point2voxel_map, int32_t* order, int32_t* seg, int32_t* counts,
                    int32_t* voxel_coors, int64_t* num, void* stream);

/* Same gradient, produced in VOXEL order from the forward's grouping (order, seg as in vox_scatter_reduce): each voxel's
 * gradient row is read once and streamed to its points, instead of being re-gathered once per point — about half the
 * HBM traffic of the map-ordered form.  Points in no voxel (the prefix order[0 .. seg[0])) receive zeros.
 * Any c <= 128 (rows staged through LDS and written as a flat dword stream), or c % 4 == 0, c <= 256 with 16-byte aligned
 * grad_vox / grad_feats / argmax; GD3D_E_BADARG otherwise: use vox_scatter_backward. */
int vox_scatter_backward_grouped(const float* grad_vox, const int32_t* order, const int32_t* seg,
                                 const int32_t* argmax, int64_t n, int32_t c, int64_t v, int reduce,
                                 float* grad_feats, void* stream);

/* Library identification: returns GD3D_ABI_VERSION; *arch (if non-NULL) receives a static
 * string naming the code-object target, e.g. "gfx950". */
int gd3d_abi_version(const char** arch);

#ifdef __cplusplus
}
#endif
#endif /* GD3D_H_ */
