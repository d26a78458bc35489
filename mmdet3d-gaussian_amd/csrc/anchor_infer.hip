// anchor_infer.hip — the anchor heads' inference slice that ends in rotated NMS, for gfx950 (include/gd3d.h, ABI 4).
//
// The reference's GDAnchor3DHead inherits its inference from mmdet3d unchanged
//   /root/reference/mmdet3d_gaussian/models/dense_heads/gd_anchor3d_head.py:10  (class GDAnchor3DHead(Anchor3DHead), no get_bboxes)
// so what runs is mmdet3d's Anchor3DHead.get_bboxes_single + box3d_multiclass_nms + DeltaXYZWLHRBBoxCoder.decode + limit_period
// (third party, absent; restated from the published 0.x text by the test infrastructure): per sample ~100 framework
// launches and two host syncs per class around `nms_gpu` (BASELINE configs[4]: Waymo PointPillars, 3 x 4096 boxes).
// Here, for all samples of the batch:
//   anchor_score_kernel   fp32 sigmoid of the best class logit of every anchor, in the reference's anchor order (h, w, a): the
//                         nms_pre selection ranks what the reference ranks (saturated scores tie, ties go by index); the class
//                         maps are never permuted or copied;
//   center_infer_select   (center_infer.hip) the nms_pre best anchors per sample and level;
//   anchor_gather_kernel  per selected anchor: its C class scores (sigmoid), direction bin, the 7 deltas and its anchor ->
//                         DeltaXYZWLHR decode in the reference's fp32 operation order, the BEV box for the NMS, and the
//                         per-class score rows / validity bytes (score > score_thr, strictly) the batched NMS reads;
//   rnms_batched_scored_sets (rbox.hip) the C class problems of EVERY sample in one set of launches, no host sync;
//   anchor_collect_kernel per sample: the classes' kept lists in turn; beyond max_num detections the max_num best by score
//                         (rank by binary search over the classes' score-ordered lists in LDS: stable, equal scores keep the
//                         concatenation order); direction-bin correction of the yaw (limit_period); counts.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/gd3d_extras.h"

namespace ainfer {

constexpr int T = 256;
constexpr int MAXC = ANCHOR_INFER_MAX_CLASSES;
constexpr float PI_F = 3.14159265358979323846f;

struct ScoreArgs {
  const float* cls;     // (B, A*C, H, W)
  float* out;           // (B, N) N = H*W*A, index (cell, a)
  int A, C, HW;
};

__global__ __launch_bounds__(T) void anchor_score_kernel(const ScoreArgs a) {
  const int b = blockIdx.y;
  const int cell = blockIdx.x * T + threadIdx.x;
  if (cell >= a.HW) return;
  const float* base = a.cls + (size_t)b * a.A * a.C * a.HW + cell;
  float* o = a.out + (size_t)b * a.HW * a.A + (size_t)cell * a.A;
  for (int an = 0; an < a.A; ++an) {
    float m = base[(size_t)(an * a.C) * a.HW];
    for (int c = 1; c < a.C; ++c) {
      const float v = base[(size_t)(an * a.C + c) * a.HW];
      m = (v > m || v != v) ? v : m;      // torch.max propagates NaN
    }
    // ranked on the fp32 SIGMOID of the best logit, as the reference's `scores.max(dim=1)` is (anchor3d_head get_bboxes_single):
    // logits that saturate to the same fp32 score (x > ~17, or strongly negative) are TIES there, and the selection breaks ties
    // by index — ranking on the raw logit would order them strictly and move the nms_pre boundary.  Same expression
    // as the gather kernel's class scores below; a rounded monotone function of m, so unsaturated logits keep their order.
    o[an] = (m != m) ? m : 1.0f / (1.0f + expf(-m));
  }
}

struct GatherArgs {
  const float* cls;       // (B, A*C, H, W)
  const float* bbox;      // (B, A*7, H, W)
  const float* dir;       // (B, A*2, H, W)
  const float* anchors;   // (N, 7)
  const long long* sel_xy;  // (B, K, 2): x = anchor index; nullptr: every anchor, in order
  int A, C, HW, K, Ktot, koff, B;
  float score_thr;
  float nms_thr;
  float* boxes7;          // (B, Ktot, 7)
  float* bev5;            // (B, Ktot, 5)
  float* scoresT;         // (B, C, Ktot)
  unsigned char* valid;   // (B, C, Ktot)
  int* dirs;              // (B, Ktot)
  float* thresh;          // (B, C)
};

__global__ __launch_bounds__(T) void anchor_gather_kernel(const GatherArgs a) {
  const int b = blockIdx.y;
  const int k = blockIdx.x * T + threadIdx.x;
  if (k < a.C) a.thresh[(size_t)b * a.C + k] = a.nms_thr;
  if (k >= a.K) return;
  const long long n = a.sel_xy != nullptr ? a.sel_xy[((size_t)b * a.K + k) * 2] : (long long)k;
  const int cell = (int)(n / a.A), an = (int)(n - (long long)cell * a.A);
  const size_t slot = (size_t)b * a.Ktot + a.koff + k;
  const float* cls = a.cls + ((size_t)b * a.A * a.C + (size_t)an * a.C) * a.HW + cell;
  for (int c = 0; c < a.C; ++c) {
    const float s = 1.0f / (1.0f + expf(-cls[(size_t)c * a.HW]));
    const size_t o = ((size_t)b * a.C + c) * a.Ktot + a.koff + k;
    a.scoresT[o] = s;
    a.valid[o] = s > a.score_thr ? 1 : 0;
  }
  const float* dr = a.dir + ((size_t)b * a.A * 2 + (size_t)an * 2) * a.HW + cell;
  a.dirs[slot] = dr[a.HW] > dr[0] ? 1 : 0;             // torch.max(dim=-1)[1]: the first maximum
  const float* bp = a.bbox + ((size_t)b * a.A * 7 + (size_t)an * 7) * a.HW + cell;
  float t[7], an7[7];
#pragma unroll
  for (int j = 0; j < 7; ++j) {
    t[j] = bp[(size_t)j * a.HW];
    an7[j] = a.anchors[(size_t)n * 7 + j];
  }
  // DeltaXYZWLHRBBoxCoder.decode, operation by operation (this file is compiled with -ffp-contract=off)
  const float xa = an7[0], ya = an7[1], wa = an7[3], la = an7[4], ha = an7[5], ra = an7[6];
  const float za = an7[2] + ha / 2.0f;
  const float diagonal = sqrtf(la * la + wa * wa);
  const float xg = t[0] * diagonal + xa;
  const float yg = t[1] * diagonal + ya;
  float zg = t[2] * ha + za;
  const float lg = expf(t[4]) * la;
  const float wg = expf(t[3]) * wa;
  const float hg = expf(t[5]) * ha;
  const float rg = t[6] + ra;
  zg = zg - hg / 2.0f;
  float* o7 = a.boxes7 + slot * 7;
  o7[0] = xg; o7[1] = yg; o7[2] = zg; o7[3] = wg; o7[4] = lg; o7[5] = hg; o7[6] = rg;
  float* o5 = a.bev5 + slot * 5;                        // xywhr2xyxyr of the bev columns [x, y, dx, dy, yaw]
  const float hw = wg / 2.0f, hl = lg / 2.0f;
  o5[0] = xg - hw; o5[1] = yg - hl; o5[2] = xg + hw; o5[3] = yg + hl; o5[4] = rg;
}

struct CollectArgs {
  const float* boxes7;       // (B, Ktot, 7)
  const float* scoresT;      // (B, C, Ktot)
  const int* dirs;           // (B, Ktot)
  const long long* keep;     // (B, C, Ktot) candidate indices in class-score order, into the batch's flat arrays (b Ktot + i)
  const long long* num;      // (B, C)
  int C, Ktot, max_num;
  float dir_offset, dir_limit_offset;
  float* out_boxes;          // (B, max_num, 7)
  float* out_scores;         // (B, max_num)
  long long* out_labels;     // (B, max_num)
  long long* out_count;      // (B)
};

__global__ __launch_bounds__(T) void anchor_collect_kernel(const CollectArgs a) {
  extern __shared__ __attribute__((aligned(16))) float ssc[];   // (C, max_num) the classes' best kept scores, descending
  __shared__ int s_n[MAXC], s_off[MAXC + 1], s_failed;
  const int b = blockIdx.x, tid = threadIdx.x;
  if (tid == 0) {
    int off = 0;
    s_failed = 0;
    for (int c = 0; c < a.C; ++c) {
      const long long nk = a.num[(size_t)b * a.C + c];
      if (nk < 0) s_failed = 1;   // a class's NMS scan gave up (num_keep = -1, include/gd3d.h): no rows from it, the count says so
      s_n[c] = (int)(nk < 0 ? 0 : nk);
      s_off[c] = off;
      off += s_n[c];
    }
    s_off[a.C] = off;
  }
  __syncthreads();
  const int total = s_off[a.C];
  const bool cut = total > a.max_num;
  // only the first max_num entries of a class can be among the max_num best
  for (int c = 0; c < a.C; ++c) {
    const int m = s_n[c] < a.max_num ? s_n[c] : a.max_num;
    for (int r = tid; r < m; r += T)
      ssc[c * a.max_num + r] = a.scoresT[((size_t)b * a.C + c) * a.Ktot + (a.keep[((size_t)b * a.C + c) * a.Ktot + r] - (long long)b * a.Ktot)];
  }
  __syncthreads();
  for (int c = 0; c < a.C; ++c) {
    const int m = s_n[c] < a.max_num ? s_n[c] : a.max_num;
    for (int r = tid; r < m; r += T) {
      int pos = s_off[c] + r;
      const float s = ssc[c * a.max_num + r];
      if (cut) {
        // rank under (score descending, concatenation position ascending): entries of the own class before r all count;
        // of another class c2 those with a greater score, and with an equal score when c2 comes first
        int rank = r;
        for (int c2 = 0; c2 < a.C; ++c2) {
          if (c2 == c) continue;
          const int m2 = s_n[c2] < a.max_num ? s_n[c2] : a.max_num;
          const float* lst = ssc + c2 * a.max_num;
          int lo = 0, hi = m2;                    // first index whose score is NOT counted
          while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            const float v = lst[mid];
            const bool counts = c2 < c ? !(v < s) : v > s;     // NaN scores cannot be here: they fail `> score_thr`
            if (counts) lo = mid + 1;
            else hi = mid;
          }
          rank += lo;
        }
        pos = rank;
        if (pos >= a.max_num) continue;
      }
      const long long cand = a.keep[((size_t)b * a.C + c) * a.Ktot + r] - (long long)b * a.Ktot;
      const float* src = a.boxes7 + ((size_t)b * a.Ktot + (size_t)cand) * 7;
      float* dst = a.out_boxes + ((size_t)b * a.max_num + pos) * 7;
#pragma unroll
      for (int j = 0; j < 6; ++j) dst[j] = src[j];
      // bboxes[..., 6] = limit_period(yaw - dir_offset, dir_limit_offset, pi) + dir_offset + pi * dir_score
      const float val = src[6] - a.dir_offset;
      const float dir_rot = val - floorf(val / PI_F + a.dir_limit_offset) * PI_F;
      dst[6] = dir_rot + a.dir_offset + PI_F * (float)a.dirs[(size_t)b * a.Ktot + (size_t)cand];
      a.out_scores[(size_t)b * a.max_num + pos] = s;
      a.out_labels[(size_t)b * a.max_num + pos] = c;
    }
  }
  if (tid == 0) a.out_count[b] = s_failed ? -1 : (cut ? a.max_num : total);
}

static size_t up256(size_t x) { return (x + 255) & ~(size_t)255; }

struct Layout {
  size_t smax[ANCHOR_INFER_MAX_LEVELS], sel_s[ANCHOR_INFER_MAX_LEVELS], sel_c[ANCHOR_INFER_MAX_LEVELS], sel_xy[ANCHOR_INFER_MAX_LEVELS];
  size_t sel_ws, boxes7, bev5, scoresT, valid, dirs, thresh, keep, num, nms, total;
  int64_t K[ANCHOR_INFER_MAX_LEVELS], N[ANCHOR_INFER_MAX_LEVELS], Ktot;
  size_t sel_ws_bytes;
};

static void select_desc(const anchor_infer_desc* d, int l, int64_t K, int64_t N, const float* smax, center_infer_task& t, center_infer_desc& s) {
  t = center_infer_task();
  t.heatmap = smax;
  t.classes = 1;
  s = center_infer_desc();
  s.num_tasks = 1;
  s.batch = d->batch;
  s.height = 1;
  s.width = (int32_t)N;
  s.max_per_img = (int32_t)K;
  s.num_channels = 0;
  s.decode = 0;
  s.heat_is_logit = 0;
  s.tasks = &t;
}

static int layout(const anchor_infer_desc* d, Layout& L) {
  if (d == nullptr || d->levels == nullptr || d->num_levels < 1 || d->num_levels > ANCHOR_INFER_MAX_LEVELS) return GD3D_E_BADARG;
  if (d->batch < 1 || d->num_anchors < 1 || d->num_classes < 1 || d->num_classes > MAXC || d->max_num < 1) return GD3D_E_BADARG;
  const int64_t B = d->batch, C = d->num_classes;
  size_t o = 0;
  L.Ktot = 0;
  L.sel_ws_bytes = 256;
  for (int l = 0; l < d->num_levels; ++l) {
    const anchor_infer_level& lv = d->levels[l];
    if (lv.cls_score == nullptr || lv.bbox_pred == nullptr || lv.dir_cls_pred == nullptr || lv.anchors == nullptr || lv.height < 1 ||
        lv.width < 1)
      return GD3D_E_BADARG;
    const int64_t N = (int64_t)lv.height * lv.width * d->num_anchors;
    if (N >= 0x7fffffffLL) return GD3D_E_TOOLARGE;
    const int64_t K = (d->nms_pre > 0 && N > d->nms_pre) ? d->nms_pre : N;
    if (K > center_infer_max_k()) return GD3D_E_TOOLARGE;          // set nms_pre: every anchor would enter the NMS otherwise
    L.N[l] = N;
    L.K[l] = K;
    L.Ktot += K;
    const bool sel = K < N;
    L.smax[l] = o; o += sel ? up256(sizeof(float) * (size_t)(B * N)) : 0;
    L.sel_s[l] = o; o += sel ? up256(sizeof(float) * (size_t)(B * K)) : 0;
    L.sel_c[l] = o; o += sel ? up256(sizeof(long long) * (size_t)(B * K)) : 0;
    L.sel_xy[l] = o; o += sel ? up256(sizeof(long long) * (size_t)(B * K * 2)) : 0;
    if (sel) {
      center_infer_task t;
      center_infer_desc s;
      select_desc(d, l, K, N, nullptr, t, s);
      t.heatmap = (const float*)256;   // the size query only checks for NULL
      const size_t w = center_infer_select_workspace_bytes(&s);
      L.sel_ws_bytes = w > L.sel_ws_bytes ? w : L.sel_ws_bytes;
    }
  }
  if (L.Ktot > rnms_scored_max_n()) return GD3D_E_TOOLARGE;
  L.sel_ws = o; o += up256(L.sel_ws_bytes);
  L.boxes7 = o; o += up256(sizeof(float) * (size_t)(B * L.Ktot * 7));
  L.bev5 = o; o += up256(sizeof(float) * (size_t)(B * L.Ktot * 5));
  L.scoresT = o; o += up256(sizeof(float) * (size_t)(B * C * L.Ktot));
  L.valid = o; o += up256((size_t)(B * C * L.Ktot));
  L.dirs = o; o += up256(sizeof(int) * (size_t)(B * L.Ktot));
  L.thresh = o; o += up256(sizeof(float) * (size_t)(B * C));
  L.keep = o; o += up256(sizeof(long long) * (size_t)(B * C * L.Ktot));
  L.num = o; o += up256(sizeof(long long) * (size_t)(B * C));
  L.nms = o; o += up256(rnms_batched_scored_workspace_bytes((int32_t)(B * C), L.Ktot, L.Ktot));
  L.total = o;
  return 0;
}

}  // namespace ainfer

using namespace ainfer;

extern "C" {

size_t anchor_infer_workspace_bytes(const anchor_infer_desc* desc) {
  Layout L;
  if (layout(desc, L) != 0) return 256;
  return L.total;
}

int64_t anchor_infer_candidates(const anchor_infer_desc* desc, int64_t* byte_offsets) {
  Layout L;
  if (layout(desc, L) != 0 || byte_offsets == nullptr) return -1;
  byte_offsets[0] = (int64_t)L.boxes7;
  byte_offsets[1] = (int64_t)L.scoresT;
  byte_offsets[2] = (int64_t)L.dirs;
  return L.Ktot;
}

int anchor_infer_bboxes(const anchor_infer_desc* d, void* workspace, float* out_boxes, float* out_scores, int64_t* out_labels,
                        int64_t* out_count, void* stream) {
  Layout L;
  int rc = layout(d, L);
  if (rc != 0) return rc;
  if (workspace == nullptr || out_boxes == nullptr || out_scores == nullptr || out_labels == nullptr || out_count == nullptr ||
      ((uintptr_t)workspace & 255) != 0)
    return GD3D_E_BADARG;
  hipStream_t s = (hipStream_t)stream;
  char* w = (char*)workspace;
  const int B = d->batch, C = d->num_classes, A = d->num_anchors;
  int koff = 0;
  for (int l = 0; l < d->num_levels; ++l) {
    const anchor_infer_level& lv = d->levels[l];
    const int HW = lv.height * lv.width;
    const bool sel = L.K[l] < L.N[l];
    if (sel) {
      ScoreArgs sa;
      sa.cls = lv.cls_score;
      sa.out = (float*)(w + L.smax[l]);
      sa.A = A;
      sa.C = C;
      sa.HW = HW;
      hipLaunchKernelGGL(anchor_score_kernel, dim3((unsigned)((HW + T - 1) / T), (unsigned)B), dim3(T), 0, s, sa);
      center_infer_task t;
      center_infer_desc sd;
      select_desc(d, l, L.K[l], L.N[l], sa.out, t, sd);
      rc = center_infer_select(&sd, w + L.sel_ws, (float*)(w + L.sel_s[l]), (int64_t*)(w + L.sel_c[l]), (int64_t*)(w + L.sel_xy[l]), nullptr,
                               stream);
      if (rc != 0) return rc;
    }
    GatherArgs g;
    g.cls = lv.cls_score;
    g.bbox = lv.bbox_pred;
    g.dir = lv.dir_cls_pred;
    g.anchors = lv.anchors;
    g.sel_xy = sel ? (const long long*)(w + L.sel_xy[l]) : nullptr;
    g.A = A;
    g.C = C;
    g.HW = HW;
    g.K = (int)L.K[l];
    g.Ktot = (int)L.Ktot;
    g.koff = koff;
    g.B = B;
    g.score_thr = d->score_thr;
    g.nms_thr = d->nms_thr;
    g.boxes7 = (float*)(w + L.boxes7);
    g.bev5 = (float*)(w + L.bev5);
    g.scoresT = (float*)(w + L.scoresT);
    g.valid = (unsigned char*)(w + L.valid);
    g.dirs = (int*)(w + L.dirs);
    g.thresh = (float*)(w + L.thresh);
    const int gx = (int)((L.K[l] > C ? L.K[l] : C) + T - 1) / T;
    hipLaunchKernelGGL(anchor_gather_kernel, dim3((unsigned)gx, (unsigned)B), dim3(T), 0, s, g);
    koff += (int)L.K[l];
  }
  // the C class problems of every sample in one set of launches: sample b's groups work on its own Ktot boxes
  rc = rnms_batched_scored_sets(d->use_rotate_nms ? 0 : 1, (const float*)(w + L.bev5), (const float*)(w + L.scoresT), (const uint8_t*)(w + L.valid),
                                B, C, L.Ktot, -1, (const float*)(w + L.thresh), (int64_t*)(w + L.keep), (int64_t*)(w + L.num), w + L.nms, stream);
  if (rc != 0) return rc;
  CollectArgs c;
  c.boxes7 = (const float*)(w + L.boxes7);
  c.scoresT = (const float*)(w + L.scoresT);
  c.dirs = (const int*)(w + L.dirs);
  c.keep = (const long long*)(w + L.keep);
  c.num = (const long long*)(w + L.num);
  c.C = C;
  c.Ktot = (int)L.Ktot;
  c.max_num = d->max_num;
  c.dir_offset = d->dir_offset;
  c.dir_limit_offset = d->dir_limit_offset;
  c.out_boxes = out_boxes;
  c.out_scores = out_scores;
  c.out_labels = (long long*)out_labels;
  c.out_count = (long long*)out_count;
  const size_t lds = sizeof(float) * (size_t)C * (size_t)d->max_num;
  if (lds > 64 * 1024) return GD3D_E_TOOLARGE;
  hipLaunchKernelGGL(anchor_collect_kernel, dim3((unsigned)B), dim3(T), lds, s, c);
  return (int)hipGetLastError();
}

}  // extern "C"
