// anchor_targets.hip — target assignment of the anchor heads for a whole batch on gfx950 (include/gd3d.h, ABI 4).
//
// The reference's GDAnchor3DHead.loss calls `self.anchor_target_3d(...)` at
//   /root/reference/mmdet3d_gaussian/models/dense_heads/gd_anchor3d_head.py:206-214
// and inherits it from mmdet3d / mmdet (third party, absent: restated from the published text):
// per sample and per size class a nearest-BEV IoU matrix (gts x 107 k anchors), MaxIoUAssigner (max / argmax both ways, a Python
// loop over the gts with a full-row compare each), PseudoSampler (two nonzero + unique), DeltaXYZWLHR encode, direction bins and
// six scatters: ~60 launches and several host syncs per (sample, class), 18 such calls per KITTI batch of 6.
// Here, two launches for the batch, no sync:
//   pass 1  iou_max_kernel : thread per anchor of one (sample, assigner); the sample's boxes sit in LDS as nearest-BEV rectangles;
//                            per box the best overlap over the assigner's anchors as one 64-bit key (IoU bits, then lowest anchor
//                            index) — LDS atomicMax per workgroup, one global atomicMax per (workgroup, box) that any anchor touches.
//   pass 2  assign_kernel  : recomputes the thread's overlaps (same instructions, same bits), applies the assigner's rules in the
//                            reference's order (negative below neg_iou_thr, positive from pos_iou_thr on, then box after box its
//                            best anchors when that best reaches min_pos_iou — later boxes overwrite earlier ones), and writes
//                            labels, weights, encoded regression targets, direction bins in the head's (h, w, size, rotation)
//                            order; positives / negatives per sample counted with integer atomics.
// Integer atomics only: the result does not depend on scheduling.  Built with -ffp-contract=off: the IoU (+, -, x, /, max, min)
// carries the rounding of the torch elementwise ops, so thresholds and ties decide as they do there.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/gd3d.h"

namespace atgt {

constexpr int T = 256;
constexpr int MAX_GT = ANCHOR_TARGETS_MAX_GT;
constexpr float PI_F = 3.14159265358979323846f;
constexpr float QUARTER_PI_F = 0.78539816339744830962f;
constexpr float TWO_PI_F = 6.28318530717958647692f;

struct Rect {
  float x1, y1, x2, y2;
};

// LiDARInstance3DBoxes.nearest_bev: the BEV rectangle with the yaw snapped to the nearer axis
__device__ __forceinline__ Rect nearest_bev(float x, float y, float dx, float dy, float r) {
  const float nr = fabsf(r - floorf(r / PI_F + 0.5f) * PI_F);        // |limit_period(r, 0.5, pi)|
  const bool turned = nr > QUARTER_PI_F;
  const float w = turned ? dy : dx, h = turned ? dx : dy;
  Rect q;
  q.x1 = x - w / 2;
  q.y1 = y - h / 2;
  q.x2 = x + w / 2;
  q.y2 = y + h / 2;
  return q;
}

__device__ __forceinline__ float rect_area(const Rect& q) { return (q.x2 - q.x1) * (q.y2 - q.y1); }

// mmdet bbox_overlaps(mode='iou', eps=1e-6) of a ground-truth rectangle g (area ag) and an anchor rectangle a (area aa)
__device__ __forceinline__ float iou_of(const Rect& g, float ag, const Rect& a, float aa) {
  const float w = fmaxf(fminf(g.x2, a.x2) - fmaxf(g.x1, a.x1), 0.0f);
  const float h = fmaxf(fminf(g.y2, a.y2) - fmaxf(g.y1, a.y1), 0.0f);
  const float overlap = w * h;
  const float uni = fmaxf(ag + aa - overlap, 1e-6f);
  return overlap / uni;
}

struct Args {
  anchor_targets_desc d;
  const float* anchors;        // (cells, S, R, 7)
  const float* gt;             // (G_total, 7)
  const long long* gt_labels;  // (G_total)
  unsigned long long* keys;    // (Q, G_total)
  long long* labels;           // (B, N)
  float* label_w;              // (B, N)
  float* bbox_t;               // (B, N, 7)
  float* bbox_w;               // (B, N, 7)
  long long* dir_t;            // (B, N)
  float* dir_w;                // (B, N)
  int* counts;                 // (B, 2)
  int g_total;
};

struct Staged {
  Rect r[MAX_GT];
  float area[MAX_GT];
  int label[MAX_GT];
};

// the sample's boxes -> LDS; returns how many of them assigner q may match
__device__ __forceinline__ int stage(const Args& a, int b, int q, Staged& s, int* s_count) {
  const int g0 = a.d.gt_start[b], G = a.d.gt_start[b + 1] - g0;
  if (threadIdx.x == 0) *s_count = 0;
  __syncthreads();
  int mine = 0;
  for (int g = threadIdx.x; g < G; g += T) {
    const float* row = a.gt + (size_t)(g0 + g) * 7;
    const Rect q4 = nearest_bev(row[0], row[1], row[3], row[4], row[6]);
    s.r[g] = q4;
    s.area[g] = rect_area(q4);
    const long long lab = a.gt_labels[g0 + g];
    const int li = (lab >= 0 && lab < 0x7fffffffLL) ? (int)lab : -1;
    s.label[g] = li;
    mine += (!a.d.assign_per_class || li == q) ? 1 : 0;
  }
  if (mine) atomicAdd(s_count, mine);
  __syncthreads();
  return G;
}

// anchor m of assigner q -> its index n in the head's order (cell, size, rotation)
__device__ __forceinline__ long long anchor_index(const anchor_targets_desc& d, int q, long long m) {
  if (d.num_assigners == 1) return m;
  return ((m / d.num_rots) * d.num_sizes + q) * d.num_rots + m % d.num_rots;
}

__global__ __launch_bounds__(T) void iou_max_kernel(const Args a) {
  __shared__ Staged s;
  __shared__ unsigned long long s_key[MAX_GT];
  __shared__ int s_count;
  const int b = blockIdx.z, q = blockIdx.y;
  const int G = stage(a, b, q, s, &s_count);
  if (G == 0 || s_count == 0) return;
  for (int g = threadIdx.x; g < G; g += T) s_key[g] = 0ull;
  __syncthreads();
  const long long M = a.d.num_assigners == 1 ? (long long)a.d.cells * a.d.num_sizes * a.d.num_rots : (long long)a.d.cells * a.d.num_rots;
  const long long m = (long long)blockIdx.x * T + threadIdx.x;
  if (m < M) {
    const float* an = a.anchors + anchor_index(a.d, q, m) * 7;
    const Rect ra = nearest_bev(an[0], an[1], an[3], an[4], an[6]);
    const float aa = rect_area(ra);
    const unsigned low = 0xffffffffu - (unsigned)m;            // equal overlaps: the lowest anchor index wins (first maximum)
    for (int g = 0; g < G; ++g) {
      if (a.d.assign_per_class && s.label[g] != q) continue;
      const float v = iou_of(s.r[g], s.area[g], ra, aa);
      if (v > 0.0f) atomicMax(&s_key[g], ((unsigned long long)__float_as_uint(v) << 32) | low);
    }
  }
  __syncthreads();
  const int g0 = a.d.gt_start[b];
  for (int g = threadIdx.x; g < G; g += T)
    if (s_key[g] != 0ull) atomicMax(&a.keys[(size_t)q * a.g_total + g0 + g], s_key[g]);
}

__global__ __launch_bounds__(T) void assign_kernel(const Args a) {
  __shared__ Staged s;
  __shared__ float s_gmax[MAX_GT];
  __shared__ unsigned s_garg[MAX_GT];
  __shared__ int s_count, s_pos[T / 64], s_neg[T / 64];
  const int b = blockIdx.z, q = blockIdx.y;
  const int G = stage(a, b, q, s, &s_count);
  const int g0 = a.d.gt_start[b];
  for (int g = threadIdx.x; g < G; g += T) {
    const unsigned long long k = a.keys[(size_t)q * a.g_total + g0 + g];
    s_gmax[g] = __uint_as_float((unsigned)(k >> 32));                       // 0 when no anchor overlaps the box
    s_garg[g] = k == 0ull ? 0u : 0xffffffffu - (unsigned)(k & 0xffffffffull);   // argmax of an all-zero row: its first entry
  }
  __syncthreads();
  const long long M = a.d.num_assigners == 1 ? (long long)a.d.cells * a.d.num_sizes * a.d.num_rots : (long long)a.d.cells * a.d.num_rots;
  const long long N = (long long)a.d.cells * a.d.num_sizes * a.d.num_rots;
  const long long m = (long long)blockIdx.x * T + threadIdx.x;
  int is_pos = 0, is_neg = 0;
  if (m < M) {
    const long long n = anchor_index(a.d, q, m);
    const float* an = a.anchors + n * 7;
    int assigned = 0;                                   // no box for this assigner: every anchor is a negative
    if (s_count > 0) {
      const Rect ra = nearest_bev(an[0], an[1], an[3], an[4], an[6]);
      const float aa = rect_area(ra);
      const float pos_thr = a.d.pos_iou_thr[q], neg_thr = a.d.neg_iou_thr[q], min_pos = a.d.min_pos_iou[q];
      float best = -1.0f;
      int arg = -1, low = -1;
      for (int g = 0; g < G; ++g) {
        if (a.d.assign_per_class && s.label[g] != q) continue;
        const float v = iou_of(s.r[g], s.area[g], ra, aa);
        if (v > best) {
          best = v;
          arg = g;
        }
        if (a.d.match_low_quality && s_gmax[g] >= min_pos) {
          const bool hit = a.d.gt_max_assign_all ? v == s_gmax[g] : (unsigned)m == s_garg[g];
          if (hit) low = g;                             // the reference's loop runs box after box: the last one stays
        }
      }
      assigned = -1;
      if (best >= 0.0f && best < neg_thr) assigned = 0;
      if (best >= pos_thr) assigned = arg + 1;
      if (low >= 0) assigned = low + 1;
    }
    const size_t o = (size_t)b * N + n;
    float t[7] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    long long lab = a.d.num_classes, dt = 0;
    float lw = 0.0f, w = 0.0f;
    if (assigned > 0) {
      const int g = assigned - 1;
      const float* gt = a.gt + (size_t)(g0 + g) * 7;
      // DeltaXYZWLHRBBoxCoder.encode(anchor, box)
      const float za = an[2] + an[5] / 2, zg = gt[2] + gt[5] / 2;
      const float diagonal = sqrtf(an[4] * an[4] + an[3] * an[3]);
      t[0] = (gt[0] - an[0]) / diagonal;
      t[1] = (gt[1] - an[1]) / diagonal;
      t[2] = (zg - za) / an[5];
      t[3] = logf(gt[3] / an[3]);
      t[4] = logf(gt[4] / an[4]);
      t[5] = logf(gt[5] / an[5]);
      t[6] = gt[6] - an[6];
      // get_direction_target(anchor, targets, dir_offset, num_bins)
      const float rot = t[6] + an[6] - a.d.dir_offset;
      const float off = rot - floorf(rot / TWO_PI_F + 0.0f) * TWO_PI_F;
      long long bin = (long long)floorf(off / (TWO_PI_F / (float)a.d.num_dir_bins));
      bin = bin < 0 ? 0 : (bin > a.d.num_dir_bins - 1 ? a.d.num_dir_bins - 1 : bin);
      dt = bin;
      lab = s.label[g];
      lw = a.d.pos_weight <= 0.0f ? 1.0f : a.d.pos_weight;
      w = 1.0f;
      is_pos = 1;
    } else if (assigned == 0) {
      lw = 1.0f;
      is_neg = 1;
    }
    a.labels[o] = lab;
    a.label_w[o] = lw;
    a.dir_t[o] = dt;
    a.dir_w[o] = w;
#pragma unroll
    for (int j = 0; j < 7; ++j) {
      a.bbox_t[o * 7 + j] = t[j];
      a.bbox_w[o * 7 + j] = w;
    }
  }
  const unsigned long long bp = __ballot(is_pos), bn = __ballot(is_neg);
  if ((threadIdx.x & 63) == 0) {
    s_pos[threadIdx.x >> 6] = __popcll(bp);
    s_neg[threadIdx.x >> 6] = __popcll(bn);
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    const int p = s_pos[0] + s_pos[1] + s_pos[2] + s_pos[3], ng = s_neg[0] + s_neg[1] + s_neg[2] + s_neg[3];
    if (p) atomicAdd(&a.counts[b * 2], p);
    if (ng) atomicAdd(&a.counts[b * 2 + 1], ng);
  }
}

}  // namespace atgt

using namespace atgt;

extern "C" {

int32_t anchor_targets_max_gt(void) { return MAX_GT; }

size_t anchor_targets_workspace_bytes(int32_t num_assigners, int32_t gt_total) {
  if (num_assigners < 1 || gt_total < 0) return 256;
  return (((size_t)num_assigners * (size_t)(gt_total > 0 ? gt_total : 1) * 8) + 255) & ~(size_t)255;
}

int anchor_targets_build(const anchor_targets_desc* desc, const float* anchors, const float* gt_boxes, const int64_t* gt_labels,
                         void* workspace, int64_t* labels, float* label_weights, float* bbox_targets, float* bbox_weights,
                         int64_t* dir_targets, float* dir_weights, int32_t* counts, void* stream) {
  if (desc == nullptr || anchors == nullptr || workspace == nullptr || labels == nullptr || label_weights == nullptr ||
      bbox_targets == nullptr || bbox_weights == nullptr || dir_targets == nullptr || dir_weights == nullptr || counts == nullptr)
    return GD3D_E_BADARG;
  const anchor_targets_desc& d = *desc;
  if (d.batch < 1 || d.batch > ANCHOR_TARGETS_MAX_BATCH || d.cells < 1 || d.num_sizes < 1 || d.num_sizes > ANCHOR_TARGETS_MAX_SIZES ||
      d.num_rots < 1 || d.num_classes < 1 || d.num_dir_bins < 1)
    return GD3D_E_BADARG;
  if (d.num_assigners != 1 && d.num_assigners != d.num_sizes) return GD3D_E_BADARG;
  if (d.num_assigners == 1 && d.assign_per_class) return GD3D_E_BADARG;
  if (d.gt_start[0] != 0) return GD3D_E_BADARG;
  for (int b = 0; b < d.batch; ++b) {
    const int g = d.gt_start[b + 1] - d.gt_start[b];
    if (g < 0) return GD3D_E_BADARG;
    if (g > MAX_GT) return GD3D_E_TOOLARGE;
  }
  const int g_total = d.gt_start[d.batch];
  if (g_total > 0 && (gt_boxes == nullptr || gt_labels == nullptr)) return GD3D_E_BADARG;
  const long long N = (long long)d.cells * d.num_sizes * d.num_rots;
  const long long M = d.num_assigners == 1 ? N : (long long)d.cells * d.num_rots;
  if (N * 7 >= 0x7fffffffLL || M >= 0xffffffffLL) return GD3D_E_TOOLARGE;
  hipStream_t s = (hipStream_t)stream;
  hipError_t e = hipMemsetAsync(workspace, 0, anchor_targets_workspace_bytes(d.num_assigners, g_total), s);
  if (e != hipSuccess) return (int)e;
  e = hipMemsetAsync(counts, 0, sizeof(int32_t) * 2 * d.batch, s);
  if (e != hipSuccess) return (int)e;
  Args a;
  a.d = d;
  a.anchors = anchors;
  a.gt = gt_boxes;
  a.gt_labels = (const long long*)gt_labels;
  a.keys = (unsigned long long*)workspace;
  a.labels = (long long*)labels;
  a.label_w = label_weights;
  a.bbox_t = bbox_targets;
  a.bbox_w = bbox_weights;
  a.dir_t = (long long*)dir_targets;
  a.dir_w = dir_weights;
  a.counts = counts;
  a.g_total = g_total > 0 ? g_total : 1;
  const dim3 grid((unsigned)((M + T - 1) / T), (unsigned)d.num_assigners, (unsigned)d.batch);
  if (g_total > 0) hipLaunchKernelGGL(iou_max_kernel, grid, dim3(T), 0, s, a);
  hipLaunchKernelGGL(assign_kernel, grid, dim3(T), 0, s, a);
  return (int)hipGetLastError();
}

}  // extern "C"
