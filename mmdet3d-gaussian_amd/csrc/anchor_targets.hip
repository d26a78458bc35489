// anchor_targets.hip — target assignment of the anchor heads for a whole batch on gfx950 (include/gd3d.h, ABI 4).
//
// The reference's GDAnchor3DHead.loss calls `self.anchor_target_3d(...)` at
//   /root/reference/mmdet3d_gaussian/models/dense_heads/gd_anchor3d_head.py:206-214
// and inherits it from mmdet3d / mmdet (third party, absent: restated from the published text):
// per sample and per size class a nearest-BEV IoU matrix (gts x 107 k anchors), MaxIoUAssigner (max / argmax both ways, a Python
// loop over the gts with a full-row compare each), PseudoSampler (two nonzero + unique), DeltaXYZWLHR encode, direction bins and
// six scatters: ~60 launches and several host syncs per (sample, class), 18 such calls per KITTI batch of 6.
// Here, two launches for the batch, no sync:
//   pass 1  iou_max_kernel : a workgroup owns a tile of cells with all their anchors, walked size by size (a wave works for one
//                            assigner at a time); the sample's boxes sit in LDS as nearest-BEV rectangles;
//                            per box the best overlap over the assigner's anchors as one 64-bit key (IoU bits, then lowest anchor
//                            index) — LDS atomicMax per workgroup, one global atomicMax per (workgroup, box) that any anchor touches.
//   both passes first cull the boxes against the bounding rectangle of the tile's anchors: a tile usually meets none or a few, so
//   the inner loops are short and both passes stream (pass 2 is bound by writing the six target arrays).
//   pass 2  assign_kernel  : recomputes the thread's overlaps (same instructions, same bits), applies the assigner's rules in the
//                            reference's order (negative below neg_iou_thr, positive from pos_iou_thr on, then box after box its
//                            best anchors when that best reaches min_pos_iou — later boxes overwrite earlier ones), stages the
//                            tile's labels, weights, encoded regression targets and direction bins in LDS in the head's
//                            (h, w, size, rotation) order and writes them out as contiguous ranges; positives / negatives per
//                            sample counted with integer atomics.
// Integer atomics only: the result does not depend on scheduling.  Built with -ffp-contract=off: the IoU (+, -, x, /, max, min)
// carries the rounding of the torch elementwise ops, so thresholds and ties decide as they do there.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/gd3d_extras.h"

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
  unsigned long long* keys;    // (K, G_total), K = 1 or S (see key_rows)
  long long* labels;           // (B, N)
  float* label_w;              // (B, N)
  float* bbox_t;               // (B, N, 7)
  float* bbox_w;               // (B, N, 7)
  long long* dir_t;            // (B, N)
  float* dir_w;                // (B, N)
  int* counts;                 // (B, 2)
  int g_total, g_cap;          // all boxes; LDS capacity per sample (the largest sample, rounded up)
  int tile_cells;              // cells of a workgroup's tile
};

// A box's best overlap is kept per (assigner, box).  With assign_per_class only assigner `label` ever sees the box, with one
// assigner there is one row anyway: K = 1 row of keys; a list of assigners that all see every box needs K = S rows.
__device__ __forceinline__ int key_rows(const anchor_targets_desc& d) { return (d.num_assigners > 1 && !d.assign_per_class) ? d.num_sizes : 1; }

// dynamic LDS: [Rect r[g_cap]] [float area[g_cap]] [int label[g_cap]] then the kernel's own arrays
struct Staged {
  Rect* r;
  float* area;
  int* label;
  unsigned char* rest;
};

__device__ __forceinline__ Staged carve(unsigned char* smem, int g_cap) {
  Staged s;
  s.r = (Rect*)smem;
  s.area = (float*)(smem + (size_t)g_cap * 16);
  s.label = (int*)(smem + (size_t)g_cap * 20);
  s.rest = smem + (size_t)g_cap * 24;
  return s;
}

// the sample's boxes -> LDS as nearest-BEV rectangles
__device__ __forceinline__ int stage(const Args& a, int b, const Staged& s) {
  const int g0 = a.d.gt_start[b], G = a.d.gt_start[b + 1] - g0;
  for (int g = threadIdx.x; g < G; g += T) {
    const float* row = a.gt + (size_t)(g0 + g) * 7;
    const Rect q4 = nearest_bev(row[0], row[1], row[3], row[4], row[6]);
    s.r[g] = q4;
    s.area[g] = rect_area(q4);
    const long long lab = a.gt_labels[g0 + g];
    s.label[g] = (lab >= 0 && lab < (1 << 24) - 1) ? (int)lab : -1;
  }
  return G;
}

// A workgroup owns tile_cells consecutive cells of one sample with all their S x R anchors: items i = t, t + T, ... in size-major
// order (size q = i / (cells x R)), so that a wave works for one assigner at a time (tile_cells x R is a multiple of 64 at the
// reference's geometries) and skips the boxes of the other classes as a whole.  j = the anchor's offset in the tile in the
// head's order (cell, size, rotation).
struct Item {
  int q, j;
  bool live;
};

__device__ __forceinline__ Item item_of(const anchor_targets_desc& d, int tile_cells, int ncell, int i) {
  const int per_q = tile_cells * d.num_rots;
  Item it;
  it.q = i / per_q;
  const int rest = i - it.q * per_q;
  const int cell = rest / d.num_rots, r = rest - cell * d.num_rots;
  it.j = (cell * d.num_sizes + it.q) * d.num_rots + r;
  it.live = cell < ncell;
  return it;
}


// Boxes that can overlap an anchor of the tile: those whose rectangle meets the bounding rectangle of the tile's anchor rectangles
// (overlap > 0 needs min(g.x2, a.x2) > max(g.x1, a.x1), hence g.x2 > min_a a.x1 and g.x1 < max_a a.x2; likewise in y — whatever the
// signs of the sizes).  Every other box has overlap exactly 0 with every anchor of the tile.  s_act receives their indices in
// ascending order (the order the assigner's rules depend on); returns their number.  Ends with a barrier.
__device__ __forceinline__ int active_boxes(const Args& a, const Staged& s, int G, const float* tile, int ncell, int* s_act, float* s_red,
                                            int* s_nact) {
  const int SR = a.d.num_sizes * a.d.num_rots;
  float bx1 = INFINITY, by1 = INFINITY, bx2 = -INFINITY, by2 = -INFINITY;
  for (int j = threadIdx.x; j < ncell * SR; j += T) {
    const float* an = tile + (size_t)j * 7;
    const Rect r = nearest_bev(an[0], an[1], an[3], an[4], an[6]);
    bx1 = fminf(bx1, r.x1);
    by1 = fminf(by1, r.y1);
    bx2 = fmaxf(bx2, r.x2);
    by2 = fmaxf(by2, r.y2);
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    bx1 = fminf(bx1, __shfl_xor(bx1, off, 64));
    by1 = fminf(by1, __shfl_xor(by1, off, 64));
    bx2 = fmaxf(bx2, __shfl_xor(bx2, off, 64));
    by2 = fmaxf(by2, __shfl_xor(by2, off, 64));
  }
  if ((threadIdx.x & 63) == 0) {
    float* o = s_red + (threadIdx.x >> 6) * 4;
    o[0] = bx1; o[1] = by1; o[2] = bx2; o[3] = by2;
  }
  __syncthreads();
  if (threadIdx.x < 64) {
    bx1 = fminf(fminf(s_red[0], s_red[4]), fminf(s_red[8], s_red[12]));
    by1 = fminf(fminf(s_red[1], s_red[5]), fminf(s_red[9], s_red[13]));
    bx2 = fmaxf(fmaxf(s_red[2], s_red[6]), fmaxf(s_red[10], s_red[14]));
    by2 = fmaxf(fmaxf(s_red[3], s_red[7]), fmaxf(s_red[11], s_red[15]));
    const int lane = threadIdx.x;
    int n = 0;
    for (int base = 0; base < G; base += 64) {
      const int g = base + lane;
      bool act = false;
      if (g < G) {
        const Rect r = s.r[g];
        act = r.x2 >= bx1 && r.x1 <= bx2 && r.y2 >= by1 && r.y1 <= by2;
      }
      const unsigned long long m = __ballot(act);
      if (act) s_act[n + __popcll(m & ((1ull << lane) - 1ull))] = g;
      n += __popcll(m);
    }
    if (lane == 0) *s_nact = n;
  }
  __syncthreads();
  return *s_nact;
}

__global__ __launch_bounds__(T) void iou_max_kernel(const Args a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  __shared__ float s_red[16];
  __shared__ int s_nact;
  const Staged s = carve(smem, a.g_cap);
  const int K = key_rows(a.d);
  unsigned long long* s_key = (unsigned long long*)s.rest;          // (K, g_cap)
  int* s_act = (int*)(s.rest + (size_t)K * a.g_cap * 8);            // (g_cap)
  const int b = blockIdx.y;
  const int G = stage(a, b, s);
  if (G == 0) return;
  for (int k = threadIdx.x; k < K * a.g_cap; k += T) s_key[k] = 0ull;
  __syncthreads();
  const int cell0 = blockIdx.x * a.tile_cells;
  const int ncell = min(a.tile_cells, a.d.cells - cell0);
  const int SR = a.d.num_sizes * a.d.num_rots;
  const float* tile = a.anchors + (size_t)cell0 * SR * 7;
  const int nact = active_boxes(a, s, G, tile, ncell, s_act, s_red, &s_nact);
  if (nact == 0) return;                                            // nothing here overlaps anything: no key changes
  const bool one = a.d.num_assigners == 1;
  for (int i = threadIdx.x; i < a.tile_cells * SR; i += T) {
    const Item it = item_of(a.d, a.tile_cells, ncell, i);
    if (!it.live) continue;
    const float* an = tile + (size_t)it.j * 7;
    const Rect ra = nearest_bev(an[0], an[1], an[3], an[4], an[6]);
    const float aa = rect_area(ra);
    // position of the anchor among the anchors its assigner sees (first maximum = lowest position)
    const long long n = (long long)cell0 * SR + it.j;
    const long long m = one ? n : ((n / SR) * a.d.num_rots + n % a.d.num_rots);
    const unsigned low = 0xffffffffu - (unsigned)m;
    unsigned long long* keys = s_key + (K > 1 ? (size_t)it.q * a.g_cap : 0);
    for (int k = 0; k < nact; ++k) {
      const int g = s_act[k];
      if (a.d.assign_per_class && s.label[g] != it.q) continue;
      const float v = iou_of(s.r[g], s.area[g], ra, aa);
      if (v > 0.0f) atomicMax(&keys[g], ((unsigned long long)__float_as_uint(v) << 32) | low);
    }
  }
  __syncthreads();
  const int g0 = a.d.gt_start[b];
  for (int k = threadIdx.x; k < K * nact; k += T) {
    const int row = k / nact, g = s_act[k - row * nact];
    const unsigned long long v = s_key[(size_t)row * a.g_cap + g];
    if (v != 0ull) atomicMax(&a.keys[(size_t)row * a.g_total + g0 + g], v);
  }
}

__global__ __launch_bounds__(T) void assign_kernel(const Args a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  __shared__ int s_pos[T / 64], s_neg[T / 64], s_first[ANCHOR_TARGETS_MAX_SIZES], s_zero[ANCHOR_TARGETS_MAX_SIZES], s_nact;
  __shared__ float s_red[16];
  const Staged s = carve(smem, a.g_cap);
  const int K = key_rows(a.d);
  float* s_gmax = (float*)s.rest;                                    // (K, g_cap)
  unsigned* s_garg = (unsigned*)(s.rest + (size_t)K * a.g_cap * 4);  // (K, g_cap)
  int* s_act = (int*)(s.rest + (size_t)K * a.g_cap * 8);             // (g_cap)
  float* st = (float*)(s.rest + (size_t)K * a.g_cap * 8 + (size_t)a.g_cap * 4);   // staged outputs of the tile: (items, 9) words
  const int b = blockIdx.y;
  if (threadIdx.x < ANCHOR_TARGETS_MAX_SIZES) {
    s_first[threadIdx.x] = 0x7fffffff;
    s_zero[threadIdx.x] = -1;
  }
  const int G = stage(a, b, s);
  const int g0 = a.d.gt_start[b];
  for (int k = threadIdx.x; k < K * G; k += T) {
    const int row = k / G, g = k - row * G;
    const unsigned long long key = a.keys[(size_t)row * a.g_total + g0 + g];
    s_gmax[(size_t)row * a.g_cap + g] = __uint_as_float((unsigned)(key >> 32));                    // 0: no anchor overlaps the box
    s_garg[(size_t)row * a.g_cap + g] = key == 0ull ? 0u : 0xffffffffu - (unsigned)(key & 0xffffffffull);   // argmax of an all-zero row: its first entry
  }
  __syncthreads();
  // per assigner q: its first box (`len(gt_bboxes) > 0` of anchor_target_single_assigner, and the argmax of an anchor that overlaps
  // nothing) and its last box that NO anchor overlaps while 0 >= min_pos_iou — such a box "best-matches" every anchor at overlap 0
  const int Q = a.d.num_assigners;
  for (int g = threadIdx.x; g < G; g += T) {
    const int lab = s.label[g];
    const int q_lo = a.d.assign_per_class ? lab : 0, q_hi = a.d.assign_per_class ? lab + 1 : Q;
    if (a.d.assign_per_class && (lab < 0 || lab >= Q)) continue;
    for (int q = q_lo; q < q_hi; ++q) {
      atomicMin(&s_first[q], g);
      const float gm = s_gmax[(K > 1 ? (size_t)q * a.g_cap : 0) + g];
      if (a.d.match_low_quality && gm == 0.0f && gm >= a.d.min_pos_iou[q]) atomicMax(&s_zero[q], g);
    }
  }
  __syncthreads();
  const int cell0 = blockIdx.x * a.tile_cells;
  const int ncell = min(a.tile_cells, a.d.cells - cell0);
  const int SR = a.d.num_sizes * a.d.num_rots;
  const float* tile = a.anchors + (size_t)cell0 * SR * 7;
  const int nact = G > 0 ? active_boxes(a, s, G, tile, ncell, s_act, s_red, &s_nact) : 0;
  const bool one = a.d.num_assigners == 1;
  int n_pos = 0, n_neg = 0;
  for (int i = threadIdx.x; i < a.tile_cells * SR; i += T) {
    const Item it = item_of(a.d, a.tile_cells, ncell, i);
    if (!it.live) continue;
    const int q = one ? 0 : it.q;                       // the assigner of this anchor
    const float* an = tile + (size_t)it.j * 7;
    int assigned = 0;                                   // no box for this assigner: every anchor is a negative
    if (s_first[q] != 0x7fffffff) {
      const Rect ra = nearest_bev(an[0], an[1], an[3], an[4], an[6]);
      const float aa = rect_area(ra);
      const float pos_thr = a.d.pos_iou_thr[q], neg_thr = a.d.neg_iou_thr[q], min_pos = a.d.min_pos_iou[q];
      const long long n = (long long)cell0 * SR + it.j;
      const unsigned m = (unsigned)(one ? n : ((n / SR) * a.d.num_rots + n % a.d.num_rots));
      const float* gmax = s_gmax + (K > 1 ? (size_t)q * a.g_cap : 0);
      const unsigned* garg = s_garg + (K > 1 ? (size_t)q * a.g_cap : 0);
      // the boxes outside the active list overlap this anchor by exactly 0: the best overlap starts at 0 with the assigner's first
      // box as its argmax (first maximum), and only a strictly larger overlap moves it
      float best = 0.0f;
      int arg = s_first[q], low = -1;
      for (int k = 0; k < nact; ++k) {
        const int g = s_act[k];
        if (a.d.assign_per_class && s.label[g] != q) continue;
        const float v = iou_of(s.r[g], s.area[g], ra, aa);
        if (v > best) {
          best = v;
          arg = g;
        }
        if (a.d.match_low_quality && gmax[g] >= min_pos) {
          const bool hit = a.d.gt_max_assign_all ? v == gmax[g] : m == garg[g];
          if (hit) low = g;                             // the reference's loop runs box after box: the last one stays
        }
      }
      // a box no anchor overlaps (best overlap 0 >= min_pos_iou) matches every anchor at overlap 0 (gt_max_assign_all), or the
      // first anchor (its argmax); the last such box competes with the last hit above
      const int z = s_zero[q];
      if (z > low && (a.d.gt_max_assign_all || m == 0u)) low = z;
      assigned = -1;
      if (best >= 0.0f && best < neg_thr) assigned = 0;
      if (best >= pos_thr) assigned = arg + 1;
      if (low >= 0) assigned = low + 1;
    }
    float t[7] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    int lab = a.d.num_classes, dt = 0, pos = 0;
    float lw = 0.0f;
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
      int bin = (int)floorf(off / (TWO_PI_F / (float)a.d.num_dir_bins));
      bin = bin < 0 ? 0 : (bin > a.d.num_dir_bins - 1 ? a.d.num_dir_bins - 1 : bin);
      dt = bin;
      lab = s.label[g];
      lw = a.d.pos_weight <= 0.0f ? 1.0f : a.d.pos_weight;
      pos = 1;
      ++n_pos;
    } else if (assigned == 0) {
      lw = 1.0f;
      ++n_neg;
    }
    float* o = st + (size_t)it.j * 9;
#pragma unroll
    for (int k = 0; k < 7; ++k) o[k] = t[k];
    o[7] = lw;
    ((int*)o)[8] = ((lab + 1) & 0xffffff) | (dt << 24) | (pos << 31);   // labels in [-1, 2^24 - 2] (a box labelled -1 hands its label on when
                                                                        // the assigner is not per class), direction bins < 128
  }
  __syncthreads();
  // the tile's outputs are contiguous ranges of the six arrays: written in whole lines
  const int items = ncell * SR;
  const size_t N = (size_t)a.d.cells * SR;
  const size_t o0 = (size_t)b * N + (size_t)cell0 * SR;
  for (int j = threadIdx.x; j < items; j += T) {
    const int packed = ((const int*)st)[(size_t)j * 9 + 8];
    const float w = packed < 0 ? 1.0f : 0.0f;
    a.labels[o0 + j] = (long long)(packed & 0xffffff) - 1;
    a.dir_t[o0 + j] = (long long)((packed >> 24) & 0x7f);
    a.label_w[o0 + j] = st[(size_t)j * 9 + 7];
    a.dir_w[o0 + j] = w;
  }
  const int total = items * 7;
  const int quads = (((o0 * 7) & 3) == 0) ? total / 4 : 0;          // 16-byte stores where the tile's range is aligned for them
  float4* t4 = (float4*)(a.bbox_t + o0 * 7);
  float4* w4 = (float4*)(a.bbox_w + o0 * 7);
  for (int k4 = threadIdx.x; k4 < quads; k4 += T) {
    float tv[4], wv[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int k = k4 * 4 + e, j = k / 7, c = k - j * 7;
      tv[e] = st[(size_t)j * 9 + c];
      wv[e] = ((const int*)st)[(size_t)j * 9 + 8] < 0 ? 1.0f : 0.0f;
    }
    t4[k4] = make_float4(tv[0], tv[1], tv[2], tv[3]);
    w4[k4] = make_float4(wv[0], wv[1], wv[2], wv[3]);
  }
  for (int k = quads * 4 + threadIdx.x; k < total; k += T) {
    const int j = k / 7, c = k - j * 7;
    a.bbox_t[o0 * 7 + k] = st[(size_t)j * 9 + c];
    a.bbox_w[o0 * 7 + k] = ((const int*)st)[(size_t)j * 9 + 8] < 0 ? 1.0f : 0.0f;
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    n_pos += __shfl_down(n_pos, off, 64);
    n_neg += __shfl_down(n_neg, off, 64);
  }
  if ((threadIdx.x & 63) == 0) {
    s_pos[threadIdx.x >> 6] = n_pos;
    s_neg[threadIdx.x >> 6] = n_neg;
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

__global__ __launch_bounds__(256) static void atgt_zero_kernel(unsigned* a, long long na, unsigned* b, long long nb) {
  const long long stride = (long long)gridDim.x * 256;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < na; i += stride) a[i] = 0u;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < nb; i += stride) b[i] = 0u;
}

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
  if (d.num_assigners == 1 && d.assign_per_class && d.num_sizes != 1) return GD3D_E_BADARG;   // one assigner per class needs one per size
  if (d.gt_start[0] != 0) return GD3D_E_BADARG;
  int g_max = 0;
  for (int b = 0; b < d.batch; ++b) {
    const int g = d.gt_start[b + 1] - d.gt_start[b];
    if (g < 0) return GD3D_E_BADARG;
    if (g > MAX_GT) return GD3D_E_TOOLARGE;
    g_max = g > g_max ? g : g_max;
  }
  if (d.num_classes >= (1 << 24) - 1 || d.num_dir_bins > 127) return GD3D_E_TOOLARGE;
  const int g_total = d.gt_start[d.batch];
  if (g_total > 0 && (gt_boxes == nullptr || gt_labels == nullptr)) return GD3D_E_BADARG;
  const long long N = (long long)d.cells * d.num_sizes * d.num_rots;
  const long long M = d.num_assigners == 1 ? N : (long long)d.cells * d.num_rots;
  if (N * 7 >= 0x7fffffffLL || M >= 0xffffffffLL) return GD3D_E_TOOLARGE;
  hipStream_t s = (hipStream_t)stream;
  const int K = (d.num_assigners > 1 && !d.assign_per_class) ? d.num_sizes : 1;
  // tile: about 768 anchors (3 rounds of the 256 threads), a power of two of cells
  const int SR = d.num_sizes * d.num_rots;
  int tile_cells = 1;
  while (tile_cells * 2 * SR <= 768 && tile_cells < 512) tile_cells *= 2;   // measured at KITTI geometry: 192 / 384 / 768 / 1536 anchors per
                                                                          // tile -> 165 / 108 / 83 / 85 us for the two passes
  const int g_cap = (g_max + 3) & ~3;
  const size_t lds1 = (size_t)g_cap * 28 + (size_t)K * g_cap * 8;
  while (tile_cells > 1 && lds1 + (size_t)tile_cells * SR * 36 > 64 * 1024) tile_cells /= 2;   // many boxes: smaller tiles
  const size_t lds2 = lds1 + (size_t)tile_cells * SR * 36;
  if (lds2 > 64 * 1024) return GD3D_E_TOOLARGE;
  // cleared by a kernel, not by hipMemsetAsync: inside a captured hipGraph a memset node was found not to be reliably ordered
  // against the kernels around it on this ROCm (round 4, csrc/rbox.hip rank_place_kernel), and this call is replayed in graphs
  {
    const size_t w4 = anchor_targets_workspace_bytes(K, g_total) / 4, c4 = (size_t)2 * d.batch;
    const size_t most = w4 > c4 ? w4 : c4;
    unsigned zb = (unsigned)((most + 255) / 256);
    if (zb > 1024) zb = 1024;
    hipLaunchKernelGGL(atgt_zero_kernel, dim3(zb), dim3(256), 0, s, (unsigned*)workspace, (long long)w4, (unsigned*)counts, (long long)c4);
  }
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
  a.g_cap = g_cap;
  a.tile_cells = tile_cells;
  const dim3 grid((unsigned)((d.cells + tile_cells - 1) / tile_cells), (unsigned)d.batch);
  if (g_total > 0) hipLaunchKernelGGL(iou_max_kernel, grid, dim3(T), lds1, s, a);
  hipLaunchKernelGGL(assign_kernel, grid, dim3(T), lds2, s, a);
  return (int)hipGetLastError();
}

}  // extern "C"
