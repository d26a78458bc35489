// rbox_cpu.cpp — the `_cpu` twins of the rotated-box entry points (include/gd3d.h).
//
// The reference's pairwise IoU helpers ARE CPU code (ops/eval/affinity.cpp:8-105, numpy in and out); these twins give the
// library's counterparts a host form: csrc/rbox_device.h — the geometry of the GPU kernels, every step a single IEEE fp32
// operation in a fixed order — compiled for the host with the same `-ffp-contract=off`, one pair per loop iteration.  Because
// the operation sequence is the same source, the results are BIT-IDENTICAL to the HIP kernels' (tests/test_cpu_rbox.py on
// the CPU against the fixtures of the compiled reference; tests/test_gpu_rbox.py::test_cpu_twins_equal_the_hip_kernels).
// Host memory in and out, no stream, no HIP call.  Rows are split over `nthreads` std::threads (<= 0: hardware concurrency).
#define GD3D_HOST_TWIN 1
#include "rbox_device.h"

#include "../../include/gd3d.h"
#include "host_threads.h"

#include <cstring>
#include <vector>

namespace {

using namespace rbox;
using gd3d_host::parallel_ranges;

int rows_team(int32_t nthreads, int64_t rows, int64_t cols) {   // >= ~4096 pairs per thread; small matrices run inline
  const int64_t per = cols > 0 ? (4096 + cols - 1) / cols : 1;
  return gd3d_host::team_size(nthreads, rows, per, 2 * per);
}

template <bool IS3D>
int eval_iou_rows(const float* det, int64_t nd, const float* gt, int64_t ng, float z_offset, float* iou, int32_t nthreads) {
  if (nd < 0 || ng < 0) return GD3D_E_BADARG;
  if (nd == 0 || ng == 0) return 0;
  if (det == nullptr || gt == nullptr || iou == nullptr) return GD3D_E_BADARG;
  const bool team_ok = parallel_ranges(nd, rows_team(nthreads, nd, ng), [&](int64_t a, int64_t b) {
    HullScratch<1> hs;
    for (int64_t i = a; i < b; ++i)
      for (int64_t j = 0; j < ng; ++j) {
        float d[7], g[7];
        for (int k = 0; k < 7; ++k) {
          d[k] = det[i * 7 + k];
          g[k] = gt[j * 7 + k];
        }
        iou[i * ng + j] = eval_iou<IS3D, 1>(d, g, z_offset, hs, 0);
      }
  });
  return team_ok ? 0 : GD3D_E_HOST;
}

}  // namespace

extern "C" {

int riou_bev_xyxyr_cpu(const float* a, int64_t na, const float* b, int64_t nb, float* iou, int32_t nthreads) {
  if (na < 0 || nb < 0) return GD3D_E_BADARG;
  if (na == 0 || nb == 0) return 0;
  if (a == nullptr || b == nullptr || iou == nullptr) return GD3D_E_BADARG;
  const bool team_ok = parallel_ranges(na, rows_team(nthreads, na, nb), [&](int64_t r0, int64_t r1) {
    VertexScratch<1> vs;
    for (int64_t i = r0; i < r1; ++i) {
      OBox A;
      obox_make(a + i * 5, A);
      for (int64_t j = 0; j < nb; ++j) {
        OBox B;
        obox_make(b + j * 5, B);
        iou[i * nb + j] = iou_bev<1>(A, B, vs, 0);
      }
    }
  });
  return team_ok ? 0 : GD3D_E_HOST;
}

int riou_eval_bev_cpu(const float* det, int64_t nd, const float* gt, int64_t ng, float* iou, int32_t nthreads) {
  return eval_iou_rows<false>(det, nd, gt, ng, 0.5f, iou, nthreads);
}

int riou_eval_3d_cpu(const float* det, int64_t nd, const float* gt, int64_t ng, float z_offset, float* iou, int32_t nthreads) {
  return eval_iou_rows<true>(det, nd, gt, ng, z_offset, iou, nthreads);
}

int riou_eval_trans_bev_cpu(const float* det, int64_t nd, int32_t det_cols, const float* gt, int64_t ng, int32_t gt_cols,
                            float* dist, int32_t nthreads) {
  if (nd < 0 || ng < 0 || det_cols < 2 || gt_cols < 2) return GD3D_E_BADARG;
  if (nd == 0 || ng == 0) return 0;
  if (det == nullptr || gt == nullptr || dist == nullptr) return GD3D_E_BADARG;
  const bool team_ok = parallel_ranges(nd, rows_team(nthreads, nd, ng * 16), [&](int64_t a, int64_t b) {
    for (int64_t i = a; i < b; ++i)
      for (int64_t j = 0; j < ng; ++j) {
        const float dx = det[i * det_cols] - gt[j * gt_cols], dy = det[i * det_cols + 1] - gt[j * gt_cols + 1];
        dist[i * ng + j] = std::sqrt(dx * dx + dy * dy);   // affinity.cpp:98-100 in fp32, correctly rounded
      }
  });
  return team_ok ? 0 : GD3D_E_HOST;
}

// match_coco on the host: the matcher kernel's own statement of matcher.cpp:8-74 (csrc/eval_match.hip) — per threshold the
// detections in row order, each taking the minimum of key = (is_ignore, cost, -index) over the gts with cost <= thr that are
// still free (or crowd): a non-ignore gt beats any ignore gt, ties go to the later gt, NaN never matches, -0 == +0.
// Integer output: bit-exact with the kernel and with the reference.  Thresholds are independent: one contiguous run per thread.
int eval_match_coco_cpu(const float* cost, const float* cost_thrs, const uint8_t* is_ignore, const uint8_t* is_crowd, int64_t nd,
                        int64_t ng, int64_t nt, int32_t* matched, int32_t nthreads) {
  // the argument rules of eval_match_coco (csrc/eval_match.hip), sizes included
  if (nd < 0 || ng < 0 || nt < 0) return GD3D_E_BADARG;
  if (nt == 0 || nd == 0) return 0;
  if (matched == nullptr || cost_thrs == nullptr) return GD3D_E_BADARG;
  if (ng > 0 && (cost == nullptr || is_ignore == nullptr || is_crowd == nullptr)) return GD3D_E_BADARG;
  if (nd > 0x7fffffffLL || nt > 0x7fffffffLL || ng > 1048576) return GD3D_E_TOOLARGE;
  auto ordered = [](float v) -> uint32_t {   // order-preserving bits of a float; -0 -> +0 (the reference compares floats)
    v += 0.0f;
    uint32_t u;
    std::memcpy(&u, &v, 4);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
  };
  const bool team_ok = parallel_ranges(nt, gd3d_host::team_size(nthreads, nt, 1, nd * ng < 65536 ? (int64_t)1 << 40 : 2), [&](int64_t t0, int64_t t1) {
    std::vector<unsigned char> taken((size_t)ng);
    for (int64_t t = t0; t < t1; ++t) {
      std::fill(taken.begin(), taken.end(), (unsigned char)0);
      const float thr = cost_thrs[t];
      for (int64_t d = 0; d < nd; ++d) {
        const float* row = cost + d * ng;
        uint64_t best = ~0ull;
        for (int64_t g = 0; g < ng; ++g) {
          if (!((!taken[(size_t)g] || is_crowd[g]) && row[g] <= thr)) continue;
          const uint64_t key = ((uint64_t)(is_ignore[g] ? 1u : 0u) << 63) | ((uint64_t)ordered(row[g]) << 31) |
                               (uint64_t)(0x7fffffffu - (uint32_t)g);
          if (key < best) best = key;
        }
        int32_t m = -1;
        if (best != ~0ull) {
          m = (int32_t)(0x7fffffffu - (uint32_t)(best & 0x7fffffffull));
          taken[(size_t)m] = 1;
        }
        matched[t * nd + d] = m;
      }
    }
  });
  return team_ok ? 0 : GD3D_E_HOST;
}

// greedy NMS on score-sorted boxes: box i suppresses a later box j iff iou(box_i, box_j) > thresh (argument order i, j) — the
// decisions of rnms_bev / rnms_normal_bev, bit for bit (same geometry source; the scan is inherently serial on one thread)
static int nms_cpu(bool normal, const float* boxes_sorted, int64_t n, float thresh, int64_t* keep, int64_t* num_keep) {
  if (n < 0 || num_keep == nullptr) return GD3D_E_BADARG;
  *num_keep = 0;
  if (n == 0) return 0;
  if (boxes_sorted == nullptr || keep == nullptr) return GD3D_E_BADARG;
  std::vector<OBox> ob(normal ? 0 : (size_t)n);
  for (int64_t i = 0; i < (normal ? 0 : n); ++i) obox_make(boxes_sorted + i * 5, ob[(size_t)i]);
  std::vector<unsigned char> dead((size_t)n, 0);
  VertexScratch<1> vs;
  int64_t nk = 0;
  for (int64_t i = 0; i < n; ++i) {
    if (dead[(size_t)i]) continue;
    keep[nk++] = i;
    for (int64_t j = i + 1; j < n; ++j) {
      if (dead[(size_t)j]) continue;
      const float v = normal ? iou_normal(boxes_sorted + i * 5, boxes_sorted + j * 5) : iou_bev<1>(ob[(size_t)i], ob[(size_t)j], vs, 0);
      if (v > thresh) dead[(size_t)j] = 1;
    }
  }
  *num_keep = nk;
  return 0;
}

int rnms_bev_cpu(const float* boxes_sorted, int64_t n, float thresh, int64_t* keep, int64_t* num_keep) {
  return nms_cpu(false, boxes_sorted, n, thresh, keep, num_keep);
}

int rnms_normal_bev_cpu(const float* boxes_sorted, int64_t n, float thresh, int64_t* keep, int64_t* num_keep) {
  return nms_cpu(true, boxes_sorted, n, thresh, keep, num_keep);
}

}  // extern "C"
