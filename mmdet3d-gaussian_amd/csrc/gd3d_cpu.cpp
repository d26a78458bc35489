// gd3d_cpu.cpp — the `_cpu` twins of the Gaussian-distance loss entry points (include/gd3d.h).
//
// The reference's GDLoss.forward is device-agnostic (gaussian_distance_loss.py:280-310): on CPU tensors it runs the same
// ~110-145 ATen ops + autograd as on the GPU.  The twins keep that property for this library: the SAME per-pair closed forms
// and hand-derived gradients as the fused gfx950 kernel (csrc/gd3d_device.h, compiled here for the host through
// gd3d_host_math.h), one pass over the (N,7) rows, loss and final gradients together.  Host memory in, host memory out; no
// HIP call is made, so the twins work on a machine without a GPU.
//
// Differences from the device path, all below the tolerance the loss is graded at (1e-5):
//   * v_rcp / v_sqrt / v_rsq / v_exp / v_log are 1-ulp operations on gfx950 and IEEE-exact here;
//   * a 256-pair tile's partial sum is accumulated in fp64 in row order and rounded to fp32 once (the kernel adds fp32
//     values in a wavefront tree); the second stage is the same fixed-order fp64 sum over the tile partials, so the result
//     does not depend on the number of threads.
// Work is split into contiguous runs of tiles over `nthreads` std::threads (no OpenMP runtime is pulled into a process
// that already has torch's); training-size calls (< 64 tiles) run on the calling thread.
#define GD3D_HOST_TWIN 1
#include "gd3d_device.h"

#include <algorithm>
#include <cstring>

#include "host_threads.h"

namespace {

using namespace gd3d;

constexpr int64_t TILE = 256;           // pairs per partial sum: the workspace layout of gd3d_loss_fused
constexpr int64_t INLINE_TILES = 64;    // below this many tiles a thread team costs more than it saves

struct Job {
  const float* pred;
  const float* target;
  const float* w;    // (n)   nullable
  const float* w7;   // (n,7) nullable: row mean taken here (gaussian_distance_loss.py:295-296)
  float* loss;       // nullable
  float* gp;         // nullable
  float* gt;         // nullable
  float* partials;   // nullable: one fp32 per tile
  int64_t n;
  float scale, alpha, ia2, tau, c[3];
};

template <int LOSS, int FUN, bool FLAG, bool GT>
void run_tiles(const Job& j, int64_t t0, int64_t t1) {
  const float c[3] = {j.c[0], j.c[1], j.c[2]};
  for (int64_t t = t0; t < t1; ++t) {
    const int64_t lo = t * TILE, hi = std::min(j.n, lo + TILE);
    double acc = 0.0;
    for (int64_t i = lo; i < hi; ++i) {
      float pv[7], tv[7], g1[7], g2[7];
      for (int k = 0; k < 7; ++k) {
        pv[k] = j.pred[i * 7 + k];
        tv[k] = j.target[i * 7 + k];
      }
      float wi = 1.0f;
      if (j.w != nullptr) wi = j.w[i];
      if (j.w7 != nullptr) {   // weight.mean(dim=-1): the 7 entries summed in index order, then / 7 (as the kernel does)
        float sum = j.w7[i * 7];
        for (int k = 1; k < 7; ++k) sum += j.w7[i * 7 + k];
        wi = sum / 7.0f;
      }
      const float f = j.scale * wi;
      const float L = pair_loss<LOSS, FUN, FLAG, GT>(pv, tv, c, j.alpha, j.ia2, j.tau, f, g1, g2);
      const float fl = f * L;
      if (j.loss != nullptr) j.loss[i] = fl;
      acc += (double)fl;
      if (j.gp != nullptr)
        for (int k = 0; k < 7; ++k) j.gp[i * 7 + k] = g1[k];
      if (GT)
        for (int k = 0; k < 7; ++k) j.gt[i * 7 + k] = g2[k];
    }
    if (j.partials != nullptr) j.partials[t] = (float)acc;
  }
}

using TileFn = void (*)(const Job&, int64_t, int64_t);

template <int LOSS, int FUN>
TileFn pick_flag_gt(bool flag, bool gt) {
  switch ((flag ? 2 : 0) | (gt ? 1 : 0)) {
    case 0: return run_tiles<LOSS, FUN, false, false>;
    case 1: return run_tiles<LOSS, FUN, false, true>;
    case 2: return run_tiles<LOSS, FUN, true, false>;
    default: return run_tiles<LOSS, FUN, true, true>;
  }
}

template <int LOSS>
TileFn pick_fun(int fun, bool flag, bool gt) {
  return fun == GD3D_FUN_LOG1P ? pick_flag_gt<LOSS, GD3D_FUN_LOG1P>(flag, gt) : pick_flag_gt<LOSS, GD3D_FUN_NONE>(flag, gt);
}

// the instantiation table of csrc/gd3d_loss.hip (launch_fun / launch_kfiou): kfiou3d ignores `sqrt` (ref :228)
TileFn pick(const gd3d_params* p, bool gt) {
  const bool flag = p->flag != 0;
  switch (p->loss_type) {
    case GD3D_GWD3D: return pick_fun<GD3D_GWD3D>(p->fun, flag, gt);
    case GD3D_KLD3D: return pick_fun<GD3D_KLD3D>(p->fun, flag, gt);
    case GD3D_BD3D: return pick_fun<GD3D_BD3D>(p->fun, flag, gt);
    case GD3D_JD3D: return pick_fun<GD3D_JD3D>(p->fun, flag, gt);
    case GD3D_KLD3D_SYMMAX: return pick_fun<GD3D_KLD3D_SYMMAX>(p->fun, flag, gt);
    case GD3D_KLD3D_SYMMIN: return pick_fun<GD3D_KLD3D_SYMMIN>(p->fun, flag, gt);
    default:
      switch (p->fun) {
        case GD3D_FUN_EXPM1: return pick_flag_gt<GD3D_KFIOU3D, GD3D_FUN_EXPM1>(false, gt);
        case GD3D_FUN_NLOG: return pick_flag_gt<GD3D_KFIOU3D, GD3D_FUN_NLOG>(false, gt);
        default: return pick_flag_gt<GD3D_KFIOU3D, GD3D_FUN_NONE>(false, gt);
      }
  }
}

int team_size(int32_t nthreads, int64_t tiles) {   // at least 16 tiles (4096 pairs) per thread; < 64 tiles run inline
  return gd3d_host::team_size(nthreads, tiles, 16, INLINE_TILES);
}
using gd3d_host::parallel_ranges;

double sum_partials(const float* partials, int64_t tiles) {
  double s = 0.0;
  for (int64_t t = 0; t < tiles; ++t) s += (double)partials[t];
  return s;
}

}  // namespace

extern "C" {

int gd3d_loss_fused_cpu(const gd3d_params* p, const float* pred, const float* target, const float* row_weight,
                        const float* weight7, int64_t n, float scale, float* loss, float* loss_sum, float* grad_pred,
                        float* grad_target, void* workspace, int32_t nthreads) {
  // the argument rules of gd3d_loss_fused (csrc/gd3d_loss.hip loss_launch), minus stream and device-pointer concerns
  if (row_weight != nullptr && weight7 != nullptr) return GD3D_E_BADARG;
  if (p == nullptr || n < 0) return GD3D_E_BADARG;
  if (n > 0 && (pred == nullptr || target == nullptr)) return GD3D_E_BADARG;
  if (p->loss_type < 0 || p->loss_type >= GD3D_NUM_LOSS_TYPES) return GD3D_E_BADARG;
  if (p->loss_type == GD3D_KFIOU3D) {
    if (p->fun != GD3D_FUN_NONE && p->fun != GD3D_FUN_EXPM1 && p->fun != GD3D_FUN_NLOG) return GD3D_E_BADARG;
  } else if (p->fun != GD3D_FUN_NONE && p->fun != GD3D_FUN_LOG1P) {
    return GD3D_E_BADARG;
  }
  if (loss_sum != nullptr && workspace == nullptr) return GD3D_E_BADARG;
  const int64_t tiles = (n + TILE - 1) / TILE;
  if (tiles > 0x7fffffffLL) return GD3D_E_TOOLARGE;
  if (n == 0) {
    if (loss_sum != nullptr) *loss_sum = 0.0f;
    return 0;
  }
  if (loss_sum == nullptr && loss == nullptr && grad_pred == nullptr && grad_target == nullptr && workspace == nullptr)
    return 0;
  Job j;
  j.pred = pred;
  j.target = target;
  j.w = row_weight;
  j.w7 = weight7;
  j.loss = loss;
  j.gp = grad_pred;
  j.gt = grad_target;
  j.partials = (float*)workspace;
  j.n = n;
  j.scale = scale;
  j.alpha = p->alpha;
  j.ia2 = gd3d_inv_alpha2(p->alpha);
  j.tau = p->tau;
  for (int k = 0; k < 3; ++k) j.c[k] = p->center_offset[k];
  const TileFn fn = pick(p, grad_target != nullptr);
  if (!parallel_ranges(tiles, team_size(nthreads, tiles), [&](int64_t a, int64_t b) { fn(j, a, b); })) return GD3D_E_HOST;
  if (loss_sum != nullptr) *loss_sum = (float)sum_partials(j.partials, tiles);
  return 0;
}

int gd3d_loss_reduce_cpu(const void* workspace, int64_t n, float* loss_sum) {
  if (n < 0 || loss_sum == nullptr || (n > 0 && workspace == nullptr)) return GD3D_E_BADARG;
  *loss_sum = (float)sum_partials((const float*)workspace, (n + TILE - 1) / TILE);
  return 0;
}

int gd3d_scale_rows_cpu(float* grad, const float* g, int per_row, int64_t n, int32_t nthreads) {
  if (n < 0 || g == nullptr || (n > 0 && grad == nullptr)) return GD3D_E_BADARG;
  if (!per_row && g[0] == 1.0f) return 0;   // loss.backward(): nothing to scale (what the device kernel's early exit does)
  const int64_t tiles = (n + TILE - 1) / TILE;
  const bool ok = parallel_ranges(tiles, team_size(nthreads, tiles), [&](int64_t a, int64_t b) {
    const int64_t hi = std::min(n, b * TILE);
    for (int64_t i = a * TILE; i < hi; ++i) {
      const float s = per_row ? g[i] : g[0];
      for (int k = 0; k < 7; ++k) grad[i * 7 + k] *= s;
    }
  });
  return ok ? 0 : GD3D_E_HOST;
}

}  // extern "C"
