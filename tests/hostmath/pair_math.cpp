// TEST INFRASTRUCTURE: csrc/gd3d_device.h compiled for the host (see hip/hip_runtime.h beside this file).
//   /opt/rocm/lib/llvm/bin/clang++ -O1 -std=c++17 -shared -fPIC -I tests/hostmath -I <repo> tests/hostmath/pair_math.cpp -o libpairmath.so
#include "mmdet3d-gaussian_amd/csrc/gd3d_device.h"

using namespace gd3d;

template <int LOSS, int FUN, bool FLAG>
static void run(const float* pred, const float* target, long n, const float* c, float alpha, float tau, float scale,
                float* loss, float* gp, float* gt) {
  const float cc[3] = {c[0], c[1], c[2]};
  for (long i = 0; i < n; ++i) {
    float pv[7], tv[7], g1[7], g2[7];
    for (int k = 0; k < 7; ++k) {
      pv[k] = pred[i * 7 + k];
      tv[k] = target[i * 7 + k];
    }
    const float L = pair_loss<LOSS, FUN, FLAG, true>(pv, tv, cc, alpha, gd3d_inv_alpha2(alpha), tau, scale, g1, g2);
    loss[i] = scale * L;
    for (int k = 0; k < 7; ++k) {
      gp[i * 7 + k] = g1[k];
      gt[i * 7 + k] = g2[k];
    }
  }
}

template <int LOSS>
static int by_fun(int fun, bool flag, const float* p, const float* t, long n, const float* c, float alpha, float tau,
                  float scale, float* loss, float* gp, float* gt) {
  if (LOSS == GD3D_KFIOU3D) {
    switch (fun) {
      case GD3D_FUN_EXPM1: run<LOSS, GD3D_FUN_EXPM1, false>(p, t, n, c, alpha, tau, scale, loss, gp, gt); return 0;
      case GD3D_FUN_NLOG: run<LOSS, GD3D_FUN_NLOG, false>(p, t, n, c, alpha, tau, scale, loss, gp, gt); return 0;
      default: run<LOSS, GD3D_FUN_NONE, false>(p, t, n, c, alpha, tau, scale, loss, gp, gt); return 0;
    }
  }
  if (fun == GD3D_FUN_LOG1P) {
    if (flag) run<LOSS, GD3D_FUN_LOG1P, true>(p, t, n, c, alpha, tau, scale, loss, gp, gt);
    else run<LOSS, GD3D_FUN_LOG1P, false>(p, t, n, c, alpha, tau, scale, loss, gp, gt);
  } else {
    if (flag) run<LOSS, GD3D_FUN_NONE, true>(p, t, n, c, alpha, tau, scale, loss, gp, gt);
    else run<LOSS, GD3D_FUN_NONE, false>(p, t, n, c, alpha, tau, scale, loss, gp, gt);
  }
  return 0;
}

// the dispatch of csrc/gd3d_loss.hip (launch_fun / launch_kfiou), one pair per loop iteration
extern "C" int hostmath_pairs(const gd3d_params* prm, const float* pred, const float* target, long n, float scale,
                              float* loss, float* gp, float* gt) {
  const float* c = prm->center_offset;
  const bool flag = prm->flag != 0;
  switch (prm->loss_type) {
    case GD3D_GWD3D: return by_fun<GD3D_GWD3D>(prm->fun, flag, pred, target, n, c, prm->alpha, prm->tau, scale, loss, gp, gt);
    case GD3D_KLD3D: return by_fun<GD3D_KLD3D>(prm->fun, flag, pred, target, n, c, prm->alpha, prm->tau, scale, loss, gp, gt);
    case GD3D_BD3D: return by_fun<GD3D_BD3D>(prm->fun, flag, pred, target, n, c, prm->alpha, prm->tau, scale, loss, gp, gt);
    case GD3D_JD3D: return by_fun<GD3D_JD3D>(prm->fun, flag, pred, target, n, c, prm->alpha, prm->tau, scale, loss, gp, gt);
    case GD3D_KLD3D_SYMMAX: return by_fun<GD3D_KLD3D_SYMMAX>(prm->fun, flag, pred, target, n, c, prm->alpha, prm->tau, scale, loss, gp, gt);
    case GD3D_KLD3D_SYMMIN: return by_fun<GD3D_KLD3D_SYMMIN>(prm->fun, flag, pred, target, n, c, prm->alpha, prm->tau, scale, loss, gp, gt);
    case GD3D_KFIOU3D: return by_fun<GD3D_KFIOU3D>(prm->fun, false, pred, target, n, c, prm->alpha, prm->tau, scale, loss, gp, gt);
    default: return 10001;
  }
}
