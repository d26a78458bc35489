// coder_device.h — the CenterPoint decode arithmetic shared by coders.hip (the coder's own entry points) and
// center_infer.hip (the inference slice): one statement of
//   /root/reference/mmdet3d_gaussian/core/bbox/coders/centerpoint_bbox_coders.py:98-108 and
//   /root/reference/mmdet3d_gaussian/core/bbox/coders/centerpoint_bbox_yaw_coders.py:32-50
// in the rounding sequence of the torch elementwise ops (both translation units are built with -ffp-contract=off).
#pragma once
#include <hip/hip_runtime.h>

namespace gdcoder {

constexpr float HALF_PI = 1.57079632679489661923f;

struct Geom {
  float osf, vs0, vs1, pc0, pc1;
  int norm_bbox;
};

struct Core {
  float x, y, d0, d1, d2, yaw;
  int k;   // parity of the quarter turns (1: w and l swapped)
};

// p0, p1: cell offsets; p3..p5: (log) dims; yaw, ds, dc: yaw and the (sin, cos) direction channels (yaw coder only)
__device__ __forceinline__ Core decode_core(float p0, float p1, float p3, float p4, float p5, float yaw, float ds, float dc,
                                            float lx, float ly, const Geom& g, int correct_yaw) {
  Core c;
  c.x = (p0 + lx) * g.osf * g.vs0 + g.pc0;
  c.y = (p1 + ly) * g.osf * g.vs1 + g.pc1;
  float d0 = p3, d1 = p4, d2 = p5;
  if (g.norm_bbox) {
    d0 = expf(d0);
    d1 = expf(d1);
    d2 = expf(d2);
  }
  int k = 0;
  if (correct_yaw) {
    const float dir = atan2f(ds, dc);
    const float nr = floorf((dir - yaw) / HALF_PI + 0.5f);
    // `num_rot90.long() % 2 == 0`: parity of the truncated integer (Python % on tensors follows the divisor's sign: -1 % 2 = 1)
    const long long kl = (long long)nr;
    k = (int)(kl & 1);
    yaw = yaw + nr * HALF_PI;
    if (k) {
      const float t = d0;
      d0 = d1;
      d1 = t;
    }
  }
  c.d0 = d0;
  c.d1 = d1;
  c.d2 = d2;
  c.yaw = yaw;
  c.k = k;
  return c;
}

}  // namespace gdcoder
