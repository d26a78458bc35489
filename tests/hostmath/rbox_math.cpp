// TEST INFRASTRUCTURE: csrc/rbox_device.h (rotated-rectangle geometry of the NMS and eval-IoU kernels) compiled for the
// host through the stand-in hip/hip_runtime.h beside this file.  One "thread" (NT = 1, t = 0).
//   g++ -O1 -std=c++17 -ffp-contract=off -shared -fPIC -I tests/hostmath -I <repo> tests/hostmath/rbox_math.cpp
#include "mmdet3d-gaussian_amd/csrc/rbox_device.h"

using namespace rbox;

// mmdet3d boxes_iou_bev on [x1,y1,x2,y2,ry] rows: what riou_xyxyr_kernel and the NMS mask kernels evaluate per pair
extern "C" void hostmath_iou_xyxyr(const float* a, long na, const float* b, long nb, float* out) {
  static VertexScratch<1> vs;
  for (long i = 0; i < na; ++i) {
    OBox A;
    obox_make(a + i * 5, A);
    for (long j = 0; j < nb; ++j) {
      OBox B;
      obox_make(b + j * 5, B);
      out[i * nb + j] = iou_bev<1>(A, B, vs, 0);
    }
  }
}

// ops/eval iou_bev / iou_3d on [x,y,z,w,l,h,yaw] rows: what riou_eval_kernel evaluates per pair
extern "C" void hostmath_eval_iou(const float* det, long nd, const float* gt, long ng, int is3d, float z_offset, float* out) {
  static HullScratch<1> hs;
  for (long i = 0; i < nd; ++i)
    for (long j = 0; j < ng; ++j) {
      float d[7], g[7];
      for (int k = 0; k < 7; ++k) {
        d[k] = det[i * 7 + k];
        g[k] = gt[j * 7 + k];
      }
      out[i * ng + j] = is3d ? eval_iou<true, 1>(d, g, z_offset, hs, 0) : eval_iou<false, 1>(d, g, z_offset, hs, 0);
    }
}
