// coders.hip — CenterPointBBoxYawCoder, CenterPointBBoxCoderRev and PointBBoxYawCoder on the device.
//
// Replaces the ~15 elementwise ATen ops (+ their autograd nodes) of
//   /root/reference/mmdet3d_gaussian/core/bbox/coders/centerpoint_bbox_yaw_coders.py:11-16 (encode), :18-56 (decode)
// by one launch each.  One thread per box; rows are short (9-11 floats), the calls are latency-bound (K <= 500 per
// sample), so there is no LDS tiling: every thread reads its row and writes its row.
//   decode: x = (p0 + loc0) * out_size_factor * voxel_size[0] + pc_range[0], y likewise, z = p2,
//           dim = exp(p3..p5) if norm_bbox else p3..p5, yaw = p6;
//           correct_yaw: k = floor((atan2(p7, p8) - yaw) / (pi/2) + 0.5) (no gradient), yaw += k * pi/2, and w <-> h when k
//           is odd; the remaining columns p9.. are passed through.
//   The training-time decode that feeds GDLoss does not come through here: it runs inside the fused loss kernel
//   (gd3d_loss.hip, GD3D_PRO_CENTER).  This file serves inference (CenterGDHead.get_bboxes :244) and target encoding.
#include <hip/hip_runtime.h>

#include "../../include/gd3d.h"
#include "coder_device.h"

namespace gdcoder {

constexpr int T = 256;

struct DecArgs {
  const float* locs;   // (n,2)
  const float* preds;  // (n,c)
  float* out;          // (n,co)
  int* swapk;          // (n) nullable: num_rot90 of the forward, for the backward
  long long n;
  int c, co, mode;     // mode 0: yaw as it is, 1: correct_yaw, 2: CenterPointBBoxCoderRev (rot = atan2(p6, p7))
  Geom g;
};

__global__ __launch_bounds__(T) void center_decode_kernel(const DecArgs a) {
  const long long i = (long long)blockIdx.x * T + threadIdx.x;
  if (i >= a.n) return;
  const float* p = a.preds + i * a.c;
  float* o = a.out + i * a.co;
  const bool rev = a.mode == 2;
  const Core c = decode_core(p[0], p[1], p[3], p[4], p[5], rev ? 0.f : p[6], a.mode == 1 ? p[7] : 0.f, a.mode == 1 ? p[8] : 0.f,
                             a.locs[i * 2], a.locs[i * 2 + 1], a.g, a.mode == 1);
  o[0] = c.x;
  o[1] = c.y;
  o[2] = p[2];
  o[3] = c.d0;
  o[4] = c.d1;
  o[5] = c.d2;
  o[6] = rev ? atan2f(p[6], p[7]) : c.yaw;
  const int first = rev ? 8 : 9;             // the columns after the rotation channels pass through
  for (int j = first; j < a.c; ++j) o[7 + (j - first)] = p[j];
  if (a.swapk != nullptr) a.swapk[i] = c.k;
}

// backward of decode wrt preds: gp (n,c) from go (n,co), the decoded output and the swap flags
struct DecBwdArgs {
  const float* go;    // (n,co)
  const float* out;   // (n,co) decoded rows of the forward (dims = exp(p) when norm_bbox)
  const int* swapk;   // (n) nullable
  float* gp;          // (n,c)
  long long n;
  int c, co, norm_bbox;
  float osf, vs0, vs1;
};

__global__ __launch_bounds__(T) void center_decode_bwd_kernel(const DecBwdArgs a) {
  const long long i = (long long)blockIdx.x * T + threadIdx.x;
  if (i >= a.n) return;
  const float* g = a.go + i * a.co;
  const float* o = a.out + i * a.co;
  float* gp = a.gp + i * a.c;
  gp[0] = g[0] * a.osf * a.vs0;
  gp[1] = g[1] * a.osf * a.vs1;
  gp[2] = g[2];
  const bool sw = a.swapk != nullptr && a.swapk[i] != 0;
  // out[3] = dim[sw ? 1 : 0], out[4] = dim[sw ? 0 : 1]
  const float g3 = sw ? g[4] : g[3], g4 = sw ? g[3] : g[4];
  const float e3 = sw ? o[4] : o[3], e4 = sw ? o[3] : o[4];
  gp[3] = a.norm_bbox ? g3 * e3 : g3;
  gp[4] = a.norm_bbox ? g4 * e4 : g4;
  gp[5] = a.norm_bbox ? g[5] * o[5] : g[5];
  gp[6] = g[6];
  if (a.c > 7) gp[7] = 0.0f;
  if (a.c > 8) gp[8] = 0.0f;
  for (int j = 9; j < a.c; ++j) gp[j] = g[7 + (j - 9)];
}

// encode: (n,c) boxes [x,y,z,w,l,h,yaw, others...] -> (n,c+2) [first 7, sin yaw, cos yaw, others]
__global__ __launch_bounds__(T) void center_encode_kernel(const float* __restrict__ boxes, float* __restrict__ out,
                                                         long long n, int c) {
  const long long i = (long long)blockIdx.x * T + threadIdx.x;
  if (i >= n) return;
  const float* b = boxes + i * c;
  float* o = out + i * (c + 2);
  for (int j = 0; j < 7; ++j) o[j] = b[j];
  float s, co;
  sincosf(b[6], &s, &co);
  o[7] = s;
  o[8] = co;
  for (int j = 7; j < c; ++j) o[j + 2] = b[j];
}


// PointBBoxYawCoder.decode (/root/reference/mmdet3d_gaussian/core/bbox/coders/point_bbox_yaw_coders.py:19-52): the prior of a
// box is a point (px, py) and a scale s;  x = p0 * s + px, y = p1 * s + py, z = p2, dims = exp(p3..p5) with the first two
// times s, yaw = p6 and the same quarter-turn correction from the (sin, cos) channels p7, p8 as the CenterPoint yaw coder.
struct PointArgs {
  const float* priors;  // (n,3)
  const float* preds;   // (n,c)
  float* out;           // (n,co)
  int* swapk;           // (n) nullable
  long long n;
  int c, co, correct_yaw;
};

__global__ __launch_bounds__(T) void point_decode_kernel(const PointArgs a) {
  const long long i = (long long)blockIdx.x * T + threadIdx.x;
  if (i >= a.n) return;
  const float* p = a.preds + i * a.c;
  const float* q = a.priors + i * 3;
  float* o = a.out + i * a.co;
  const float s = q[2];
  o[0] = p[0] * s + q[0];
  o[1] = p[1] * s + q[1];
  o[2] = p[2];
  float d0 = expf(p[3]) * s, d1 = expf(p[4]) * s;
  float yaw = p[6];
  int k = 0;
  if (a.correct_yaw) {
    const float dir = atan2f(p[7], p[8]);
    const float nr = floorf((dir - yaw) / HALF_PI + 0.5f);
    k = (int)((long long)nr & 1);           // parity of `num_rot90.long()` (:44)
    yaw = yaw + nr * HALF_PI;
    if (k) {
      const float t = d0;
      d0 = d1;
      d1 = t;
    }
  }
  o[3] = d0;
  o[4] = d1;
  o[5] = expf(p[5]);
  o[6] = yaw;
  for (int j = 9; j < a.c; ++j) o[7 + (j - 9)] = p[j];
  if (a.swapk != nullptr) a.swapk[i] = k;
}

struct PointBwdArgs {
  const float* go;      // (n,co)
  const float* out;     // (n,co)
  const float* priors;  // (n,3)
  const int* swapk;     // (n) nullable
  float* gp;            // (n,c)
  long long n;
  int c, co;
};

__global__ __launch_bounds__(T) void point_decode_bwd_kernel(const PointBwdArgs a) {
  const long long i = (long long)blockIdx.x * T + threadIdx.x;
  if (i >= a.n) return;
  const float* g = a.go + i * a.co;
  const float* o = a.out + i * a.co;
  float* gp = a.gp + i * a.c;
  const float s = a.priors[i * 3 + 2];
  gp[0] = g[0] * s;
  gp[1] = g[1] * s;
  gp[2] = g[2];
  const bool sw = a.swapk != nullptr && a.swapk[i] != 0;
  // out[3] = exp(p[sw ? 4 : 3]) s, out[4] = exp(p[sw ? 3 : 4]) s: each is its own derivative
  gp[3] = sw ? g[4] * o[4] : g[3] * o[3];
  gp[4] = sw ? g[3] * o[3] : g[4] * o[4];
  gp[5] = g[5] * o[5];
  gp[6] = g[6];
  if (a.c > 7) gp[7] = 0.0f;
  if (a.c > 8) gp[8] = 0.0f;
  for (int j = 9; j < a.c; ++j) gp[j] = g[7 + (j - 9)];
}

// PVRCNNBboxHead.get_bboxes, the decode in front of its NMS (/root/reference/mmdet3d_gaussian/models/roi_heads/bbox_heads/pvrcnn_bbox_head.py:
// 376-386): DeltaXYZWLHRBBoxCoder.decode (mmdet3d, third party) of the head's residuals against the roi moved to the origin, the
// centre turned about z by the roi's yaw (mmdet3d rotation_3d_in_axis, third party: counter-clockwise in 1.0, its transpose in 0.x —
// `clockwise` selects) and moved back; plus the [x1, y1, x2, y2, yaw] rectangle multi_class_nms feeds to nms_gpu (:447-448).
struct RoiArgs {
  const float* rois;   // (n, stride) rows [.., x, y, z, dx, dy, dz, yaw] starting at column `first`
  const float* pred;   // (n, 7)
  float* boxes;        // (n, 7)
  float* bev;          // (n, 5) nullable
  long long n;
  int stride, first, clockwise;
};

__global__ __launch_bounds__(T) void roi_decode_kernel(const RoiArgs a) {
  const long long i = (long long)blockIdx.x * T + threadIdx.x;
  if (i >= a.n) return;
  const float* r = a.rois + i * a.stride + a.first;
  const float* t = a.pred + i * 7;
  const float wa = r[3], la = r[4], ha = r[5], ra = r[6];
  // decode against the anchor (0, 0, 0, wa, la, ha, ra)
  const float za = 0.0f + ha / 2;
  const float diagonal = sqrtf(la * la + wa * wa);
  const float xl = t[0] * diagonal + 0.0f;
  const float yl = t[1] * diagonal + 0.0f;
  float zg = t[2] * ha + za;
  const float lg = expf(t[4]) * la;
  const float wg = expf(t[3]) * wa;
  const float hg = expf(t[5]) * ha;
  const float rg = t[6] + ra;
  zg = zg - hg / 2;
  float sn, cs;
  sincosf(ra, &sn, &cs);
  const float xr = a.clockwise ? xl * cs + yl * sn : xl * cs - yl * sn;
  const float yr = a.clockwise ? yl * cs - xl * sn : xl * sn + yl * cs;
  const float x = xr + r[0], y = yr + r[1], z = zg + r[2];
  float* o = a.boxes + i * 7;
  o[0] = x; o[1] = y; o[2] = z; o[3] = wg; o[4] = lg; o[5] = hg; o[6] = rg;
  if (a.bev != nullptr) {
    float* b = a.bev + i * 5;
    b[0] = x - wg / 2; b[1] = y - lg / 2; b[2] = x + wg / 2; b[3] = y + lg / 2; b[4] = rg;
  }
}

}  // namespace gdcoder

using namespace gdcoder;

extern "C" {

int coder_center_decode(const gd3d_prologue* coder, const float* locs, const float* preds, int64_t n, int32_t c,
                        int32_t correct_yaw, float* out, int32_t* num_rot_parity, void* stream) {
  if (coder == nullptr || n < 0 || c < 7 || correct_yaw < 0 || correct_yaw > 2) return GD3D_E_BADARG;
  if (correct_yaw == 1 && c < 9) return GD3D_E_BADARG;
  if (correct_yaw == 2 && c < 8) return GD3D_E_BADARG;
  if (n == 0) return 0;
  if (locs == nullptr || preds == nullptr || out == nullptr) return GD3D_E_BADARG;
  DecArgs a;
  a.locs = locs;
  a.preds = preds;
  a.out = out;
  a.swapk = (int*)num_rot_parity;
  a.n = n;
  a.c = c;
  a.co = correct_yaw == 2 ? c - 1 : 7 + (c > 9 ? c - 9 : 0);
  a.mode = correct_yaw;
  a.g.norm_bbox = coder->norm_bbox;
  a.g.osf = coder->out_size_factor;
  a.g.vs0 = coder->voxel_size[0];
  a.g.vs1 = coder->voxel_size[1];
  a.g.pc0 = coder->pc_range[0];
  a.g.pc1 = coder->pc_range[1];
  const long long nb = (n + T - 1) / T;
  if (nb > 0x7fffffffLL) return GD3D_E_TOOLARGE;
  hipLaunchKernelGGL(center_decode_kernel, dim3((unsigned)nb), dim3(T), 0, (hipStream_t)stream, a);
  return (int)hipGetLastError();
}

int coder_center_decode_backward(const gd3d_prologue* coder, const float* grad_out, const float* out,
                                 const int32_t* num_rot_parity, int64_t n, int32_t c, float* grad_preds, void* stream) {
  if (coder == nullptr || n < 0 || c < 7) return GD3D_E_BADARG;
  if (n == 0) return 0;
  if (grad_out == nullptr || out == nullptr || grad_preds == nullptr) return GD3D_E_BADARG;
  DecBwdArgs a;
  a.go = grad_out;
  a.out = out;
  a.swapk = (const int*)num_rot_parity;
  a.gp = grad_preds;
  a.n = n;
  a.c = c;
  a.co = 7 + (c > 9 ? c - 9 : 0);
  a.norm_bbox = coder->norm_bbox;
  a.osf = coder->out_size_factor;
  a.vs0 = coder->voxel_size[0];
  a.vs1 = coder->voxel_size[1];
  const long long nb = (n + T - 1) / T;
  if (nb > 0x7fffffffLL) return GD3D_E_TOOLARGE;
  hipLaunchKernelGGL(center_decode_bwd_kernel, dim3((unsigned)nb), dim3(T), 0, (hipStream_t)stream, a);
  return (int)hipGetLastError();
}

int coder_center_encode(const float* boxes, int64_t n, int32_t c, float* out, void* stream) {
  if (n < 0 || c < 7) return GD3D_E_BADARG;
  if (n == 0) return 0;
  if (boxes == nullptr || out == nullptr) return GD3D_E_BADARG;
  const long long nb = (n + T - 1) / T;
  if (nb > 0x7fffffffLL) return GD3D_E_TOOLARGE;
  hipLaunchKernelGGL(center_encode_kernel, dim3((unsigned)nb), dim3(T), 0, (hipStream_t)stream, boxes, out, (long long)n,
                     (int)c);
  return (int)hipGetLastError();
}

int coder_point_decode(const float* priors, const float* preds, int64_t n, int32_t c, int32_t correct_yaw, float* out,
                       int32_t* num_rot_parity, void* stream) {
  if (n < 0 || c < 7 || correct_yaw < 0 || correct_yaw > 1) return GD3D_E_BADARG;
  if (correct_yaw == 1 && c < 9) return GD3D_E_BADARG;
  if (n == 0) return 0;
  if (priors == nullptr || preds == nullptr || out == nullptr) return GD3D_E_BADARG;
  PointArgs a;
  a.priors = priors;
  a.preds = preds;
  a.out = out;
  a.swapk = (int*)num_rot_parity;
  a.n = n;
  a.c = c;
  a.co = 7 + (c > 9 ? c - 9 : 0);
  a.correct_yaw = correct_yaw;
  const long long nb = (n + T - 1) / T;
  if (nb > 0x7fffffffLL) return GD3D_E_TOOLARGE;
  hipLaunchKernelGGL(point_decode_kernel, dim3((unsigned)nb), dim3(T), 0, (hipStream_t)stream, a);
  return (int)hipGetLastError();
}

int coder_point_decode_backward(const float* priors, const float* grad_out, const float* out, const int32_t* num_rot_parity,
                                int64_t n, int32_t c, float* grad_preds, void* stream) {
  if (n < 0 || c < 7) return GD3D_E_BADARG;
  if (n == 0) return 0;
  if (priors == nullptr || grad_out == nullptr || out == nullptr || grad_preds == nullptr) return GD3D_E_BADARG;
  PointBwdArgs a;
  a.go = grad_out;
  a.out = out;
  a.priors = priors;
  a.swapk = (const int*)num_rot_parity;
  a.gp = grad_preds;
  a.n = n;
  a.c = c;
  a.co = 7 + (c > 9 ? c - 9 : 0);
  const long long nb = (n + T - 1) / T;
  if (nb > 0x7fffffffLL) return GD3D_E_TOOLARGE;
  hipLaunchKernelGGL(point_decode_bwd_kernel, dim3((unsigned)nb), dim3(T), 0, (hipStream_t)stream, a);
  return (int)hipGetLastError();
}

int coder_roi_decode(const float* rois, int32_t roi_stride, int32_t first_col, const float* bbox_pred, int64_t n, int32_t clockwise,
                     float* boxes, float* bev_xyxyr, void* stream) {
  if (n < 0 || roi_stride < 7 || first_col < 0 || first_col + 7 > roi_stride) return GD3D_E_BADARG;
  if (n == 0) return 0;
  if (rois == nullptr || bbox_pred == nullptr || boxes == nullptr) return GD3D_E_BADARG;
  RoiArgs a;
  a.rois = rois;
  a.pred = bbox_pred;
  a.boxes = boxes;
  a.bev = bev_xyxyr;
  a.n = n;
  a.stride = roi_stride;
  a.first = first_col;
  a.clockwise = clockwise != 0;
  const long long nb = (n + T - 1) / T;
  if (nb > 0x7fffffffLL) return GD3D_E_TOOLARGE;
  hipLaunchKernelGGL(roi_decode_kernel, dim3((unsigned)nb), dim3(T), 0, (hipStream_t)stream, a);
  return (int)hipGetLastError();
}

}  // extern "C"
