// rbox_device.h — rotated-rectangle geometry on the device (gfx950).
//
// Part 1: the mmdet3d 0.x `iou3d` BEV overlap on [x1,y1,x2,y2,ry] boxes (the op behind `nms_gpu`
//         called at /root/reference/mmdet3d_gaussian/models/dense_heads/gd_centerpoint_head.py:340-345):
//         corners rotated about the centre, 4x4 segment intersections, corner-in-box tests with a
//         1e-5 margin, vertices ordered by atan2 about their mean, shoelace area.
// Part 2: the reference's own CPU eval geometry (ops/eval/rbox_utils.hpp:52-302): 24-point
//         intersection set, Graham hull, fan area.
//
// Every arithmetic step of part 1 is a single IEEE fp32 operation in a fixed order (this file is
// compiled with -ffp-contract=off and correctly rounded division), including the sin/cos/atan2
// polynomial sequences, so the suppression mask is reproducible bit for bit by a CPU evaluation of
// the same sequence — which is how the keep indices are verified.
#pragma once
#ifdef GD3D_HOST_TWIN   // csrc/rbox_cpu.cpp: the `_cpu` twins run this same geometry on the host
#include "gd3d_host_math.h"
#else
#include <hip/hip_runtime.h>
#endif

namespace rbox {

#define RB_DEV __device__ __forceinline__
// phase stamps inside box_overlap for tools/clip_probe.hip (cycle accounting of one clipping pass); nothing in the product
#ifndef RB_STAMP
#define RB_STAMP(k) do { } while (0)
#endif

// ------------------------------------------------------------------ fixed-sequence fp32 math
RB_DEV void fx_sincosf(float x, float& s, float& c) {
  const float q = rintf(x * 0.63661977236758134f);
  float r = x - q * 1.5703125f;
  r = r - q * 4.837512969970703125e-4f;
  r = r - q * 7.54978995489188216e-8f;
  const int n = (int)q & 3;
  const float z = r * r;
  const float ps = ((-1.9515295891e-4f * z + 8.3321608736e-3f) * z - 1.6666654611e-1f) * z * r + r;
  const float pc =
      ((2.443315711809948e-5f * z - 1.388731625493765e-3f) * z + 4.166664568298827e-2f) * z * z - 0.5f * z + 1.0f;
  float sv = (n & 1) ? pc : ps;
  float cv = (n & 1) ? ps : pc;
  if (n & 2) sv = -sv;
  if ((n + 1) & 2) cv = -cv;
  s = sv;
  c = cv;
}

// atan2 as select chains instead of branches: every candidate value is computed with the operations of the branch it
// belongs to (so the selected one is bit-identical to the branchy evaluation; divisions by zero on untaken candidates
// produce inf / NaN that are discarded), and independent evaluations can overlap in the pipeline — a lone wave spent
// ~680 dependent-issue cycles per call on the branchy form (tools/clip_probe.hip).
RB_DEV float fx_atanf_pos(float t) {
  const bool hi = t > 2.414213562373095f, mid = t > 0.4142135623730950f;
  const float x_hi = -(1.0f / t);
  const float x_mid = (t - 1.0f) / (t + 1.0f);
  const float y0 = hi ? 1.5707963267948966f : (mid ? 0.7853981633974483f : 0.0f);
  const float x = hi ? x_hi : (mid ? x_mid : t);
  const float z = x * x;
  const float p =
      (((8.05374449538e-2f * z - 1.38776856032e-1f) * z + 1.99777106478e-1f) * z - 3.33329491539e-1f) * z * x + x;
  return y0 + p;
}

RB_DEV float fx_atan2f(float y, float x) {
  const float PI_F = 3.14159265358979323846f;
  const float t = y / x;
  const float ap = fx_atanf_pos(t < 0.0f ? -t : t);
  float a = t < 0.0f ? -ap : ap;
  const float shifted = (y < 0.0f) ? a - PI_F : a + PI_F;
  a = (x < 0.0f) ? shifted : a;
  const float axis = (y > 0.0f) ? 1.5707963267948966f : ((y < 0.0f) ? -1.5707963267948966f : 0.0f);
  return (x == 0.0f) ? axis : a;
}

// ------------------------------------------------------------------ part 1: iou3d overlap
struct Pt {
  float x, y;
};

// oriented box in the form the pair test consumes: 16 floats = 64 bytes
struct OBox {
  float x1, y1, x2, y2;
  float cx, cy, co, si;
  Pt c[4];
};

RB_DEV float fmin2(float a, float b) { return a < b ? a : b; }
RB_DEV float fmax2(float a, float b) { return a > b ? a : b; }

RB_DEV void obox_make(const float* b, OBox& o) {
  o.x1 = b[0];
  o.y1 = b[1];
  o.x2 = b[2];
  o.y2 = b[3];
  o.cx = (b[0] + b[2]) / 2.0f;
  o.cy = (b[1] + b[3]) / 2.0f;
  fx_sincosf(b[4], o.si, o.co);
  const float xs[4] = {b[0], b[2], b[2], b[0]}, ys[4] = {b[1], b[1], b[3], b[3]};
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const float dx = xs[k] - o.cx, dy = ys[k] - o.cy;
    o.c[k].x = dx * o.co + dy * o.si + o.cx;
    o.c[k].y = -dx * o.si + dy * o.co + o.cy;
  }
}

RB_DEV float cross3(Pt p1, Pt p2, Pt p0) { return (p1.x - p0.x) * (p2.y - p0.y) - (p2.x - p0.x) * (p1.y - p0.y); }

RB_DEV bool seg_intersection(Pt p1, Pt p0, Pt q1, Pt q0, Pt& ans) {
  const bool rc = fmin2(p0.x, p1.x) <= fmax2(q0.x, q1.x) && fmin2(q0.x, q1.x) <= fmax2(p0.x, p1.x) &&
                  fmin2(p0.y, p1.y) <= fmax2(q0.y, q1.y) && fmin2(q0.y, q1.y) <= fmax2(p0.y, p1.y);
  if (!rc) return false;
  const float s1 = cross3(q0, p1, p0);
  const float s2 = cross3(p1, q1, p0);
  const float s3 = cross3(p0, q1, q0);
  const float s4 = cross3(q1, p1, q0);
  if (!(s1 * s2 > 0.0f && s3 * s4 > 0.0f)) return false;
  const float s5 = cross3(q1, p1, p0);
  if (fabsf(s5 - s1) > 1e-8f) {
    ans.x = (s5 * q0.x - s1 * q1.x) / (s5 - s1);
    ans.y = (s5 * q0.y - s1 * q1.y) / (s5 - s1);
  } else {
    const float a0 = p0.y - p1.y, b0 = p1.x - p0.x, c0 = p0.x * p1.y - p1.x * p0.y;
    const float a1 = q0.y - q1.y, b1 = q1.x - q0.x, c1 = q0.x * q1.y - q1.x * q0.y;
    const float D = a0 * b1 - a1 * b0;
    ans.x = (b0 * c1 - b1 * c0) / D;
    ans.y = (a1 * c0 - a0 * c1) / D;
  }
  return true;
}

RB_DEV bool in_box(const OBox& o, Pt p) {
  const float MARGIN = 1e-5f;
  const float dx = p.x - o.cx, dy = p.y - o.cy;
  const float rx = dx * o.co + dy * (-o.si) + o.cx;
  const float ry = -dx * (-o.si) + dy * o.co + o.cy;
  return rx > o.x1 - MARGIN && rx < o.x2 + MARGIN && ry > o.y1 - MARGIN && ry < o.y2 + MARGIN;
}

// Per-thread vertex scratch lives in LDS, laid out [slot][thread] (consecutive lanes on consecutive
// banks): runtime-indexed per-thread arrays would otherwise spill to scratch memory.
template <int NT>
struct VertexScratch {
  float x[16][NT];
  float y[16][NT];
  float a[16][NT];
};

template <int NT>
RB_DEV float box_overlap(const OBox& A, const OBox& B, VertexScratch<NT>& vs, int t) {
  // exact early-out: disjoint bounding circles (with slack far above the 1e-5 in-box margin) give
  // cnt == 0, hence overlap exactly 0, in the full algorithm as well.
  {
    const float ddx = A.cx - B.cx, ddy = A.cy - B.cy;
    const float ra = fabsf(A.x2 - A.x1) + fabsf(A.y2 - A.y1);  // >= diagonal >= 2 * radius
    const float rb = fabsf(B.x2 - B.x1) + fabsf(B.y2 - B.y1);
    const float reach = 0.5f * (ra + rb) + 1e-2f;
    if (ddx * ddx + ddy * ddy > reach * reach * 1.0001f) return 0.0f;
  }
  RB_STAMP(1);
  float pcx = 0.0f, pcy = 0.0f;
  int cnt = 0;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      Pt ans;
      if (seg_intersection(A.c[(i + 1) & 3], A.c[i], B.c[(j + 1) & 3], B.c[j], ans) && cnt < 16) {
        pcx = pcx + ans.x;
        pcy = pcy + ans.y;
        vs.x[cnt][t] = ans.x;
        vs.y[cnt][t] = ans.y;
        ++cnt;
      }
    }
  }
  RB_STAMP(2);
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    if (in_box(A, B.c[k]) && cnt < 16) {
      pcx = pcx + B.c[k].x;
      pcy = pcy + B.c[k].y;
      vs.x[cnt][t] = B.c[k].x;
      vs.y[cnt][t] = B.c[k].y;
      ++cnt;
    }
    if (in_box(B, A.c[k]) && cnt < 16) {
      pcx = pcx + A.c[k].x;
      pcy = pcy + A.c[k].y;
      vs.x[cnt][t] = A.c[k].x;
      vs.y[cnt][t] = A.c[k].y;
      ++cnt;
    }
  }
  RB_STAMP(3);
  if (cnt == 0) return 0.0f;
  pcx = pcx / (float)cnt;
  pcy = pcy / (float)cnt;
  // Fast path for <= 8 vertices (two convex quadrilaterals meet in at most 8 points; more only arise when a point is
  // collected both as an intersection and as a corner): the reference sequence — angle of every vertex about the
  // centroid, STABLE bubble sort by angle, fan area in sorted order — evaluated on registers.  The 8 angle evaluations
  // are independent (a lone wave otherwise spends ~680 dependent-issue cycles per vertex, tools/clip_probe.hip), the sort
  // is a 19-exchange network on the key (angle, collection index) — a strict total order, so the network yields exactly
  // the permutation the stable bubble sort does, ties and +-0 included — and the area sum runs in that order with the
  // reference's operations.  A NaN angle (NaN / inf input boxes) has no place in a total order: those pairs, and the
  // ones with more than 8 vertices, take the literal sequence on the LDS arrays below.
  if (cnt <= 8) {
    float X[8], Y[8], K[8];
    int I[8];
    bool ordered = true;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      X[k] = vs.x[k][t];  // slots >= cnt hold stale values: keyed +inf below, never used by the area
      Y[k] = vs.y[k][t];
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const float ang = fx_atan2f(Y[k] - pcy, X[k] - pcx);
      const bool valid = k < cnt;
      ordered = ordered && !(valid && ang != ang);
      K[k] = valid ? ang : __builtin_inff();
      I[k] = k;
    }
    if (ordered) {
#define RB_CE(i, j)                                                          \
  {                                                                          \
    const bool sw = (K[i] > K[j]) || (K[i] == K[j] && I[i] > I[j]);          \
    const float ka = K[i], xa = X[i], ya = Y[i];                             \
    const int ia = I[i];                                                     \
    K[i] = sw ? K[j] : ka;  X[i] = sw ? X[j] : xa;  Y[i] = sw ? Y[j] : ya;  I[i] = sw ? I[j] : ia; \
    K[j] = sw ? ka : K[j];  X[j] = sw ? xa : X[j];  Y[j] = sw ? ya : Y[j];  I[j] = sw ? ia : I[j]; \
  }
      RB_CE(0, 2) RB_CE(1, 3) RB_CE(4, 6) RB_CE(5, 7)
      RB_CE(0, 4) RB_CE(1, 5) RB_CE(2, 6) RB_CE(3, 7)
      RB_CE(0, 1) RB_CE(2, 3) RB_CE(4, 5) RB_CE(6, 7)
      RB_CE(2, 4) RB_CE(3, 5)
      RB_CE(1, 4) RB_CE(3, 6)
      RB_CE(1, 2) RB_CE(3, 4) RB_CE(5, 6)
#undef RB_CE
      RB_STAMP(4);
      RB_STAMP(5);
      float area = 0.0f;
      const float x0 = X[0], y0 = Y[0];
#pragma unroll
      for (int k = 0; k < 7; ++k) {
        const float ax = X[k] - x0, ay = Y[k] - y0;
        const float bx = X[k + 1] - x0, by = Y[k + 1] - y0;
        const float next = area + (ax * by - ay * bx);
        area = (k < cnt - 1) ? next : area;
      }
      RB_STAMP(6);
      return fabsf(area) / 2.0f;
    }
  }
  for (int k = 0; k < cnt; ++k) vs.a[k][t] = fx_atan2f(vs.y[k][t] - pcy, vs.x[k][t] - pcx);
  RB_STAMP(4);
  for (int j = 0; j < cnt - 1; ++j)
    for (int i = 0; i < cnt - j - 1; ++i) {
      const float a0 = vs.a[i][t], a1 = vs.a[i + 1][t];
      if (a0 > a1) {
        const float x0 = vs.x[i][t], y0 = vs.y[i][t];
        vs.x[i][t] = vs.x[i + 1][t];
        vs.y[i][t] = vs.y[i + 1][t];
        vs.a[i][t] = a1;
        vs.x[i + 1][t] = x0;
        vs.y[i + 1][t] = y0;
        vs.a[i + 1][t] = a0;
      }
    }
  RB_STAMP(5);
  float area = 0.0f;
  const float x0 = vs.x[0][t], y0 = vs.y[0][t];
  for (int k = 0; k < cnt - 1; ++k) {
    const float ax = vs.x[k][t] - x0, ay = vs.y[k][t] - y0;
    const float bx = vs.x[k + 1][t] - x0, by = vs.y[k + 1][t] - y0;
    area = area + (ax * by - ay * bx);
  }
  RB_STAMP(6);
  return fabsf(area) / 2.0f;
}

template <int NT>
RB_DEV float iou_bev(const OBox& A, const OBox& B, VertexScratch<NT>& vs, int t) {
  const float sa = (A.x2 - A.x1) * (A.y2 - A.y1);
  const float sb = (B.x2 - B.x1) * (B.y2 - B.y1);
  const float so = box_overlap<NT>(A, B, vs, t);
  return so / fmax2(sa + sb - so, 1e-8f);
}

RB_DEV float iou_normal(const float* a, const float* b) {
  const float left = fmax2(a[0], b[0]), right = fmin2(a[2], b[2]);
  const float top = fmax2(a[1], b[1]), bottom = fmin2(a[3], b[3]);
  const float width = fmax2(right - left, 0.0f), height = fmax2(bottom - top, 0.0f);
  const float inter = width * height;
  const float sa = (a[2] - a[0]) * (a[3] - a[1]);
  const float sb = (b[2] - b[0]) * (b[3] - b[1]);
  return inter / fmax2(sa + sb - inter, 1e-8f);
}

// ------------------------------------------------------------------ part 2: ops/eval geometry
struct RBox {
  float xc, yc, w, h, a;
};

RB_DEV float cross2(Pt A, Pt B) { return A.x * B.y - B.x * A.y; }
RB_DEV float dot2(Pt A, Pt B) { return A.x * B.x + A.y * B.y; }
RB_DEV Pt sub2(Pt A, Pt B) { return Pt{A.x - B.x, A.y - B.y}; }

// rbox_utils.hpp:52-71 — cos/sin in double, cast to float (as the reference does)
RB_DEV void rot_vertices(const RBox& b, Pt (&p)[4]) {
  const double theta = (double)b.a;
  const float c2 = (float)cos(theta) * 0.5f, s2 = (float)sin(theta) * 0.5f;
  p[0].x = b.xc - s2 * b.h - c2 * b.w;
  p[0].y = b.yc + c2 * b.h - s2 * b.w;
  p[1].x = b.xc + s2 * b.h - c2 * b.w;
  p[1].y = b.yc - c2 * b.h - s2 * b.w;
  p[2].x = 2 * b.xc - p[0].x;
  p[2].y = 2 * b.yc - p[0].y;
  p[3].x = 2 * b.xc - p[1].x;
  p[3].y = 2 * b.yc - p[1].y;
}

template <int NT>
struct HullScratch {
  float px[24][NT], py[24][NT];  // intersection points
  float qx[24][NT], qy[24][NT];  // hull work array
  float d[24][NT];               // squared distances (pre-sort order, as in the CPU branch)
};

RB_DEV bool hull_less(Pt A, Pt B) {
  const float t = cross2(A, B);
  if (fabs((double)t) < 1e-6) return dot2(A, A) < dot2(B, B);
  return t > 0;
}

// rbox_utils.hpp:73-302, CPU (std::sort) branch semantics
template <int NT>
RB_DEV float rot_intersection(const RBox& b1, const RBox& b2, HullScratch<NT>& hs, int t) {
  Pt p1[4], p2[4], v1[4], v2[4];
  rot_vertices(b1, p1);
  rot_vertices(b2, p2);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    v1[i] = sub2(p1[(i + 1) & 3], p1[i]);
    v2[i] = sub2(p2[(i + 1) & 3], p2[i]);
  }
  int num = 0;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float det = cross2(v2[j], v1[i]);
      if (fabs((double)det) <= 1e-14) continue;
      const Pt v12 = sub2(p2[j], p1[i]);
      const float t1 = cross2(v2[j], v12) / det;
      const float t2 = cross2(v1[i], v12) / det;
      if (t1 >= 0.0f && t1 <= 1.0f && t2 >= 0.0f && t2 <= 1.0f) {
        hs.px[num][t] = p1[i].x + v1[i].x * t1;
        hs.py[num][t] = p1[i].y + v1[i].y * t1;
        ++num;
      }
    }
  }
  {
    const Pt AB = v2[0], DA = v2[3];
    const float ABAB = dot2(AB, AB), ADAD = dot2(DA, DA);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const Pt AP = sub2(p1[i], p2[0]);
      const float ab = dot2(AP, AB), ad = -dot2(AP, DA);
      if (ab >= 0 && ad >= 0 && ab <= ABAB && ad <= ADAD) {
        hs.px[num][t] = p1[i].x;
        hs.py[num][t] = p1[i].y;
        ++num;
      }
    }
  }
  {
    const Pt AB = v1[0], DA = v1[3];
    const float ABAB = dot2(AB, AB), ADAD = dot2(DA, DA);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const Pt AP = sub2(p2[i], p1[0]);
      const float ab = dot2(AP, AB), ad = -dot2(AP, DA);
      if (ab >= 0 && ad >= 0 && ab <= ABAB && ad <= ADAD) {
        hs.px[num][t] = p2[i].x;
        hs.py[num][t] = p2[i].y;
        ++num;
      }
    }
  }
  if (num <= 2) return 0.0f;

  // Graham scan (shift_to_zero = true)
  int s = 0;
  for (int i = 1; i < num; ++i) {
    const float yi = hs.py[i][t], ys = hs.py[s][t];
    if (yi < ys || (yi == ys && hs.px[i][t] < hs.px[s][t])) s = i;
  }
  const float sx = hs.px[s][t], sy = hs.py[s][t];
  for (int i = 0; i < num; ++i) {
    hs.qx[i][t] = hs.px[i][t] - sx;
    hs.qy[i][t] = hs.py[i][t] - sy;
  }
  {
    const float tx = hs.qx[0][t], ty = hs.qy[0][t];
    hs.qx[0][t] = hs.qx[s][t];
    hs.qy[0][t] = hs.qy[s][t];
    hs.qx[s][t] = tx;
    hs.qy[s][t] = ty;
  }
  for (int i = 0; i < num; ++i) {
    const Pt q = {hs.qx[i][t], hs.qy[i][t]};
    hs.d[i][t] = dot2(q, q);
  }
  for (int i = 2; i < num; ++i) {  // insertion sort of q[1..num) with the std::sort comparator
    const Pt key = {hs.qx[i][t], hs.qy[i][t]};
    int j = i - 1;
    while (j >= 1) {
      const Pt qj = {hs.qx[j][t], hs.qy[j][t]};
      if (!hull_less(key, qj)) break;
      hs.qx[j + 1][t] = qj.x;
      hs.qy[j + 1][t] = qj.y;
      --j;
    }
    hs.qx[j + 1][t] = key.x;
    hs.qy[j + 1][t] = key.y;
  }
  int k;
  for (k = 1; k < num; ++k)
    if ((double)hs.d[k][t] > 1e-8) break;  // pre-sort distances: the CPU branch never permutes dist[]
  if (k == num) return 0.0f;
  hs.qx[1][t] = hs.qx[k][t];
  hs.qy[1][t] = hs.qy[k][t];
  int m = 2;
  for (int i = k + 1; i < num; ++i) {
    const Pt qi = {hs.qx[i][t], hs.qy[i][t]};
    while (m > 1) {
      const Pt a = {hs.qx[m - 2][t], hs.qy[m - 2][t]};
      const Pt b = {hs.qx[m - 1][t], hs.qy[m - 1][t]};
      if (cross2(sub2(qi, a), sub2(b, a)) >= 0) --m;
      else break;
    }
    hs.qx[m][t] = qi.x;
    hs.qy[m][t] = qi.y;
    ++m;
  }
  if (m <= 2) return 0.0f;
  float area = 0.0f;
  const Pt q0 = {hs.qx[0][t], hs.qy[0][t]};
  for (int i = 1; i < m - 1; ++i) {
    const Pt a = {hs.qx[i][t], hs.qy[i][t]};
    const Pt b = {hs.qx[i + 1][t], hs.qy[i + 1][t]};
    area += fabsf(cross2(sub2(a, q0), sub2(b, q0)));
  }
  return area / 2.0f;
}

// IoU of one detection / ground-truth pair from the intersection above: affinity.cpp:8-81 (iou_bev :8-38, iou_3d :40-81).
// Rows are [x, y, z, w, l, h, yaw]; z_offset places the box centre between its bottom (0) and top (1) face.
template <bool IS3D, int NT>
RB_DEV float eval_iou(const float (&d)[7], const float (&g)[7], float z_offset, HullScratch<NT>& hs, int t) {
  const RBox D = {d[0], d[1], d[3], d[4], d[6]}, G = {g[0], g[1], g[3], g[4], g[6]};
  const float bev = rot_intersection<NT>(D, G, hs, t);
  const float EPSF = 1.1920928955078125e-7f;
  if (IS3D) {
    const float dzb = d[2] + (z_offset - 0.5f) * d[5], gzb = g[2] + (z_offset - 0.5f) * g[5];
    const float dzt = d[2] + (z_offset + 0.5f) * d[5], gzt = g[2] + (z_offset + 0.5f) * g[5];
    const float zb = dzb > gzb ? dzb : gzb, zt = dzt < gzt ? dzt : gzt;
    float zi = zt - zb;
    zi = zi < 0.f ? 0.f : zi;
    const float dv = d[3] * d[4] * d[5], gv = g[3] * g[4] * g[5];
    float iv = bev * zi;
    iv = iv < 0.f ? 0.f : iv;
    iv = iv > dv ? dv : iv;
    iv = iv > gv ? gv : iv;
    float uv = dv + gv - iv;
    uv = uv < EPSF ? EPSF : uv;
    return iv / uv;
  }
  const float da = d[3] * d[4], ga = g[3] * g[4];
  float inter = bev < 0.f ? 0.f : bev;
  inter = inter > da ? da : inter;
  inter = inter > ga ? ga : inter;
  float un = da + ga - inter;
  un = un < EPSF ? EPSF : un;
  return inter / un;
}

}  // namespace rbox
