/*
 * rbox_oracle.c — TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU oracles for the rotated-box half of the hot path (SURVEY.md §8 rows aN, aI).
 *
 * Part 1 — rotated BEV NMS / IoU in the mmdet3d `iou3d` box format [x1,y1,x2,y2,ry].
 *   PARITY UNPINNED.  The reference imports this op from third-party mmdet3d
 *   (`from mmdet3d.ops.iou3d.iou3d_utils import nms_gpu`,
 *   /root/reference/mmdet3d_gaussian/models/dense_heads/gd_centerpoint_head.py:9,340-345;
 *   models/roi_heads/bbox_heads/pvrcnn_bbox_head.py:12,463-464).  mmdet3d is not vendored,
 *   not pinned (no requirements file; API usage suggests 0.17-0.18) and not installed, and the
 *   reference holds no test or golden vector for it.  What follows restates the published
 *   algorithm of mmdet3d 0.x `mmdet3d/ops/iou3d/src/iou3d_kernel.cu` (itself from
 *   OpenPCDet): corners rotated about the box centre, 4x4 segment intersections, corner-in-box
 *   tests with a 1e-5 margin, vertices sorted by atan2 about their mean, shoelace area;
 *   IoU = overlap / max(sa + sb - overlap, 1e-8); NMS = 64-wide bit-mask + greedy scan, where
 *   box i suppresses a later box j iff iou_bev(box_i, box_j) > thresh (argument order i, j).
 *   sin/cos/atan2 are evaluated with the fixed polynomial sequences below (plain IEEE fp32
 *   +,-,*,/ only, no FMA contraction) so that the HIP kernels, which use the same sequences,
 *   produce bit-identical masks and therefore bit-identical keep indices.
 *
 * Part 2 — pairwise rotated IoU of 7-dof boxes, restating the reference's own CPU helpers
 *   /root/reference/mmdet3d_gaussian/ops/eval/rbox_utils.hpp:52-302 and
 *   /root/reference/mmdet3d_gaussian/ops/eval/affinity.cpp:8-81.
 *   PINNED: tests/test_oracle_rbox.py compares it with oracle/_ref (the reference sources
 *   compiled where they lie, oracle/Makefile target `ref`) and with tests/golden/riou_eval.npz
 *   generated from that build.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ fixed fp32 math */
/* sin & cos of x: Cody-Waite reduction by pi/2 (3 constants), Cephes minimax polynomials on
 * [-pi/4, pi/4].  Every operation is a single IEEE fp32 op in this order. */
static void fx_sincosf(float x, float *s, float *c) {
  float q = nearbyintf(x * 0.63661977236758134f); /* round-to-nearest-even of x * 2/pi */
  float r = x - q * 1.5703125f;
  r = r - q * 4.837512969970703125e-4f;
  r = r - q * 7.54978995489188216e-8f;
  int n = (int)q & 3;
  float z = r * r;
  float ps = ((-1.9515295891e-4f * z + 8.3321608736e-3f) * z - 1.6666654611e-1f) * z * r + r;
  float pc = ((2.443315711809948e-5f * z - 1.388731625493765e-3f) * z + 4.166664568298827e-2f) * z * z -
             0.5f * z + 1.0f;
  float sv = (n & 1) ? pc : ps;
  float cv = (n & 1) ? ps : pc;
  if (n & 2) sv = -sv;
  if ((n + 1) & 2) cv = -cv;
  *s = sv;
  *c = cv;
}

/* atan2(y, x), Cephes atanf reduction + polynomial, plain fp32 ops */
static float fx_atanf_pos(float t) { /* t >= 0 */
  float y0, x;
  if (t > 2.414213562373095f) { y0 = 1.5707963267948966f; x = -(1.0f / t); }
  else if (t > 0.4142135623730950f) { y0 = 0.7853981633974483f; x = (t - 1.0f) / (t + 1.0f); }
  else { y0 = 0.0f; x = t; }
  float z = x * x;
  float p = (((8.05374449538e-2f * z - 1.38776856032e-1f) * z + 1.99777106478e-1f) * z - 3.33329491539e-1f) * z * x + x;
  return y0 + p;
}
static float fx_atan2f(float y, float x) {
  const float PI_F = 3.14159265358979323846f;
  if (x == 0.0f) {
    if (y > 0.0f) return 1.5707963267948966f;
    if (y < 0.0f) return -1.5707963267948966f;
    return 0.0f;
  }
  float t = y / x;
  float a = t < 0.0f ? -fx_atanf_pos(-t) : fx_atanf_pos(t);
  if (x < 0.0f) a = (y < 0.0f) ? a - PI_F : a + PI_F;
  return a;
}

/* ------------------------------------------------------------------ Part 1: iou3d */
typedef struct { float x, y; } pt;

static float cross3(pt p1, pt p2, pt p0) {
  return (p1.x - p0.x) * (p2.y - p0.y) - (p2.x - p0.x) * (p1.y - p0.y);
}
static float fminf2(float a, float b) { return a < b ? a : b; }
static float fmaxf2(float a, float b) { return a > b ? a : b; }

static int rect_cross(pt p1, pt p2, pt q1, pt q2) {
  return fminf2(p1.x, p2.x) <= fmaxf2(q1.x, q2.x) && fminf2(q1.x, q2.x) <= fmaxf2(p1.x, p2.x) &&
         fminf2(p1.y, p2.y) <= fmaxf2(q1.y, q2.y) && fminf2(q1.y, q2.y) <= fmaxf2(p1.y, p2.y);
}

/* segment p0-p1 against q0-q1 */
static int seg_intersection(pt p1, pt p0, pt q1, pt q0, pt *ans) {
  if (!rect_cross(p0, p1, q0, q1)) return 0;
  float s1 = cross3(q0, p1, p0);
  float s2 = cross3(p1, q1, p0);
  float s3 = cross3(p0, q1, q0);
  float s4 = cross3(q1, p1, q0);
  if (!(s1 * s2 > 0.0f && s3 * s4 > 0.0f)) return 0;
  float s5 = cross3(q1, p1, p0);
  if (fabsf(s5 - s1) > 1e-8f) {
    ans->x = (s5 * q0.x - s1 * q1.x) / (s5 - s1);
    ans->y = (s5 * q0.y - s1 * q1.y) / (s5 - s1);
  } else {
    float a0 = p0.y - p1.y, b0 = p1.x - p0.x, c0 = p0.x * p1.y - p1.x * p0.y;
    float a1 = q0.y - q1.y, b1 = q1.x - q0.x, c1 = q0.x * q1.y - q1.x * q0.y;
    float D = a0 * b1 - a1 * b0;
    ans->x = (b0 * c1 - b1 * c0) / D;
    ans->y = (a1 * c0 - a0 * c1) / D;
  }
  return 1;
}

typedef struct { float x1, y1, x2, y2, cx, cy, co, si; pt c[5]; } obox;

/* Sensitivity knob (tests/test_nms_margin.py): the upstream CUDA op takes sin/cos from libdevice, this restatement and the HIP
 * kernels from the fixed polynomials above; the two can differ by an ulp.  mode 0: off; 1 / 2: every box's sin AND cos moved
 * one ulp up / down; 3: per box and per function a pseudo-random move in {-1, 0, +1} ulp (hash of seed and box index). */
static int g_nudge_mode = 0;
static uint32_t g_nudge_seed = 0;
void rbox_oracle_set_trig_nudge(int mode, uint32_t seed) { g_nudge_mode = mode; g_nudge_seed = seed; }
static uint32_t mix32(uint32_t x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }
static float nudge_ulps(float v, int k) {
  if (k > 0) return nextafterf(v, INFINITY);
  if (k < 0) return nextafterf(v, -INFINITY);
  return v;
}
static void nudge_trig(int64_t idx, float *s, float *c) {
  if (g_nudge_mode == 0) return;
  int ks, kc;
  if (g_nudge_mode == 1) ks = kc = 1;
  else if (g_nudge_mode == 2) ks = kc = -1;
  else {
    uint32_t h = mix32(g_nudge_seed ^ mix32((uint32_t)idx * 2654435761U + 12345U));
    ks = (int)(h % 3U) - 1;
    kc = (int)((h >> 8) % 3U) - 1;
  }
  *s = nudge_ulps(*s, ks);
  *c = nudge_ulps(*c, kc);
}

static void obox_make_i(const float *b, obox *o, int64_t idx);
static void obox_make(const float *b, obox *o) { obox_make_i(b, o, 0); }

static void obox_make_i(const float *b, obox *o, int64_t idx) {
  o->x1 = b[0]; o->y1 = b[1]; o->x2 = b[2]; o->y2 = b[3];
  o->cx = (b[0] + b[2]) / 2.0f;
  o->cy = (b[1] + b[3]) / 2.0f;
  fx_sincosf(b[4], &o->si, &o->co);
  nudge_trig(idx, &o->si, &o->co);
  const float xs[4] = {b[0], b[2], b[2], b[0]}, ys[4] = {b[1], b[1], b[3], b[3]};
  for (int k = 0; k < 4; ++k) { /* rotate_around_center */
    float dx = xs[k] - o->cx, dy = ys[k] - o->cy;
    o->c[k].x = dx * o->co + dy * o->si + o->cx;
    o->c[k].y = -dx * o->si + dy * o->co + o->cy;
  }
  o->c[4] = o->c[0];
}

/* is p inside box o (rotate p by -angle about the centre; 1e-5 margin) */
static int in_box(const obox *o, pt p) {
  const float MARGIN = 1e-5f;
  /* cos(-a) = co, sin(-a) = -si */
  float dx = p.x - o->cx, dy = p.y - o->cy;
  float rx = dx * o->co + dy * (-o->si) + o->cx;
  float ry = -dx * (-o->si) + dy * o->co + o->cy;
  return rx > o->x1 - MARGIN && rx < o->x2 + MARGIN && ry > o->y1 - MARGIN && ry < o->y2 + MARGIN;
}

static float box_overlap(const obox *a, const obox *b) {
  pt cp[16]; /* the published buffer size; writes are guarded (cnt < 16) here and in the HIP kernel */
  float pcx = 0.0f, pcy = 0.0f;
  int cnt = 0;
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) {
      pt ans;
      if (seg_intersection(a->c[i + 1], a->c[i], b->c[j + 1], b->c[j], &ans) && cnt < 16) {
        pcx = pcx + ans.x; pcy = pcy + ans.y; cp[cnt++] = ans;
      }
    }
  for (int k = 0; k < 4; ++k) {
    if (in_box(a, b->c[k]) && cnt < 16) { pcx = pcx + b->c[k].x; pcy = pcy + b->c[k].y; cp[cnt++] = b->c[k]; }
    if (in_box(b, a->c[k]) && cnt < 16) { pcx = pcx + a->c[k].x; pcy = pcy + a->c[k].y; cp[cnt++] = a->c[k]; }
  }
  if (cnt == 0) return 0.0f; /* published code divides by cnt = 0 and then sums nothing: 0 */
  pcx = pcx / (float)cnt; pcy = pcy / (float)cnt;
  float ang[16];
  for (int k = 0; k < cnt; ++k) ang[k] = fx_atan2f(cp[k].y - pcy, cp[k].x - pcx);
  for (int j = 0; j < cnt - 1; ++j) /* bubble sort, ascending angle */
    for (int i = 0; i < cnt - j - 1; ++i)
      if (ang[i] > ang[i + 1]) {
        pt tp = cp[i]; cp[i] = cp[i + 1]; cp[i + 1] = tp;
        float ta = ang[i]; ang[i] = ang[i + 1]; ang[i + 1] = ta;
      }
  float area = 0.0f;
  for (int k = 0; k < cnt - 1; ++k) {
    float ax = cp[k].x - cp[0].x, ay = cp[k].y - cp[0].y;
    float bx = cp[k + 1].x - cp[0].x, by = cp[k + 1].y - cp[0].y;
    area = area + (ax * by - ay * bx);
  }
  return fabsf(area) / 2.0f;
}

static float iou_bev_ob(const obox *a, const obox *b) {
  float sa = (a->x2 - a->x1) * (a->y2 - a->y1);
  float sb = (b->x2 - b->x1) * (b->y2 - b->y1);
  float so = box_overlap(a, b);
  return so / fmaxf2(sa + sb - so, 1e-8f);
}

static float iou_normal(const float *a, const float *b) {
  float left = fmaxf2(a[0], b[0]), right = fminf2(a[2], b[2]);
  float top = fmaxf2(a[1], b[1]), bottom = fminf2(a[3], b[3]);
  float width = fmaxf2(right - left, 0.0f), height = fmaxf2(bottom - top, 0.0f);
  float inter = width * height;
  float sa = (a[2] - a[0]) * (a[3] - a[1]);
  float sb = (b[2] - b[0]) * (b[3] - b[1]);
  return inter / fmaxf2(sa + sb - inter, 1e-8f);
}

/* keep[] receives indices (ascending) into the score-sorted box list; returns the count */
int64_t rbox_oracle_nms_bev(const float *boxes_sorted, int64_t n, float thresh, int64_t *keep) {
  if (n <= 0) return 0;
  obox *ob = (obox *)malloc((size_t)n * sizeof(obox));
  unsigned char *dead = (unsigned char *)calloc((size_t)n, 1);
  for (int64_t i = 0; i < n; ++i) obox_make_i(boxes_sorted + 5 * i, &ob[i], i);
  int64_t nk = 0;
  for (int64_t i = 0; i < n; ++i) {
    if (dead[i]) continue;
    keep[nk++] = i;
    for (int64_t j = i + 1; j < n; ++j)
      if (!dead[j] && iou_bev_ob(&ob[i], &ob[j]) > thresh) dead[j] = 1;
  }
  free(ob); free(dead);
  return nk;
}

/* Decision margins of the greedy scan: for every pair the scan EVALUATES (kept box i, a later box j still alive when i is
 * reached) the distance |iou(i,j) - thresh|.  counts[k] = number of such pairs with |margin| < edges[k] (cumulative), and
 * stats = {pairs evaluated, pairs with iou > 0, smallest |margin|, its iou}.  Same arithmetic as rbox_oracle_nms_bev. */
int64_t rbox_oracle_nms_margin(const float *boxes_sorted, int64_t n, float thresh, const double *edges, int n_edges,
                               int64_t *counts, double *stats) {
  for (int k = 0; k < n_edges; ++k) counts[k] = 0;
  stats[0] = stats[1] = 0.0; stats[2] = 1e30; stats[3] = 0.0;
  if (n <= 0) return 0;
  obox *ob = (obox *)malloc((size_t)n * sizeof(obox));
  unsigned char *dead = (unsigned char *)calloc((size_t)n, 1);
  for (int64_t i = 0; i < n; ++i) obox_make_i(boxes_sorted + 5 * i, &ob[i], i);
  int64_t nk = 0;
  for (int64_t i = 0; i < n; ++i) {
    if (dead[i]) continue;
    ++nk;
    for (int64_t j = i + 1; j < n; ++j) {
      if (dead[j]) continue;
      float iou = iou_bev_ob(&ob[i], &ob[j]);
      double m = fabs((double)iou - (double)thresh);
      stats[0] += 1.0;
      if (iou > 0.0f) stats[1] += 1.0;
      if (m < stats[2]) { stats[2] = m; stats[3] = (double)iou; }
      for (int k = 0; k < n_edges; ++k) if (m < edges[k]) counts[k] += 1;
      if (iou > thresh) dead[j] = 1;
    }
  }
  free(ob); free(dead);
  return nk;
}

int64_t rbox_oracle_nms_normal(const float *boxes_sorted, int64_t n, float thresh, int64_t *keep) {
  if (n <= 0) return 0;
  unsigned char *dead = (unsigned char *)calloc((size_t)n, 1);
  int64_t nk = 0;
  for (int64_t i = 0; i < n; ++i) {
    if (dead[i]) continue;
    keep[nk++] = i;
    for (int64_t j = i + 1; j < n; ++j)
      if (!dead[j] && iou_normal(boxes_sorted + 5 * i, boxes_sorted + 5 * j) > thresh) dead[j] = 1;
  }
  free(dead);
  return nk;
}

/* full suppression bit-mask, layout of the published kernel: mask[i*cb + c] bit k set iff
 * j = 64c+k > i (same 64-block) or any j in other blocks, and iou(i,j) > thresh. */
void rbox_oracle_nms_mask(const float *boxes_sorted, int64_t n, float thresh, uint64_t *mask) {
  int64_t cb = (n + 63) / 64;
  obox *ob = (obox *)malloc((size_t)(n > 0 ? n : 1) * sizeof(obox));
  for (int64_t i = 0; i < n; ++i) obox_make(boxes_sorted + 5 * i, &ob[i]);
  memset(mask, 0, (size_t)(n * cb) * sizeof(uint64_t));
  for (int64_t i = 0; i < n; ++i)
    for (int64_t j = 0; j < n; ++j) {
      if (j / 64 == i / 64 && j <= i) continue;
      if (iou_bev_ob(&ob[i], &ob[j]) > thresh) mask[i * cb + j / 64] |= 1ULL << (j % 64);
    }
  free(ob);
}

void rbox_oracle_iou_bev_xyxyr(const float *a, int64_t na, const float *b, int64_t nb, float *out) {
  obox *oa = (obox *)malloc((size_t)(na > 0 ? na : 1) * sizeof(obox));
  obox *obb = (obox *)malloc((size_t)(nb > 0 ? nb : 1) * sizeof(obox));
  for (int64_t i = 0; i < na; ++i) obox_make(a + 5 * i, &oa[i]);
  for (int64_t j = 0; j < nb; ++j) obox_make(b + 5 * j, &obb[j]);
  for (int64_t i = 0; i < na; ++i)
    for (int64_t j = 0; j < nb; ++j) out[i * nb + j] = iou_bev_ob(&oa[i], &obb[j]);
  free(oa); free(obb);
}

/* ------------------------------------------------------------------ Part 2: ops/eval */
typedef struct { float xc, yc, w, h, a; } rbox;

static float cross2(pt A, pt B) { return A.x * B.y - B.x * A.y; }
static float dot2(pt A, pt B) { return A.x * B.x + A.y * B.y; }
static pt sub2(pt A, pt B) { pt r = {A.x - B.x, A.y - B.y}; return r; }

/* rbox_utils.hpp:52-71 (angle in radians; cos/sin in double, then cast) */
static void rot_vertices(const rbox *b, pt *p) {
  double theta = (double)b->a;
  float c2 = (float)cos(theta) * 0.5f, s2 = (float)sin(theta) * 0.5f;
  p[0].x = b->xc - s2 * b->h - c2 * b->w;
  p[0].y = b->yc + c2 * b->h - s2 * b->w;
  p[1].x = b->xc + s2 * b->h - c2 * b->w;
  p[1].y = b->yc - c2 * b->h - s2 * b->w;
  p[2].x = 2 * b->xc - p[0].x;
  p[2].y = 2 * b->yc - p[0].y;
  p[3].x = 2 * b->xc - p[1].x;
  p[3].y = 2 * b->yc - p[1].y;
}

/* rbox_utils.hpp:73-151 */
static int isect_points(const pt *p1, const pt *p2, pt *out) {
  pt v1[4], v2[4];
  for (int i = 0; i < 4; ++i) { v1[i] = sub2(p1[(i + 1) % 4], p1[i]); v2[i] = sub2(p2[(i + 1) % 4], p2[i]); }
  int num = 0;
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) {
      float det = cross2(v2[j], v1[i]);
      if (fabs((double)det) <= 1e-14) continue;
      pt v12 = sub2(p2[j], p1[i]);
      float t1 = cross2(v2[j], v12) / det;
      float t2 = cross2(v1[i], v12) / det;
      if (t1 >= 0.0f && t1 <= 1.0f && t2 >= 0.0f && t2 <= 1.0f) {
        out[num].x = p1[i].x + v1[i].x * t1;
        out[num].y = p1[i].y + v1[i].y * t1;
        ++num;
      }
    }
  for (int pass = 0; pass < 2; ++pass) { /* vertices of rect1 in rect2, then the reverse */
    const pt *P = pass ? p2 : p1, *Q = pass ? p1 : p2;
    const pt *V = pass ? v1 : v2;
    pt AB = V[0], DA = V[3];
    float ABAB = dot2(AB, AB), ADAD = dot2(DA, DA);
    for (int i = 0; i < 4; ++i) {
      pt AP = sub2(P[i], Q[0]);
      float ab = dot2(AP, AB), ad = -dot2(AP, DA);
      if (ab >= 0 && ad >= 0 && ab <= ABAB && ad <= ADAD) out[num++] = P[i];
    }
  }
  return num;
}

/* comparator of the CPU std::sort branch, rbox_utils.hpp:209-218 */
static int hull_less(pt A, pt B) {
  float t = cross2(A, B);
  if (fabs((double)t) < 1e-6) return dot2(A, A) < dot2(B, B);
  return t > 0;
}

/* rbox_utils.hpp:153-264, CPU branch, shift_to_zero = true.  Note (faithful quirk): the CPU
 * branch sorts q but NOT dist, so step 4 reads the pre-sort distances. */
static int hull_graham(const pt *p, int n_in, pt *q) {
  int t = 0;
  for (int i = 1; i < n_in; ++i)
    if (p[i].y < p[t].y || (p[i].y == p[t].y && p[i].x < p[t].x)) t = i;
  pt start = p[t];
  for (int i = 0; i < n_in; ++i) q[i] = sub2(p[i], start);
  pt tmp = q[0]; q[0] = q[t]; q[t] = tmp;
  float dist[24];
  for (int i = 0; i < n_in; ++i) dist[i] = dot2(q[i], q[i]);
  for (int i = 2; i < n_in; ++i) { /* insertion sort of q[1..n_in) */
    pt key = q[i];
    int j = i - 1;
    while (j >= 1 && hull_less(key, q[j])) { q[j + 1] = q[j]; --j; }
    q[j + 1] = key;
  }
  int k;
  for (k = 1; k < n_in; ++k)
    if (dist[k] > 1e-8) break;
  if (k == n_in) { q[0] = p[t]; return 1; }
  q[1] = q[k];
  int m = 2;
  for (int i = k + 1; i < n_in; ++i) {
    while (m > 1 && cross2(sub2(q[i], q[m - 2]), sub2(q[m - 1], q[m - 2])) >= 0) --m;
    q[m++] = q[i];
  }
  return m;
}

/* rbox_utils.hpp:266-302 */
static float rot_intersection(const rbox *b1, const rbox *b2) {
  pt p1[4], p2[4], ip[24], op[24];
  rot_vertices(b1, p1);
  rot_vertices(b2, p2);
  int num = isect_points(p1, p2, ip);
  if (num <= 2) return 0.0f;
  int m = hull_graham(ip, num, op);
  if (m <= 2) return 0.0f;
  float area = 0.0f;
  for (int i = 1; i < m - 1; ++i)
    area += fabsf(cross2(sub2(op[i], op[0]), sub2(op[i + 1], op[0])));
  return (float)(area / 2.0);
}

/* affinity.cpp:51-81 */
void rbox_oracle_eval_iou_bev(const float *det, int64_t nd, const float *gt, int64_t ng, float *out) {
  for (int64_t di = 0; di < nd; ++di)
    for (int64_t gi = 0; gi < ng; ++gi) {
      const float *d = det + 7 * di, *g = gt + 7 * gi;
      rbox D = {d[0], d[1], d[3], d[4], d[6]}, G = {g[0], g[1], g[3], g[4], g[6]};
      float da = d[3] * d[4], ga = g[3] * g[4];
      float inter = rot_intersection(&D, &G);
      inter = inter < 0.f ? 0.f : inter;
      inter = inter > da ? da : inter;
      inter = inter > ga ? ga : inter;
      float un = da + ga - inter;
      un = un < 1.1920928955078125e-7f ? 1.1920928955078125e-7f : un;
      out[di * ng + gi] = inter / un;
    }
}

/* affinity.cpp:8-49 */
void rbox_oracle_eval_iou_3d(const float *det, int64_t nd, const float *gt, int64_t ng, float z_offset,
                             float *out) {
  for (int64_t di = 0; di < nd; ++di)
    for (int64_t gi = 0; gi < ng; ++gi) {
      const float *d = det + 7 * di, *g = gt + 7 * gi;
      rbox D = {d[0], d[1], d[3], d[4], d[6]}, G = {g[0], g[1], g[3], g[4], g[6]};
      float bev = rot_intersection(&D, &G);
      float dzb = d[2] + (z_offset - 0.5f) * d[5], gzb = g[2] + (z_offset - 0.5f) * g[5];
      float dzt = d[2] + (z_offset + 0.5f) * d[5], gzt = g[2] + (z_offset + 0.5f) * g[5];
      float zb = dzb > gzb ? dzb : gzb, zt = dzt < gzt ? dzt : gzt;
      float zi = zt - zb;
      zi = zi < 0.f ? 0.f : zi;
      float dv = d[3] * d[4] * d[5], gv = g[3] * g[4] * g[5];
      float iv = bev * zi;
      iv = iv < 0.f ? 0.f : iv;
      iv = iv > dv ? dv : iv;
      iv = iv > gv ? gv : iv;
      float uv = dv + gv - iv;
      uv = uv < 1.1920928955078125e-7f ? 1.1920928955078125e-7f : uv;
      out[di * ng + gi] = iv / uv;
    }
}

/* affinity.cpp:83-105 — BEV centre distance (LidarCenterTransBEV).  `cols` = row length of det / gt (the reference
 * reads columns 0 and 1 of whatever 2-D arrays it is given).  Whether the reference's unqualified `sqrt` binds to the
 * float or the double overload does not matter: a double sqrt of a float argument rounded back to float equals the
 * correctly rounded float sqrt (53 >= 2*24 + 2).  Pinned against oracle/_ref bit for bit. */
void rbox_oracle_trans_bev(const float *det, int64_t nd, int64_t dcols, const float *gt, int64_t ng, int64_t gcols,
                           float *out) {
  for (int64_t di = 0; di < nd; ++di)
    for (int64_t gi = 0; gi < ng; ++gi) {
      const float dx = det[di * dcols] - gt[gi * gcols], dy = det[di * dcols + 1] - gt[gi * gcols + 1];
      out[di * ng + gi] = sqrtf(dx * dx + dy * dy);
    }
}

/* matcher.cpp:8-74 — COCO-style greedy matching of detections (rows, in the given order) to ground truths for every
 * cost threshold: a detection takes the cheapest still-free (or crowd) gt with cost <= thr, non-ignore gts beating
 * ignore ones; ties go to the LATER gt (`<=`).  matched (n_thr, n_det) int32, -1 = unmatched. */
void rbox_oracle_match_coco(const float *cost, const float *thrs, const uint8_t *is_ignore, const uint8_t *is_crowd,
                            int64_t nd, int64_t ng, int64_t nt, int32_t *matched) {
  uint8_t *taken = (uint8_t *)calloc((size_t)(nt * ng > 0 ? nt * ng : 1), 1);
  for (int64_t t = 0; t < nt; ++t)
    for (int64_t d = 0; d < nd; ++d) {
      const float thr = thrs[t];
      float best = thr;
      int64_t m = -1;
      for (int64_t g = 0; g < ng; ++g) {
        if (taken[t * ng + g] && !is_crowd[g]) continue;
        const float v = cost[d * ng + g];
        if (m == -1) {
          if (v <= best) { best = v; m = g; }
        } else if (is_ignore[m]) {
          if (!is_ignore[g]) {
            if (v <= thr) { best = v; m = g; }
          } else if (v <= best) { best = v; m = g; }
        } else if (!is_ignore[g] && v <= best) { best = v; m = g; }
      }
      if (m != -1) taken[t * ng + m] = 1;
      matched[t * nd + d] = (int32_t)m;
    }
  free(taken);
}
