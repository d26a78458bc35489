// gd3d_device.h — per-pair register math of the fused Gaussian-distance kernels (gfx950).
//
// One thread owns one (pred, target) pair: 14 input floats in registers, the loss value and
// the 7 (or 14) gradient floats out.  Everything is closed-form 2x2 algebra on the entries of
//   Sigma = R diag(a^2, b^2) R^T :  S11 = A c^2 + B s^2,  S12 = (A-B) s c,  S22 = A s^2 + B c^2
// which is what the reference builds with (N,2,2) bmm chains
// (/root/reference/mmdet3d_gaussian/models/losses/gaussian_distance_loss.py:8-21, 86-87).
// The backward is hand-derived reverse mode; clamp / sqrt-at-zero / max-tie rules are those
// torch autograd applies to the reference graph (clamp: pass-through on the closed interval;
// clamp(0).sqrt(): 0 for u < 0, +inf slope at u == 0; maximum/minimum: ties split 1/2).
#pragma once
#ifdef GD3D_HOST_TWIN   // csrc/gd3d_cpu.cpp: the `_cpu` twins run this same per-pair math on the host
#include "gd3d_host_math.h"
#else
#include <hip/hip_runtime.h>
#endif

#include "../../include/gd3d.h"

namespace gd3d {

#define GD_DEV __device__ __forceinline__

// ------------------------------------------------------------------ scalar helpers
GD_DEV float frcp(float x) { return __builtin_amdgcn_rcpf(x); }      // v_rcp_f32, 1 ulp
GD_DEV float fsqrt(float x) { return __builtin_amdgcn_sqrtf(x); }    // v_sqrt_f32, 1 ulp
GD_DEV float frsq(float x) { return __builtin_amdgcn_rsqf(x); }      // v_rsq_f32, 1 ulp
GD_DEV float fexp2(float x) { return __builtin_amdgcn_exp2f(x); }    // v_exp_f32

// sin & cos.  Cody-Waite reduction by pi/2 with FMAs + Cephes minimax polynomials on
// [-pi/4, pi/4] (abs error < 1.5e-7 for |x| <= 8192).  Beyond that (never in practice for a
// yaw) fall back to the full-range library routine.
// quadrant n: sin takes the sign of bit 1 of n, cos that of bit 1 of n + 1 = bit 1 ^ bit 0 — applied as sign-bit XORs
// (two shifts and two three-input bit operations instead of two compares and two negating selects)
GD_DEV void quadrant_signs(int n, float sv, float cv, float& s, float& c) {
  const unsigned t30 = (unsigned)n << 30, t31 = (unsigned)n << 31;
  s = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, sv) ^ (t30 & 0x80000000u));
  c = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, cv) ^ ((t30 ^ t31) & 0x80000000u));
}

GD_DEV void sincos_poly(float x, float& s, float& c);
GD_DEV void sincos_f(float x, float& s, float& c) {
  if (__builtin_expect(!(fabsf(x) <= 8192.0f), 0)) {
    sincosf(x, &s, &c);
    return;
  }
  sincos_poly(x, s, c);
}
// the polynomial path alone, for callers that have already bounded |x| (<= 8192)
GD_DEV void sincos_poly(float x, float& s, float& c) {
  const float q = rintf(x * 0.63661977236758134f);
  float r = fmaf(q, -1.5703125f, x);
  r = fmaf(q, -4.837512969970703125e-4f, r);
  r = fmaf(q, -7.54978995489188216e-8f, r);
  const int n = (int)q;
  const float z = r * r;
  float ps = fmaf(fmaf(-1.9515295891e-4f, z, 8.3321608736e-3f), z, -1.6666654611e-1f);
  ps = fmaf(ps * z, r, r);
  float pc = fmaf(fmaf(2.443315711809948e-5f, z, -1.388731625493765e-3f), z, 4.166664568298827e-2f);
  pc = fmaf(pc * z, z, fmaf(-0.5f, z, 1.0f));
  const float sv = (n & 1) ? pc : ps;
  const float cv = (n & 1) ? ps : pc;
  quadrant_signs(n, sv, cv, s, c);
}

// The same for TWO angles (|x| <= 8192 each): range reduction and both polynomials run as packed fp32 (v_pk_mul_f32 /
// v_pk_fma_f32 on two-element vectors: one issue slot for both angles); only rounding, the integer quadrant and the final
// selects are per angle.  Bit-identical to two sincos_poly calls.
typedef float v2f __attribute__((ext_vector_type(2)));
GD_DEV v2f fma2(v2f a, v2f b, v2f c) { return __builtin_elementwise_fma(a, b, c); }
GD_DEV v2f splat2(float v) { return v2f{v, v}; }
GD_DEV void sincos_poly2(float x0, float x1, float& s0, float& c0, float& s1, float& c1) {
  const v2f x = {x0, x1};
  const v2f xq = x * splat2(0.63661977236758134f);
  const v2f q = {rintf(xq.x), rintf(xq.y)};
  v2f r = fma2(q, splat2(-1.5703125f), x);
  r = fma2(q, splat2(-4.837512969970703125e-4f), r);
  r = fma2(q, splat2(-7.54978995489188216e-8f), r);
  const v2f z = r * r;
  v2f ps = fma2(fma2(splat2(-1.9515295891e-4f), z, splat2(8.3321608736e-3f)), z, splat2(-1.6666654611e-1f));
  ps = fma2(ps * z, r, r);
  v2f pc = fma2(fma2(splat2(2.443315711809948e-5f), z, splat2(-1.388731625493765e-3f)), z, splat2(4.166664568298827e-2f));
  pc = fma2(pc * z, z, fma2(splat2(-0.5f), z, splat2(1.0f)));
  const int n0 = (int)q.x, n1 = (int)q.y;
  const float sv0 = (n0 & 1) ? pc.x : ps.x, cv0 = (n0 & 1) ? ps.x : pc.x;
  const float sv1 = (n1 & 1) ? pc.y : ps.y, cv1 = (n1 & 1) ? ps.y : pc.y;
  quadrant_signs(n0, sv0, cv0, s0, c0);
  quadrant_signs(n1, sv1, cv1, s1, c1);
}

GD_DEV float flog2(float x) { return __builtin_amdgcn_logf(x); }     // v_log_f32 (base 2), 1 ulp
constexpr float LN2 = 0.6931471805599453f;
static inline float gd3d_inv_alpha2(float alpha) { return 1.0f / (alpha * alpha); }   // host side, once per launch

// log1p with the (1+d) rounding error folded back in; *inv_u returns 1/(1+d) (= d log1p / dd) from the same v_rcp
GD_DEV float log1p_f(float d, float& inv_u) {
  const float u = 1.0f + d;
  inv_u = frcp(u);
  const float lost = d - (u - 1.0f);                 // d = +inf: inf - inf; torch.log1p(inf) = inf, so no correction there
  const float corr = (inv_u == 0.0f) ? 0.0f : lost * inv_u;
  return fmaf(flog2(u), LN2, corr);
}

// ------------------------------------------------------------------ box -> Gaussian
// Offset of the gravity centres, C_p - C_t with C = xyz + c * dims (UNCLAMPED dims, ref :12), taken as
// (xyz_p - xyz_t) + c * (dims_p - dims_t): the differences first.  The reference rounds each centre (x up to 70 m: half an
// ulp is 4e-6 m) and subtracts; for boxes that are close the two inner differences here are exact (Sterbenz), so the offset
// keeps the inputs' full precision — with a non-zero c on x or y this is what limited the reference's own fp32 (and round
// 2's kernel) to 1e-5 on the KITTI family.  NaN / inf dims still reach the loss through it (0 * NaN, 0 * inf are NaN).
struct Offs {
  float dX, dY, dZ;
};
GD_DEV void center_delta(const float (&pv)[7], const float (&tv)[7], const float (&c)[3], Offs& o) {
  o.dX = fmaf(c[0], pv[3] - tv[3], pv[0] - tv[0]);
  o.dY = fmaf(c[1], pv[4] - tv[4], pv[1] - tv[1]);
  o.dZ = fmaf(c[2], pv[5] - tv[5], pv[2] - tv[2]);
}

struct Box {
  float a, b, e;     // half extents of clamped dims (ref :13-14, :19-20)
  float mw, mh, ml;  // 0.5 * clamp pass-through mask
  float co, si;
  float A, B;
  float S11, S12, S22;
};

struct Adj {
  float gX, gY, gZ, ga, gb, ge, gr;
};

GD_DEV float half_clamp(float v, float& m) {
  // 0.5 * torch.clamp(v, 1e-7, 1e7) in one v_med3_f32; the clamp passes gradient iff it changed nothing.
  // A NaN dim: med3 returns a bound and m = 0; the NaN still reaches the loss through the gravity
  // centre (X = x + c*w uses the raw dim, even for c = 0) except in kfiou3d, which re-injects it.
  const float cl = __builtin_amdgcn_fmed3f(v, 1e-7f, 1e7f);
  m = (cl == v) ? 0.5f : 0.0f;
  return 0.5f * cl;
}

GD_DEV void rotdiag(float dA, float dB, float co, float si, float& m11, float& m12, float& m22) {
  const float cc = co * co, ss = si * si;
  m11 = fmaf(dA, cc, dB * ss);
  m12 = (dA - dB) * (si * co);
  m22 = fmaf(dA, ss, dB * cc);
}

GD_DEV void rotdiag_bwd(float dA, float dB, float co, float si, float m12, float g11, float g12,
                        float g22, float& g_dA, float& g_dB, float& g_r) {
  const float cc = co * co, ss = si * si, sc = si * co;
  g_dA = fmaf(g11, cc, fmaf(g12, sc, g22 * ss));
  g_dB = fmaf(g11, ss, fmaf(-g12, sc, g22 * cc));
  g_r = fmaf(g22 - g11, 2.0f * m12, g12 * (dA - dB) * (cc - ss));
}

// SIGMA = false skips sin/cos and the covariance entries (gwd3d only needs the yaw DIFFERENCE, see gwd<>)
template <bool SIGMA>
GD_DEV void box_make(const float (&v)[7], const float (&c)[3], Box& o) {
  o.a = half_clamp(v[3], o.mw);
  o.b = half_clamp(v[4], o.mh);
  o.e = half_clamp(v[5], o.ml);
  o.A = o.a * o.a;
  o.B = o.b * o.b;
  if (SIGMA) {
    sincos_f(v[6], o.si, o.co);
    rotdiag(o.A, o.B, o.co, o.si, o.S11, o.S12, o.S22);
  }
}

GD_DEV void sigma_bwd(const Box& bx, float g11, float g12, float g22, Adj& g) {
  float gA, gB, gr;
  rotdiag_bwd(bx.A, bx.B, bx.co, bx.si, bx.S12, g11, g12, g22, gA, gB, gr);
  g.ga = fmaf(gA, 2.0f * bx.a, g.ga);
  g.gb = fmaf(gB, 2.0f * bx.b, g.gb);
  g.gr += gr;
}

GD_DEV void box_grad(const Box& bx, const Adj& g, const float (&c)[3], float f, float (&out)[7]) {
  out[0] = f * g.gX;
  out[1] = f * g.gY;
  out[2] = f * g.gZ;
  out[3] = f * fmaf(c[0], g.gX, bx.mw * g.ga);
  out[4] = f * fmaf(c[1], g.gY, bx.mh * g.gb);
  out[5] = f * fmaf(c[2], g.gZ, bx.ml * g.ge);
  out[6] = f * g.gr;
}

GD_DEV void adj_zero(Adj& g) { g.gX = g.gY = g.gZ = g.ga = g.gb = g.ge = g.gr = 0.0f; }

// ------------------------------------------------------------------ postprocess (ref :24-39)
template <int FUN>
GD_DEV float post(float d, float tau, float& deriv) {
  float f, df;
  if (FUN == GD3D_FUN_LOG1P) {
    f = log1p_f(d, df);
  } else if (FUN == GD3D_FUN_EXPM1) {
    f = expm1f(d);
    df = f + 1.0f;
  } else if (FUN == GD3D_FUN_NLOG) {
    const float t = 1.0f - d + 1e-7f;
    f = -logf(t);
    df = frcp(t);
  } else {
    f = d;
    df = 1.0f;
  }
  if (tau >= 1.0f) {  // wave-uniform (kernel argument)
    const float iq = frcp(tau + f);
    deriv = tau * iq * iq * df;
    return fmaf(-tau, iq, 1.0f);
  }
  deriv = df;
  return f;
}

// clamp(0).sqrt(): value and d sqrt/du under autograd rules (slope 0 for u < 0, +inf at u == 0).
// u * 0 keeps a NaN a NaN where clamp(0) would (torch.clamp propagates NaN) and is 0 otherwise.
// One v_rsq_f32 serves both: sqrt(u) = u * rsq(u), 1/(2 sqrt(u)) = rsq(u)/2 (rsq(0) = +inf as required).
GD_DEV float sqrt0(float u, float& dsu) {
  const float r = frsq(u);
  const float s = (u > 0.0f) ? ((r == 0.0f) ? u : u * r) : u * 0.0f;   // rsq(+inf) = 0: sqrt(inf) = inf, not inf * 0
  dsu = (u >= 0.0f) ? 0.5f * r : 0.0f;
  return s;
}

// ------------------------------------------------------------------ gwd3d (ref :42-106)
// The reference forms whlr = tr(Sp) + tr(St) - 2 sqrt(tr(Sp St) + 2 sqrt(det Sp det St)) + (ep-et)^2 from the covariance
// entries (:81-97), which cancels catastrophically when the boxes are close (its own fp32 result is off by ~2e-4 there,
// SURVEY.md §4).  The same quantity in a cancellation-free closed form, with d = r_p - r_t, r0 = ap at + bp bt and
// K = (ap^2-bp^2)(at^2-bt^2):
//     tr(Sp St) + 2 ap bp at bt = r0^2 - K sin^2 d =: q
//     tr(Sp) + tr(St) - 2 sqrt(q) = (ap-at)^2 + (bp-bt)^2 + 2 K sin^2 d / (r0 + sqrt q)
// (identity noted in SURVEY.md §8a; every term is a product of small differences, none is a difference of large
// numbers).  One sin/cos of the yaw difference replaces two sin/cos + two R S^2 R^T.  Gradients by direct
// differentiation of the same form:
//     d/d ap = 2 [(ap-at) + at m + ap (At-Bt) sin^2 d / sqrt q],  m = 1 - r0/sqrt q = -K sin^2 d / ((r0 + sqrt q) sqrt q)
//     d/d bp = 2 [(bp-bt) + bt m - bp (At-Bt) sin^2 d / sqrt q],  d/d r_p = 2 K sin d cos d / sqrt q = - d/d r_t
template <int FUN, bool NORMALIZE, bool GT>
GD_DEV float gwd(const Box& p, const Box& t, const Offs& o, float yaw_p, float yaw_t, float alpha, float tau, Adj& gp, Adj& gt) {
  const float dX = o.dX, dY = o.dY, dZ = o.dZ;
  const float dxyz = fmaf(dX, dX, fmaf(dY, dY, dZ * dZ));
  float sd, cd;
  if (__builtin_expect(!(fabsf(yaw_p) <= 16.0f && fabsf(yaw_t) <= 16.0f), 0)) {
    // fl(yaw_p - yaw_t) is off by up to ulp(yaw)/2 (1e-6 rad at 16, 0.03 rad at 1e6): for yaws that far out take the
    // sine and cosine of the difference from the two angles themselves, as the reference's per-box rotations do
    float sp, cp, st, ct;
    sincos_f(yaw_p, sp, cp);
    sincos_f(yaw_t, st, ct);
    sd = fmaf(sp, ct, -cp * st);
    cd = fmaf(cp, ct, sp * st);
  } else {
    sincos_poly(yaw_p - yaw_t, sd, cd);   // |difference| <= 32: no range check of its own (one branch on this path, as before)
  }
  const float s2 = sd * sd;
  const float r0 = fmaf(p.a, t.a, p.b * t.b);
  const float dAp = (p.a - p.b) * (p.a + p.b), dAt = (t.a - t.b) * (t.a + t.b);
  const float K = dAp * dAt;
  const float Ks2 = K * s2;
  const float q = fmaf(r0, r0, -Ks2);  // >= (ap bt + bp at)^2 > 0 for clamped dims; NaN inputs stay NaN
  const float rq = frsq(q);
  const float sq = q * rq;
  const float irs = frcp(r0 + sq);
  const float da = p.a - t.a, db = p.b - t.b, de = p.e - t.e;
  const float whlr = fmaf(de, de, fmaf(da, da, fmaf(db, db, 2.0f * Ks2 * irs)));
  const float a2 = alpha * alpha;
  const float u = fmaf(a2, whlr, dxyz);
  float ddist;
  const float dist = sqrt0(u, ddist);
  float dn = dist, iscale = 1.0f;
  const float Dp = p.a * p.b, Dt = t.a * t.b;
  if (NORMALIZE) {
    // 2 exp((ln D + ln ep + ln et)/6) = 2 * 2^((log2 D + log2(ep et))/6)
    const float L2 = flog2(Dp * Dt) + flog2(p.e * t.e);
    iscale = 0.5f * fexp2(L2 * (-1.0f / 6.0f));
    dn = dist * iscale;
  }
  float dpost;
  const float out = post<FUN>(dn, tau, dpost);

  const float g_dist = dpost * iscale;
  const float g_L = NORMALIZE ? -dpost * dn * (1.0f / 6.0f) : 0.0f;
  const float g_u = g_dist * ddist;
  const float g_w2 = 2.0f * g_u * a2;          // 2 x adjoint of whlr
  const float m = -Ks2 * irs * rq;             // 1 - r0 / sqrt(q)
  const float e_t = dAt * s2 * rq, e_p = dAp * s2 * rq;
  gp.gX = 2.0f * g_u * dX;
  gp.gY = 2.0f * g_u * dY;
  gp.gZ = 2.0f * g_u * dZ;
  gp.ga = fmaf(g_w2, fmaf(p.a, e_t, fmaf(t.a, m, da)), g_L * frcp(p.a));
  gp.gb = fmaf(g_w2, fmaf(-p.b, e_t, fmaf(t.b, m, db)), g_L * frcp(p.b));
  gp.ge = fmaf(g_w2, de, g_L * frcp(p.e));
  gp.gr = g_w2 * K * sd * cd * rq;
  if (GT) {
    gt.gX = -gp.gX;
    gt.gY = -gp.gY;
    gt.gZ = -gp.gZ;
    gt.ga = fmaf(g_w2, fmaf(t.a, e_p, fmaf(p.a, m, -da)), g_L * frcp(t.a));
    gt.gb = fmaf(g_w2, fmaf(-t.b, e_p, fmaf(p.b, m, -db)), g_L * frcp(t.b));
    gt.ge = fmaf(-g_w2, de, g_L * frcp(t.e));
    gt.gr = -gp.gr;
  }
  return out;
}

// ------------------------------------------------------------------ shared pair geometry (kld / jd / sym / bd)
// Everything below works in the frame of ONE box and in the yaw DIFFERENCE d = r_p - r_t, like gwd<> above: the
// reference's covariance entries (R diag R^T, (N,2,2) bmm chains) only ever enter through rotation-invariant
// combinations, and written in d and in differences of the extents those combinations have no cancellation left:
// the reference's own fp32 loses 3-4 digits on near-identical boxes (1.5 - 1.5 + O(delta^2)), these forms do not.
struct Geo {
  float dX, dY, dZ;  // C_p - C_t (gravity centres)
  float sd, cd;      // sin / cos of d = yaw_p - yaw_t
  float sp, cp;      // sin / cos of yaw_p
};

GD_DEV void geo_make(const Offs& o, float yaw_p, float yaw_t, Geo& g) {
  g.dX = o.dX;
  g.dY = o.dY;
  g.dZ = o.dZ;
  if (__builtin_expect(!(fabsf(yaw_p) <= 16.0f && fabsf(yaw_t) <= 16.0f), 0)) {
    // fl(yaw_p - yaw_t) is off by up to ulp(yaw)/2 (1e-6 rad at 16, 0.03 rad at 1e6): out there take the difference's sine
    // and cosine from the two angles themselves, as the reference's per-box rotations do (cf. gwd<>)
    float st, ct;
    sincos_f(yaw_p, g.sp, g.cp);
    sincos_f(yaw_t, st, ct);
    g.sd = fmaf(g.sp, ct, -g.cp * st);
    g.cd = fmaf(g.cp, ct, g.sp * st);
  } else {
    sincos_poly2(yaw_p - yaw_t, yaw_p, g.sd, g.cd, g.sp, g.cp);   // (the difference is exact when the yaws are close: Sterbenz)
  }
}

// the target box in the pred role: its own frame (yaw_t = yaw_p - d) and the mirrored offset and yaw difference
struct GeoT {
  float st, ct;
};
GD_DEV void geo_swap(const Geo& G, GeoT& o) {
  o.ct = fmaf(G.cp, G.cd, G.sp * G.sd);
  o.st = fmaf(G.sp, G.cd, -G.cp * G.sd);
}

// f(rho) = (rho^2 - 1)/2 - ln(rho) for rho = num / den, the per-extent term of KL between Gaussians: >= 0, zero and flat
// at rho = 1, where (rho^2-1)/2 and ln(rho) cancel.  With delta = rho - 1 and z = (num-den)/(num+den) = delta/(2+delta):
//     ln(rho) = 2 atanh(z) = 2z + 2z^3/3 + 2z^5/5 + ...,   delta - 2z = delta z   (exactly)
//     f = delta^2/2 + delta z - 2 z^3 (1/3 + z^2/5 + z^4/7 + z^6/9 + z^8/11)
// every term a product of differences.  Five series terms are good to 1.2e-7 of f for |z| <= 1/3 (rho in [1/2, 2]); lanes
// outside take the direct expression (no cancellation there: |ln rho| > 0.69).  *delta returns rho - 1.
GD_DEV bool ratio_far(float d, float sum) { return !(3.0f * fabsf(d) <= sum); }   // |z| > 1/3; also true for NaN
GD_DEV float ratio_direct(float num, float iden, float delta) {
  return fmaf(delta, fmaf(0.5f, delta, 1.0f), -LN2 * flog2(num * iden));
}
GD_DEV float ratio_series(float num, float den, float iden, float& delta) {
  const float d = num - den, sum = num + den;
  delta = d * iden;
  const float z = d * frcp(sum);
  const float w = z * z;
  float P = fmaf(w, 1.0f / 11.0f, 1.0f / 9.0f);
  P = fmaf(w, P, 1.0f / 7.0f);
  P = fmaf(w, P, 1.0f / 5.0f);
  P = fmaf(w, P, 1.0f / 3.0f);
  return fmaf(delta, fmaf(0.5f, delta, z), -2.0f * (z * w) * P);
}
// two ratios at once (the a and the b extents), packed fp32; bit-identical to two ratio_series calls
GD_DEV v2f ratio_series2(v2f num, v2f den, v2f iden, v2f& delta) {
  const v2f d = num - den, sum = num + den;
  delta = d * iden;
  const v2f z = d * v2f{frcp(sum.x), frcp(sum.y)};
  const v2f w = z * z;
  v2f P = fma2(w, splat2(1.0f / 11.0f), splat2(1.0f / 9.0f));
  P = fma2(w, P, splat2(1.0f / 7.0f));
  P = fma2(w, P, splat2(1.0f / 5.0f));
  P = fma2(w, P, splat2(1.0f / 3.0f));
  return fma2(delta, fma2(splat2(0.5f), delta, z), splat2(-2.0f) * (z * w) * P);
}

// The reference's own evaluation of [dX dY] M [dX dY]^T (two bmm's, ref :119-121, :168-170).  Used only when a centre offset
// is not finite, to decide whether the reference's quadratic form is NaN (inf - inf or 0 * inf between ITS world-frame
// terms) rather than +inf: the box-frame forms below group the same infinities differently.
GD_DEV float quad_ref(float dX, float dY, float m11, float m12, float m22) {
  const float r0 = dX * m11 + dY * m12, r1 = dX * m12 + dY * m22;
  return r0 * dX + r1 * dY;
}
GD_DEV bool offset_not_finite(float dX, float dY) { return !(fabsf(dX) + fabsf(dY) < __builtin_inff()); }

// ------------------------------------------------------------------ kld3d core (ref :109-137)
// kl(q, r) = kld3d_loss(pred = q, target = r) before sqrt / postprocess = KL(N_r || N_q), with
//   (u, v) = R_q^T (C_q - C_r)           centre offset in q's frame
//   xyz    = (u^2/Aq + v^2/Bq + dZ^2/Eq) / 2
//   whlr   = f(ar/aq) + f(br/bq) + f(er/eq) + sin^2(d) (Ar-Br)(Aq-Bq) / (2 Aq Bq)
// (tr(Sq^-1 Sr) = Ar/Aq + Br/Bq + sin^2 d (Ar-Br)(1/Bq - 1/Aq); the -1.5 and the six logs of ref :126-136 are inside f.)
struct KlI {
  float iaq, ibq, ieq;  // 1 / half extents of q
  float u, v;
  float U, V, W;        // u/aq, v/bq, dZ/eq
  float da, db, de;     // rho - 1 of the three extent ratios r/q
  float dAr, dAq;       // Ar - Br, Aq - Bq
  float iAB2;           // 1 / (Aq Bq)
};

// dX,dY,dZ = C_q - C_r; (cq, sq) = cos / sin of yaw_q; s = sin(yaw_q - yaw_r)
GD_DEV float kl_fwd(const Box& q, const Box& r, float dX, float dY, float dZ, float cq, float sq, float s, float ia2,
                    KlI& k) {
  k.iaq = frcp(q.a);
  k.ibq = frcp(q.b);
  k.ieq = frcp(q.e);
  k.u = fmaf(cq, dX, sq * dY);
  k.v = fmaf(cq, dY, -sq * dX);
  k.U = k.u * k.iaq;
  k.V = k.v * k.ibq;
  k.W = dZ * k.ieq;
  float xyz2 = fmaf(k.U, k.U, fmaf(k.V, k.V, k.W * k.W));
  v2f dab;
  v2f fab = ratio_series2(v2f{r.a, r.b}, v2f{q.a, q.b}, v2f{k.iaq, k.ibq}, dab);
  float fe = ratio_series(r.e, q.e, k.ieq, k.de);
  k.da = dab.x;
  k.db = dab.y;
  // ONE rare branch for everything the fast forms do not cover: an extent ratio outside [1/2, 2] (direct expression) and a
  // centre offset that is not finite (NaN decision of the reference's own quadratic form)
  const bool far_a = ratio_far(r.a - q.a, r.a + q.a), far_b = ratio_far(r.b - q.b, r.b + q.b), far_e = ratio_far(r.e - q.e, r.e + q.e);
  const bool bad_off = offset_not_finite(dX, dY);
  if (__builtin_expect(far_a || far_b || far_e || bad_off, 0)) {
    if (far_a) fab.x = ratio_direct(r.a, k.iaq, k.da);
    if (far_b) fab.y = ratio_direct(r.b, k.ibq, k.db);
    if (far_e) fe = ratio_direct(r.e, k.ieq, k.de);
    if (bad_off) {
      const float iA = k.iaq * k.iaq, iB = k.ibq * k.ibq;
      const float qr = quad_ref(dX, dY, fmaf(iA, cq * cq, iB * sq * sq), (iA - iB) * (sq * cq), fmaf(iA, sq * sq, iB * cq * cq));
      if (qr != qr) xyz2 = qr;
    }
  }
  const float fa = fab.x, fb = fab.y;
  k.dAr = (r.a - r.b) * (r.a + r.b);
  k.dAq = (q.a - q.b) * (q.a + q.b);
  const float iAB = k.iaq * k.ibq;
  k.iAB2 = iAB * iAB;
  const float T = 0.5f * (s * s) * (k.dAr * k.dAq) * k.iAB2;
  return fmaf(0.5f * ia2, xyz2, (fa + fb) + (fe + T));
}

// g (rho^2 - 1) of an extent ratio rho = r / q, from the accurately formed d = rho - 1.
// Test hooks (defined by tools/build_variants.py only, never in the product build; profiles/r05_gate_bites.txt shows the parity
// gates failing on them): GD_TEST_NAIVE_RATIO forms rho^2 - 1 the textbook way, which cancels on late-training (near-identical)
// boxes; GD_TEST_PERTURB_KL_BWD=<eps> scales everything kl_bwd accumulates by (1 + eps).
#ifdef GD_TEST_NAIVE_RATIO
#define GD_RATIO_TERM(g, d, r, iq) ((g) * (((r) * (iq)) * ((r) * (iq)) - 1.0f))
#else
#define GD_RATIO_TERM(g, d, r, iq) ((g) * (d) * (2.0f + (d)))
#endif

// accumulates into gq (if GQ) and gr (if GR) with upstream g; c = cos(yaw_q - yaw_r)
template <bool GQ, bool GR>
GD_DEV void kl_bwd(const Box& q, const Box& r, float cq, float sq, float s, float c, float ia2, const KlI& k, float g,
                   Adj& gq, Adj& gr) {
#ifdef GD_TEST_PERTURB_KL_BWD
  g *= 1.0f + (float)(GD_TEST_PERTURB_KL_BWD);
#endif
  const float gx = g * ia2;
  const float gu = gx * k.U * k.iaq, gv = gx * k.V * k.ibq;      // d/du, d/dv
  const float gDX = fmaf(cq, gu, -sq * gv), gDY = fmaf(sq, gu, cq * gv), gDZ = gx * k.W * k.ieq;
  const float s2 = s * s;
  const float gsc = g * (s * c) * (k.dAr * k.dAq) * k.iAB2;       // d/d yaw_q of the sin^2 term (= - d/d yaw_r)
  if (GQ) {
    gq.gX += gDX;
    gq.gY += gDY;
    gq.gZ += gDZ;
    gq.gr += fmaf(gu, k.v, fmaf(-gv, k.u, gsc));                  // u' = v, v' = -u under a turn of q
    const float t2 = g * s2 * k.dAr;                              // d T / d aq = s^2 (Ar-Br) / (Aq aq), d T / d bq = -(...)/(Bq bq)
    gq.ga += k.iaq * (fmaf(t2, k.iaq * k.iaq, GD_RATIO_TERM(-g, k.da, r.a, k.iaq)) - gx * k.U * k.U);
    gq.gb += k.ibq * (fmaf(-t2, k.ibq * k.ibq, GD_RATIO_TERM(-g, k.db, r.b, k.ibq)) - gx * k.V * k.V);
    gq.ge += k.ieq * (GD_RATIO_TERM(-g, k.de, r.e, k.ieq) - gx * k.W * k.W);
  }
  if (GR) {
    gr.gX -= gDX;
    gr.gY -= gDY;
    gr.gZ -= gDZ;
    gr.gr -= gsc;
    const float t3 = g * s2 * k.dAq * k.iAB2;                     // d T / d ar = s^2 ar (Aq-Bq)/(Aq Bq)
    gr.ga += fmaf(GD_RATIO_TERM(g, k.da, r.a, k.iaq), frcp(r.a), t3 * r.a);  // d f / d ar = (rho^2 - 1) / ar
    gr.gb += fmaf(GD_RATIO_TERM(g, k.db, r.b, k.ibq), frcp(r.b), -t3 * r.b);
    gr.ge += GD_RATIO_TERM(g, k.de, r.e, k.ieq) * frcp(r.e);
  }
}

template <int FUN, bool SQRT, bool GT>
GD_DEV float kld(const Box& p, const Box& t, const Geo& G, float ia2, float tau, Adj& gp, Adj& gt) {
  KlI k;
  float d = kl_fwd(p, t, G.dX, G.dY, G.dZ, G.cp, G.sp, G.sd, ia2, k), ds = 1.0f;
  if (SQRT) d = sqrt0(d, ds);
  float dpost;
  const float out = post<FUN>(d, tau, dpost);
  adj_zero(gp);
  adj_zero(gt);
  kl_bwd<true, GT>(p, t, G.cp, G.sp, G.sd, G.cd, ia2, k, dpost * ds, gp, gt);
  return out;
}

// ------------------------------------------------------------------ jd3d (ref :189-198)
template <int FUN, bool SQRT, bool GT>
GD_DEV float jd(const Box& p, const Box& t, const Geo& G, float ia2, float tau, Adj& gp, Adj& gt) {
  KlI k1, k2;
  GeoT S;
  geo_swap(G, S);
  float v = kl_fwd(p, t, G.dX, G.dY, G.dZ, G.cp, G.sp, G.sd, ia2, k1);
  v = v + kl_fwd(t, p, -G.dX, -G.dY, -G.dZ, S.ct, S.st, -G.sd, ia2, k2);
  v = v * 0.5f;
  float ds = 1.0f;
  if (SQRT) v = sqrt0(v, ds);
  float dpost;
  const float out = post<FUN>(v, tau, dpost);
  const float g = dpost * ds * 0.5f;
  adj_zero(gp);
  adj_zero(gt);
  kl_bwd<true, GT>(p, t, G.cp, G.sp, G.sd, G.cd, ia2, k1, g, gp, gt);
  kl_bwd<GT, true>(t, p, S.ct, S.st, -G.sd, G.cd, ia2, k2, g, gt, gp);
  return out;
}

// ------------------------------------------------------------------ symmax / symmin (ref :201-224)
template <int FUN, bool SQRT, bool GT, bool WANT_MAX>
GD_DEV float sym(const Box& p, const Box& t, const Geo& G, float ia2, float tau, Adj& gp, Adj& gt) {
  KlI k1, k2;
  GeoT S;
  geo_swap(G, S);
  float v1 = kl_fwd(p, t, G.dX, G.dY, G.dZ, G.cp, G.sp, G.sd, ia2, k1);
  float v2 = kl_fwd(t, p, -G.dX, -G.dY, -G.dZ, S.ct, S.st, -G.sd, ia2, k2);
  float ds1 = 1.0f, ds2 = 1.0f;
  if (SQRT) {
    v1 = sqrt0(v1, ds1);
    v2 = sqrt0(v2, ds2);
  }
  float m, f1, f2;
  if (v1 == v2) {
    m = v1;
    f1 = f2 = 0.5f;
  } else if ((v1 > v2) == WANT_MAX) {
    m = v1;
    f1 = 1.0f;
    f2 = 0.0f;
  } else {
    m = v2;
    f1 = 0.0f;
    f2 = 1.0f;
  }
  const bool poisoned = (v1 != v1) || (v2 != v2);
  if (poisoned) {
    m = v1 + v2;
    f1 = f2 = 0.0f;
  }
  float dpost;
  const float out = post<FUN>(m, tau, dpost);
  if (poisoned) {
    // torch.maximum / minimum backward: where(a == b, g/2, g).masked_fill(a < b, 0) — with a NaN operand neither mask
    // is set, so BOTH inputs receive the upstream gradient, which is NaN here (postprocess of a NaN): every gradient
    // entry of the row is NaN in the reference, not 0
    const float qn = m * 0.0f;   // NaN
    gp.gX = gp.gY = gp.gZ = gp.ga = gp.gb = gp.ge = gp.gr = qn;
    gt = gp;
    return out;
  }
  adj_zero(gp);
  adj_zero(gt);
  // a zero factor must contribute exactly 0 (not 0 * inf): branch instead of multiply
  if (f1 != 0.0f) kl_bwd<true, GT>(p, t, G.cp, G.sp, G.sd, G.cd, ia2, k1, dpost * f1 * ds1, gp, gt);
  if (f2 != 0.0f) kl_bwd<GT, true>(t, p, S.ct, S.st, -G.sd, G.cd, ia2, k2, dpost * f2 * ds2, gt, gp);
  return out;
}

// ------------------------------------------------------------------ bd3d (ref :144-186)
// In p's frame, with m = Sigma_p + Sigma_t there (twice the reference's Sigma), Ks = (At-Bt) sin^2 d:
//   m11 = Ap + At - Ks,  m22 = Bp + Bt + Ks,  m12 = -(At-Bt) sin d cos d
//   det4 = det m = (Ap+At)(Bp+Bt) + sin^2 d (Ap-Bp)(At-Bt)                  (= 4 det Sigma; sum of products, no cancellation)
//   xyz  = ( [u v] adj(m) [u v]^T / det4 + dZ^2 / (Ep+Et) ) / 4
//   whlr = ln( (1+qa)(1+qb)(1+qe)(1+qs) ) / 2,   qa = (ap-at)^2 / (2 ap at), ..., (1+qa)(1+qb) qs = sin^2 d (Ap-Bp)(At-Bt) / (4 ap at bp bt)
// (ref :174-180's five logs of O(1) quantities cancel to O(delta^2) on similar boxes; here the logarithm's argument is 1 + a sum
// of squares of differences.)  The reference clamps det Sigma at 1e-7 (:158): lanes that hit the clamp take its expression.
template <int FUN, bool SQRT, bool GT>
GD_DEV float bd(const Box& p, const Box& t, const Geo& G, float ia2, float tau, Adj& gp, Adj& gt) {
  const float u = fmaf(G.cp, G.dX, G.sp * G.dY), v = fmaf(G.cp, G.dY, -G.sp * G.dX);
  const float s = G.sd, c = G.cd, s2 = s * s, sc = s * c;
  const float Ep = p.e * p.e, Et = t.e * t.e;
  const float SA = p.A + t.A, SB = p.B + t.B, SE = Ep + Et;
  const float dAp = (p.a - p.b) * (p.a + p.b), dAt = (t.a - t.b) * (t.a + t.b);
  const float Ks = dAt * s2;
  const float m11 = SA - Ks, m22 = SB + Ks, m12 = -dAt * sc;
  const float KK = dAp * dAt;
  const float det4_raw = fmaf(SA, SB, s2 * KK);
  const bool clamped = !(det4_raw >= 4e-7f);          // det Sigma < 1e-7 (a NaN det can only come from NaN extents: impossible)
  const float det4 = __builtin_fmaxf(det4_raw, 4e-7f);
  const float idet4 = frcp(det4);
  const float N = fmaf(m22 * u, u, fmaf(-2.0f * m12 * u, v, m11 * v * v));
  const float iSE = frcp(SE);
  const float Nid = N * idet4;
  const bool bad_off = offset_not_finite(G.dX, G.dY);
  float xyz = 0.25f * fmaf(G.dZ * G.dZ, iSE, Nid);
  const float da = p.a - t.a, db = p.b - t.b, de = p.e - t.e;
  const float iPa = frcp(p.a * t.a), iPb = frcp(p.b * t.b), iPe = frcp(p.e * t.e);
  float whlr;
  {
    const float qa = 0.5f * da * da * iPa, qb = 0.5f * db * db * iPb, qe = 0.5f * de * de * iPe;
    const float Q1 = fmaf(qa, qb, qa + qb);
    const float Q2 = fmaf(0.25f * s2 * KK, iPa * iPb, Q1);
    const float Q = fmaf(Q2, qe, Q2 + qe);
    float unused;
    whlr = 0.5f * log1p_f(Q, unused);
  }
  if (__builtin_expect(clamped || bad_off, 0)) {   // ONE rare branch
    if (clamped)   // ref :158: det := 1e-7
      whlr = (0.5f * LN2) * ((flog2(1e-7f) + flog2(0.5f * SE)) - (flog2(p.a * p.b * p.e) + flog2(t.a * t.b * t.e)));
    if (bad_off) {
      GeoT S;
      geo_swap(G, S);
      float p11, p12, p22, t11, t12, t22;
      rotdiag(p.A, p.B, G.cp, G.sp, p11, p12, p22);
      rotdiag(t.A, t.B, S.ct, S.st, t11, t12, t22);
      const float qr = quad_ref(G.dX, G.dY, (p22 + t22) * idet4, -(p12 + t12) * idet4, (p11 + t11) * idet4);
      if (qr != qr) xyz = qr;   // the reference's quadratic form is NaN: so is the loss
    }
  }
  float d = fmaf(xyz, ia2, whlr), ds = 1.0f;
  if (SQRT) d = sqrt0(d, ds);
  float dpost;
  const float out = post<FUN>(d, tau, dpost);

  // Reverse mode.  The upstream factor is multiplied into u and v BEFORE anything is squared: with a centre 1e20 m away the
  // forward value overflows to its saturated 1.0 and g is exactly 0; 0 x (finite input) stays 0 as in the reference's autograd
  // graph, 0 x (overflowed intermediate) would be NaN.
  const float g = dpost * ds, h = 0.25f * g * ia2;
  const float mdet = clamped ? 0.0f : 1.0f;
  const float r = h * idet4;                                        // d xyz / d N
  const float ru = r * u, rv = r * v;
  const float gu = 2.0f * fmaf(m22, ru, -m12 * rv), gv = 2.0f * fmaf(m11, rv, -m12 * ru);
  const float rN = 0.5f * fmaf(gu, u, gv * v) * idet4 * mdet;       // r N / det4 = - d xyz / d det4
  const float gDZ = 2.0f * h * G.dZ * iSE;
  const float gDX = fmaf(G.cp, gu, -G.sp * gv), gDY = fmaf(G.sp, gu, G.cp * gv);
  const float gw = 0.5f * g * idet4 * mdet;                         // g / (2 det4): weight of the whlr numerators below
  // d/dd of N / det4 (u, v fixed): dN/dd = (At-Bt) (2 s c (u^2 - v^2) + 2 u v (c^2 - s^2)), d det4/dd = 2 s c (Ap-Bp)(At-Bt);
  // d whlr / dd = s c (Ap-Bp)(At-Bt) / det4
  const float gd = fmaf(2.0f * dAt, fmaf(sc, fmaf(ru, u, -rv * v), ru * v * ((c - s) * (c + s))), 2.0f * sc * KK * (gw - rN));
  const float gEz = -gDZ * G.dZ * iSE;                              // d xyz / d Ep = d xyz / d Et (before the factor e)
  const float sApBp = p.A + p.B, sAtBt = t.A + t.B;
  gp.gX = gDX;
  gp.gY = gDY;
  gp.gZ = gDZ;
  gp.gr = fmaf(gu, v, -gv * u) + gd;
  // d whlr / d ap = [(Ap-At)(Bp+Bt) + s^2 (At-Bt)(Ap+Bp)] / (2 ap det4)   (clamped: - 1 / (2 ap)), 1 / ap = at / (ap at)
  // (gw carries the clamp mask: with the determinant clamped the bracket drops out and cl = -g/2 is what is left)
  const float cl = clamped ? -0.5f * g : 0.0f;
  gp.ga = fmaf(2.0f * p.a, fmaf(rv, v, -rN * m22), fmaf(gw, fmaf(da * (p.a + t.a), SB, s2 * dAt * sApBp), cl) * (iPa * t.a));
  gp.gb = fmaf(2.0f * p.b, fmaf(ru, u, -rN * m11), fmaf(gw, fmaf(db * (p.b + t.b), SA, -s2 * dAt * sApBp), cl) * (iPb * t.b));
  gp.ge = fmaf(gEz, p.e, 0.5f * g * de * (p.e + t.e) * iSE * (iPe * t.e));
  if (GT) {
    const float su = fmaf(s, u, c * v), cu = fmaf(c, u, -s * v);
    const float rsu = fmaf(s, ru, c * rv), rcu = fmaf(c, ru, -s * rv);
    gt.gX = -gDX;
    gt.gY = -gDY;
    gt.gZ = -gDZ;
    gt.gr = -gd;
    gt.ga = fmaf(2.0f * t.a, fmaf(rsu, su, -rN * fmaf(s2, dAp, SB)),
                 fmaf(gw, fmaf(-da * (p.a + t.a), SB, s2 * dAp * sAtBt), cl) * (iPa * p.a));
    gt.gb = fmaf(2.0f * t.b, fmaf(rcu, cu, -rN * fmaf(-s2, dAp, SA)),
                 fmaf(gw, fmaf(-db * (p.b + t.b), SA, -s2 * dAp * sAtBt), cl) * (iPb * p.b));
    gt.ge = fmaf(gEz, t.e, -0.5f * g * de * (p.e + t.e) * iSE * (iPe * p.e));
  }
  return out;
}

// ------------------------------------------------------------------ kfiou3d (ref :227-248)
template <int FUN, bool GT>
GD_DEV float kfiou(const Box& p, const Box& t, float nanp, Adj& gp, Adj& gt) {
  const float S11 = p.S11 + t.S11, S12 = p.S12 + t.S12, S22 = p.S22 + t.S22;
  const float det2 = fmaf(S11, S22, -S12 * S12);
  const float detl = fmaf(p.e, p.e, t.e * t.e);
  const float det = det2 * detl;
  // kfiou3d never touches the centre, so a NaN DIM (which half_clamp swallows) is re-injected here: nanp is 0, or NaN
  // when one of the six raw dims is NaN (torch.clamp propagates NaN); positions and yaw are legitimately ignored
  const float vp = p.a * p.b * p.e + nanp, vt = t.a * t.b * t.e;
  const float m = det >= 1e-7f ? 1.0f : 0.0f;
  float detc = det >= 1e-7f ? det : 1e-7f;
  detc = (det != det) ? det : detc;
  const float isq = frsq(detc);
  const float inter = vp * vt * isq;
  const float un_raw = vp + vt - inter;
  const float mu = un_raw >= 1e-7f ? 1.0f : 0.0f;
  float un = un_raw >= 1e-7f ? un_raw : 1e-7f;
  un = (un_raw != un_raw) ? un_raw : un;
  const float iun = frcp(un);
  const float k = inter * iun;
  const float d = fmaf(-4.656854249492381f, k, 1.0f);
  float dpost;
  const float out = post<FUN>(d, 0.0f, dpost);  // tau is fixed to 0.0 (ref :247)

  const float g_k = -4.656854249492381f * dpost;
  float g_inter = g_k * iun;
  const float g_unraw = -g_k * k * iun * mu;
  g_inter -= g_unraw;
  const float g_vp = fmaf(g_inter, vt * isq, g_unraw);
  const float g_vt = fmaf(g_inter, vp * isq, g_unraw);
  const float g_det = -0.5f * g_inter * inter * isq * isq * m;  // d inter / d detc = -inter / (2 detc)
  const float g_det2 = g_det * detl, g_detl = g_det * det2;
  const float gS11 = g_det2 * S22, gS22 = g_det2 * S11, gS12 = -2.0f * g_det2 * S12;
  gp.gX = gp.gY = gp.gZ = 0.0f;
  gp.ga = g_vp * p.b * p.e;
  gp.gb = g_vp * p.a * p.e;
  gp.ge = fmaf(g_vp, p.a * p.b, 2.0f * g_detl * p.e);
  gp.gr = 0.0f;
  sigma_bwd(p, gS11, gS12, gS22, gp);
  if (GT) {
    gt.gX = gt.gY = gt.gZ = 0.0f;
    gt.ga = g_vt * t.b * t.e;
    gt.gb = g_vt * t.a * t.e;
    gt.ge = fmaf(g_vt, t.a * t.b, 2.0f * g_detl * t.e);
    gt.gr = 0.0f;
    sigma_bwd(t, gS11, gS12, gS22, gt);
  }
  return out;
}

// ------------------------------------------------------------------ one pair
// Returns L_i; fills gpred[7] (and gtgt[7] if GT) with f * dL_i/d(row).  ia2 = 1 / alpha^2 (uniform: formed once on the
// host, gd3d_inv_alpha2, instead of a v_mul + v_rcp in every lane).
template <int LOSS, int FUN, bool FLAG, bool GT>
GD_DEV float pair_loss(const float (&pv)[7], const float (&tv)[7], const float (&c)[3], float alpha, float ia2,
                       float tau, float f, float (&gpred)[7], float (&gtgt)[7]) {
  Box p, t;
  Adj gp, gt;
  constexpr bool SIGMA = LOSS == GD3D_KFIOU3D;   // the only loss left that wants the covariance entries themselves
  box_make<SIGMA>(pv, c, p);
  box_make<SIGMA>(tv, c, t);
  Offs o;
  Geo G;
  if (LOSS != GD3D_KFIOU3D) center_delta(pv, tv, c, o);
  if (LOSS != GD3D_GWD3D && LOSS != GD3D_KFIOU3D) geo_make(o, pv[6], tv[6], G);
  float out;
  if (LOSS == GD3D_GWD3D) out = gwd<FUN, FLAG, GT>(p, t, o, pv[6], tv[6], alpha, tau, gp, gt);
  else if (LOSS == GD3D_KLD3D) out = kld<FUN, FLAG, GT>(p, t, G, ia2, tau, gp, gt);
  else if (LOSS == GD3D_BD3D) out = bd<FUN, FLAG, GT>(p, t, G, ia2, tau, gp, gt);
  else if (LOSS == GD3D_JD3D) out = jd<FUN, FLAG, GT>(p, t, G, ia2, tau, gp, gt);
  else if (LOSS == GD3D_KLD3D_SYMMAX) out = sym<FUN, FLAG, GT, true>(p, t, G, ia2, tau, gp, gt);
  else if (LOSS == GD3D_KLD3D_SYMMIN) out = sym<FUN, FLAG, GT, false>(p, t, G, ia2, tau, gp, gt);
  else {
    const bool dim_nan = (pv[3] != pv[3]) || (pv[4] != pv[4]) || (pv[5] != pv[5]) || (tv[3] != tv[3]) ||
                         (tv[4] != tv[4]) || (tv[5] != tv[5]);
    out = kfiou<FUN, GT>(p, t, dim_nan ? __builtin_nanf("") : 0.0f, gp, gt);
  }
  box_grad(p, gp, c, f, gpred);
  if (GT) box_grad(t, gt, c, f, gtgt);
  return out;
}

}  // namespace gd3d
