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
#include <hip/hip_runtime.h>

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
  s = (n & 2) ? -sv : sv;
  c = ((n + 1) & 2) ? -cv : cv;
}

GD_DEV float flog2(float x) { return __builtin_amdgcn_logf(x); }     // v_log_f32 (base 2), 1 ulp
constexpr float LN2 = 0.6931471805599453f;

// log1p with the (1+d) rounding error folded back in; *inv_u returns 1/(1+d) (= d log1p / dd) from the same v_rcp
GD_DEV float log1p_f(float d, float& inv_u) {
  const float u = 1.0f + d;
  inv_u = frcp(u);
  const float lost = d - (u - 1.0f);                 // d = +inf: inf - inf; torch.log1p(inf) = inf, so no correction there
  const float corr = (inv_u == 0.0f) ? 0.0f : lost * inv_u;
  return fmaf(flog2(u), LN2, corr);
}

// ------------------------------------------------------------------ box -> Gaussian
struct Box {
  float X, Y, Z;     // gravity centre (UNCLAMPED dims, ref :12)
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
  o.X = fmaf(c[0], v[3], v[0]);
  o.Y = fmaf(c[1], v[4], v[1]);
  o.Z = fmaf(c[2], v[5], v[2]);
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
GD_DEV float gwd(const Box& p, const Box& t, float yaw_p, float yaw_t, float alpha, float tau, Adj& gp, Adj& gt) {
  const float dX = p.X - t.X, dY = p.Y - t.Y, dZ = p.Z - t.Z;
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

// ------------------------------------------------------------------ kld3d core (ref :109-137)
struct KldI {
  float iap, ibp, iep, iA, iB, iE, P11, P12, P22, dX, dY, dZ;
};

GD_DEV float kld_fwd(const Box& p, const Box& t, float ia2, KldI& k) {
  k.iap = frcp(p.a);
  k.ibp = frcp(p.b);
  k.iep = frcp(p.e);
  k.iA = k.iap * k.iap;
  k.iB = k.ibp * k.ibp;
  k.iE = k.iep * k.iep;
  rotdiag(k.iA, k.iB, p.co, p.si, k.P11, k.P12, k.P22);
  k.dX = p.X - t.X;
  k.dY = p.Y - t.Y;
  k.dZ = p.Z - t.Z;
  const float quad = fmaf(k.dX * k.dX, k.P11, fmaf(2.0f * k.dX * k.dY, k.P12, k.dY * k.dY * k.P22));
  const float xyz = 0.5f * fmaf(k.dZ * k.dZ, k.iE, quad);
  const float tr = fmaf(k.P11, t.S11, fmaf(2.0f * k.P12, t.S12, k.P22 * t.S22));
  float whlr = 0.5f * fmaf(k.iE, t.e * t.e, tr);
  // (ln ap + ln bp + ln ep) - (ln at + ln bt + ln et) as one log of a ratio would overflow for
  // clamped dims; keep two logs of products (each product is within fp32 range: >= 1.25e-22)
  const float lp = flog2(p.a * p.b * p.e);
  const float lt = flog2(t.a * t.b * t.e);
  whlr = fmaf(lp - lt, LN2, whlr) - 1.5f;
  return fmaf(xyz, ia2, whlr);
}

// accumulates into gp (always) and gt (if GT) with upstream g
template <bool GP, bool GT>
GD_DEV void kld_bwd(const Box& p, const Box& t, float ia2, const KldI& k, float g, Adj& gp, Adj& gt) {
  const float g_xyz = g * ia2, g_w = g;
  const float g_quad = 0.5f * g_xyz;
  const float g_dX = 2.0f * g_quad * fmaf(k.dX, k.P11, k.dY * k.P12);
  const float g_dY = 2.0f * g_quad * fmaf(k.dX, k.P12, k.dY * k.P22);
  const float g_dZ = g_xyz * k.dZ * k.iE;
  if (GP) {
    const float g_iE = 0.5f * fmaf(g_xyz, k.dZ * k.dZ, g_w * (t.e * t.e));
    gp.gX += g_dX;
    gp.gY += g_dY;
    gp.gZ += g_dZ;
    const float gP11 = fmaf(g_quad * k.dX, k.dX, 0.5f * g_w * t.S11);
    const float gP12 = fmaf(2.0f * g_quad * k.dX, k.dY, g_w * t.S12);
    const float gP22 = fmaf(g_quad * k.dY, k.dY, 0.5f * g_w * t.S22);
    float g_iA, g_iB, g_r;
    rotdiag_bwd(k.iA, k.iB, p.co, p.si, k.P12, gP11, gP12, gP22, g_iA, g_iB, g_r);
    gp.gr += g_r;
    gp.ga += fmaf(g_iA, -2.0f * k.iA * k.iap, g_w * k.iap);
    gp.gb += fmaf(g_iB, -2.0f * k.iB * k.ibp, g_w * k.ibp);
    gp.ge += fmaf(g_iE, -2.0f * k.iE * k.iep, g_w * k.iep);
  }
  if (GT) {
    gt.gX -= g_dX;
    gt.gY -= g_dY;
    gt.gZ -= g_dZ;
    sigma_bwd(t, 0.5f * g_w * k.P11, g_w * k.P12, 0.5f * g_w * k.P22, gt);
    gt.ga -= g_w * frcp(t.a);
    gt.gb -= g_w * frcp(t.b);
    gt.ge += g_w * fmaf(k.iE, t.e, -frcp(t.e));
  }
}

template <int FUN, bool SQRT, bool GT>
GD_DEV float kld(const Box& p, const Box& t, float alpha, float tau, Adj& gp, Adj& gt) {
  KldI k;
  const float ia2 = frcp(alpha * alpha);
  float d = kld_fwd(p, t, ia2, k), ds = 1.0f;
  if (SQRT) d = sqrt0(d, ds);
  float dpost;
  const float out = post<FUN>(d, tau, dpost);
  adj_zero(gp);
  adj_zero(gt);
  kld_bwd<true, GT>(p, t, ia2, k, dpost * ds, gp, gt);
  return out;
}

// ------------------------------------------------------------------ jd3d (ref :189-198)
template <int FUN, bool SQRT, bool GT>
GD_DEV float jd(const Box& p, const Box& t, float alpha, float tau, Adj& gp, Adj& gt) {
  KldI k1, k2;
  const float ia2 = frcp(alpha * alpha);
  float v = kld_fwd(p, t, ia2, k1);
  v = v + kld_fwd(t, p, ia2, k2);
  v = v * 0.5f;
  float ds = 1.0f;
  if (SQRT) v = sqrt0(v, ds);
  float dpost;
  const float out = post<FUN>(v, tau, dpost);
  const float g = dpost * ds * 0.5f;
  adj_zero(gp);
  adj_zero(gt);
  kld_bwd<true, GT>(p, t, ia2, k1, g, gp, gt);
  kld_bwd<GT, true>(t, p, ia2, k2, g, gt, gp);
  return out;
}

// ------------------------------------------------------------------ symmax / symmin (ref :201-224)
template <int FUN, bool SQRT, bool GT, bool WANT_MAX>
GD_DEV float sym(const Box& p, const Box& t, float alpha, float tau, Adj& gp, Adj& gt) {
  KldI k1, k2;
  const float ia2 = frcp(alpha * alpha);
  float v1 = kld_fwd(p, t, ia2, k1), v2 = kld_fwd(t, p, ia2, k2);
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
  if (f1 != 0.0f) kld_bwd<true, GT>(p, t, ia2, k1, dpost * f1 * ds1, gp, gt);
  if (f2 != 0.0f) kld_bwd<GT, true>(t, p, ia2, k2, dpost * f2 * ds2, gt, gp);
  return out;
}

// ------------------------------------------------------------------ bd3d (ref :144-186)
template <int FUN, bool SQRT, bool GT>
GD_DEV float bd(const Box& p, const Box& t, float alpha, float tau, Adj& gp, Adj& gt) {
  const float ia2 = frcp(alpha * alpha);
  const float S11 = 0.5f * (p.S11 + t.S11), S12 = 0.5f * (p.S12 + t.S12), S22 = 0.5f * (p.S22 + t.S22);
  const float Ep = p.e * p.e, Et = t.e * t.e, Sl = 0.5f * (Ep + Et);
  const float det_raw = fmaf(S11, S22, -S12 * S12);
  // clamp(min=1e-7): a NaN det_raw can only come from NaN inputs, which already poison dX/dY/dZ -> v_max is enough
  const float mdet = det_raw >= 1e-7f ? 1.0f : 0.0f;
  const float det = __builtin_fmaxf(det_raw, 1e-7f);
  const float idet = frcp(det);
  const float I11 = S22 * idet, I12 = -S12 * idet, I22 = S11 * idet;
  const float dX = p.X - t.X, dY = p.Y - t.Y, dZ = p.Z - t.Z;
  const float quad = fmaf(dX * dX, I11, fmaf(2.0f * dX * dY, I12, dY * dY * I22));
  const float iSl = frcp(Sl);
  const float xyz = 0.125f * fmaf(dZ * dZ, iSl, quad);
  // 0.5(ln det + ln Sl) - 0.25(ln Ap + ln Bp + ln Ep) - 0.25(ln At + ln Bt + ln Et)
  //   = 0.5 (ln det + ln Sl) - 0.5 (ln(ap bp ep) + ln(at bt et))
  const float whlr = (0.5f * LN2) * ((flog2(det) + flog2(Sl)) - (flog2(p.a * p.b * p.e) + flog2(t.a * t.b * t.e)));
  float d = fmaf(xyz, ia2, whlr), ds = 1.0f;
  if (SQRT) d = sqrt0(d, ds);
  float dpost;
  const float out = post<FUN>(d, tau, dpost);

  const float g = dpost * ds, g_xyz = g * ia2, g_w = g;
  const float g_quad = 0.125f * g_xyz;
  const float g_Sl = fmaf(-0.125f * g_xyz * dZ * dZ, iSl * iSl, 0.5f * g_w * iSl);
  const float g_dX = 2.0f * g_quad * fmaf(dX, I11, dY * I12);
  const float g_dY = 2.0f * g_quad * fmaf(dX, I12, dY * I22);
  const float g_dZ = 0.25f * g_xyz * dZ * iSl;
  const float gI11 = g_quad * dX * dX, gI12 = 2.0f * g_quad * dX * dY, gI22 = g_quad * dY * dY;
  const float g_idet = fmaf(gI11, S22, fmaf(-gI12, S12, gI22 * S11));
  const float g_det = fmaf(-g_idet * idet, idet, 0.5f * g_w * idet) * mdet;
  const float gS11 = 0.5f * fmaf(gI22, idet, g_det * S22);
  const float gS22 = 0.5f * fmaf(gI11, idet, g_det * S11);
  const float gS12 = 0.5f * fmaf(-gI12, idet, -2.0f * g_det * S12);
  gp.gX = g_dX;
  gp.gY = g_dY;
  gp.gZ = g_dZ;
  gp.ga = -0.5f * g_w * frcp(p.a);
  gp.gb = -0.5f * g_w * frcp(p.b);
  gp.ge = fmaf(g_Sl, p.e, -0.5f * g_w * frcp(p.e));
  gp.gr = 0.0f;
  sigma_bwd(p, gS11, gS12, gS22, gp);
  if (GT) {
    gt.gX = -g_dX;
    gt.gY = -g_dY;
    gt.gZ = -g_dZ;
    gt.ga = -0.5f * g_w * frcp(t.a);
    gt.gb = -0.5f * g_w * frcp(t.b);
    gt.ge = fmaf(g_Sl, t.e, -0.5f * g_w * frcp(t.e));
    gt.gr = 0.0f;
    sigma_bwd(t, gS11, gS12, gS22, gt);
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
// Returns L_i; fills gpred[7] (and gtgt[7] if GT) with f * dL_i/d(row).
template <int LOSS, int FUN, bool FLAG, bool GT>
GD_DEV float pair_loss(const float (&pv)[7], const float (&tv)[7], const float (&c)[3], float alpha,
                       float tau, float f, float (&gpred)[7], float (&gtgt)[7]) {
  Box p, t;
  Adj gp, gt;
  box_make<LOSS != GD3D_GWD3D>(pv, c, p);
  box_make<LOSS != GD3D_GWD3D>(tv, c, t);
  float out;
  if (LOSS == GD3D_GWD3D) out = gwd<FUN, FLAG, GT>(p, t, pv[6], tv[6], alpha, tau, gp, gt);
  else if (LOSS == GD3D_KLD3D) out = kld<FUN, FLAG, GT>(p, t, alpha, tau, gp, gt);
  else if (LOSS == GD3D_BD3D) out = bd<FUN, FLAG, GT>(p, t, alpha, tau, gp, gt);
  else if (LOSS == GD3D_JD3D) out = jd<FUN, FLAG, GT>(p, t, alpha, tau, gp, gt);
  else if (LOSS == GD3D_KLD3D_SYMMAX) out = sym<FUN, FLAG, GT, true>(p, t, alpha, tau, gp, gt);
  else if (LOSS == GD3D_KLD3D_SYMMIN) out = sym<FUN, FLAG, GT, false>(p, t, alpha, tau, gp, gt);
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
