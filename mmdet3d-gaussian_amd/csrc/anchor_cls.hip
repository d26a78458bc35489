// anchor_cls.hip — the classification (sigmoid focal) and direction (2-way cross entropy) terms of the anchor heads' loss,
// forward + gradient in ONE pass over the head's NCHW maps, for gfx950 (include/gd3d.h, ABI 4).
//
// Reference: GDAnchor3DHead.loss_single
//   /root/reference/mmdet3d_gaussian/models/dense_heads/gd_anchor3d_head.py:84-92   cls_score.permute(0,2,3,1).reshape(-1, C) (a copy of
//       all class maps) -> self.loss_cls(cls_score, labels, label_weights, avg_factor=num_total_samples)
//   :143-149   self.loss_dir(pos_dir_cls_preds, pos_dir_targets, pos_dir_weights, avg_factor=num_total_samples) on the positives
//       (labels in [0, num_classes), :101-103), gathered from the permuted direction maps
// with mmdet's FocalLoss(use_sigmoid=True) and CrossEntropyLoss(use_sigmoid=False) (third party, absent: restated):
//   t = [label == c];  p = sigmoid(x);  loss = BCEwithLogits(x, t) * (alpha t + (1 - alpha)(1 - t)) * ((1 - p) t + p (1 - t))^gamma
//   * label_weight;   dir: (logsumexp(d) - d[target]) * dir_weight;   both summed and scaled by loss_weight / avg_factor.
// ~25 elementwise launches forward and as many backward over B x A x C x H x W logits (5.8 M at KITTI geometry, batch 6), plus
// the permuted copies.  Here: a workgroup owns a tile of cells and all their anchors; every class logit is read where the head left it and its
// gradient written in the same layout (8 bytes per logit + the anchor's label and weight); per-workgroup (cls, dir) partial
// sums are added in a fixed order in fp64 by a one-workgroup finish kernel.  No float atomics, no host sync.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/gd3d_extras.h"

namespace acls {

constexpr int T = 256;      // threads of a workgroup
constexpr int TC = 128;     // cells of a workgroup's tile: TC * A (cell, anchor) items, T of them in flight at a time
constexpr int MAX_A = 32;   // anchors per cell the label stage in LDS is sized for (TC * A * 8 bytes)

struct Args {
  const float* cls;          // (B, A*C, H, W)
  const float* dir;          // (B, A*2, H, W) nullable
  const long long* labels;   // (B, N) N = H*W*A, anchor order (h, w, a)
  const float* label_w;      // (B, N)
  const long long* dir_t;    // (B, N) nullable with dir
  const float* dir_w;        // (B, N)
  float* gcls;               // nullable
  float* gdir;               // nullable
  float* partial;            // (blocks, 2)
  int A, C, HW;
  float gamma, alpha, cls_scale, dir_scale;
  const float* avg_dev;      // nullable: the two scales are w_cls / *avg_dev, w_dir / *avg_dev (divided in double, rounded once)
  double w_cls, w_dir;
};

__device__ __forceinline__ void scales_of(const Args& a, float& cs, float& ds) {
  cs = a.cls_scale;
  ds = a.dir_scale;
  if (a.avg_dev != nullptr) {
    const double avg = (double)*a.avg_dev;
    cs = (float)(a.w_cls / avg);
    ds = (float)(a.w_dir / avg);
  }
}

__device__ __forceinline__ float powg(float x, float g) {
  if (g == 2.0f) return x * x;
  if (g == 1.0f) return x;
  if (g == 0.0f) return 1.0f;
  return powf(x, g);
}

// log(1 + e) for e in (0, 1] in ~10 instructions: the alternating series below 1/16 (next term e^6/6: 1.6e-7 relative),
// the hardware log of the rounded sum above it (the rounding of 1 + e is 6e-8 against a result >= 0.06: 1e-6 relative).
// Both sides are evaluated and one selected: no divergence.
__device__ __forceinline__ float log1p_unit(float e) {
  const float series = e * (1.0f + e * (-0.5f + e * (1.0f / 3.0f + e * (-0.25f + e * 0.2f))));
  const float direct = __logf(1.0f + e);
  return e < 0.0625f ? series : direct;
}

// One class logit of one anchor: loss and d loss / d logit of mmdet's sigmoid focal loss, from ONE exp, ONE reciprocal and ONE
// log1p (the kernel is bound by VALU issue, not by HBM, with the library's expf / log1pf / IEEE division: ~200 instructions per
// logit against ~40 here, measured 36 us -> see DESIGN.md 3.11):
// e = exp(-|x|);  sigmoid and its complement without cancellation on either side;  -log p = softplus(-x), -log q = softplus(x).
__device__ __forceinline__ void focal(float x, bool is_t, float gamma, float alpha, float& l, float& g) {
  const float e = __expf(-fabsf(x));
  const float r = __frcp_rn(1.0f + e);
  const float big = r, small = e * r;
  const float p = x >= 0.0f ? big : small, q = x >= 0.0f ? small : big;
  const float l1p = log1p_unit(e);
  // with s = pt = (1 - p) t + p (1 - t), o = 1 - s, x' = x for t = 0 and -x for t = 1 (so that s = sigmoid(x')):
  //   loss = coef s^gamma softplus(x'),   d loss / d x' = coef s^gamma (s + gamma o softplus(x')),   d x' / d x = -1 for t = 1
  const float s_ = is_t ? q : p, o_ = is_t ? p : q;
  const float coef = is_t ? alpha : 1.0f - alpha;
  const float nl = fmaxf(is_t ? -x : x, 0.0f) + l1p;
  const float sg = coef * powg(s_, gamma);
  l = sg * nl;
  const float gm = sg * (s_ + gamma * o_ * nl);
  g = is_t ? -gm : gm;
}

// A workgroup owns TC consecutive cells of one sample and all their anchors.  The anchors' labels and weights are one
// contiguous range of the (h, w, a)-ordered arrays: staged once, coalesced, in LDS (labels outside [0, C) become -1: no class
// matches them and they are no positives, as in the reference's one-hot and its positive mask).  Then thread t walks the items
// (anchor plane an, cell cl) = (i / TC, i % TC), i = t, t + T, ...: a wave reads 64 consecutive cells of one plane (256 B) per
// class, every load of an item independent of the others.
template <int CT>
__global__ __launch_bounds__(T) void cls_dir_kernel(const Args a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  __shared__ float sc[T / 64], sd[T / 64];
  const int A = a.A, C = CT ? CT : a.C, HW = a.HW;
  int* s_lab = (int*)smem;
  float* s_w = (float*)(smem + (size_t)TC * A * sizeof(int));
  const int b = blockIdx.y;
  const int cell0 = blockIdx.x * TC;
  const int ncell = min(TC, HW - cell0);
  const size_t base = (size_t)b * HW * A + (size_t)cell0 * A;
  for (int i = threadIdx.x; i < ncell * A; i += T) {
    const long long lab = a.labels[base + i];
    s_lab[i] = (lab >= 0 && lab < C) ? (int)lab : -1;
    s_w[i] = a.label_w[base + i];
  }
  __syncthreads();
  float lc = 0.0f, ld = 0.0f;
  const float gamma = a.gamma, alpha = a.alpha;
  float cls_scale, dir_scale;
  scales_of(a, cls_scale, dir_scale);
#pragma unroll 2
  for (int i = threadIdx.x; i < TC * A; i += T) {
    const int an = i / TC, cl = i % TC;
    if (cl >= ncell) continue;
    const int lab = s_lab[cl * A + an];
    const float w = s_w[cl * A + an];
    const size_t o = ((size_t)b * A * C + (size_t)an * C) * HW + cell0 + cl;
    if (CT) {
      float x[CT ? CT : 1];
#pragma unroll
      for (int c = 0; c < CT; ++c) x[c] = a.cls[o + (size_t)c * HW];
#pragma unroll
      for (int c = 0; c < CT; ++c) {
        float l, g;
        focal(x[c], lab == c, gamma, alpha, l, g);
        lc += l * w;
        if (a.gcls != nullptr) a.gcls[o + (size_t)c * HW] = g * w * cls_scale;
      }
    } else {
      for (int c = 0; c < C; ++c) {
        float l, g;
        focal(a.cls[o + (size_t)c * HW], lab == c, gamma, alpha, l, g);
        lc += l * w;
        if (a.gcls != nullptr) a.gcls[o + (size_t)c * HW] = g * w * cls_scale;
      }
    }
    if (a.dir != nullptr) {
      const size_t o0 = ((size_t)b * A * 2 + (size_t)an * 2) * HW + cell0 + cl, o1 = o0 + HW;
      float g0 = 0.0f, g1 = 0.0f;
      if (lab >= 0) {                    // a positive anchor (:101-103)
        const size_t n = base + (size_t)cl * A + an;
        const float d0 = a.dir[o0], d1 = a.dir[o1];
        const long long k = a.dir_t[n];
        const float dw = a.dir_w[n];
        // a direction target outside [0, 2) on a positive anchor: F.cross_entropy raises in the reference (a device-side assert on
        // the GPU); the sync-free analogue here is a NaN direction loss and NaN direction gradients for that anchor — loud in
        // the first step instead of a silently finite "bin 1" 
        const float z = (k == 0 ? d1 - d0 : d0 - d1) + ((k == 0 || k == 1) ? 0.0f : __builtin_nanf(""));   // wrong bin minus right bin
        const float e = expf(-fabsf(z));
        ld += (fmaxf(z, 0.0f) + log1pf(e)) * dw;                 // logsumexp(d) - d[k] = softplus(z)
        const float r = 1.0f / (1.0f + e);
        const float pw = z >= 0.0f ? r : e * r;                  // softmax probability of the wrong bin
        const float gs = pw * dw * dir_scale;
        g0 = k == 0 ? -gs : gs;
        g1 = -g0;
      }
      if (a.gdir != nullptr) {
        a.gdir[o0] = g0;
        a.gdir[o1] = g1;
      }
    }
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    lc += __shfl_down(lc, off, 64);
    ld += __shfl_down(ld, off, 64);
  }
  if ((threadIdx.x & 63) == 0) {
    sc[threadIdx.x >> 6] = lc;
    sd[threadIdx.x >> 6] = ld;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    const size_t blk = (size_t)blockIdx.y * gridDim.x + blockIdx.x;
    a.partial[blk * 2] = (sc[0] + sc[1]) + (sc[2] + sc[3]);
    a.partial[blk * 2 + 1] = (sd[0] + sd[1]) + (sd[2] + sd[3]);
  }
}

__global__ __launch_bounds__(T) void cls_dir_finish_kernel(const float* __restrict__ partial, int blocks, const Args sc,
                                                           float* __restrict__ losses) {
  __shared__ double s0[T], s1[T];
  double a = 0.0, b = 0.0;
  for (int i = threadIdx.x; i < blocks; i += T) {      // fixed assignment and order: deterministic
    a += (double)partial[(size_t)i * 2];
    b += (double)partial[(size_t)i * 2 + 1];
  }
  s0[threadIdx.x] = a;
  s1[threadIdx.x] = b;
  __syncthreads();
  for (int off = T / 2; off >= 1; off >>= 1) {
    if ((int)threadIdx.x < off) {
      s0[threadIdx.x] += s0[threadIdx.x + off];
      s1[threadIdx.x] += s1[threadIdx.x + off];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    float cls_scale, dir_scale;
    scales_of(sc, cls_scale, dir_scale);
    losses[0] = (float)(s0[0] * (double)cls_scale);
    losses[1] = (float)(s1[0] * (double)dir_scale);
  }
}

}  // namespace acls

using namespace acls;

extern "C" {

size_t gd3d_anchor_cls_dir_workspace_bytes(int32_t batch, int32_t height, int32_t width) {
  if (batch < 1 || height < 1 || width < 1) return 256;
  const size_t blocks = (size_t)batch * (((size_t)height * width + TC - 1) / TC);
  return (blocks * 2 * sizeof(float) + 255) & ~(size_t)255;
}

static int cls_dir_impl(const float* cls_score, const float* dir_cls_preds, const int64_t* labels, const float* label_weights,
                        const int64_t* dir_targets, const float* dir_weights, int32_t batch, int32_t num_anchors, int32_t num_classes,
                        int32_t height, int32_t width, float gamma, float alpha, float cls_scale, float dir_scale, const float* avg_dev,
                        double cls_weight, double dir_weight, float* grad_cls, float* grad_dir, float* losses, void* workspace,
                        void* stream) {
  if (batch < 1 || num_anchors < 1 || num_classes < 1 || height < 1 || width < 1) return GD3D_E_BADARG;
  if (num_anchors > MAX_A || batch > 65535) return GD3D_E_TOOLARGE;
  if (cls_score == nullptr || labels == nullptr || label_weights == nullptr || losses == nullptr || workspace == nullptr) return GD3D_E_BADARG;
  if (dir_cls_preds != nullptr && (dir_targets == nullptr || dir_weights == nullptr)) return GD3D_E_BADARG;
  if (dir_cls_preds == nullptr && grad_dir != nullptr) return GD3D_E_BADARG;
  const long long HW = (long long)height * width;
  if (HW * num_anchors * (num_classes > 2 ? num_classes : 2) >= 0x7fffffffLL) return GD3D_E_TOOLARGE;
  Args a;
  a.cls = cls_score;
  a.dir = dir_cls_preds;
  a.labels = (const long long*)labels;
  a.label_w = label_weights;
  a.dir_t = (const long long*)dir_targets;
  a.dir_w = dir_weights;
  a.gcls = grad_cls;
  a.gdir = grad_dir;
  a.partial = (float*)workspace;
  a.A = num_anchors;
  a.C = num_classes;
  a.HW = (int)HW;
  a.gamma = gamma;
  a.alpha = alpha;
  a.cls_scale = cls_scale;
  a.dir_scale = dir_scale;
  a.avg_dev = avg_dev;
  a.w_cls = cls_weight;
  a.w_dir = dir_weight;
  const unsigned gx = (unsigned)((HW + TC - 1) / TC);
  hipStream_t s = (hipStream_t)stream;
  const size_t lds = (size_t)TC * num_anchors * 8;
  const dim3 grid(gx, (unsigned)batch), block(T);
  switch (num_classes) {
    case 1: hipLaunchKernelGGL(cls_dir_kernel<1>, grid, block, lds, s, a); break;
    case 2: hipLaunchKernelGGL(cls_dir_kernel<2>, grid, block, lds, s, a); break;
    case 3: hipLaunchKernelGGL(cls_dir_kernel<3>, grid, block, lds, s, a); break;
    case 4: hipLaunchKernelGGL(cls_dir_kernel<4>, grid, block, lds, s, a); break;
    default: hipLaunchKernelGGL(cls_dir_kernel<0>, grid, block, lds, s, a); break;
  }
  hipLaunchKernelGGL(cls_dir_finish_kernel, dim3(1), dim3(T), 0, s, (const float*)workspace, (int)(gx * (unsigned)batch), a, losses);
  return (int)hipGetLastError();
}

int gd3d_anchor_cls_dir_loss(const float* cls_score, const float* dir_cls_preds, const int64_t* labels, const float* label_weights,
                             const int64_t* dir_targets, const float* dir_weights, int32_t batch, int32_t num_anchors,
                             int32_t num_classes, int32_t height, int32_t width, float gamma, float alpha, float cls_scale,
                             float dir_scale, float* grad_cls, float* grad_dir, float* losses, void* workspace, void* stream) {
  return cls_dir_impl(cls_score, dir_cls_preds, labels, label_weights, dir_targets, dir_weights, batch, num_anchors, num_classes, height, width,
                      gamma, alpha, cls_scale, dir_scale, nullptr, 0.0, 0.0, grad_cls, grad_dir, losses, workspace, stream);
}

int gd3d_anchor_cls_dir_loss_dyn(const float* cls_score, const float* dir_cls_preds, const int64_t* labels, const float* label_weights,
                                 const int64_t* dir_targets, const float* dir_weights, int32_t batch, int32_t num_anchors,
                                 int32_t num_classes, int32_t height, int32_t width, float gamma, float alpha, double cls_weight,
                                 double dir_weight, const float* avg_dev, float* grad_cls, float* grad_dir, float* losses,
                                 void* workspace, void* stream) {
  if (avg_dev == nullptr) return GD3D_E_BADARG;
  return cls_dir_impl(cls_score, dir_cls_preds, labels, label_weights, dir_targets, dir_weights, batch, num_anchors, num_classes, height, width,
                      gamma, alpha, 0.0f, 0.0f, avg_dev, cls_weight, dir_weight, grad_cls, grad_dir, losses, workspace, stream);
}

}  // extern "C"
