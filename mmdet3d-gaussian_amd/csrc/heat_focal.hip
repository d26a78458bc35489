// heat_focal.hip — the heat-map classification loss of the CenterPoint heads, forward + gradient in one pass, for gfx950
// (include/gd3d.h, ABI 4): the remaining term of CenterGDHead.loss next to the regression losses of gd3d_center_head_loss.
//
// Reference, per task (/root/reference/mmdet3d_gaussian/models/dense_heads/gd_centerpoint_head.py:403-411):
//     heatmap  = clip_sigmoid(preds_dict[0]['heatmap'])                     mmdet3d: clamp(sigmoid(x), 1e-4, 1 - 1e-4)
//     num_pos  = heatmaps[task_id].eq(1).float().sum().item()               (a host sync per task)
//     loss     = self.loss_cls(heatmap, heatmaps[task_id], avg_factor=max(num_pos, 1))
// with loss_cls = mmdet's GaussianFocalLoss (third party, absent: restated from the published 2.x text):
//     pos = -log(p + 1e-12) * (1 - p)^alpha * [t == 1],   neg = -log(1 - p + 1e-12) * p^alpha * (1 - t)^gamma,
//     loss = loss_weight * sum(pos + neg) / avg_factor                       (alpha 2, gamma 4 by default)
// i.e. ~20 elementwise launches forward, as many backward, and a sync per task, over B x C x H x W cells per task.
//
// Here, for ALL tasks: focal_kernel reads logit and target once and writes the raw gradient d(sum)/d(logit) (12 bytes per
// cell; at 10.5 M cells the pass runs at 4 TB/s, bound by its exp / log / reciprocal chain rather than by memory, and at the
// reference's batch sizes it is launch-bound) plus one (loss sum, positive count) partial per workgroup; focal_finish_kernel (one workgroup per task)
// adds the partials in a fixed order in fp64 — no float atomics — and leaves loss = w * sum / max(num_pos, 1) and the factor
// w / max(num_pos, 1) on the device; the autograd backward is one in-place scaling of the raw gradient by upstream * factor.
// Nothing is read back.  d clamp / d sigmoid is 1 exactly where torch.clamp passes its gradient (lo <= s <= hi).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/gd3d_extras.h"

namespace hfocal {

constexpr int T = 256;
constexpr int PER_THREAD = 8;                 // two 16-byte loads of each operand per thread (16 per thread, all issued up
                                              // front, nontemporal: measured slower, 31 -> 36 us at 10.5 M cells)
constexpr int TILE = T * PER_THREAD;
constexpr int MAXT = GD3D_HEAT_FOCAL_MAX_TASKS;

struct Task {
  const float* logits;
  const float* target;
  float* grad;           // nullable
  long long n;
  int block0;            // first workgroup of this task in the grid
  int blocks;
};

struct Args {
  Task task[MAXT];
  int Tn;
  float alpha, gamma, clip, log_eps;
  float* partial;        // (total blocks, 2): loss sum, positive count
};

__device__ __forceinline__ float powa(float x, float e) {
  if (e == 2.0f) return x * x;
  if (e == 4.0f) {
    const float x2 = x * x;
    return x2 * x2;
  }
  if (e == 1.0f) return x;
  return powf(x, e);
}

// one cell: loss and d loss / d logit.  Hardware exp / log / reciprocal (1 ulp each: the loss is graded at 1e-5) and the
// positive-cell terms behind a branch (a few dozen cells per map): at batch 64 the pass is otherwise bound by its own
// arithmetic (39 us for 126 MB) instead of by the memory system.
__device__ __forceinline__ void cell(float x, float t, const Args& a, float& loss, float& g, float& pos) {
  const float s = __frcp_rn(1.0f + __expf(-x));
  const float lo = a.clip, hi = 1.0f - a.clip;
  const float p = fminf(fmaxf(s, lo), hi);
  const bool pass = s >= lo && s <= hi;                 // where clamp hands its gradient through
  const float q = 1.0f - p;
  const float lq = __logf(q + a.log_eps);
  const float pa1 = a.alpha == 2.0f ? p : powa(p, a.alpha - 1.0f);
  const float pa = a.alpha == 2.0f ? p * p : powa(p, a.alpha);
  const float nw = powa(1.0f - t, a.gamma);
  loss = -lq * pa * nw;
  // d/dp: neg  [p^a / (q + e) - a p^(a-1) log(q + e)] * nw;   pos  -(q^a) / (p + e) + a q^(a-1) log(p + e)
  float dp = (pa * __frcp_rn(q + a.log_eps) - a.alpha * pa1 * lq) * nw;
  pos = 0.0f;
  if (t == 1.0f) {
    pos = 1.0f;
    const float lp = __logf(p + a.log_eps);
    const float qa1 = a.alpha == 2.0f ? q : powa(q, a.alpha - 1.0f);
    const float qa = a.alpha == 2.0f ? q * q : powa(q, a.alpha);
    loss += -lp * qa;
    dp += -qa * __frcp_rn(p + a.log_eps) + a.alpha * qa1 * lp;
  }
  g = pass ? dp * s * (1.0f - s) : 0.0f;
}

__global__ __launch_bounds__(T) void focal_kernel(const Args a) {
  __shared__ float sl[T / 64], sp[T / 64];
  int t = 0;
  for (int q = 1; q < a.Tn; ++q) t += (int)blockIdx.x >= a.task[q].block0 ? 1 : 0;
  const Task& tk = a.task[t];
  const long long base = (long long)((int)blockIdx.x - tk.block0) * TILE;
  float lsum = 0.0f, psum = 0.0f;
  const bool vec = ((((uintptr_t)tk.logits | (uintptr_t)tk.target | (uintptr_t)tk.grad) & 15) == 0) && base + TILE <= tk.n;
  if (vec) {
    const float4* x4 = (const float4*)(tk.logits + base);
    const float4* t4 = (const float4*)(tk.target + base);
    float4* g4 = (float4*)(tk.grad != nullptr ? tk.grad + base : nullptr);
#pragma unroll
    for (int u = 0; u < PER_THREAD / 4; ++u) {
      const int i = u * T + threadIdx.x;
      const float4 x = x4[i], tt = t4[i];
      float4 g;
      float l, p;
      cell(x.x, tt.x, a, l, g.x, p); lsum += l; psum += p;
      cell(x.y, tt.y, a, l, g.y, p); lsum += l; psum += p;
      cell(x.z, tt.z, a, l, g.z, p); lsum += l; psum += p;
      cell(x.w, tt.w, a, l, g.w, p); lsum += l; psum += p;
      if (g4 != nullptr) g4[i] = g;
    }
  } else {
    for (int u = 0; u < PER_THREAD; ++u) {
      const long long i = base + (long long)u * T + threadIdx.x;
      if (i < tk.n) {
        float l, g, p;
        cell(tk.logits[i], tk.target[i], a, l, g, p);
        lsum += l;
        psum += p;
        if (tk.grad != nullptr) tk.grad[i] = g;
      }
    }
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    lsum += __shfl_down(lsum, off, 64);
    psum += __shfl_down(psum, off, 64);
  }
  if ((threadIdx.x & 63) == 0) {
    sl[threadIdx.x >> 6] = lsum;
    sp[threadIdx.x >> 6] = psum;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    a.partial[(size_t)blockIdx.x * 2] = (sl[0] + sl[1]) + (sl[2] + sl[3]);
    a.partial[(size_t)blockIdx.x * 2 + 1] = (sp[0] + sp[1]) + (sp[2] + sp[3]);
  }
}

struct FinArgs {
  int block0[MAXT + 1];
  const float* partial;
  float loss_weight;
  float* losses;      // (Tn)
  float* factor;      // (Tn): loss_weight / max(num_pos, 1)
  float* num_pos;     // (Tn)
};

__global__ __launch_bounds__(T) void focal_finish_kernel(const FinArgs a) {
  __shared__ double sl[T], sp[T];
  const int t = blockIdx.x;
  const int b0 = a.block0[t], b1 = a.block0[t + 1];
  double l = 0.0, p = 0.0;
  for (int i = b0 + threadIdx.x; i < b1; i += T) {          // fixed assignment, fixed order: deterministic
    l += (double)a.partial[(size_t)i * 2];
    p += (double)a.partial[(size_t)i * 2 + 1];
  }
  sl[threadIdx.x] = l;
  sp[threadIdx.x] = p;
  __syncthreads();
  for (int off = T / 2; off >= 1; off >>= 1) {
    if ((int)threadIdx.x < off) {
      sl[threadIdx.x] += sl[threadIdx.x + off];
      sp[threadIdx.x] += sp[threadIdx.x + off];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const double avg = sp[0] > 1.0 ? sp[0] : 1.0;
    a.losses[t] = (float)((double)a.loss_weight * sl[0] / avg);
    a.factor[t] = (float)((double)a.loss_weight / avg);
    a.num_pos[t] = (float)sp[0];
  }
}

struct ScaleArgs {
  Task task[MAXT];
  int Tn;
  const float* factor;     // (Tn)
  const float* upstream;   // (Tn) gradient of the caller's scalar wrt each task's loss
};

__global__ __launch_bounds__(T) void focal_scale_kernel(const ScaleArgs a) {
  int t = 0;
  for (int q = 1; q < a.Tn; ++q) t += (int)blockIdx.x >= a.task[q].block0 ? 1 : 0;
  const Task& tk = a.task[t];
  if (tk.grad == nullptr) return;
  const float f = a.factor[t] * a.upstream[t];
  const long long base = (long long)((int)blockIdx.x - tk.block0) * TILE;
  for (int u = 0; u < PER_THREAD; ++u) {
    const long long i = base + (long long)u * T + threadIdx.x;
    if (i < tk.n) tk.grad[i] *= f;
  }
}

static int fill(const gd3d_heat_focal_task* tasks, int32_t n, Task* out, int* total) {
  int blocks = 0;
  for (int t = 0; t < n; ++t) {
    if (tasks[t].n < 0 || (tasks[t].n > 0 && (tasks[t].logits == nullptr || tasks[t].target == nullptr))) return GD3D_E_BADARG;
    const long long nb = (tasks[t].n + TILE - 1) / TILE;
    if (nb + blocks > 0x7fffffffLL) return GD3D_E_TOOLARGE;
    out[t].logits = tasks[t].logits;
    out[t].target = tasks[t].target;
    out[t].grad = tasks[t].grad;
    out[t].n = tasks[t].n;
    out[t].block0 = blocks;
    out[t].blocks = (int)nb;
    blocks += (int)nb;
  }
  *total = blocks;
  return 0;
}

}  // namespace hfocal

using namespace hfocal;

extern "C" {

size_t gd3d_heat_focal_workspace_bytes(const gd3d_heat_focal_task* tasks, int32_t num_tasks) {
  if (tasks == nullptr || num_tasks < 1 || num_tasks > MAXT) return 256;
  Task tk[MAXT];
  int blocks = 0;
  if (fill(tasks, num_tasks, tk, &blocks) != 0) return 256;
  return (((size_t)blocks * 2 * sizeof(float)) + 255) & ~(size_t)255;
}

int gd3d_heat_focal_loss(const gd3d_heat_focal_task* tasks, int32_t num_tasks, float alpha, float gamma, float clip_eps,
                         float log_eps, float loss_weight, float* losses, float* factor, float* num_pos, void* workspace,
                         void* stream) {
  if (tasks == nullptr || num_tasks < 1 || num_tasks > MAXT || losses == nullptr || factor == nullptr || num_pos == nullptr ||
      workspace == nullptr)
    return GD3D_E_BADARG;
  Args a = {};
  int blocks = 0;
  const int rc = fill(tasks, num_tasks, a.task, &blocks);
  if (rc != 0) return rc;
  a.Tn = num_tasks;
  a.alpha = alpha;
  a.gamma = gamma;
  a.clip = clip_eps;
  a.log_eps = log_eps;
  a.partial = (float*)workspace;
  hipStream_t s = (hipStream_t)stream;
  if (blocks > 0) hipLaunchKernelGGL(focal_kernel, dim3((unsigned)blocks), dim3(T), 0, s, a);
  FinArgs f = {};
  for (int t = 0; t < num_tasks; ++t) f.block0[t] = a.task[t].block0;
  f.block0[num_tasks] = blocks;
  f.partial = a.partial;
  f.loss_weight = loss_weight;
  f.losses = losses;
  f.factor = factor;
  f.num_pos = num_pos;
  hipLaunchKernelGGL(focal_finish_kernel, dim3((unsigned)num_tasks), dim3(T), 0, s, f);
  return (int)hipGetLastError();
}

int gd3d_heat_focal_scale(const gd3d_heat_focal_task* tasks, int32_t num_tasks, const float* factor, const float* upstream,
                          void* stream) {
  if (tasks == nullptr || num_tasks < 1 || num_tasks > MAXT || factor == nullptr || upstream == nullptr) return GD3D_E_BADARG;
  ScaleArgs a = {};
  int blocks = 0;
  const int rc = fill(tasks, num_tasks, a.task, &blocks);
  if (rc != 0) return rc;
  a.Tn = num_tasks;
  a.factor = factor;
  a.upstream = upstream;
  if (blocks > 0) hipLaunchKernelGGL(focal_scale_kernel, dim3((unsigned)blocks), dim3(T), 0, (hipStream_t)stream, a);
  return (int)hipGetLastError();
}

}  // extern "C"
