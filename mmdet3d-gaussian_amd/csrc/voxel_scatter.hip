// voxel_scatter.hip — dynamic point-to-voxel scatter-reduce (max / mean / sum), forward and backward, for gfx950
// (SURVEY.md §8f-4).  Replaces the reference's atomic kernels
//   /root/reference/mmdet3d_gaussian/ops/voxel/src/scatter_points_cuda.cu:80-179 (feats_reduce_kernel,
//   add_reduce_traceback_grad_kernel, max_reduce_traceback_scatter_idx_kernel, max_reduce_scatter_grad_kernel)
// behind ops/voxel/scatter.py:29-72 (`scatter_reduce`).
//
// CDNA4 design: no float CAS / atomicAdd.  The host groups the points by voxel once per `Scatter`
// (`order` = stable argsort of the point->voxel map, `seg` = segment starts), then
//   forward : a sub-wave of CP = pow2 >= C lanes owns one voxel, lane = channel; it walks the voxel's points in
//             ascending point index (4 independent row loads in flight) and keeps sum / max (+ the arg max point id)
//             in registers; rows are read as contiguous C*4-byte runs.  Deterministic (fixed order), one store per
//             output element.
//   backward: ONE gather pass over the points for all three modes, every output element written exactly once with
//             16-byte accesses: sum/mean = the voxel gradient row by the map; max = the same, masked by
//             argmax[voxel, ch] == point id (the forward's recorded arg max = the point the reference picks: atomicMin
//             over equal-to-max points = smallest point index).  No zero-fill pass.
// HBM-bound: reads N*C*4 + N*4, writes V*C*4 (+ V*C*4 arg max).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "../../include/gd3d.h"

namespace vox {

#ifndef SCATTER_U
#define SCATTER_U 4  // independent row loads in flight per lane (forward)
#endif

template <int REDUCE>
__global__ __launch_bounds__(256) void reduce_kernel(const float* __restrict__ feats, const int* __restrict__ order,
                                                     const int* __restrict__ seg, int c, int cp, long long v,
                                                     float* __restrict__ out, int* __restrict__ argmax) {
  const int lane = threadIdx.x & 63;
  const long long wave = ((long long)blockIdx.x * 256 + threadIdx.x) >> 6;
  const int groups = 64 / cp;  // voxels per wave
  const int sub = lane / cp, ch0 = lane - sub * cp;
  const long long vox = wave * groups + sub;
  if (vox >= v) return;
  const int b = seg[vox], e = seg[vox + 1];
  for (int ch = ch0; ch < c; ch += cp) {  // cp == 64 and c > 64: several channel passes
    float acc = (REDUCE == GD3D_REDUCE_MAX) ? -__builtin_inff() : 0.0f;
    int arg = -1;
    int k = b;
    for (; k + 4 <= e; k += 4) {
      int pid[4];
      float x[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) pid[u] = order[k + u];
#pragma unroll
      for (int u = 0; u < 4; ++u) x[u] = feats[(long long)pid[u] * c + ch];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (REDUCE == GD3D_REDUCE_MAX) {
          if (x[u] > acc) {  // strict: the first (smallest) point index wins ties; NaN never wins (fmaxf semantics)
            acc = x[u];
            arg = pid[u];
          }
        } else {
          acc += x[u];
        }
      }
    }
    for (; k < e; ++k) {
      const int pid = order[k];
      const float x = feats[(long long)pid * c + ch];
      if (REDUCE == GD3D_REDUCE_MAX) {
        if (x > acc) {
          acc = x;
          arg = pid;
        }
      } else {
        acc += x;
      }
    }
    if (REDUCE == GD3D_REDUCE_MEAN) acc = acc / (float)(e - b);
    out[vox * c + ch] = acc;
    if (REDUCE == GD3D_REDUCE_MAX && argmax != nullptr) argmax[vox * c + ch] = arg;
  }
}

// Vectorised forward (c % 4 == 0, 16-byte aligned rows, c <= 256): a lane owns FOUR consecutive channels and a sub-wave
// of LP = pow2 >= c/4 lanes owns one voxel, so a wave walks 64 / LP voxels at once (c = 64: 4 voxels, 16 row loads of
// 256 B in flight per wave instead of 4).  The scalar kernel above is latency-bound: a voxel averages ~9 points, i.e.
// seg -> order -> rows is 3-5 dependent round trips for 2.4 KB, ~0.4 KB in flight per wave (measured 3.7 TB/s, the same
// with physically sorted rows, so not a gather problem).  Same visiting order, same arithmetic -> same bits.
template <int REDUCE, int U>
__global__ __launch_bounds__(256) void reduce_v4_kernel(const float* __restrict__ feats, const int* __restrict__ order,
                                                        const int* __restrict__ seg, int c, int lp, long long v,
                                                        float* __restrict__ out, int* __restrict__ argmax) {
  const int lane = threadIdx.x & 63;
  const long long wave = ((long long)blockIdx.x * 256 + threadIdx.x) >> 6;
  const int groups = 64 / lp;  // voxels per wave
  const int sub = lane / lp, q = lane - sub * lp;
  const long long vox = wave * groups + sub;
  if (vox >= v || q * 4 >= c) return;
  const int b = seg[vox], e = seg[vox + 1];
  const float init = (REDUCE == GD3D_REDUCE_MAX) ? -__builtin_inff() : 0.0f;
  float acc[4] = {init, init, init, init};
  int arg[4] = {-1, -1, -1, -1};
  const float* base = feats + q * 4;
  int k = b;
  for (; k + U <= e; k += U) {
    int pid[U];
    float4 x[U];
#pragma unroll
    for (int u = 0; u < U; ++u) pid[u] = order[k + u];
#pragma unroll
    for (int u = 0; u < U; ++u) x[u] = *reinterpret_cast<const float4*>(base + (long long)pid[u] * c);
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const float xv[4] = {x[u].x, x[u].y, x[u].z, x[u].w};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (REDUCE == GD3D_REDUCE_MAX) {
          if (xv[j] > acc[j]) {  // strict: the first (smallest) point index wins ties; NaN never wins
            acc[j] = xv[j];
            arg[j] = pid[u];
          }
        } else {
          acc[j] += xv[j];
        }
      }
    }
  }
  for (; k < e; ++k) {
    const int pid = order[k];
    const float4 x = *reinterpret_cast<const float4*>(base + (long long)pid * c);
    const float xv[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (REDUCE == GD3D_REDUCE_MAX) {
        if (xv[j] > acc[j]) {
          acc[j] = xv[j];
          arg[j] = pid;
        }
      } else {
        acc[j] += xv[j];
      }
    }
  }
  if (REDUCE == GD3D_REDUCE_MEAN) {
    const float cnt = (float)(e - b);
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] = acc[j] / cnt;
  }
  float4 o;
  o.x = acc[0]; o.y = acc[1]; o.z = acc[2]; o.w = acc[3];
  *reinterpret_cast<float4*>(out + vox * c + q * 4) = o;
  if (REDUCE == GD3D_REDUCE_MAX && argmax != nullptr) {
    int4 a;
    a.x = arg[0]; a.y = arg[1]; a.z = arg[2]; a.w = arg[3];
    *reinterpret_cast<int4*>(argmax + vox * c + q * 4) = a;
  }
}

// Backward as ONE gather pass over the points, every output element written exactly once (no zero-fill pass):
//   sum : grad_feats[i, ch] = map[i] >= 0 ? grad_vox[map[i], ch] : 0
//   mean: ... / count[map[i]]
//   max : ... only where argmax[map[i], ch] == i (the forward's recorded arg max), else 0
// VEC = 4: a thread moves 4 consecutive channels with 16-byte accesses (c % 4 == 0, 16-byte aligned rows).
typedef float v4f_t __attribute__((ext_vector_type(4)));
#ifndef GATHER_GU_SCALAR
#define GATHER_GU_SCALAR 4   // 8 measured no faster (c = 10 sum 39.8 vs 40.5 us) and slower for mean (50.5 vs 45 us)
#endif

template <int MODE, int VEC>
__global__ __launch_bounds__(256) void gather_grad_kernel(const float* __restrict__ gvox, const int* __restrict__ map,
                                                          const int* __restrict__ count, const int* __restrict__ argmax,
                                                          long long n, int c, int shift, unsigned magic,
                                                          float* __restrict__ gfeats) {
  // GU items per thread, loads staged (all map loads, then all row loads, then the stores): map -> row is a dependent
  // chain, one item per thread leaves 1 KB in flight per wave and the kernel latency-bound (3.5 TB/s at c = 64).
  // The gradient is written once and not re-read here: nontemporal stores keep the gathered voxel rows in cache.
  constexpr int GU = VEC == 4 ? 4 : GATHER_GU_SCALAR;   // dword items: twice as many in flight (the chain is latency-bound)
  const int cv = c / VEC;  // lanes per point row; shift = log2(cv) when cv is a power of two, else -1
  const long long total = n * cv;
  long long idx[GU], pt[GU];
  int ch[GU], m[GU];
#pragma unroll
  for (int u = 0; u < GU; ++u) {
    idx[u] = ((long long)blockIdx.x * GU + u) * 256 + threadIdx.x;
    const long long ic = idx[u] < total ? idx[u] : total - 1;
    // index / lanes-per-row: a shift for powers of two; otherwise ONE multiply-high when the host has checked that the
    // index range allows it (magic = floor(2^32 / cv) + 1 is exact below 2^32 / cv), else a 64-bit division.  The division
    // used to be taken for every c that is not a power of two and made the c = 10 kernel VALU-bound (40 us for 88 MB).
    pt[u] = shift >= 0 ? (ic >> shift) : (magic != 0u ? (long long)__umulhi((unsigned)ic, magic) : (long long)((unsigned long long)ic / (unsigned)cv));
    ch[u] = (int)(ic - pt[u] * cv) * VEC;
  }
#pragma unroll
  for (int u = 0; u < GU; ++u) m[u] = map[pt[u]];
  float g[GU][VEC];
  int am[GU][VEC];
  float cnt[GU];
#pragma unroll
  for (int u = 0; u < GU; ++u) {
    const long long src = (long long)(m[u] >= 0 ? m[u] : 0) * c + ch[u];
    if (VEC == 4) {
      const float4 v = *reinterpret_cast<const float4*>(gvox + src);
      g[u][0] = v.x; g[u][1] = v.y; g[u][2] = v.z; g[u][3] = v.w;
    } else {
      g[u][0] = gvox[src];
    }
    if (MODE == GD3D_REDUCE_MEAN) cnt[u] = (float)count[m[u] >= 0 ? m[u] : 0];
    if (MODE == GD3D_REDUCE_MAX) {
      if (VEC == 4) {
        const int4 a4 = *reinterpret_cast<const int4*>(argmax + src);
        am[u][0] = a4.x; am[u][1] = a4.y; am[u][2] = a4.z; am[u][3] = a4.w;
      } else {
        am[u][0] = argmax[src];
      }
    }
  }
#pragma unroll
  for (int u = 0; u < GU; ++u) {
    if (idx[u] >= total) continue;
#pragma unroll
    for (int k = 0; k < VEC; ++k) {
      float x = m[u] >= 0 ? g[u][k] : 0.0f;
      if (MODE == GD3D_REDUCE_MEAN) x = m[u] >= 0 ? x / cnt[u] : 0.0f;
      if (MODE == GD3D_REDUCE_MAX) x = (m[u] >= 0 && am[u][k] == (int)pt[u]) ? x : 0.0f;
      g[u][k] = x;
    }
    float* dst = gfeats + pt[u] * c + ch[u];
    if (VEC == 4) {
      v4f_t o = {g[u][0], g[u][1], g[u][2], g[u][3]};
      __builtin_nontemporal_store(o, reinterpret_cast<v4f_t*>(dst));
    } else {
      __builtin_nontemporal_store(g[u][0], dst);
    }
  }
}

// Backward in VOXEL order (needs the forward's grouping): the sub-wave that owns a voxel loads its gradient row (and
// arg-max row) ONCE and streams it to the voxel's points — 16-byte nontemporal stores, 256 B contiguous per point at
// c = 64.  The map-ordered gather above re-reads a voxel row once per point from beyond the L2 (N*C*4 extra bytes:
// measured 143 us at 2 M x 64, of which only 512 MB are the mandatory writes); this form moves V*C*4 + N*4 + N*C*4.
// Points that belong to no voxel (map = -1) are the prefix order[0 .. seg[0]); the trailing ZERO_BLOCKS workgroups
// zero their rows (seg[0] is read on the device: no host sync to size anything).
constexpr int ZERO_BLOCKS = 64;
template <int MODE, int U>
__global__ __launch_bounds__(256) void spread_grad_v4_kernel(const float* __restrict__ gvox, const int* __restrict__ order,
                                                             const int* __restrict__ seg, const int* __restrict__ argmax,
                                                             int c, int lp, long long v, unsigned vox_blocks,
                                                             float* __restrict__ gfeats) {
  if (blockIdx.x >= vox_blocks) {
    const long long n0 = seg[0];
    const int cq = c >> 2;
    const long long stride = (long long)(gridDim.x - vox_blocks) * 256;
    for (long long idx = (long long)(blockIdx.x - vox_blocks) * 256 + threadIdx.x; idx < n0 * cq; idx += stride) {
      const long long i = idx / cq;
      const int q = (int)(idx - i * cq);
      const v4f_t z = {0.f, 0.f, 0.f, 0.f};
      __builtin_nontemporal_store(z, reinterpret_cast<v4f_t*>(gfeats + (long long)order[i] * c + q * 4));
    }
    return;
  }
  const int lane = threadIdx.x & 63;
  const long long wave = ((long long)blockIdx.x * 256 + threadIdx.x) >> 6;
  const int groups = 64 / lp;
  const int sub = lane / lp, q = lane - sub * lp;
  const long long vox = wave * groups + sub;
  if (vox >= v || q * 4 >= c) return;
  const int b = seg[vox], e = seg[vox + 1];
  const float4 g4 = *reinterpret_cast<const float4*>(gvox + vox * c + q * 4);
  float g[4] = {g4.x, g4.y, g4.z, g4.w};
  int a[4] = {0, 0, 0, 0};
  if (MODE == GD3D_REDUCE_MEAN) {
    const float cnt = (float)(e - b);
#pragma unroll
    for (int j = 0; j < 4; ++j) g[j] = g[j] / cnt;
  }
  if (MODE == GD3D_REDUCE_MAX) {
    const int4 a4 = *reinterpret_cast<const int4*>(argmax + vox * c + q * 4);
    a[0] = a4.x; a[1] = a4.y; a[2] = a4.z; a[3] = a4.w;
  }
  float* base = gfeats + q * 4;
  int k = b;
  for (; k + U <= e; k += U) {
    int pid[U];
#pragma unroll
    for (int u = 0; u < U; ++u) pid[u] = order[k + u];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      v4f_t o = {g[0], g[1], g[2], g[3]};
      if (MODE == GD3D_REDUCE_MAX) {
        o.x = a[0] == pid[u] ? g[0] : 0.0f;
        o.y = a[1] == pid[u] ? g[1] : 0.0f;
        o.z = a[2] == pid[u] ? g[2] : 0.0f;
        o.w = a[3] == pid[u] ? g[3] : 0.0f;
      }
      __builtin_nontemporal_store(o, reinterpret_cast<v4f_t*>(base + (long long)pid[u] * c));
    }
  }
  for (; k < e; ++k) {
    const int pid = order[k];
    v4f_t o = {g[0], g[1], g[2], g[3]};
    if (MODE == GD3D_REDUCE_MAX) {
      o.x = a[0] == pid ? g[0] : 0.0f;
      o.y = a[1] == pid ? g[1] : 0.0f;
      o.z = a[2] == pid ? g[2] : 0.0f;
      o.w = a[3] == pid ? g[3] : 0.0f;
    }
    __builtin_nontemporal_store(o, reinterpret_cast<v4f_t*>(base + (long long)pid * c));
  }
}

// ------------------------------------------------------------------------------------------------------------------
// Rows that are not whole 16-byte vectors (c = 3, 9, 10: the xyz mean and 3x3 covariance reduces of the reference's
// PointVoxelStatsCalculator, models/voxel_encoders/utils.py:56,63-64, and its 10-channel pillar features).
//
// FORWARD.  These rows stay on reduce_kernel above.  Round 4 built the alternative the sub-wave kernel's 640 B in flight per
// wave suggested — a workgroup gathers the rows of 51 consecutive voxels into LDS as one flat dword stream (64 consecutive
// dwords per wave instruction, 8 independent loads per thread) and reduces from LDS in the same visiting order — and
// measured the SAME time (c = 10, 2 M random points -> 214 K voxels: 50.9-55.6 us against 50.2-51.4 us; c = 16: 52-60 us
// against 44-47 us for the 16-byte vector kernel), so it was removed.  The counters say why (profiles/r04_scatter_pmc.txt):
// 2.2-2.6 M L2-miss requests per launch, i.e. every 40-byte row of the 80 MB point array costs 1.1-1.3 fetches of a 128-byte
// line beyond the XCD's 4 MB L2, 280-330 MB of line traffic at 6.5 TB/s — the rate the memory system gives uniformly random
// rows (MI355X_MICROARCH.md, "Indexed rows": 7.4-8.6 TB/s for wide rows from the Infinity Cache).  A gather of randomly
// placed narrow rows is bound by LINES, not by algorithmic bytes; more loads in flight cannot change it.
//
// BACKWARD.  For rows narrower than a line the map-ordered gather (gather_grad_kernel) is the right form: its writes stream
// and the V * c gradient rows it gathers stay in L2.  The voxel-ordered form below exists so that
// vox_scatter_backward_grouped is total over c <= 128; its stores are partial lines at random rows (c = 10: 118 us against
// 40 us), and scatter.py does not pick it for c < 32.
constexpr int LDS_MAX_TP = 1024;        // points per tile
constexpr int LDS_MAX_GV = 128;         // voxels per workgroup, at most
constexpr int LDS_PAIRS = 2;            // (voxel, channel) pairs per thread: GV * c <= 512

struct LdsPlan {
  int gv, tp;
  unsigned magic;   // floor(2^32 / c) + 1: f / c == __umulhi(f, magic) for f < 2^32 / c
  size_t lds;
  long long blocks;
};

static bool lds_plan(int c, long long n, long long v, LdsPlan& p) {
  if (c > 128 || v <= 0) return false;
  p.tp = LDS_MAX_TP;
  p.gv = 256 * LDS_PAIRS / c;
  if (p.gv > LDS_MAX_GV) p.gv = LDS_MAX_GV;
  // most workgroups should need ONE tile: voxels per workgroup from the average run length (n / v, known on the host)
  const double avg = (double)n / (double)v;
  const int fit = (int)(0.8 * p.tp / (avg > 1.0 ? avg : 1.0));
  if (fit < p.gv) p.gv = fit < 1 ? 1 : fit;
  p.magic = (unsigned)(0x100000000ULL / (unsigned)c) + 1u;
  p.blocks = (v + p.gv - 1) / p.gv;
  // segl (LDS_MAX_GV + 4 ints) | pids (tp ints) | vloc (tp ints) | gradient rows and arg-max rows (GV * c each)
  p.lds = (size_t)(LDS_MAX_GV + 4 + 2 * p.tp + 2 * 256 * LDS_PAIRS) * 4;
  return true;
}

// Backward in voxel order for any c <= 128: a workgroup owns GV consecutive voxels = one contiguous run of `order`; its GV gradient rows (one contiguous run of GV * c floats; the mean's
// division and the max's arg-max row with them) are staged in LDS once, then the run's points are written as a flat dword
// stream (64 consecutive dwords per wave store = 64 / c whole point rows), nontemporal.
template <int MODE>
__global__ __launch_bounds__(256) void spread_lds_kernel(const float* __restrict__ gvox, const int* __restrict__ order,
                                                         const int* __restrict__ seg, const int* __restrict__ argmax, int c,
                                                         int gv, int tp, unsigned magic, long long v, unsigned vox_blocks,
                                                         float* __restrict__ gfeats) {
  const int tid = threadIdx.x;
  if (blockIdx.x >= vox_blocks) {  // points in no voxel: the prefix order[0 .. seg[0]) gets zero rows
    const long long n0 = seg[0];
    const long long stride = (long long)(gridDim.x - vox_blocks) * 256;
    for (long long idx = (long long)(blockIdx.x - vox_blocks) * 256 + tid; idx < n0 * c; idx += stride) {
      const long long i = idx / c;
      __builtin_nontemporal_store(0.0f, gfeats + (long long)order[i] * c + (idx - i * c));
    }
    return;
  }
  extern __shared__ int lds_i[];
  int* const segl = lds_i;
  int* const pids = lds_i + LDS_MAX_GV + 4;
  int* const vloc = pids + tp;
  float* const grow = reinterpret_cast<float*>(vloc + tp);
  int* const arow = reinterpret_cast<int*>(grow) + 256 * LDS_PAIRS;
  const long long v0 = (long long)blockIdx.x * gv;
  const int nv = (int)((v - v0) < gv ? (v - v0) : gv);
  for (int i = tid; i <= nv; i += 256) segl[i] = seg[v0 + i];
  __syncthreads();
  const int pb = segl[0], pe = segl[nv];
  const int npairs = nv * c;
  for (int pr = tid; pr < npairs; pr += 256) {
    float g = gvox[v0 * c + pr];
    if (MODE == GD3D_REDUCE_MEAN) {
      const int lv = (int)__umulhi((unsigned)pr, magic);
      g = g / (float)(segl[lv + 1] - segl[lv]);
    }
    grow[pr] = g;
    if (MODE == GD3D_REDUCE_MAX) arow[pr] = argmax[v0 * c + pr];
  }
  for (int t0 = pb; t0 < pe; t0 += tp) {
    const int np = (pe - t0) < tp ? (pe - t0) : tp;
    __syncthreads();  // rows staged / the previous tile's pids and vloc consumed
    for (int i = tid; i < np; i += 256) {
      pids[i] = order[t0 + i];
      int lo = 0, hi = nv;  // the voxel of position t0 + i: the last lv with segl[lv] <= t0 + i
      while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (segl[mid] <= t0 + i) lo = mid;
        else hi = mid;
      }
      vloc[i] = lo;
    }
    __syncthreads();
    const int nf = np * c;
    for (int f = tid; f < nf; f += 256) {
      const int p = (int)__umulhi((unsigned)f, magic);
      const int chn = f - p * c;
      const int pid = pids[p];
      const int src = vloc[p] * c + chn;
      float g = grow[src];
      if (MODE == GD3D_REDUCE_MAX) g = (arow[src] == pid) ? g : 0.0f;
      __builtin_nontemporal_store(g, gfeats + (long long)pid * c + chn);
    }
  }
}

// max backward for narrow rows: zero fill + one 4-byte store per (voxel, channel) at the recorded arg max.  Measured
// (2 M points -> 214 K voxels): c = 16: 40 us vs 85 us for the masked gather; c = 64: 320 us vs 202 us -> used for c < 32.
__global__ __launch_bounds__(256) void max_grad_kernel(const float* __restrict__ gvox, const int* __restrict__ argmax,
                                                       long long v, int c, float* __restrict__ gfeats) {
  const long long total = v * c;
  const long long stride = (long long)gridDim.x * 256;
  for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += stride) {
    const int pid = argmax[idx];
    if (pid >= 0) gfeats[(long long)pid * c + (idx % c)] = gvox[idx];
  }
}

static int pow2_at_least(int c) {
  int p = 1;
  while (p < c && p < 64) p <<= 1;
  return p;
}

template <int MODE>
static void launch_backward(bool vec, unsigned blocks, hipStream_t s, const float* gv, const int32_t* map, const int32_t* count,
                            const int32_t* argmax, long long n, int c, float* gf) {
  const int cv = vec ? c / 4 : c;
  int shift = -1;
  if ((cv & (cv - 1)) == 0) {
    shift = 0;
    while ((1 << shift) < cv) ++shift;
  }
  const unsigned long long total = (unsigned long long)n * (unsigned)cv;
  const unsigned magic = (shift < 0 && total < 0x100000000ULL / (unsigned)cv) ? (unsigned)(0x100000000ULL / (unsigned)cv) + 1u : 0u;
  if (vec) hipLaunchKernelGGL((gather_grad_kernel<MODE, 4>), dim3(blocks), dim3(256), 0, s, gv, map, count, argmax, n, c, shift, magic, gf);
  else hipLaunchKernelGGL((gather_grad_kernel<MODE, 1>), dim3(blocks), dim3(256), 0, s, gv, map, count, argmax, n, c, shift, magic, gf);
}

// Clears (or fills) small or large device buffers from a KERNEL.  Not hipMemsetAsync: inside a captured hipGraph a memset node was
// found not to be reliably ordered against the kernels around it on this ROCm (profiles/r04_nms_queue_ab.txt) —
// rule of this library: no memset nodes in paths a caller may capture.
__global__ __launch_bounds__(256) void fill_words_kernel(unsigned* __restrict__ p, long long nwords, unsigned value) {
  const long long stride = (long long)gridDim.x * 256;
  long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if ((((uintptr_t)p) & 15) == 0) {
    uint4* p4 = reinterpret_cast<uint4*>(p);
    const long long nv = nwords >> 2;
    const uint4 v4 = make_uint4(value, value, value, value);
    for (long long k = i; k < nv; k += stride) p4[k] = v4;
    for (long long k = (nv << 2) + i; k < nwords; k += stride) p[k] = value;
    return;
  }
  for (; i < nwords; i += stride) p[i] = value;
}
static int fill_words(void* p, size_t bytes, unsigned value, hipStream_t s) {   // bytes: a multiple of 4
  const long long nwords = (long long)(bytes / 4);
  if (nwords == 0) return 0;
  long long blocks = (nwords / 4 + 255) / 256;
  blocks = blocks < 1 ? 1 : (blocks > 4096 ? 4096 : blocks);
  hipLaunchKernelGGL(fill_words_kernel, dim3((unsigned)blocks), dim3(256), 0, s, (unsigned*)p, nwords, value);
  return (int)hipGetLastError();
}

}  // namespace vox

using namespace vox;

extern "C" {

int vox_scatter_reduce(const float* feats, const int32_t* order, const int32_t* seg, int64_t n, int32_t c, int64_t v,
                       int reduce, float* out, int32_t* argmax, void* stream) {
  if (n < 0 || v < 0 || c <= 0) return GD3D_E_BADARG;
  if (reduce != GD3D_REDUCE_SUM && reduce != GD3D_REDUCE_MEAN && reduce != GD3D_REDUCE_MAX) return GD3D_E_BADARG;
  if (v == 0) return 0;
  if (feats == nullptr || order == nullptr || seg == nullptr || out == nullptr) return GD3D_E_BADARG;
  if (n > 0x7fffffffLL) return GD3D_E_TOOLARGE;  // point ids are int32, as in the reference
  hipStream_t s = (hipStream_t)stream;
  const bool vec = (c % 4 == 0) && c <= 256 &&
                   ((((uintptr_t)feats | (uintptr_t)out | (uintptr_t)argmax) & 15) == 0);
  if (vec) {
    const int lp = pow2_at_least(c / 4);
    const long long vw = (v + (64 / lp) - 1) / (64 / lp);
    const long long vb = (vw + 3) / 4;
    if (vb > 0x7fffffffLL) return GD3D_E_TOOLARGE;
    const dim3 vgrid((unsigned)vb), vblk(256);
    if (reduce == GD3D_REDUCE_MAX)
      hipLaunchKernelGGL((reduce_v4_kernel<GD3D_REDUCE_MAX, SCATTER_U>), vgrid, vblk, 0, s, feats, order, seg, (int)c, lp, (long long)v, out, argmax);
    else if (reduce == GD3D_REDUCE_MEAN)
      hipLaunchKernelGGL((reduce_v4_kernel<GD3D_REDUCE_MEAN, SCATTER_U>), vgrid, vblk, 0, s, feats, order, seg, (int)c, lp, (long long)v, out, argmax);
    else
      hipLaunchKernelGGL((reduce_v4_kernel<GD3D_REDUCE_SUM, SCATTER_U>), vgrid, vblk, 0, s, feats, order, seg, (int)c, lp, (long long)v, out, argmax);
    return (int)hipGetLastError();
  }
  const int cp = pow2_at_least(c);
  const long long waves = (v + (64 / cp) - 1) / (64 / cp);
  const long long blocks = (waves + 3) / 4;
  if (blocks > 0x7fffffffLL) return GD3D_E_TOOLARGE;
  const dim3 grid((unsigned)blocks), blk(256);
  if (reduce == GD3D_REDUCE_MAX)
    hipLaunchKernelGGL((reduce_kernel<GD3D_REDUCE_MAX>), grid, blk, 0, s, feats, order, seg, (int)c, cp, (long long)v, out, argmax);
  else if (reduce == GD3D_REDUCE_MEAN)
    hipLaunchKernelGGL((reduce_kernel<GD3D_REDUCE_MEAN>), grid, blk, 0, s, feats, order, seg, (int)c, cp, (long long)v, out, argmax);
  else
    hipLaunchKernelGGL((reduce_kernel<GD3D_REDUCE_SUM>), grid, blk, 0, s, feats, order, seg, (int)c, cp, (long long)v, out, argmax);
  return (int)hipGetLastError();
}

int vox_scatter_backward(const float* grad_vox, const int32_t* map, const int32_t* count, const int32_t* argmax, int64_t n,
                         int32_t c, int64_t v, int reduce, float* grad_feats, void* stream) {
  if (n < 0 || v < 0 || c <= 0) return GD3D_E_BADARG;
  if (reduce != GD3D_REDUCE_SUM && reduce != GD3D_REDUCE_MEAN && reduce != GD3D_REDUCE_MAX) return GD3D_E_BADARG;
  if (n == 0) return 0;
  if (grad_feats == nullptr) return GD3D_E_BADARG;
  hipStream_t s = (hipStream_t)stream;
  if (v == 0) return fill_words(grad_feats, (size_t)n * c * sizeof(float), 0u, s);
  if (grad_vox == nullptr || map == nullptr) return GD3D_E_BADARG;
  if (reduce == GD3D_REDUCE_MAX && argmax == nullptr) return GD3D_E_BADARG;
  if (reduce == GD3D_REDUCE_MEAN && count == nullptr) return GD3D_E_BADARG;
  const bool vec = (c % 4 == 0) && ((((uintptr_t)grad_vox | (uintptr_t)grad_feats | (uintptr_t)argmax) & 15) == 0);
  const long long per_block = 256LL * (vec ? 4 : GATHER_GU_SCALAR);             // 256 threads x GU items
  const long long blocks = ((long long)n * (vec ? c / 4 : c) + per_block - 1) / per_block;
  if (blocks > 0x7fffffffLL) return GD3D_E_TOOLARGE;
  const unsigned nb = (unsigned)blocks;
  if (reduce == GD3D_REDUCE_MAX && c < 32) {
    const int e = fill_words(grad_feats, (size_t)n * c * sizeof(float), 0u, s);
    if (e != 0) return e;
    long long mb = (v * c + 255) / 256;
    if (mb > 8192) mb = 8192;
    hipLaunchKernelGGL(vox::max_grad_kernel, dim3((unsigned)mb), dim3(256), 0, s, grad_vox, argmax, (long long)v, (int)c, grad_feats);
  } else if (reduce == GD3D_REDUCE_MAX) launch_backward<GD3D_REDUCE_MAX>(vec, nb, s, grad_vox, map, count, argmax, n, c, grad_feats);
  else if (reduce == GD3D_REDUCE_MEAN) launch_backward<GD3D_REDUCE_MEAN>(vec, nb, s, grad_vox, map, count, argmax, n, c, grad_feats);
  else launch_backward<GD3D_REDUCE_SUM>(vec, nb, s, grad_vox, map, count, argmax, n, c, grad_feats);
  return (int)hipGetLastError();
}

int vox_scatter_backward_grouped(const float* grad_vox, const int32_t* order, const int32_t* seg, const int32_t* argmax,
                                 int64_t n, int32_t c, int64_t v, int reduce, float* grad_feats, void* stream) {
  if (n < 0 || v < 0 || c <= 0) return GD3D_E_BADARG;
  if (reduce != GD3D_REDUCE_SUM && reduce != GD3D_REDUCE_MEAN && reduce != GD3D_REDUCE_MAX) return GD3D_E_BADARG;
  if (n == 0) return 0;
  if (grad_feats == nullptr) return GD3D_E_BADARG;
  hipStream_t s = (hipStream_t)stream;
  if (v == 0) return fill_words(grad_feats, (size_t)n * c * sizeof(float), 0u, s);
  if (grad_vox == nullptr || order == nullptr || seg == nullptr) return GD3D_E_BADARG;
  if (reduce == GD3D_REDUCE_MAX && argmax == nullptr) return GD3D_E_BADARG;
  const bool vec = (c % 4) == 0 && c <= 256 && ((((uintptr_t)grad_vox | (uintptr_t)grad_feats | (uintptr_t)argmax) & 15) == 0);
  LdsPlan plan;
  if (!vec && lds_plan(c, n, v, plan)) {
    if (plan.blocks + ZERO_BLOCKS > 0x7fffffffLL) return GD3D_E_TOOLARGE;
    const dim3 lgrid((unsigned)(plan.blocks + ZERO_BLOCKS)), lblk(256);
    if (reduce == GD3D_REDUCE_MAX)
      hipLaunchKernelGGL((spread_lds_kernel<GD3D_REDUCE_MAX>), lgrid, lblk, plan.lds, s, grad_vox, order, seg, argmax, (int)c, plan.gv, plan.tp, plan.magic, (long long)v, (unsigned)plan.blocks, grad_feats);
    else if (reduce == GD3D_REDUCE_MEAN)
      hipLaunchKernelGGL((spread_lds_kernel<GD3D_REDUCE_MEAN>), lgrid, lblk, plan.lds, s, grad_vox, order, seg, argmax, (int)c, plan.gv, plan.tp, plan.magic, (long long)v, (unsigned)plan.blocks, grad_feats);
    else
      hipLaunchKernelGGL((spread_lds_kernel<GD3D_REDUCE_SUM>), lgrid, lblk, plan.lds, s, grad_vox, order, seg, argmax, (int)c, plan.gv, plan.tp, plan.magic, (long long)v, (unsigned)plan.blocks, grad_feats);
    return (int)hipGetLastError();
  }
  if (!vec) return GD3D_E_BADARG;  // c > 128 and not 16-byte rows: callers use vox_scatter_backward (map order)
  const int lp = pow2_at_least(c / 4);
  const long long vw = (v + (64 / lp) - 1) / (64 / lp);
  const long long vb = (vw + 3) / 4;
  if (vb + ZERO_BLOCKS > 0x7fffffffLL) return GD3D_E_TOOLARGE;
  const dim3 grid((unsigned)(vb + ZERO_BLOCKS)), blk(256);
  if (reduce == GD3D_REDUCE_MAX)
    hipLaunchKernelGGL((spread_grad_v4_kernel<GD3D_REDUCE_MAX, SCATTER_U>), grid, blk, 0, s, grad_vox, order, seg, argmax, (int)c, lp, (long long)v, (unsigned)vb, grad_feats);
  else if (reduce == GD3D_REDUCE_MEAN)
    hipLaunchKernelGGL((spread_grad_v4_kernel<GD3D_REDUCE_MEAN, SCATTER_U>), grid, blk, 0, s, grad_vox, order, seg, argmax, (int)c, lp, (long long)v, (unsigned)vb, grad_feats);
  else
    hipLaunchKernelGGL((spread_grad_v4_kernel<GD3D_REDUCE_SUM, SCATTER_U>), grid, blk, 0, s, grad_vox, order, seg, argmax, (int)c, lp, (long long)v, (unsigned)vb, grad_feats);
  return (int)hipGetLastError();
}

}  // extern "C"
