// eval_match.hip — the two remaining pieces of the reference's CPU evaluation helpers on gfx950:
//   * trans_bev   (ops/eval/affinity.cpp:83-105): BEV centre distance matrix, one thread per (det, gt);
//   * match_coco  (ops/eval/matcher.cpp:8-74):    COCO-style greedy matching per cost threshold.
// The matcher consumes the (D,G) affinity the IoU kernels of rbox.hip leave in HBM, so an evaluation pass moves only
// the (T,D) int32 result to the host.  Compiled with -ffp-contract=off (distances bit-equal to the CPU evaluation).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/gd3d.h"

namespace evalm {

__global__ __launch_bounds__(256) void trans_bev_kernel(const float* __restrict__ det, long long nd, int dcols,
                                                        const float* __restrict__ gt, long long ng, int gcols,
                                                        float* __restrict__ out) {
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= nd * ng) return;
  const long long di = idx / ng, gi = idx - di * ng;
  const float dx = det[di * dcols] - gt[gi * gcols], dy = det[di * dcols + 1] - gt[gi * gcols + 1];
  out[idx] = sqrtf(dx * dx + dy * dy);  // correctly rounded (hipcc default); __fsqrt_rn maps to the native approximation
}

// The sequential inner loop of matcher.cpp:30-63 (state = current match and its cost, `<=` so that ties go to the
// later gt, non-ignore beats ignore, an ignore match is overridden by ANY non-ignore gt within the threshold) is
// equivalent to one lexicographic minimum over the eligible gts with cost <= thr:
//     key = (is_ignore, cost, -index)
// so one WAVE owns one threshold, walks the detections in order (the greedy part is inherently serial) and finds each
// detection's gt with a 64-lane strided scan + a wave-wide 64-bit min.  Taken flags live in LDS as a bitmask.
//   key bits: [63] ignore | [62:31] order-preserving cost | [30:0] ~index   (G < 2^31)
__device__ __forceinline__ unsigned int ordered_bits(float v) {
  v += 0.0f;  // -0 -> +0: the reference compares floats, where the two are equal
  const unsigned int u = __float_as_uint(v);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

// wave-wide unsigned min on the DPP network (row_shr 1/2/4/8 scan inside each row of 16, then row_bcast 15 / 31 carry
// the row results; lane 63 ends up with the total).  Lanes that receive nothing keep the identity 0xffffffff.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ unsigned int dpp_min(unsigned int v) {
  const unsigned int moved = (unsigned int)__builtin_amdgcn_update_dpp((int)0xffffffffu, (int)v, CTRL, ROW_MASK, 0xf, false);
  return moved < v ? moved : v;
}
__device__ __forceinline__ unsigned int wave_min_u32(unsigned int v) {
  v = dpp_min<0x111, 0xf>(v);
  v = dpp_min<0x112, 0xf>(v);
  v = dpp_min<0x114, 0xf>(v);
  v = dpp_min<0x118, 0xf>(v);
  v = dpp_min<0x142, 0xa>(v);
  v = dpp_min<0x143, 0xc>(v);
  return (unsigned int)__builtin_amdgcn_readlane((int)v, 63);
}
// 64-bit min = min of the high words, then min of the low words among the lanes that tie on the high word
__device__ __forceinline__ unsigned long long wave_min_u64(unsigned long long k) {
  const unsigned int hi = (unsigned int)(k >> 32), lo = (unsigned int)k;
  const unsigned int mh = wave_min_u32(hi);
  const unsigned int ml = wave_min_u32(hi == mh ? lo : 0xffffffffu);
  return ((unsigned long long)mh << 32) | ml;
}

constexpr unsigned long long KEY_NONE = ~0ull;
constexpr int KB = 4;  // 64-gt chunks whose loads are issued together (256 gts per batch)
constexpr int PF = 8;  // rows of prefetch distance (generic kernel)
constexpr int RB = 32; // detections per register block (narrow kernels)

// one candidate gt: eligible iff (free or crowd) and cost <= thr (NaN never matches, as in the reference)
template <bool REG>
__device__ __forceinline__ unsigned long long consider(unsigned long long best, float v, unsigned int fl, int g, int ng,
                                                       float thr, unsigned int mytaken, int chunk,
                                                       const unsigned int* taken) {
  const int gc = g < ng ? g : 0;
  const unsigned int tk = REG ? (mytaken >> chunk) & 1u : (taken[gc >> 5] >> (gc & 31)) & 1u;
  const bool ok = g < ng && (!tk || (fl & 2u)) && v <= thr;
  const unsigned long long key = ((unsigned long long)(fl & 1u) << 63) | ((unsigned long long)ordered_bits(v) << 31) |
                                 (unsigned long long)(0x7fffffffu - (unsigned int)g);
  return (ok && key < best) ? key : best;
}

// REG: ng <= 2048 -> lane l keeps the taken bits of its own gts (g = 64*chunk + l -> bit `chunk`) in one register and the
// loop touches no memory besides the prefetched cost rows; otherwise the bits live in LDS.
// WIDE: ng > 256 -> the rest of each row is read in the loop; compiled out otherwise so that the per-detection path is
// straight-line code and the compiler's waitcnt insertion waits for the ring slot in use only (a loop in that path
// makes it fall back to vmcnt(0), which serialises every step behind an L2/HBM round trip).
template <bool REG, bool WIDE>
__global__ __launch_bounds__(64) void match_coco_kernel(const float* __restrict__ cost, const float* __restrict__ thrs,
                                                        const unsigned char* __restrict__ is_ignore,
                                                        const unsigned char* __restrict__ is_crowd, int nd, int ng,
                                                        int* __restrict__ matched) {
  extern __shared__ unsigned int taken[];  // ceil(ng / 32) words, this threshold's gt_matched row
  const int t = blockIdx.x, lane = threadIdx.x;
  if (ng <= 0 || nd <= 0) return;  // host never launches these; the clamped loads below assume >= 1 row and column
  const int words = (ng + 31) >> 5;
  if (!REG) {
    for (int w = lane; w < words; w += 64) taken[w] = 0u;
    __builtin_amdgcn_s_barrier();
  }
  unsigned int mytaken = 0u;
  const float thr = thrs[t];
  // The serial loop over the detections must never wait on memory: the first 256 gts of the next PF rows sit in a
  // register ring (a row's loads are issued PF detections before it is used: ~PF x 100 cycles of cover for an L2/HBM
  // round trip), their flags for the whole run.  With G <= 256 (the usual per-class evaluation) that is everything.
  float ring[PF][KB];
  unsigned int flag0[KB];  // bit 0 ignore, bit 1 crowd
#pragma unroll
  for (int u = 0; u < KB; ++u) {
    const int g = u * 64 + lane;
    flag0[u] = g < ng ? ((is_ignore[g] ? 1u : 0u) | (is_crowd[g] ? 2u : 0u)) : 0u;
#pragma unroll
    for (int p = 0; p < PF; ++p) ring[p][u] = cost[(size_t)min(p, nd - 1) * ng + min(g, ng - 1)];  // clamped: branch-free
  }
  int mreg = -1;
  for (int d0 = 0; d0 < nd; d0 += PF) {
#pragma unroll
    for (int p = 0; p < PF; ++p) {
      const int d = d0 + p;
      if (d >= nd) break;
      const float* row = cost + (size_t)d * ng;
      unsigned long long best = KEY_NONE;
      {  // gts 0..255 from the ring; refill the slot with row d + PF (unconditional, index-clamped loads: straight-line
         // code keeps the compiler's vmcnt bookkeeping exact, so it waits for THIS row only)
        float v[KB];
#pragma unroll
        for (int u = 0; u < KB; ++u) v[u] = ring[p][u];
        const float* nrow = cost + (size_t)min(d + PF, nd - 1) * ng;
#pragma unroll
        for (int u = 0; u < KB; ++u) ring[p][u] = nrow[min(u * 64 + lane, ng - 1)];
#pragma unroll
        for (int u = 0; u < KB; ++u) best = consider<REG>(best, v[u], flag0[u], u * 64 + lane, ng, thr, mytaken, u, taken);
      }
      for (int g0 = 64 * KB; WIDE && g0 < ng; g0 += 64 * KB) {  // wide problems: the rest of the row, batch by batch
        float v[KB];
        unsigned int fl[KB];
#pragma unroll
        for (int u = 0; u < KB; ++u) {
          const int g = min(g0 + u * 64 + lane, ng - 1);
          v[u] = row[g];
          fl[u] = (is_ignore[g] ? 1u : 0u) | (is_crowd[g] ? 2u : 0u);
        }
#pragma unroll
        for (int u = 0; u < KB; ++u)
          best = consider<REG>(best, v[u], fl[u], g0 + u * 64 + lane, ng, thr, mytaken, (g0 >> 6) + u, taken);
      }
      int m = -1;
      if (__ballot(best != KEY_NONE) != 0ull) {   // (uniform) nobody eligible — the usual detection of an evaluation: no reduction
        best = wave_min_u64(best);
        m = (int)(0x7fffffffu - (unsigned int)(best & 0x7fffffffull));
      }
      mreg = lane == (d & 63) ? m : mreg;  // results leave in coalesced 64-detection stores, not one store per step
      if (m >= 0) {
        if (REG) {
          if (lane == (m & 63)) mytaken |= 1u << (m >> 6);
        } else if (lane == 0) {
          taken[m >> 5] |= 1u << (m & 31);  // one wave: its LDS operations execute in order, no fence needed
        }
      }
      __builtin_amdgcn_wave_barrier();
    }
    if (((d0 + PF) & 63) == 0 && d0 + PF <= nd) matched[(size_t)t * nd + (d0 + PF - 64) + lane] = mreg;
  }
  const int tail0 = nd & ~63;
  if (tail0 + lane < nd) matched[(size_t)t * nd + tail0 + lane] = mreg;
}

// ---- narrow problems (ng <= 64 * KBT, KBT = 1 or 4): the usual per-class, per-frame evaluation --------------------
// A lone wave issues one VALU instruction every 4+ cycles, so the serial loop is INSTRUCTION bound: the generic kernel
// above spends ~190 VALU instructions per detection on 64-bit keys.  Here the bookkeeping moves to the scalar unit:
// eligibility, the taken set, the ignore / crowd flags and the ignore-priority rule are 64-bit lane MASKS in SGPRs
// (v_cmp writes such a mask directly), the vector unit only compares, selects and runs one 32-bit DPP min.
//   ok[u]   = (cost <= thr) & valid & (~taken | crowd)            scalar
//   sel[u]  = any(ok & ~ignore) ? ok & ~ignore : ok               scalar  (a non-ignore gt beats every ignore gt)
//   mn      = wave-min over sel lanes of the order-preserving cost bits      (6 DPP steps)
//   m       = highest chunk, highest lane with bits == mn in sel             (ties go to the later gt)
template <int KBT>
__global__ __launch_bounds__(64) void match_coco_small_kernel(const float* __restrict__ cost,
                                                              const float* __restrict__ thrs,
                                                              const unsigned char* __restrict__ is_ignore,
                                                              const unsigned char* __restrict__ is_crowd, int nd, int ng,
                                                              int* __restrict__ matched) {
  const int t = blockIdx.x, lane = threadIdx.x;
  if (ng <= 0 || nd <= 0) return;
  const float thr = thrs[t];
  unsigned long long valid[KBT], nonign[KBT], crowd[KBT], taken[KBT];
#pragma unroll
  for (int u = 0; u < KBT; ++u) {
    const int g = u * 64 + lane, gc = min(g, ng - 1);
    valid[u] = __ballot(g < ng);
    nonign[u] = __ballot(g < ng && !is_ignore[gc]);
    crowd[u] = __ballot(g < ng && is_crowd[gc]);
    taken[u] = 0ull;
  }
  // Cost rows reach the serial loop through REGISTER BLOCKS of RB detections: the loads of block b+1 (RB x KBT per lane,
  // index-clamped, all in flight together) are issued before block b is processed and first needed RB detections
  // later, so the single wait per block is a plain vmcnt(0) that has long been satisfied.  (A per-detection register
  // ring needs counted vmcnt waits, which the compiler gives up on as soon as the loop body has a branch.)
  int gcl[KBT];
#pragma unroll
  for (int u = 0; u < KBT; ++u) gcl[u] = min(u * 64 + lane, ng - 1);
  float cur[RB][KBT], nxt[RB][KBT];
#pragma unroll
  for (int r = 0; r < RB; ++r)
#pragma unroll
    for (int u = 0; u < KBT; ++u) cur[r][u] = cost[(size_t)min(r, nd - 1) * ng + gcl[u]];
  int mreg = -1;
  for (int b0 = 0; b0 < nd; b0 += RB) {
#pragma unroll
    for (int r = 0; r < RB; ++r)
#pragma unroll
      for (int u = 0; u < KBT; ++u) nxt[r][u] = cost[(size_t)min(b0 + RB + r, nd - 1) * ng + gcl[u]];
#pragma unroll
    for (int r = 0; r < RB; ++r) {  // branch-free body: rows past nd are computed and discarded
      const int d = b0 + r;
      const bool live = d < nd;
      unsigned long long ok[KBT], anyn = 0ull, anyo = 0ull;
#pragma unroll
      for (int u = 0; u < KBT; ++u) {
        ok[u] = __ballot(cur[r][u] <= thr) & valid[u] & (~taken[u] | crowd[u]);  // NaN never matches
        anyn |= ok[u] & nonign[u];
        anyo |= ok[u];
      }
      if (anyo == 0ull) {   // uniform, and the usual case in an evaluation (most detections touch no ground truth within the
        // threshold): unmatched, nothing else changes — ~10 instructions instead of the ~100 of the selection below (the
        // block-wise prefetch does not mind the branch: its single wait is at the end of the block)
        mreg = (live && lane == (d & 63)) ? -1 : mreg;
        continue;
      }
      unsigned int ord[KBT], best = 0xffffffffu;
#pragma unroll
      for (int u = 0; u < KBT; ++u) {
        ok[u] &= anyn ? nonign[u] : ~0ull;
        ord[u] = ordered_bits(cur[r][u]);
        const unsigned int cand = __builtin_amdgcn_inverse_ballot_w64(ok[u]) ? ord[u] : 0xffffffffu;
        best = cand < best ? cand : best;
      }
      const unsigned int mn = wave_min_u32(best);
      int m = -1;
#pragma unroll
      for (int u = KBT - 1; u >= 0; --u) {
        const unsigned long long eq = __ballot(ord[u] == mn) & ok[u];
        m = (m < 0 && eq) ? u * 64 + 63 - __builtin_clzll(eq | 1ull) : m;
      }
      m = (anyo && live) ? m : -1;
#pragma unroll
      for (int u = 0; u < KBT; ++u) taken[u] |= (m >= 0 && (m >> 6) == u) ? 1ull << (m & 63) : 0ull;
      mreg = (live && lane == (d & 63)) ? m : mreg;
    }
    if (((b0 + RB) & 63) == 0 && b0 + RB <= nd) matched[(size_t)t * nd + (b0 + RB - 64) + lane] = mreg;
#pragma unroll
    for (int r = 0; r < RB; ++r)
#pragma unroll
      for (int u = 0; u < KBT; ++u) cur[r][u] = nxt[r][u];
  }
  const int tail0 = nd & ~63;
  if (tail0 + lane < nd) matched[(size_t)t * nd + tail0 + lane] = mreg;
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

}  // namespace evalm

extern "C" {

int riou_eval_trans_bev(const float* det, int64_t nd, int32_t det_cols, const float* gt, int64_t ng, int32_t gt_cols,
                        float* dist, void* stream) {
  if (nd < 0 || ng < 0 || det_cols < 2 || gt_cols < 2) return GD3D_E_BADARG;
  if (nd == 0 || ng == 0) return 0;
  if (det == nullptr || gt == nullptr || dist == nullptr) return GD3D_E_BADARG;
  const long long total = (long long)nd * ng;
  const long long blocks = (total + 255) / 256;
  if (blocks > 0x7fffffffLL) return GD3D_E_TOOLARGE;
  hipLaunchKernelGGL(evalm::trans_bev_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, det, (long long)nd,
                     det_cols, gt, (long long)ng, gt_cols, dist);
  return (int)hipGetLastError();
}

int eval_match_coco(const float* cost, const float* cost_thrs, const uint8_t* is_ignore, const uint8_t* is_crowd,
                    int64_t nd, int64_t ng, int64_t nt, int32_t* matched, void* stream) {
  if (nd < 0 || ng < 0 || nt < 0) return GD3D_E_BADARG;
  if (nt == 0 || nd == 0) return 0;
  if (matched == nullptr || cost_thrs == nullptr) return GD3D_E_BADARG;
  if (ng == 0)  // no ground truth: every detection is unmatched (-1 = all-ones bytes); the kernels index gt ng - 1
    return evalm::fill_words(matched, sizeof(int32_t) * (size_t)nt * (size_t)nd, 0xffffffffu, (hipStream_t)stream);
  if (cost == nullptr || is_ignore == nullptr || is_crowd == nullptr) return GD3D_E_BADARG;
  if (nd > 0x7fffffffLL || nt > 0x7fffffffLL) return GD3D_E_TOOLARGE;
  if (ng > 1048576) return GD3D_E_TOOLARGE;  // taken bitmask: 128 KiB of the 160 KiB LDS
  const size_t lds = (size_t)((ng + 31) / 32 + 1) * sizeof(unsigned int);
  if (lds > 48 * 1024) {
    hipError_t e = hipFuncSetAttribute((const void*)evalm::match_coco_kernel<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)lds);
    if (e != hipSuccess) return (int)e;
  }
  if (ng <= 64)
    hipLaunchKernelGGL((evalm::match_coco_small_kernel<1>), dim3((unsigned)nt), dim3(64), 0, (hipStream_t)stream, cost,
                       cost_thrs, is_ignore, is_crowd, (int)nd, (int)ng, (int*)matched);
  else if (ng <= 256)
    hipLaunchKernelGGL((evalm::match_coco_small_kernel<4>), dim3((unsigned)nt), dim3(64), 0, (hipStream_t)stream, cost,
                       cost_thrs, is_ignore, is_crowd, (int)nd, (int)ng, (int*)matched);
  else if (ng <= 2048)
    hipLaunchKernelGGL((evalm::match_coco_kernel<true, true>), dim3((unsigned)nt), dim3(64), 16, (hipStream_t)stream, cost,
                       cost_thrs, is_ignore, is_crowd, (int)nd, (int)ng, (int*)matched);
  else
    hipLaunchKernelGGL((evalm::match_coco_kernel<false, true>), dim3((unsigned)nt), dim3(64), lds, (hipStream_t)stream, cost,
                       cost_thrs, is_ignore, is_crowd, (int)nd, (int)ng, (int*)matched);
  return (int)hipGetLastError();
}

}  // extern "C"
