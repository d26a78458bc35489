// center_infer.hip — the CenterPoint inference slice that ends in rotated NMS, for gfx950 (include/gd3d.h, ABI 4).
//
// What the reference does per task with ~60 framework launches, a Python loop over samples and B host syncs
//   /root/reference/mmdet3d_gaussian/models/dense_heads/gd_centerpoint_head.py:218-303 (get_bboxes), :305-361
//   /root/reference/mmdet3d_gaussian/core/bbox/coders/centerpoint_bbox_coders.py:23-58 (_topk, select_best), :87-112 (decode)
//   /root/reference/mmdet3d_gaussian/core/bbox/coders/centerpoint_bbox_yaw_coders.py:18-56 (decode, correct_yaw)
// is here one selection kernel + the batched NMS of rbox.hip + one merge kernel for ALL tasks and samples.
//
// select_kernel, one 1024-thread workgroup per (task, sample) group, on order-preserving 32-bit keys of the LOGITS (sigmoid is
// monotonic: the heat map is read, never rewritten):
//   1. candidates.  A threshold from a strided sample of 1024 cells (sorted in LDS; the sample's R-th best, R a few standard
//      deviations above the rank at which the K-th best is expected), then ONE pass over the map that keeps the cells
//      >= threshold (per thread a count, one wave scan and one LDS atomic per wave and batch of 32 cells).  The pass proves
//      itself: with K <= M <= 8192 survivors the K best are among them whatever the sample looked like.  Otherwise (a
//      sample that misled, masses of equal keys, maps of a few hundred thousand cells with K in the thousands) the EXACT
//      radix select runs: 12 + 12 + 8 bit levels, each a histogram pass in LDS (peer-aggregated adds) and a pass that
//      appends what lies above the boundary bin; with all 32 bits resolved and still too many equal keys — a constant map —
//      the needed ties are taken in index order by an ordered block scan.
//   2. order.  Bitonic sort of the candidates in LDS on (key, ~index): descending score, equal scores in ascending flat
//      index — a total order, so the result does not depend on the order the atomics appended in.  Register blocked (three
//      compare-exchange steps per LDS round trip), the list padded against bank conflicts.
//   3. per selected cell: class / y / x, sigmoid of its logit, gather of the head channels at that cell from the separate
//      maps (no concatenated copy of the maps exists), decode, score and centre-range mask, ORDERED compaction of the
//      survivors (they stay in score order: the NMS needs no second sort), NMS boxes [x1, y1, x2, y2, yaw] / circle centres.
// One workgroup per group is the right grain for the reference's geometry (128 x 128 cells x 1-2 classes, K = 500: 34-47 us,
// tools/center_infer_phases.py).  Maps above 131072 cells per group (468 x 468 x 3 with K = 4096 took 0.9 ms that way: one CU
// streaming the map) take the WIDE form: wide_sample_kernel (the threshold, one workgroup per group), wide_filter_kernel (the
// filtering pass by slices of 32768 cells over the whole chip, candidates appended to a list in global memory), then
// select_kernel starts from that list — thinned by a second threshold, sampled from the candidates themselves, when it exceeds
// the LDS buffer — and falls back to the exact select over the map when the list proves nothing (fewer than K entries, overflow).
// merge_kernel, one thread per output row: the kept rows of every task, in task order, z moved to the box bottom, labels
// shifted by the task's class offset, and the number of detections of the sample.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/gd3d_extras.h"
#include "coder_device.h"
#include "lds_sort.h"
#include "rbox_device.h"

namespace cinfer {

using ldssort::PH;
using ldssort::bitonic_desc;

constexpr int T = 1024;
constexpr int WAVES = T / 64;
constexpr int MAXC = CENTER_INFER_MAX_CHANNELS;
constexpr int MAXT = CENTER_INFER_MAX_TASKS;
constexpr int BINS = 4096;
constexpr int SORT_CAP = 8192;
constexpr int MAX_K = 4096;
constexpr int WIDE_N = 131072;       // above this many cells per group the map is filtered by many workgroups
constexpr int WIDE_SLICE = 32768;    // cells per workgroup of that pass
// the candidate list is stored with one pad entry after every 8 (PH below): a thread of the sort that owns 8 consecutive
// entries (64 bytes) would otherwise share its two LDS banks with 31 other lanes of its wave
constexpr size_t LDS_BYTES = sizeof(int) * BINS + sizeof(unsigned long long) * (SORT_CAP + SORT_CAP / 8);


struct Task {
  const float* heat;
  const float* chan[MAXC];
  long long bstride[MAXC];
  int classes, label_offset;
  float nms_thresh;
  int pad;
};

struct SelArgs {
  Task task[MAXT];
  int B, H, W, K, nchan, decode, sigmoid, use_thr, use_range, circle, co, cap, group0;
  gdcoder::Geom geom;
  float score_thr, lo[3], hi[3];
  // raw selection (nullable): (G,K), (G,K), (G,K,2), (G,K,nchan)
  float* sel_scores;
  long long* sel_cls;
  long long* sel_xy;
  float* sel_preds;
  // compacted, decoded survivors (nullable as a set): (G,K,co), (G,K), (G,K), (G,K,5|2), (G), (G,cap), (G)
  float* boxes;
  float* scores;
  int* labels;
  float* nmsbox;
  int* counts;
  long long* order;
  float* thresh;
  rbox::OBox* obox;    // nullable: (G, cap) oriented boxes of the survivors for rnms_batched_prepared (rotate NMS)
  long long* clocks;   // nullable: (groups, 8) s_memrealtime stamps of the kernel's phases (center_infer_debug_clocks)
  // wide form (maps above WIDE_N cells): the threshold and the filtering pass ran in their own chip-wide launches
  int wide, wcap;
  unsigned* wtau;               // (G) threshold keys
  int* wcount;                  // (G) candidates found (may exceed wcap: overflow)
  unsigned long long* wlist;    // (G, wcap) candidates, unordered
};

__device__ __forceinline__ unsigned key_of(float v) {
  unsigned u = __float_as_uint(v);
  if (v != v) return 0xffffffffu;          // NaN ranks first (torch.topk)
  if (u == 0x80000000u) u = 0u;            // -0.0 == +0.0
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

// inclusive prefix sum over the 1024 threads (wave shuffles + one LDS hop); `part` is a 16-int scratch; returns the
// inclusive sum of thread `tid`, *total = sum over the block.  Two barriers.
__device__ __forceinline__ int block_scan(int v, int* part, int* total) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int x = v;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const int y = __shfl_up(x, off, 64);
    if (lane >= off) x += y;
  }
  __syncthreads();                           // the previous use of `part` is over
  if (lane == 63) part[wave] = x;
  __syncthreads();
  int base = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < WAVES; ++w) {
    const int p = part[w];
    base += w < wave ? p : 0;
    tot += p;
  }
  *total = tot;
  return x + base;
}

// wave-aggregated append of `item` by the lanes with `take` set; order inside the list is irrelevant (sorted later)
__device__ __forceinline__ void append(bool take, unsigned long long item, unsigned long long* list, int* counter) {
  const unsigned long long m = __ballot(take);
  if (m == 0) return;
  const int lane = threadIdx.x & 63;
  const int leader = __ffsll((long long)m) - 1;
  int base = 0;
  if (lane == leader) base = atomicAdd(counter, __popcll(m));
  base = __builtin_amdgcn_readlane(base, __builtin_amdgcn_readfirstlane(leader));
  const int pos = base + __popcll(m & ((1ull << lane) - 1ull));
  if (take && pos < SORT_CAP) list[PH(pos)] = item;      // the counter keeps counting: the caller sees the overflow
}

// histogram add with two rounds of peer aggregation: trained heat maps put most cells into a handful of exponent bins, and 64
// lanes adding to one LDS word are served one after the other
__device__ __forceinline__ void hist_add(int* hist, int bin, bool valid) {
  const int lane = threadIdx.x & 63;
  unsigned long long todo = __ballot(valid);
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    if (todo == 0) break;
    const int leader = __ffsll((long long)todo) - 1;
    const int lb = __builtin_amdgcn_readlane(bin, __builtin_amdgcn_readfirstlane(leader));
    const unsigned long long same = __ballot(valid && bin == lb) & todo;
    if (lane == leader) atomicAdd(&hist[lb], __popcll(same));
    todo &= ~same;
  }
  if ((todo >> lane) & 1ull) atomicAdd(&hist[bin], 1);
}

// f(key, index, valid) over the N cells of one group, the same number of calls in every lane (f may hold wave collectives):
// up to three scalar cells to reach 16-byte alignment, then four independent 16-byte loads per thread and step, then the tail
template <class F>
__device__ __forceinline__ void scan_keys(const float* __restrict__ heat, int N, F&& f) {
  const int tid = threadIdx.x;
  int head = (int)((4u - (unsigned)(((uintptr_t)heat >> 2) & 3u)) & 3u);
  head = head < N ? head : N;
  if (head > 0) {
    const bool v = tid < head;
    f(v ? key_of(heat[tid]) : 0u, tid, v);
  }
  const int nvec = (N - head) >> 2;
  const float4* __restrict__ hv = (const float4*)(heat + head);
  for (int base = 0; base < nvec; base += 4 * T) {
    float4 x[4];
    bool v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int q = base + u * T + tid;
      v[u] = q < nvec;
      x[u] = v[u] ? hv[q] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = head + 4 * (base + u * T + tid);
      f(key_of(x[u].x), i, v[u]);
      f(key_of(x[u].y), i + 1, v[u]);
      f(key_of(x[u].z), i + 2, v[u]);
      f(key_of(x[u].w), i + 3, v[u]);
    }
  }
  const int done = head + 4 * nvec;
  if (done < N) {
    const int i = done + tid;
    const bool v = i < N;
    f(v ? key_of(heat[i]) : 0u, i, v);
  }
}

__device__ __forceinline__ unsigned long long pack(unsigned key, unsigned idx) {
  return ((unsigned long long)key << 32) | (unsigned long long)(0xffffffffu - idx);
}

// One pass over the map that keeps the cells with value >= tau (NaN counts as greater than everything): per thread and
// batch a count, one wave scan and ONE LDS atomic per wave, then the kept cells are written as (key, ~index) entries.
// Entries beyond SORT_CAP are dropped while the counter keeps counting: the caller sees the overflow.
__device__ __forceinline__ int wave_excl_scan(int v, int* total) {
  const int lane = threadIdx.x & 63;
  int x = v;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const int y = __shfl_up(x, off, 64);
    if (lane >= off) x += y;
  }
  *total = __builtin_amdgcn_readlane(x, 63);
  return x - v;
}

// the float whose key is `k` (0: -inf, keeps everything; the NaN key gives a NaN, and `!(v < NaN)` keeps everything as well: the
// pass then overflows unless the map is small, and the exact select takes over)
__device__ __forceinline__ float key_to_float(unsigned k) {
  return k == 0u ? -__builtin_inff() : __uint_as_float((k & 0x80000000u) ? (k ^ 0x80000000u) : ~k);
}

// LDS_LIST: the padded LDS list of one workgroup (capacity SORT_CAP); otherwise a plain list in global memory (capacity
// `cap`) that many workgroups append to, `i0` = flat index of heat[0] in the group's map
template <bool LDS_LIST>
__device__ __forceinline__ void emit(float v, int i, bool take, int& pos, unsigned long long* list, int cap) {
  if (take) {
    if (pos < cap) list[LDS_LIST ? PH(pos) : pos] = pack(key_of(v), (unsigned)i);
    ++pos;
  }
}

template <bool LDS_LIST>
__device__ __forceinline__ void filter_pass(const float* __restrict__ heat, int N, float tau, unsigned long long* list, int* counter,
                                            int cap = SORT_CAP, int i0 = 0) {
  const int tid = threadIdx.x, lane = tid & 63;
  int head = (int)((4u - (unsigned)(((uintptr_t)heat >> 2) & 3u)) & 3u);
  head = head < N ? head : N;
  const int nvec = (N - head) >> 2;
  const int done = head + 4 * nvec;
  // the (at most 3 + 3) unaligned cells at both ends: one cell per thread
  {
    const int i = tid < head ? tid : done + (tid - head);
    const bool valid = tid < head + (N - done);
    const float v = valid ? heat[i] : 0.f;
    const bool take = valid && !(v < tau);
    int tot;
    const int ex = wave_excl_scan(take ? 1 : 0, &tot);
    if (tot > 0) {
      int base = 0;
      if (lane == 0) base = atomicAdd(counter, tot);
      int pos = __builtin_amdgcn_readfirstlane(base) + ex;
      emit<LDS_LIST>(v, i0 + i, take, pos, list, cap);
    }
  }
  const float4* __restrict__ hv = (const float4*)(heat + head);
  constexpr int U = 8;                        // 16-byte loads in flight per thread
  for (int b0 = 0; b0 < nvec; b0 += U * T) {
    float4 x[U];
    bool ok[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int q = b0 + u * T + tid;
      ok[u] = q < nvec;
      x[u] = ok[u] ? hv[q] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    int cnt = 0;
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (ok[u]) cnt += (!(x[u].x < tau) ? 1 : 0) + (!(x[u].y < tau) ? 1 : 0) + (!(x[u].z < tau) ? 1 : 0) + (!(x[u].w < tau) ? 1 : 0);
    }
    int tot;
    const int ex = wave_excl_scan(cnt, &tot);
    if (tot == 0) continue;                   // wave-uniform
    int base = 0;
    if (lane == 0) base = atomicAdd(counter, tot);
    int pos = __builtin_amdgcn_readfirstlane(base) + ex;
    if (cnt > 0) {
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int i = i0 + head + 4 * (b0 + u * T + tid);
        emit<LDS_LIST>(x[u].x, i, ok[u] && !(x[u].x < tau), pos, list, cap);
        emit<LDS_LIST>(x[u].y, i + 1, ok[u] && !(x[u].y < tau), pos, list, cap);
        emit<LDS_LIST>(x[u].z, i + 2, ok[u] && !(x[u].z < tau), pos, list, cap);
        emit<LDS_LIST>(x[u].w, i + 3, ok[u] && !(x[u].w < tau), pos, list, cap);
      }
    }
  }
}

// The exact radix select (any input): 12 + 12 + 8 bit levels over the keys in global memory.  Leaves >= K candidates that
// contain the K best in list[0 .. *counter) (unordered), or exactly the K best when the ordered tie scan was needed.
__device__ __forceinline__ void radix_select(const float* __restrict__ heat, int N, int K, int* hist, unsigned long long* cand,
                                             int* part, int* s_bin, int* s_above, int* s_cnt, int* s_nsel) {
  const int tid = threadIdx.x;
  int need = K;            // still to be found below the resolved prefix
  unsigned prefix = 0;     // the resolved high bits, right-aligned
  int done = 0;            // how many bits that is
  for (int lvl = 0; lvl < 3; ++lvl) {
    const int bits = lvl < 2 ? 12 : 8;
    const int shift = 32 - done - bits;
    const unsigned mask = (1u << bits) - 1u;
    for (int i = tid; i < BINS; i += T) hist[i] = 0;
    __syncthreads();
    scan_keys(heat, N, [&](unsigned k, int, bool valid) {
      hist_add(hist, (int)((k >> shift) & mask), valid && (done == 0 || (k >> (32 - done)) == prefix));
    });
    __syncthreads();
    // boundary bin: thread tid owns the four bins 4*(1023 - tid) .. +3, so that the block scan runs from the top bin down
    const int top = 4 * (T - 1 - tid) + 3;
    int own = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) own += hist[top - q];
    int total;
    const int incl = block_scan(own, part, &total);
    int excl = incl - own;
    if (excl < need && need <= incl) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int c = hist[top - q];
        if (excl < need && need <= excl + c) {
          *s_bin = top - q;
          *s_above = excl;
          *s_cnt = c;
        }
        excl += c;
      }
    }
    __syncthreads();
    const int bin = *s_bin, above = *s_above, cnt = *s_cnt;
    const int nsel = K - need;               // appended so far
    const bool fits = nsel + above + cnt <= SORT_CAP;
    // append pass: everything under the prefix above the boundary bin, and the boundary bin itself when the lot fits
    scan_keys(heat, N, [&](unsigned k, int i, bool valid) {
      bool take = false;
      if (valid && (done == 0 || (k >> (32 - done)) == prefix)) {
        const int d = (int)((k >> shift) & mask);
        take = d > bin || (fits && d == bin);
      }
      append(take, pack(k, (unsigned)i), cand, s_nsel);
    });
    __syncthreads();
    if (fits) break;
    need -= above;
    prefix = (prefix << bits) | (unsigned)bin;
    done += bits;
    if (lvl == 2) {
      // every key bit is resolved and more than SORT_CAP cells share the boundary key: take the `need` lowest indices
      int taken = 0;
      for (int base = 0; base < N && taken < need; base += T) {
        const int i = base + tid;
        const bool tie = i < N && key_of(heat[i]) == prefix;
        int tot;
        const int inc = block_scan(tie ? 1 : 0, part, &tot);
        const int rank = taken + inc - 1;
        if (tie && rank < need) cand[PH((K - need) + rank)] = pack(prefix, (unsigned)i);
        taken += tot;
      }
      __syncthreads();
      if (tid == 0) *s_nsel = K;
      __syncthreads();
    }
  }
}

__global__ __launch_bounds__(T) void select_kernel(const SelArgs a) {
  // 16-byte aligned: the static __shared__ words below would otherwise push the dynamic block to an offset that is no multiple
  // of 8, and every 64-bit LDS access of the sort becomes a misaligned one (measured: 2.5x slower)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  int* hist = (int*)smem;
  unsigned long long* cand = (unsigned long long*)(smem + sizeof(int) * BINS);
  __shared__ int part[WAVES];
  __shared__ int s_bin, s_above, s_cnt, s_nsel;
  __shared__ unsigned s_tau;

  const int tid = threadIdx.x;
  const int g = blockIdx.x;                 // group inside this launch
  const int t = g / a.B, b = g - t * a.B;
  const Task& tk = a.task[t];
  const int HW = a.H * a.W;
  const int N = tk.classes * HW;
  const float* heat = tk.heat + (size_t)b * (size_t)N;
  const int K = a.K;
  long long* clk = a.clocks != nullptr ? a.clocks + (size_t)(a.group0 + g) * 8 : nullptr;
#define STAMP(i) do { if (clk != nullptr && tid == 0) clk[i] = (long long)wall_clock64(); } while (0)
  STAMP(0);

  // --- 1. candidates: a superset of the K best in cand[0 .. M) -------------------------------------------------------------
  // Fast path: a threshold from a strided sample of 1024 cells (the sample's R-th best key, R a few standard deviations above
  // the rank the K-th best is expected at), then ONE pass over the map that keeps the keys >= threshold.  The pass proves
  // itself: with K <= M <= SORT_CAP survivors the K best are among them, whatever the sample looked like; otherwise (sample
  // not representative, masses of equal keys) the exact radix select below runs instead.
  // Wide form: threshold and pass already ran chip-wide (wide_sample_kernel, wide_filter_kernel) and left M candidates in
  // global memory; more than fit the LDS buffer are thinned by a second threshold, sampled from the candidates themselves.
  if (tid == 0) {
    s_nsel = 0;
    s_tau = 0u;
  }
  unsigned long long* samp = (unsigned long long*)hist;      // 1152 x 8 bytes of the histogram's 16 KB
  bool fast = false;
  int M = 0;
  if (a.wide) {
    STAMP(7);
    STAMP(1);
    const int G = a.group0 + g;
    const int Mw = a.wcount[G];
    const unsigned long long* wl = a.wlist + (size_t)G * (size_t)a.wcap;
    if (Mw >= K && Mw <= a.wcap) {
      fast = true;
      if (Mw > SORT_CAP) {
        const float rho = (float)K * (float)T / (float)Mw;
        const int R = (int)ceilf(rho + 4.0f * sqrtf(rho) + 8.0f);
        if (R < T) {
          samp[PH(tid)] = wl[(unsigned)(((unsigned long long)tid * (unsigned long long)Mw) / (unsigned long long)T)];
          __syncthreads();
          bitonic_desc(samp, T);
          if (tid == 0) s_tau = (unsigned)(samp[PH(R - 1)] >> 32);
        }
      }
      __syncthreads();
      const unsigned tk = s_tau;
      for (int base = 0; base < Mw; base += T) {
        const int i = base + tid;
        const unsigned long long e = i < Mw ? wl[i] : 0ull;
        append(i < Mw && (unsigned)(e >> 32) >= tk, e, cand, &s_nsel);
      }
      __syncthreads();
      M = s_nsel;
    }
  } else {
    fast = N <= SORT_CAP;                   // tau = 0 keeps everything
    if (N > 2048) {
      const float rho = (float)K * (float)T / (float)N;
      const int R = (int)ceilf(rho + 4.0f * sqrtf(rho) + 8.0f);
      if (R < T) {
        fast = true;
        const unsigned si = (unsigned)(((unsigned long long)tid * (unsigned long long)N) / (unsigned long long)T);
        samp[PH(tid)] = pack(key_of(heat[si]), si);
        __syncthreads();
        STAMP(7);
        bitonic_desc(samp, T);
        if (tid == 0) s_tau = (unsigned)(samp[PH(R - 1)] >> 32);
      }
    }
    __syncthreads();
    STAMP(1);
    if (fast) {
      filter_pass<true>(heat, N, key_to_float(s_tau), cand, &s_nsel);
      __syncthreads();
      M = s_nsel;
    }
  }
  if (fast && (M < K || M > SORT_CAP)) {
    fast = false;
    __syncthreads();
    if (tid == 0) s_nsel = 0;
    __syncthreads();
  }
  STAMP(2);
  if (!fast) {
    radix_select(heat, N, K, hist, cand, part, &s_bin, &s_above, &s_cnt, &s_nsel);
    M = s_nsel;
  }
  STAMP(3);
  if (clk != nullptr && tid == 0) clk[6] = M;
  // --- 2. order: descending (key, ~index) ----------------------------------------------------------------------------------
  const unsigned long long* best = cand;
  bitonic_desc(cand, M);
  STAMP(4);

  // the K best, in order: gather, decode, mask, ordered compaction
  const int G = a.group0 + g;              // global group index: output rows
  const size_t row0 = (size_t)G * (size_t)K;
  int kept = 0;
  for (int base = 0; base < K; base += T) {
    const int r = base + tid;
    bool ok = false;
    float box[MAXC];
    float score = 0.f;
    int cls = 0;
    if (r < K) {
      const unsigned idx = 0xffffffffu - (unsigned)best[PH(r)];
      const float v = heat[idx];
      score = a.sigmoid ? 1.0f / (1.0f + expf(-v)) : v;
      cls = (int)(idx / (unsigned)HW);
      const int cell = (int)(idx - (unsigned)cls * (unsigned)HW);
      const int y = cell / a.W, x = cell - y * a.W;
      float p[MAXC];
#pragma unroll
      for (int j = 0; j < MAXC; ++j) {
        p[j] = 0.f;
        if (j < a.nchan) p[j] = tk.chan[j] != nullptr ? tk.chan[j][(size_t)b * (size_t)tk.bstride[j] + (size_t)cell] : 0.5f;
      }
      if (a.sel_scores != nullptr) {
        a.sel_scores[row0 + r] = score;
        a.sel_cls[row0 + r] = cls;
        a.sel_xy[(row0 + r) * 2] = x;
        a.sel_xy[(row0 + r) * 2 + 1] = y;
#pragma unroll
        for (int j = 0; j < MAXC; ++j)
          if (j < a.nchan) a.sel_preds[(row0 + r) * a.nchan + j] = p[j];
      }
      if (a.decode != 0) {
        if (a.decode == 1) {
          const gdcoder::Core c = gdcoder::decode_core(p[0], p[1], p[3], p[4], p[5], 0.f, 0.f, 0.f, (float)x, (float)y, a.geom, 0);
          box[0] = c.x; box[1] = c.y; box[2] = p[2]; box[3] = c.d0; box[4] = c.d1; box[5] = c.d2;
          box[6] = atan2f(p[6], p[7]);
#pragma unroll
          for (int j = 8; j < MAXC; ++j) box[j - 1] = p[j];
        } else {
          const gdcoder::Core c = gdcoder::decode_core(p[0], p[1], p[3], p[4], p[5], p[6], p[7], p[8], (float)x, (float)y, a.geom, 1);
          box[0] = c.x; box[1] = c.y; box[2] = p[2]; box[3] = c.d0; box[4] = c.d1; box[5] = c.d2;
          box[6] = c.yaw;
#pragma unroll
          for (int j = 9; j < MAXC; ++j) box[j - 2] = p[j];
        }
        ok = !a.use_thr || score >= a.score_thr;
        if (a.use_range) {
#pragma unroll
          for (int q = 0; q < 3; ++q) {
            const float ge = box[q] >= a.lo[q] ? 1.0f : 0.0f;     // `.ge(lo)` ... `.le(hi)` on the boolean, as the reference writes it
            ok = ok && ge <= a.hi[q];
          }
        }
      }
    }
    if (a.decode != 0) {                    // uniform
      int tot;
      const int inc = block_scan(ok ? 1 : 0, part, &tot);
      if (ok) {
        const size_t o = row0 + (size_t)(kept + inc - 1);
#pragma unroll
        for (int j = 0; j < MAXC; ++j)
          if (j < a.co) a.boxes[o * a.co + j] = box[j];
        a.scores[o] = score;
        a.labels[o] = cls;
        if (a.circle) {
          a.nmsbox[o * 2] = box[0];
          a.nmsbox[o * 2 + 1] = box[1];
        } else {                            // xywhr2xyxyr of the bev columns [x, y, dx, dy, yaw]
          const float hw = box[3] / 2.0f, hh = box[4] / 2.0f;
          a.nmsbox[o * 5] = box[0] - hw;
          a.nmsbox[o * 5 + 1] = box[1] - hh;
          a.nmsbox[o * 5 + 2] = box[0] + hw;
          a.nmsbox[o * 5 + 3] = box[1] + hh;
          a.nmsbox[o * 5 + 4] = box[6];
          const int pcomp = kept + inc - 1;
          if (a.obox != nullptr && pcomp < a.cap) {   // what obox_prep_kernel would compute from that row in a launch of its own
            const float nb[5] = {box[0] - hw, box[1] - hh, box[0] + hw, box[1] + hh, box[6]};
            rbox::OBox ob;
            rbox::obox_make(nb, ob);
            a.obox[(size_t)G * a.cap + pcomp] = ob;
          }
        }
      }
      kept += tot;
    }
  }
  if (a.decode != 0) {
    for (int r = tid; r < a.cap; r += T) a.order[(size_t)G * a.cap + r] = (long long)(row0 + r);   // already in score order
    if (tid == 0) {
      a.counts[G] = kept;
      a.thresh[G] = tk.nms_thresh;
    }
  }
  STAMP(5);
#undef STAMP
}

// ---- wide form: maps above WIDE_N cells per group -------------------------------------------------------------------------
// wide_sample_kernel (one workgroup per group): the threshold from 1024 strided cells, as above.
// wide_filter_kernel (grid: slices of WIDE_SLICE cells x groups): the filtering pass, appending to the group's list in global
// memory (one global atomic per wave and batch).  select_kernel then starts from that list.
struct WideArgs {
  Task task[MAXT];
  int B, H, W, K, group0, wcap;
  unsigned* wtau;
  int* wcount;
  unsigned long long* wlist;
};

__global__ __launch_bounds__(T) void wide_sample_kernel(const WideArgs a) {
  __shared__ __attribute__((aligned(16))) unsigned long long samp[T + T / 8];
  const int tid = threadIdx.x, g = blockIdx.x;
  const int t = g / a.B, b = g - t * a.B;
  const Task& tk = a.task[t];
  const int N = tk.classes * a.H * a.W;
  const float* heat = tk.heat + (size_t)b * (size_t)N;
  const float rho = (float)a.K * (float)T / (float)N;
  const int R = (int)ceilf(rho + 4.0f * sqrtf(rho) + 8.0f);
  unsigned tau = 0u;                        // R >= T: keep everything (the list then overflows and the exact select runs)
  if (R < T) {
    const unsigned si = (unsigned)(((unsigned long long)tid * (unsigned long long)N) / (unsigned long long)T);
    samp[PH(tid)] = pack(key_of(heat[si]), si);
    __syncthreads();
    bitonic_desc(samp, T);
    tau = (unsigned)(samp[PH(R - 1)] >> 32);
  }
  if (tid == 0) {
    a.wtau[a.group0 + g] = tau;
    a.wcount[a.group0 + g] = 0;
  }
}

__global__ __launch_bounds__(T) void wide_filter_kernel(const WideArgs a) {
  const int g = blockIdx.y;
  const int t = g / a.B, b = g - t * a.B;
  const Task& tk = a.task[t];
  const int N = tk.classes * a.H * a.W;
  const int i0 = blockIdx.x * WIDE_SLICE;
  if (i0 >= N) return;
  const int n = N - i0 < WIDE_SLICE ? N - i0 : WIDE_SLICE;
  const int G = a.group0 + g;
  filter_pass<false>(tk.heat + (size_t)b * (size_t)N + i0, n, key_to_float(a.wtau[G]), a.wlist + (size_t)G * (size_t)a.wcap,
                     a.wcount + G, a.wcap, i0);
}

struct MergeArgs {
  const float* boxes;        // (G,K,co)
  const float* scores;       // (G,K)
  const int* labels;         // (G,K)
  const long long* keep;     // (G,cap) flat row indices
  const long long* num_keep; // (G)
  int B, Tn, cap, post, co;
  int label_offset[MAXT * 4];
  float* out_boxes;          // (B, Tn*post, co)
  float* out_scores;
  int* out_labels;
  long long* out_count;
};

__global__ __launch_bounds__(256) void merge_kernel(const MergeArgs a) {
  // blockIdx.y = sample; one thread per output row slot (task, r): the row offset of a task is the sum of the kept counts of
  // the tasks before it (a handful of cached loads per thread)
  const int b = blockIdx.y;
  const int slot = blockIdx.x * 256 + threadIdx.x;
  const int t = slot / a.post, r = slot - t * a.post;
  int off = 0, mine = 0, total = 0;
  bool failed = false;   // a group's NMS scan gave up (num_keep = -1, include/gd3d.h): no rows from it, and the sample's count says so
  for (int q = 0; q < a.Tn; ++q) {
    const long long nk = a.num_keep[q * a.B + b];
    failed |= nk < 0;
    const int n = (int)(nk < 0 ? 0 : (nk < a.post ? nk : a.post));
    off += q < t ? n : 0;
    mine = q == t ? n : mine;
    total += n;
  }
  if (slot == 0) a.out_count[b] = failed ? -1 : total;
  if (t >= a.Tn || r >= mine) return;
  const int G = t * a.B + b;
  const size_t src = (size_t)a.keep[(size_t)G * a.cap + r];
  const size_t dst = (size_t)b * ((size_t)a.Tn * a.post) + (size_t)(off + r);
  for (int j = 0; j < a.co; ++j) {
    float v = a.boxes[src * a.co + j];
    if (j == 2) v = v - a.boxes[src * a.co + 5] * 0.5f;      // gravity centre -> bottom centre (:289)
    a.out_boxes[dst * a.co + j] = v;
  }
  a.out_scores[dst] = a.scores[src];
  a.out_labels[dst] = a.labels[src] + a.label_offset[t];
}

// clock probe: `iters` dependent FMAs per thread, wall time in 10 ns ticks of block 0 (tools/center_infer_phases.py compares a
// one-workgroup launch with a chip-filling one: the ratio says what a latency-bound single-workgroup kernel loses to the clock
// the power management grants a nearly idle chip)
__global__ __launch_bounds__(256) void clock_probe_kernel(long long* out, int iters) {
  float x = (float)threadIdx.x;
  const long long t0 = (long long)wall_clock64();
  for (int i = 0; i < iters; ++i) x = __builtin_fmaf(x, 1.0000001f, 0.5f);
  const long long t1 = (long long)wall_clock64();
  if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
  if (x == 12345.678f) out[1] = 1;     // keeps the chain alive
}

static size_t up256(size_t x) { return (x + 255) & ~(size_t)255; }

struct Layout {
  size_t boxes, scores, labels, nmsbox, counts, order, thresh, keep, num_keep, nms, total;
  int64_t G, K, cap, post;
  int co;
};

// the wide form's buffers: (G) threshold keys, (G) counters, (G, wcap) candidate lists; all zero-sized below WIDE_N cells
struct WideLayout {
  size_t tau, count, list, total;
  int wcap;
  bool wide;
};

static void wide_layout(const center_infer_desc* d, WideLayout& W) {
  int64_t nmax = 0;
  for (int t = 0; t < d->num_tasks; ++t) {
    const int64_t n = (int64_t)d->tasks[t].classes * d->height * d->width;
    nmax = n > nmax ? n : nmax;
  }
  W.wide = nmax > WIDE_N;
  W.wcap = 8 * d->max_per_img + 32768;
  const size_t G = (size_t)d->num_tasks * (size_t)d->batch;
  size_t o = 0;
  W.tau = o; o += W.wide ? up256(sizeof(unsigned) * G) : 0;
  W.count = o; o += W.wide ? up256(sizeof(int) * G) : 0;
  W.list = o; o += W.wide ? up256(sizeof(unsigned long long) * G * (size_t)W.wcap) : 0;
  W.total = o;
}

static int check(const center_infer_desc* d, bool need_decode) {
  if (d == nullptr || d->tasks == nullptr) return GD3D_E_BADARG;
  if (d->num_tasks < 1 || d->batch < 1 || d->height < 1 || d->width < 1) return GD3D_E_BADARG;
  if (d->num_channels < 0 || d->num_channels > MAXC) return GD3D_E_BADARG;
  if (d->max_per_img < 1) return GD3D_E_BADARG;
  if (d->max_per_img > MAX_K) return GD3D_E_TOOLARGE;
  const int64_t hw = (int64_t)d->height * d->width;
  if (d->max_per_img > hw) return GD3D_E_BADARG;     // torch.topk(K) over H*W cells of a class raises in the reference
  for (int t = 0; t < d->num_tasks; ++t) {
    if (d->tasks[t].heatmap == nullptr || d->tasks[t].classes < 1) return GD3D_E_BADARG;
    if (hw * d->tasks[t].classes >= 0x7fffffffLL) return GD3D_E_TOOLARGE;
  }
  if (need_decode) {
    if (d->decode == 1 ? d->num_channels < 8 : d->decode == 2 ? d->num_channels < 9 : true) return GD3D_E_BADARG;
    if (d->nms_type != 0 && d->nms_type != 2) return GD3D_E_BADARG;
  }
  return 0;
}

static void layout(const center_infer_desc* d, Layout& L) {
  L.G = (int64_t)d->num_tasks * d->batch;
  L.K = d->max_per_img;
  L.cap = (d->pre_max_size >= 0 && d->pre_max_size < L.K) ? d->pre_max_size : L.K;
  L.post = (d->post_max_size >= 0 && d->post_max_size < L.cap) ? d->post_max_size : L.cap;
  L.co = d->decode == 1 ? d->num_channels - 1 : d->num_channels - 2;
  const int64_t cap1 = L.cap > 0 ? L.cap : 1;
  size_t o = 0;
  L.boxes = o; o += up256(sizeof(float) * (size_t)(L.G * L.K * L.co));
  L.scores = o; o += up256(sizeof(float) * (size_t)(L.G * L.K));
  L.labels = o; o += up256(sizeof(int) * (size_t)(L.G * L.K));
  L.nmsbox = o; o += up256(sizeof(float) * (size_t)(L.G * L.K * 5));
  L.counts = o; o += up256(sizeof(int) * (size_t)L.G);
  L.order = o; o += up256(sizeof(long long) * (size_t)(L.G * cap1));
  L.thresh = o; o += up256(sizeof(float) * (size_t)L.G);
  L.keep = o; o += up256(sizeof(long long) * (size_t)(L.G * cap1));
  L.num_keep = o; o += up256(sizeof(long long) * (size_t)L.G);
  L.nms = o; o += up256(rnms_batched_workspace_bytes((int32_t)L.G, cap1));
  L.total = o;
}

static long long* g_clocks = nullptr;

static void fill_common(const center_infer_desc* d, SelArgs& a) {
  a.clocks = g_clocks;
  a.B = d->batch;
  a.H = d->height;
  a.W = d->width;
  a.K = d->max_per_img;
  a.nchan = d->num_channels;
  a.sigmoid = d->heat_is_logit;
  a.geom.osf = d->out_size_factor;
  a.geom.vs0 = d->voxel_size[0];
  a.geom.vs1 = d->voxel_size[1];
  a.geom.pc0 = d->pc_range[0];
  a.geom.pc1 = d->pc_range[1];
  a.geom.norm_bbox = d->norm_bbox;
}

// the tasks go through the kernel arguments, MAXT per launch
static int launch_select(const center_infer_desc* d, SelArgs& a, void* wide_ws, hipStream_t s) {
  WideLayout WL;
  wide_layout(d, WL);
  if (WL.wide && wide_ws == nullptr) return GD3D_E_BADARG;
  WideArgs w = {};
  a.wide = WL.wide ? 1 : 0;
  if (WL.wide) {
    a.wcap = w.wcap = WL.wcap;
    a.wtau = w.wtau = (unsigned*)((char*)wide_ws + WL.tau);
    a.wcount = w.wcount = (int*)((char*)wide_ws + WL.count);
    a.wlist = w.wlist = (unsigned long long*)((char*)wide_ws + WL.list);
    w.B = d->batch;
    w.H = d->height;
    w.W = d->width;
    w.K = d->max_per_img;
  }
  // more than 64 KB of dynamic LDS has to be allowed per function AND per device (one process may drive several)
  static bool attr_set[64] = {};
  int devid = 0;
  if (hipGetDevice(&devid) != hipSuccess) return GD3D_E_BADARG;
  if (devid < 0 || devid >= 64 || !attr_set[devid]) {
    const hipError_t e = hipFuncSetAttribute((const void*)select_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_BYTES);
    if (e != hipSuccess) return (int)e;
    if (devid >= 0 && devid < 64) attr_set[devid] = true;
  }
  for (int t0 = 0; t0 < d->num_tasks; t0 += MAXT) {
    const int nt = d->num_tasks - t0 < MAXT ? d->num_tasks - t0 : MAXT;
    for (int t = 0; t < nt; ++t) {
      const center_infer_task& src = d->tasks[t0 + t];
      Task& dst = a.task[t];
      dst.heat = src.heatmap;
      for (int j = 0; j < MAXC; ++j) {
        dst.chan[j] = j < d->num_channels ? src.channel[j] : nullptr;
        dst.bstride[j] = j < d->num_channels ? (long long)src.sample_stride[j] : 0;
      }
      dst.classes = src.classes;
      dst.label_offset = src.label_offset;
      dst.nms_thresh = src.nms_thresh;
      dst.pad = 0;
    }
    a.group0 = t0 * d->batch;
    if (WL.wide) {
      int64_t nmax = 0;
      for (int t = 0; t < nt; ++t) {
        w.task[t] = a.task[t];
        const int64_t n = (int64_t)a.task[t].classes * d->height * d->width;
        nmax = n > nmax ? n : nmax;
      }
      w.group0 = a.group0;
      hipLaunchKernelGGL(wide_sample_kernel, dim3((unsigned)(nt * d->batch)), dim3(T), 0, s, w);
      hipLaunchKernelGGL(wide_filter_kernel, dim3((unsigned)((nmax + WIDE_SLICE - 1) / WIDE_SLICE), (unsigned)(nt * d->batch)), dim3(T), 0, s, w);
    }
    hipLaunchKernelGGL(select_kernel, dim3((unsigned)(nt * d->batch)), dim3(T), LDS_BYTES, s, a);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return (int)e;
  }
  return 0;
}

}  // namespace cinfer

using namespace cinfer;

extern "C" {

int center_infer_max_k(void) { return MAX_K; }

int64_t center_infer_rows_per_task(const center_infer_desc* desc) {
  if (desc == nullptr) return 0;
  Layout L;
  layout(desc, L);
  return L.post;
}

size_t center_infer_workspace_bytes(const center_infer_desc* desc) {
  if (check(desc, true) != 0) return 256;
  Layout L;
  layout(desc, L);
  WideLayout W;
  wide_layout(desc, W);
  return L.total + W.total;
}

size_t center_infer_select_workspace_bytes(const center_infer_desc* desc) {
  if (check(desc, false) != 0) return 256;
  WideLayout W;
  wide_layout(desc, W);
  return W.total > 0 ? W.total : 256;
}

int center_infer_debug_clock_probe(int64_t* device_out, int32_t blocks, int32_t iters, void* stream) {
  if (device_out == nullptr || blocks < 1 || iters < 1) return GD3D_E_BADARG;
  hipLaunchKernelGGL(clock_probe_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (long long*)device_out, (int)iters);
  return (int)hipGetLastError();
}

int center_infer_debug_clocks(int64_t* device_buffer) {
  g_clocks = (long long*)device_buffer;
  return 0;
}

int center_infer_candidates(const center_infer_desc* desc, int64_t* byte_offsets) {
  const int rc = check(desc, true);
  if (rc != 0) return rc;
  if (byte_offsets == nullptr) return GD3D_E_BADARG;
  Layout L;
  layout(desc, L);
  byte_offsets[0] = (int64_t)L.boxes;
  byte_offsets[1] = (int64_t)L.scores;
  byte_offsets[2] = (int64_t)L.labels;
  byte_offsets[3] = (int64_t)L.counts;
  return 0;
}

int center_infer_select(const center_infer_desc* d, void* workspace, float* sel_scores, int64_t* sel_cls, int64_t* sel_xy,
                        float* sel_preds, void* stream) {
  const int rc = check(d, false);
  if (rc != 0) return rc;
  if (sel_scores == nullptr || sel_cls == nullptr || sel_xy == nullptr || (d->num_channels > 0 && sel_preds == nullptr))
    return GD3D_E_BADARG;
  SelArgs a = {};
  fill_common(d, a);
  a.decode = 0;
  a.sel_scores = sel_scores;
  a.sel_cls = (long long*)sel_cls;
  a.sel_xy = (long long*)sel_xy;
  a.sel_preds = sel_preds;
  if (workspace != nullptr && ((uintptr_t)workspace & 255) != 0) return GD3D_E_BADARG;
  return launch_select(d, a, workspace, (hipStream_t)stream);
}

int center_infer_bboxes(const center_infer_desc* d, void* workspace, float* out_boxes, float* out_scores, int32_t* out_labels,
                        int64_t* out_count, void* stream) {
  int rc = check(d, true);
  if (rc != 0) return rc;
  if (workspace == nullptr || out_count == nullptr || ((uintptr_t)workspace & 255) != 0) return GD3D_E_BADARG;
  if (d->num_tasks > MAXT * 4) return GD3D_E_TOOLARGE;
  Layout L;
  layout(d, L);
  hipStream_t s = (hipStream_t)stream;
  if (L.post == 0) return (int)hipMemsetAsync(out_count, 0, sizeof(int64_t) * (size_t)d->batch, s);
  if (out_boxes == nullptr || out_scores == nullptr || out_labels == nullptr) return GD3D_E_BADARG;
  char* w = (char*)workspace;
  SelArgs a = {};
  fill_common(d, a);
  a.decode = d->decode;
  a.use_thr = d->use_score_threshold;
  a.use_range = d->use_limit_range;
  a.circle = d->nms_type == 2;
  a.co = L.co;
  a.cap = (int)L.cap;
  a.score_thr = d->score_threshold;
  for (int q = 0; q < 3; ++q) {
    a.lo[q] = d->limit_range[q];
    a.hi[q] = d->limit_range[q + 3];
  }
  a.boxes = (float*)(w + L.boxes);
  a.scores = (float*)(w + L.scores);
  a.labels = (int*)(w + L.labels);
  a.nmsbox = (float*)(w + L.nmsbox);
  a.counts = (int*)(w + L.counts);
  a.order = (long long*)(w + L.order);
  a.thresh = (float*)(w + L.thresh);
  a.obox = a.circle ? nullptr : (rbox::OBox*)(w + L.nms);      // the NMS workspace starts with its (G, cap) oriented boxes
  rc = launch_select(d, a, w + L.total, s);
  if (rc != 0) return rc;
  if (a.circle)
    rc = rnms_batched(2, a.nmsbox, (const int64_t*)a.order, a.counts, (int32_t)L.G, L.cap, a.thresh, (int64_t*)(w + L.keep),
                      (int64_t*)(w + L.num_keep), w + L.nms, stream);
  else
    rc = rnms_batched_prepared(a.nmsbox, (const int64_t*)a.order, a.counts, (int32_t)L.G, L.cap, a.thresh,
                               (int64_t*)(w + L.keep), (int64_t*)(w + L.num_keep), w + L.nms, stream);
  if (rc != 0) return rc;
  MergeArgs m = {};
  m.boxes = a.boxes;
  m.scores = a.scores;
  m.labels = a.labels;
  m.keep = (const long long*)(w + L.keep);
  m.num_keep = (const long long*)(w + L.num_keep);
  m.B = d->batch;
  m.Tn = d->num_tasks;
  m.cap = (int)L.cap;
  m.post = (int)L.post;
  m.co = L.co;
  for (int t = 0; t < d->num_tasks; ++t) m.label_offset[t] = d->tasks[t].label_offset;
  m.out_boxes = out_boxes;
  m.out_scores = out_scores;
  m.out_labels = out_labels;
  m.out_count = (long long*)out_count;
  hipLaunchKernelGGL(merge_kernel, dim3((unsigned)(((int64_t)d->num_tasks * L.post + 255) / 256), (unsigned)d->batch), dim3(256), 0, s, m);
  return (int)hipGetLastError();
}

}  // extern "C"
