// rbox.hip — rotated BEV NMS and pairwise rotated IoU kernels for gfx950 + their C-ABI entry points
// (include/gd3d.h).  Geometry: rbox_device.h.  Compiled with -ffp-contract=off (see build.py).
//
// NMS (replaces mmdet3d iou3d_cuda.nms_gpu, whose mask goes D->H and is scanned on the host):
//   0. (nms_gpu path, <= 16384 candidates) rank_place_kernel: the score order by counting larger (score, -index) keys — no
//      sort — with the per-box prep scattered straight to its rank, one launch.
//   1. obox_prep_kernel (pre-sorted / caller-ordered paths): one thread per box: sin/cos + rotated corners once ->
//      64-byte OBox records.
//   2. nms_mask_compact_kernel (rotated) / nms_mask_kernel (axis-aligned, circle): one wave per (8..64 row boxes,
//      64-box column block at or right of their own block).  Rotated: the cheap bounding-circle test for every pair
//      first, survivors queued in LDS, then the full polygon-clipping predicate with the live lanes packed densely;
//      per-thread polygon vertices live in LDS [slot][thread] (12 KiB per wave).  VALU-throughput-bound integer+fp32
//      work (no HBM roofline: N=4096 reads 256 KB, writes 1 MB).
//      From 768 boxes on (rotated): the QUEUED form — nms_circle_queue_kernel (circle tests only, survivors to a queue in HBM) and
//      nms_clip_queue_kernel (the full predicate, every pass full), which also appends each positive pair of different blocks to the
//      earlier box's near / far VICTIM LIST.
//   3. the greedy scan, ONE 16-wave workgroup per group that never leaves the device:
//      nms_list_or_scan_kernel (groups of <= 16384 boxes): the LIST scan — one state byte per box in LDS, a resolver wave per block
//      (alive bytes -> in-block fixed point -> kept; marks the kept boxes' near victims through addresses from an LDS ring), twelve
//      helper waves (far victims, ring fill, kept ids); no mask rows, no barrier in the loop.  A full victim list makes the same
//      launch run the classic scan:
//      nms_scan_kernel (classic; also beyond 16384 boxes, two-level with nms_propagate_kernel): a resolver wave solves each block
//      from an LDS ring, three phase-shifted groups of row waves OR the mask rows of the boxes just kept into the removed-set (LDS);
//      one LDS-only barrier per block.
#include <hip/hip_runtime.h>
#include <type_traits>
#include <stdint.h>

#include "../../include/gd3d.h"
#include <stdlib.h>

#include "rbox_device.h"

namespace rbox {

enum { MODE_ROT = 0, MODE_NORMAL = 1, MODE_CIRCLE = 2 };

// One launch serves G independent NMS problems ("groups": classes, samples, tasks) over a shared box array:
// group g owns order[g*cap .. g*cap + n_g), n_g = counts[g] read ON THE DEVICE (no host sync to size the launch;
// grids are sized for `cap` and surplus workgroups leave after one scalar load).  G = 1 with counts == NULL is the plain call.
struct NmsArgs {
  const float* boxes;          // (N,5) [x1,y1,x2,y2,r]; MODE_CIRCLE: (N,2) centres
  const long long* order;      // (G, cap) score order per group (indices into boxes); NULL (G = 1 only): identity
  const int* counts;           // (G) device, nullable
  const float* thresh_dev;     // (G) device, nullable -> thresh / thresh_d
  int n, cap, cbs, rows;       // cbs = ceil(cap / 64): mask row stride in words
  float thresh;
  double thresh_d;             // MODE_CIRCLE: numba compares the float32 distance with a float64 threshold
};

__device__ __forceinline__ int group_n(const NmsArgs& a, int g) {
  if (a.counts == nullptr) return a.n;
  const int c = a.counts[g];
  return c < 0 ? 0 : (c > a.cap ? a.cap : c);
}

// `order` (nullable): score order computed by the caller; box i of the NMS is boxes[order[i]] (saves the gather pass)
// `zero_words` (nullable): control words of the queued mask form, cleared here so that no separate fill sits in the stream
// (zero_n words PER GROUP; the first workgroup of group blockIdx.y clears that group's words)
__device__ __forceinline__ void zero_control_words(unsigned* zero_words, int zero_n) {
  if (zero_words != nullptr && blockIdx.x == 0)
    for (int k = threadIdx.x; k < zero_n; k += blockDim.x) zero_words[(size_t)blockIdx.y * zero_n + k] = 0u;
}

__global__ __launch_bounds__(256) void zero_words_kernel(unsigned* words, int per_group) { zero_control_words(words, per_group); }

__global__ __launch_bounds__(256) void obox_prep_kernel(const NmsArgs a, OBox* __restrict__ out, unsigned* zero_words, int zero_n) {
  zero_control_words(zero_words, zero_n);
  const int g = blockIdx.y;
  const int n = group_n(a, g);
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const size_t src = a.order != nullptr ? (size_t)a.order[(size_t)g * a.cap + i] : (size_t)i;
  float b[5];
#pragma unroll
  for (int k = 0; k < 5; ++k) b[k] = a.boxes[src * 5 + k];
  OBox o;
  obox_make(b, o);
  out[(size_t)g * a.cap + i] = o;
}

// Score order + prep without torch.sort, for up to RANK_MAX candidates (the heads cut to nms_pre before they call nms_gpu).
// Order = scores descending, ties by ascending index, NaN scores first: what torch.sort(descending=True, stable=True)
// yields.  Every box gets one UNIQUE 64-bit key — (order-preserving map of the float) << 32 | ~index — so its position
// in the order is simply the number of larger keys.  That count is embarrassingly parallel (a single-workgroup bitonic
// sort of 4096 keys is LDS-bandwidth-bound at ~50 us; rocPRIM's radix sort behind torch.sort takes 16-24 us + launches):
// rank_place_kernel below.
constexpr int RANK_MAX = 16384;


__device__ __forceinline__ unsigned long long score_key(float s, unsigned idx) {
  unsigned u = __float_as_uint(s);
  if (s != s) u = 0xfffffffeu;              // any NaN: greatest (+inf maps to 0xff800000); NOT 0xffffffff: rank_place forms u + 1
  else {
    if (u == 0x80000000u) u = 0u;           // -0.0 == +0.0 for the comparison torch.sort makes
    u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
  }
  return ((unsigned long long)u << 32) | (unsigned long long)(0xffffffffu - idx);
}

// lt += #{keys of the lane's row of 16 that are < m}, two instructions per step: step T subtracts m from the key T places away
// in the row — row_ror:T as a DPP operand of v_sub_co_u32 (VOP2 takes DPP on gfx9, VOPC does not) — the borrow (key < m) lands in
// VCC and v_addc adds it.  Sixteen independent rotations of ONE register, no dependent chain, no wait states between the pairs
// (a VCC written by one VALU instruction may be the next one's carry-in).  The leading s_nop covers both DPP hazards (source
// VGPR written by the VALU instruction before: 2 wait states; EXEC written by a VALU instruction: 5) for whatever code the
// compiler puts in front of the block.  Measured on the chip before use (profiles/r06_nms_batched.txt): v_sub_co_u32_dpp
// computes dpp(src0) - src1 as written; v_subREV_co_u32_dpp does NOT compute src1 - dpp(src0): the rotation goes to the
// MINUEND there too (it gives dpp(src1) - src0), which pairs every key with the wrong box.
__device__ __forceinline__ void count_row_keys_below(unsigned ku, unsigned m, int& lt) {
  unsigned tmp;
  asm("s_nop 4\n"
      "v_sub_co_u32 %1, vcc, %2, %3\n v_addc_co_u32 %0, vcc, 0, %0, vcc\n"
      "v_sub_co_u32_dpp %1, vcc, %2, %3 row_ror:1 row_mask:0xf bank_mask:0xf\n v_addc_co_u32 %0, vcc, 0, %0, vcc\n"
      "v_sub_co_u32_dpp %1, vcc, %2, %3 row_ror:2 row_mask:0xf bank_mask:0xf\n v_addc_co_u32 %0, vcc, 0, %0, vcc\n"
      "v_sub_co_u32_dpp %1, vcc, %2, %3 row_ror:3 row_mask:0xf bank_mask:0xf\n v_addc_co_u32 %0, vcc, 0, %0, vcc\n"
      "v_sub_co_u32_dpp %1, vcc, %2, %3 row_ror:4 row_mask:0xf bank_mask:0xf\n v_addc_co_u32 %0, vcc, 0, %0, vcc\n"
      "v_sub_co_u32_dpp %1, vcc, %2, %3 row_ror:5 row_mask:0xf bank_mask:0xf\n v_addc_co_u32 %0, vcc, 0, %0, vcc\n"
      "v_sub_co_u32_dpp %1, vcc, %2, %3 row_ror:6 row_mask:0xf bank_mask:0xf\n v_addc_co_u32 %0, vcc, 0, %0, vcc\n"
      "v_sub_co_u32_dpp %1, vcc, %2, %3 row_ror:7 row_mask:0xf bank_mask:0xf\n v_addc_co_u32 %0, vcc, 0, %0, vcc\n"
      "v_sub_co_u32_dpp %1, vcc, %2, %3 row_ror:8 row_mask:0xf bank_mask:0xf\n v_addc_co_u32 %0, vcc, 0, %0, vcc\n"
      "v_sub_co_u32_dpp %1, vcc, %2, %3 row_ror:9 row_mask:0xf bank_mask:0xf\n v_addc_co_u32 %0, vcc, 0, %0, vcc\n"
      "v_sub_co_u32_dpp %1, vcc, %2, %3 row_ror:10 row_mask:0xf bank_mask:0xf\n v_addc_co_u32 %0, vcc, 0, %0, vcc\n"
      "v_sub_co_u32_dpp %1, vcc, %2, %3 row_ror:11 row_mask:0xf bank_mask:0xf\n v_addc_co_u32 %0, vcc, 0, %0, vcc\n"
      "v_sub_co_u32_dpp %1, vcc, %2, %3 row_ror:12 row_mask:0xf bank_mask:0xf\n v_addc_co_u32 %0, vcc, 0, %0, vcc\n"
      "v_sub_co_u32_dpp %1, vcc, %2, %3 row_ror:13 row_mask:0xf bank_mask:0xf\n v_addc_co_u32 %0, vcc, 0, %0, vcc\n"
      "v_sub_co_u32_dpp %1, vcc, %2, %3 row_ror:14 row_mask:0xf bank_mask:0xf\n v_addc_co_u32 %0, vcc, 0, %0, vcc\n"
      "v_sub_co_u32_dpp %1, vcc, %2, %3 row_ror:15 row_mask:0xf bank_mask:0xf\n v_addc_co_u32 %0, vcc, 0, %0, vcc\n"
      : "+v"(lt), "=&v"(tmp)
      : "v"(ku), "v"(m)
      : "vcc");
}

// rank_place_kernel: a 16-wave workgroup owns SIXTEEN boxes of a group and ALL of the group's keys — wave w counts, for the
// workgroup's boxes, the keys of the w-th sixteenth that are greater.  A wave is four DPP rows of 16 lanes: lane (row q, b) holds the
// workgroup's box b; of every 64 keys the wave loads (lane l builds key jb + l in registers) row q owns keys 16 q .. 16 q + 15 and
// rotates them through its lanes (v_mov_b32_dpp row_ror:1): sixteen steps show every box all 64 keys, one compare per lane and step.
// The four rows' counts meet by two lane exchanges, the sixteen waves' in LDS, and wave 0 places its boxes right away: order[r] = i
// and, for rotated NMS, the OBox record of box i in slot r (this IS the prep kernel, scattered).  No atomics, deterministic.
// (Until round 5 a workgroup owned 64 boxes, one per lane, the 64 keys broadcast by v_readlane: the same number of compare
// instructions, but n / 64 workgroups — 64 at n = 4096 — kept a quarter of the CUs busy; with n / 16 workgroups the kernel covers
// the chip: 14.9 -> 12.0 us averaged over n = 1000 / 4096 / 9000 (6.5 -> 5.2, 31.1 -> 26.4 at the ends).  Rounds 2-3 ran it as two launches — partial counts per 256-key slice in HBM, then a scatter kernel.)
// counts[g] (nullable) = min(#valid boxes of the group, n_keep): every workgroup sees all of the group's valid flags while it
// counts, so workgroup 0 WRITES the number — nothing is cleared and then added to (the round-3 form cleared counts with a memset
// that a captured hipGraph did not order reliably: profiles/r04_nms_queue_ab.txt).
// blockIdx.y = group.  Dense form: scores / valid are (G, n) rows; a box that is not `valid` in its group (nullable mask) gets
// key 0: below every real key, never placed.  Segmented form (seg != nullptr, (G+1) int32 on the device): group g owns the boxes
// [seg[g], seg[g+1]) of ONE flat score array and ranks only those — O(sum n_g^2) compares instead of the dense (G, G n) matrices;
// `n` is then the LARGEST group size (grid extent), indices inside a group are local.  gps > 0: every gps consecutive groups share
// one set of n boxes, set k at rows [k n, (k + 1) n) of the flat box array.
template <bool PREP>
__global__ __launch_bounds__(1024) void rank_place_kernel(const float* __restrict__ boxes, const float* __restrict__ scores_,
                                                          const unsigned char* __restrict__ valid_, const int* __restrict__ seg,
                                                          int n, int n_keep, long long* __restrict__ order_,
                                                          OBox* __restrict__ ob_, int* __restrict__ counts, int gps,
                                                          unsigned* zero_words, int zero_n) {
  __shared__ int spart[16][16];
  __shared__ int svalid[16];
  zero_control_words(zero_words, zero_n);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nw = (int)(blockDim.x >> 6);                               // waves per workgroup: 16; 4 in the masked dense form (below)
  const int g = blockIdx.y;
  const int sbase = seg != nullptr ? seg[g] : 0;                       // first score / box of the group in the flat arrays
  const int ng = seg != nullptr ? seg[g + 1] - sbase : n;
  if (ng <= 0) {                                                       // uniform: an empty group places nothing
    if (counts != nullptr && blockIdx.x == 0 && tid == 0) counts[g] = 0;
    return;
  }
  if ((int)blockIdx.x * 16 >= ng) return;                              // uniform: no box of this group in this workgroup
  const size_t grow = (size_t)g * n;
  const float* scores = seg != nullptr ? scores_ + sbase : scores_ + grow;
  const unsigned char* valid = (valid_ != nullptr && seg == nullptr) ? valid_ + grow : nullptr;
  const int i = blockIdx.x * 16 + (lane & 15);
  // dense form with a validity mask (multi-class NMS over shared boxes): a workgroup none of whose boxes takes part in this group
  // has nothing to place (workgroup 0 stays: it counts the group's valid boxes), and below a chunk of 64 keys without a valid one
  // is skipped — the work follows the group's own size, not the size of the shared box array
  if (valid != nullptr && blockIdx.x != 0 && __ballot(i < ng && valid[min(i, ng - 1)] != 0) == 0ull) return;   // uniform over the workgroup
  // The compares (round 6).  A key is (order-preserving 32-bit image u of the score, ~index): box i's rank = #{u_j > u_i} +
  // #{u_j == u_i, j < i}.  The workgroup's sixteen boxes lie in ONE chunk of 64 keys, C0; for every other chunk the index part is
  // decided by the chunk alone — keys of an EARLIER chunk count from u_j >= u_i, keys of a LATER chunk from u_j > u_i, i.e.
  // u_j >= u_i + 1 (images end at 0xfffffffe: nothing wraps) — so one 32-bit compare against a per-chunk uniform choice of
  // threshold m decides, counted as its complement: every lane sees 16 keys per chunk, #{u_j >= m} = 16 - #{u_j < m} (the 0 of
  // an unused key is below every m: real images start at 0x007fffff, the image of -inf).  v_sub + v_addc per 64 pairs where the
  // 64-bit keys cost a 64-bit compare, a select, an add and two dependent rotations: 2 against ~6 instructions and their wait
  // states per step.  Only chunk C0 compares whole keys.  (One class alone, 4096 keys: 7.4 -> 7.7 us, unchanged — 4.2 us of that
  // is the dispatch floor; the masked three-class form needed it together with fewer waves: profiles/r06_nms_batched.txt.)
  const int C0 = (int)(blockIdx.x * 16u) >> 6;
  const unsigned long long mine = i < ng ? score_key(scores[i], (unsigned)i) : ~0ull;
  const unsigned mu = (unsigned)(mine >> 32);
  int cnt = 0, nvalid = 0, lt = 0, n32 = 0;
  auto compare = [&](int c, bool use, float sc, int j) {
    if (c != C0) {   // uniform
      const unsigned ku = use ? (unsigned)(score_key(sc, 0u) >> 32) : 0u;
      const unsigned m = c < C0 ? mu : mu + 1u;
      count_row_keys_below(ku, m, lt);
      ++n32;
    } else {
      const unsigned long long kj = use ? score_key(sc, (unsigned)j) : 0ull;   // 0 is below every real key
      unsigned klo = (unsigned)kj, khi = (unsigned)(kj >> 32);
#pragma unroll
      for (int t = 0; t < 16; ++t) {
        cnt += (((unsigned long long)khi << 32) | klo) > mine ? 1 : 0;
        klo = (unsigned)__builtin_amdgcn_update_dpp(0, (int)klo, 0x121, 0xf, 0xf, false);   // row_ror:1
        khi = (unsigned)__builtin_amdgcn_update_dpp(0, (int)khi, 0x121, 0xf, 0xf, false);
      }
    }
  };
  if (valid == nullptr) {
    const int q = (((ng + nw - 1) / nw) + 63) & ~63;                     // keys per wave: a sixteenth, in whole chunks of 64
    const int b = wave * q, e = min(b + q, ng);
    for (int jb0 = b; jb0 < e; jb0 += 4 * 64) {   // four chunks' scores in flight together (a lane past the end re-reads the last key)
      float sc[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) sc[u] = scores[min(jb0 + 64 * u + lane, ng - 1)];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int jb = jb0 + 64 * u;
        if (jb >= e) break;   // (uniform)
        const int j = jb + lane;
        const bool use = j < e;
        nvalid += __popcll(__ballot(use));
        compare(jb >> 6, use, sc[u], j);
      }
    }
  } else {
    // Dense form with a validity mask: the group's valid keys are a SUBSET of the shared array (multi-class NMS: class c's 4096
    // candidates among 12 288 boxes).  Round 6 (profiles/r06_nms_batched.txt): 25.6 -> 16.9 us for 3 x 4096 of 12 288 —
    //  * the chunks are dealt INTERLEAVED (wave w: chunks w, w + nw, ...: any run of valid boxes spreads over all waves) and
    //    eight at a time, all sixteen loads independent: one memory round trip per eight chunks where the contiguous sixteenth
    //    per wave paid two per chunk (flag byte, then score) and left ten of sixteen waves without a valid key;
    //  * the launch uses FOUR waves per workgroup in this form: two thirds of the workgroups have no valid box of their group and
    //    leave after one byte load, but every wave of theirs costs dispatch (sweep: 128 threads 23.5 us, 256: 16.9, 512: 16.4,
    //    1024: 22.8; without the 32-bit compares below the busy workgroups were VALU-bound and four waves gained nothing);
    //  * a chunk without a valid key is skipped.
    constexpr int RU = 8;
    const int nchunk = (ng + 63) >> 6;
    for (int c0 = wave; c0 < nchunk; c0 += nw * RU) {
      // sixteen independent loads (eight flag bytes, eight scores; a lane past the end re-reads the group's last key): ONE memory
      // round trip per eight chunks — a score load that waits for its flag serialises the chunks (measured: 0.7 us per chunk)
      unsigned char fb[RU];
      float sc[RU];
#pragma unroll
      for (int u = 0; u < RU; ++u) {
        const int j = min(((c0 + nw * u) << 6) + lane, ng - 1);
        fb[u] = valid[j];
        sc[u] = scores[j];
      }
      bool use[RU];
      unsigned long long usem[RU];
#pragma unroll
      for (int u = 0; u < RU; ++u) {
        use[u] = ((c0 + nw * u) << 6) + lane < ng && fb[u] != 0;
        usem[u] = __ballot(use[u]);
      }
#pragma unroll
      for (int u = 0; u < RU; ++u) {
        nvalid += __popcll(usem[u]);
        if (usem[u] == 0ull) continue;   // (uniform)
        compare(c0 + nw * u, use[u], sc[u], ((c0 + nw * u) << 6) + lane);
      }
    }
  }
  cnt += 16 * n32 - lt;         // chunks compared through the 32-bit images: 16 keys each per lane, less those below the threshold
  cnt += __shfl_xor(cnt, 16);   // the four rows hold the same boxes
  cnt += __shfl_xor(cnt, 32);
  if (lane < 16) spart[wave][lane] = cnt;
  if (lane == 0) svalid[wave] = nvalid;
  __syncthreads();
  if (wave != 0) return;
  if (counts != nullptr && blockIdx.x == 0 && lane == 0) {
    int total = 0;
    for (int w = 0; w < nw; ++w) total += svalid[w];
    counts[g] = min(total, n_keep);
  }
  if (lane < 16 && i < ng && (valid == nullptr || valid[i] != 0)) {
    int r = 0;
    for (int w = 0; w < nw; ++w) r += spart[w][lane];
    if (r < n_keep) {
      const int bbase = seg != nullptr ? sbase : (gps > 0 ? (g / gps) * n : 0);
      order_[(size_t)g * n_keep + r] = (long long)(bbase + i);
      if (PREP) {
        float bx[5];
#pragma unroll
        for (int k = 0; k < 5; ++k) bx[k] = boxes[(size_t)(bbase + i) * 5 + k];
        OBox o;
        obox_make(bx, o);
        ob_[(size_t)g * n_keep + r] = o;
      }
    }
  }
}

// Victim lists of the LIST scan (round 5; one group of known size, n <= LIST_MAX_N: nms_list_body below).  Per box i two lists of
// 16-bit ids of boxes of LATER 64-blocks whose IoU with i exceeds the threshold, in any order, with a counter each (nothing is
// initialised but the counters; readers mask by the count):
//   near list: victims in the next LIST_K blocks (<= LIST_NEAR entries): marked by the scan's resolver wave itself;
//   far list : victims beyond (<= LIST_FAR entries): marked by helper waves.
constexpr int LIST_K = 8;   // (measured, scan kernel: n = 4096 thr 0.25: 17.9 / 16.5 / 16.2 / 16.4 us at 4 / 6 / 8 / 12; n = 9000 thr 0.7: 30.7 / 29.7 / 28.8 at 4 / 8 / 12)
constexpr int LIST_NEAR = 16;
constexpr int LIST_FAR = 64;
constexpr unsigned LIST_MAX_N = 16384;          // the list scan keeps one state BYTE per box in LDS
constexpr unsigned short LIST_DUMMY = 0x4040u;  // first of 64 scratch state bytes (never boxes), four bytes apart, one per lane
struct QueueArgs {
  unsigned* queue;   // (G, QUEUE_SHARDS, scap)
  unsigned* ctl;     // (G, CTL_WORDS): [s * CTL_STRIDE] entries reserved in shard s (may exceed scap); [QUEUE_SHARDS * CTL_STRIDE] overflowed block pairs
  unsigned* ovl;     // (G, npairs) overflowed block pair ids
  unsigned scap, npairs;   // scap: entries per shard
  unsigned short* lists;   // per group: (cap, LIST_NEAR) near lists, then (cap, LIST_FAR) far lists — or nullptr: no lists wanted
  unsigned* lcnt;          // per group a block of `lblock` words: (cap, 2) entries appended per box to its near / far list (may exceed the
                           // capacity: the list is then incomplete), then the group's FAILURE WORD (+ padding): set when the lists cannot
                           // be used — a full list, or block pairs that went to the overflow list
  unsigned lblock;
};
__device__ __forceinline__ unsigned* list_counts(const QueueArgs& q, int g) { return q.lcnt + (size_t)g * q.lblock; }
__device__ __forceinline__ unsigned* list_fail(const QueueArgs& q, const NmsArgs&, int g) { return list_counts(q, g) + (q.lblock - 64u); }

// one more entry of box i's near or far victim list (j: a box of a LATER 64-block that i suppresses if i is kept); `pos` from the
// counter (the caller's atomicAdd: per pair, or wave-aggregated)
__device__ __forceinline__ void list_put(const QueueArgs& q, const NmsArgs& a, int g, int i, int j, bool far, unsigned pos) {
  unsigned short* const glists = q.lists + (size_t)g * a.cap * (LIST_NEAR + LIST_FAR);
  if (far) {
    if (pos < (unsigned)LIST_FAR) glists[(size_t)a.cap * LIST_NEAR + (size_t)i * LIST_FAR + pos] = (unsigned short)j;
    else *list_fail(q, a, g) = 1u;   // the list is incomplete: the list scan must not run (the classic scan does)
  } else {
    if (pos < (unsigned)LIST_NEAR) glists[(size_t)i * LIST_NEAR + pos] = (unsigned short)j;
    else *list_fail(q, a, g) = 1u;
  }
}


// Axis-aligned and circle NMS (cheap predicates, no polygon scratch): one WAVE per (row box i, 64-box column block
// c >= block of i): lane l tests box i against box 64c + l and the wave-wide ballot IS the 64-bit mask word — no partial
// words, no barrier.  A wave walks `rows` (1, 2, 4 or 8; host-chosen) consecutive row boxes against the same 64 column
// boxes (loaded once): 1 keeps small problems latency-short, 8 keeps large ones from being workgroup-dispatch bound;
// blockIdx.x = (upper-triangle block pair) * (64 / rows) + row group; blockIdx.y = group.
// On a DIAGONAL block the lanes left of the row box are not idle: lane j < i evaluates the same predicate with the
// operands in greedy order (box j first, box i second — bit for bit what row j's wave computes for its lane i), so the
// ballot also yields "which earlier boxes of my block suppress box i".  That word goes to colm[i]; the scan resolves a
// 64-box block from these column words in a few wave-parallel steps instead of one scalar step per kept box.
// (Rotated boxes went through this kernel too until the compacted form below replaced it: n = 4096 99 -> 29 us,
// n = 9000 293 -> 97 us, n = 1000 25 -> 21 us, same mask bits.)
template <int MODE>
__global__ __launch_bounds__(64) void nms_mask_kernel(const NmsArgs a, const OBox* __restrict__ ob_,
                                                      unsigned long long* __restrict__ mask_,
                                                      unsigned long long* __restrict__ colm_, const QueueArgs q) {
  static_assert(MODE == MODE_NORMAL || MODE == MODE_CIRCLE, "rotated boxes: nms_mask_compact_kernel");
  const int lane = threadIdx.x;
  const int g = blockIdx.y;
  const int n = group_n(a, g);
  const int cb = (n + 63) >> 6;
  const int rows = a.rows;
  const int groups = 64 / rows;
  const unsigned pair = blockIdx.x / groups;
  if (pair >= (unsigned)(cb * (cb + 1) / 2)) return;  // grid is sized for `cap`
  const int r0 = (int)(blockIdx.x % groups) * rows;
  const long long* order = a.order != nullptr ? a.order + (size_t)g * a.cap : nullptr;
  unsigned long long* mask = mask_ + (size_t)g * a.cap * a.cbs;
  const float thresh = a.thresh_dev != nullptr ? a.thresh_dev[g] : a.thresh;
  const double thresh_d = a.thresh_dev != nullptr ? (double)a.thresh_dev[g] : a.thresh_d;
  // pair -> (rb, c): pairs before row block rb: rb*cb - rb(rb-1)/2
  int rb = (int)((2.0f * cb + 1.0f - sqrtf((2.0f * cb + 1.0f) * (2.0f * cb + 1.0f) - 8.0f * (float)pair)) * 0.5f);
  rb = max(0, min(rb, cb - 1));
  while (rb > 0 && (unsigned)(rb * cb - rb * (rb - 1) / 2) > pair) --rb;
  while ((unsigned)((rb + 1) * cb - (rb + 1) * rb / 2) <= pair) ++rb;
  const int c = rb + (int)(pair - (unsigned)(rb * cb - rb * (rb - 1) / 2));
  const int j = c * 64 + lane;
  float braw[5];
  if (j < n) {
    const size_t sj = order != nullptr ? (size_t)order[j] : (size_t)j;
    if constexpr (MODE == MODE_NORMAL) {
#pragma unroll
      for (int k = 0; k < 5; ++k) braw[k] = a.boxes[sj * 5 + k];
    } else {
      braw[0] = a.boxes[sj * 2];
      braw[1] = a.boxes[sj * 2 + 1];
    }
  }
  unsigned hits = 0u;   // bit r: this lane's box is a hit of row r (rows <= 8)
  int total = 0;        // lane r: hits of row r
  for (int r = 0; r < rows; ++r) {
    const int i = rb * 64 + r0 + r;  // wave-uniform
    if (i >= n) break;
    const bool act = j < n && j != i;   // (off-diagonal blocks: j > i always)
    const bool low = j < i;             // diagonal block only: lane box precedes the row box -> it goes first
    bool hit = false;
    if constexpr (MODE == MODE_NORMAL) {
      if (act) {
        const size_t si = order != nullptr ? (size_t)order[i] : (size_t)i;
        float ar[5];
#pragma unroll
        for (int k = 0; k < 5; ++k) ar[k] = a.boxes[si * 5 + k];
        hit = (low ? iou_normal(braw, ar) : iou_normal(ar, braw)) > thresh;
      }
    } else {
      if (act) {  // mmdet3d circle_nms: dist = (x_i - x_j)^2 + (y_i - y_j)^2 ; suppressed iff dist <= thresh
        const size_t si = order != nullptr ? (size_t)order[i] : (size_t)i;
        const float xi = a.boxes[si * 2], yi = a.boxes[si * 2 + 1];
        const float dx = low ? braw[0] - xi : xi - braw[0], dy = low ? braw[1] - yi : yi - braw[1];
        const float dist = dx * dx + dy * dy;
        hit = (double)dist <= thresh_d;
      }
    }
    const unsigned long long word = __ballot(hit);
    if (lane == 0) {
      if (rb == c) {
        const int il = i & 63;
        const unsigned long long below = (1ull << il) - 1ull;
        mask[(size_t)i * a.cbs + c] = word & ~(below | (1ull << il));
        colm_[(size_t)g * a.cap + i] = word & below;
      } else {
        mask[(size_t)i * a.cbs + c] = word;
      }
    }
    // the list scan's victim lists: remembered per row (bit r of `hits`, the row's hit count in lane r), appended after the loop
    if (r < 8) {
      hits |= hit ? (1u << r) : 0u;
      if (lane == r) total = __popcll(word);
    }
  }
  // the list scan's victim lists (q.lists: counters and failure word were zeroed before this kernel): the hits of a LATER block go
  // to the row boxes' near or far lists.  ONE returning atomic for all of the wave's rows (lane r reserves row r's entries): a
  // counter update per row inside the loop above put a memory round trip between the rows (20 -> 25 us at n = 4096)
  if (q.lists != nullptr && rb != c) {   // uniform
    const bool far = c - rb > LIST_K;
    const int irow = rb * 64 + r0 + lane;   // lane r: row r of this wave
    unsigned base = 0u;
    if (lane < min(rows, 8) && total > 0) base = atomicAdd(&list_counts(q, g)[2 * irow + (far ? 1 : 0)], (unsigned)total);
    if (__ballot(hits != 0u) != 0ull) {
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        if (r < rows) {
          const unsigned long long word = __ballot((hits >> r) & 1u);
          const unsigned b = (unsigned)__builtin_amdgcn_readlane((int)base, r);
          if ((hits >> r) & 1u) list_put(q, a, g, rb * 64 + r0 + r, j, far, b + (unsigned)__popcll(word & ((1ull << lane) - 1ull)));
        }
      }
    }
  }
}

// Rotated mode, compacted: the same mask words, with the expensive lanes packed densely.
// In nms_mask_kernel a wave pays a whole polygon-clipping pass whenever ANY of its 64 lanes survives the bounding-circle
// test; on score-sorted detector output ~1 % of the pairs do, i.e. about every second (row, 64 columns) wave runs a pass
// with one or two live lanes.  Here a wave owns `rows` (8..64) consecutive row boxes x one 64-box column block and works
// in two phases per 16-row chunk:
//   1. the circle test alone for every (row, lane) pair (~12 VALU per row, straight-line; the row boxes' centre / extent
//      are broadcast from lanes by v_readlane): every lane keeps the 16-bit candidate mask of ITS column, the survivors
//      are then appended to an LDS queue as (row << 6 | column) in bulk (DPP scan of the per-lane counts);
//   2. whenever >= 64 candidates are queued (and once more at the end) lane l takes candidate l: loads both 64-byte
//      records, runs the FULL predicate (iou_bev, which repeats the circle test — one code path, bit-identical
//      decisions) and ORs its bit into the row's word in LDS.  Every clipping pass but the last has 64 live lanes.
// The circle test is symmetric in its operands (squared differences, commutative sums), so on a DIAGONAL block it also
// selects the (earlier box, row box) pairs that are evaluated in greedy operand order for colm[] — as in the plain kernel.
// A pair that fails the circle test has overlap exactly 0 and IoU +0, which is "> thresh" only for thresh < 0: for such a
// threshold (or a NaN one) every valid pair is queued, so the result stays that of the plain kernel.
constexpr int CQ_ROWS = 16;                 // rows per chunk between drains (<= 32: one bit per row in a lane's mask)
constexpr int CQ_CAP = CQ_ROWS * 64 + 64;   // worst case of one chunk + the carried remainder (< 64)

// block pair index of the upper triangle (row-major over row blocks) -> (row block rb, column block c >= rb)
__device__ __forceinline__ void pair_blocks(unsigned pair, int cb, int& rb, int& c) {
  rb = (int)((2.0f * cb + 1.0f - sqrtf((2.0f * cb + 1.0f) * (2.0f * cb + 1.0f) - 8.0f * (float)pair)) * 0.5f);
  rb = max(0, min(rb, cb - 1));
  while (rb > 0 && (unsigned)(rb * cb - rb * (rb - 1) / 2) > pair) --rb;
  while ((unsigned)((rb + 1) * cb - (rb + 1) * rb / 2) <= pair) ++rb;
  c = rb + (int)(pair - (unsigned)(rb * cb - rb * (rb - 1) / 2));
}

struct CompactLds {
  VertexScratch<64> vs;
  unsigned short queue[CQ_CAP];
  unsigned long long words[64];
};

// one wave: `rows` row boxes (from row r0 of row block rb) x column block c of group g -> final mask words (and colm on a
// diagonal block).  The whole job of nms_mask_compact_kernel for one workgroup; also the overflow path of the queued form.
__device__ __forceinline__ void compact_pair(const NmsArgs& a, const OBox* __restrict__ ob, unsigned long long* __restrict__ mask,
                                             unsigned long long* __restrict__ colm, int n, int rb, int c, int r0, int rows,
                                             float thresh, CompactLds& L) {
  VertexScratch<64>& vs = L.vs;
  unsigned short* const queue = L.queue;
  unsigned long long* const words = L.words;
  const int lane = threadIdx.x & 63;
  const bool all_pairs = !(thresh >= 0.0f);
  const int i0 = rb * 64 + r0;  // first row box of this wave
  if (i0 >= n) return;
  const int nrows = min(rows, n - i0);
  const int j = c * 64 + lane;
  const bool jv = j < n;
  float bcx = 0.0f, bcy = 0.0f, bext = 0.0f;
  if (jv) {
    const OBox& B = ob[j];
    bcx = B.cx;
    bcy = B.cy;
    bext = fabsf(B.x2 - B.x1) + fabsf(B.y2 - B.y1);
  }
  // lane r also holds row box i0 + r's centre / extent: the row loop reads them with v_readlane instead of one
  // dependent scalar-load round trip per row (64 rows x ~500 cycles was most of phase 1)
  float rcx = 0.0f, rcy = 0.0f, rext = 0.0f;
  if (lane < nrows) {
    const OBox& R = ob[i0 + lane];
    rcx = R.cx;
    rcy = R.cy;
    rext = fabsf(R.x2 - R.x1) + fabsf(R.y2 - R.y1);
  }
  words[lane] = 0ull;
  __syncthreads();
  int qn = 0;  // queued candidates (wave-uniform)
  for (int rbase = 0; rbase < nrows; rbase += CQ_ROWS) {
    const int rend = min(rbase + CQ_ROWS, nrows);
    // circle tests of the chunk, straight-line: lane l (column box j) tests itself against the chunk's 16 row boxes
    // (centre / extent broadcast by v_readlane) and keeps ITS OWN 16-bit candidate mask — no ballot, no branch, no LDS
    // in the loop, rows independent of each other.  (A per-row ballot + divergent queue append ran at ~310 cycles per
    // row for a lone wave — mixed SALU/VALU dependencies and three branches per row — half of a wave's life.)
    unsigned colbits = 0u;
    const int jdiag = j - i0 - rbase;  // lane's column box IS row box (rbase + k)  <=>  k == jdiag
#pragma unroll
    for (int k = 0; k < CQ_ROWS; ++k) {
      const int r = rbase + k;  // < 64 always; rows >= nrows are masked off below
      const float acx = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(rcx), r));
      const float acy = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(rcy), r));
      const float aext = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(rext), r));
      // box_overlap's early-out, same operations (it is symmetric in the two boxes)
      const float ddx = acx - bcx, ddy = acy - bcy;
      const float reach = 0.5f * (aext + bext) + 1e-2f;
      const bool near = !(ddx * ddx + ddy * ddy > reach * reach * 1.0001f);
      colbits |= (near && k != jdiag) ? (1u << k) : 0u;
    }
    if (all_pairs) colbits = ~(jdiag >= 0 && jdiag < CQ_ROWS ? (1u << jdiag) : 0u);
    colbits &= (rend - rbase >= 32) ? 0xffffffffu : ((1u << (rend - rbase)) - 1u);
    if (!jv) colbits = 0u;
    // queue append in bulk: inclusive scan of the 64 per-lane counts on the DPP network, then every lane walks the set
    // bits of its own mask (a handful at detector densities)
    const int cntl = __popc(colbits);
    int incl = cntl;
    incl += __builtin_amdgcn_update_dpp(0, incl, 0x111, 0xf, 0xf, true);   // row_shr:1 (out-of-row reads 0)
    incl += __builtin_amdgcn_update_dpp(0, incl, 0x112, 0xf, 0xf, true);   // row_shr:2
    incl += __builtin_amdgcn_update_dpp(0, incl, 0x114, 0xf, 0xf, true);   // row_shr:4
    incl += __builtin_amdgcn_update_dpp(0, incl, 0x118, 0xf, 0xf, true);   // row_shr:8
    incl += __builtin_amdgcn_update_dpp(0, incl, 0x142, 0xa, 0xf, false);  // row_bcast:15 -> rows 1, 3
    incl += __builtin_amdgcn_update_dpp(0, incl, 0x143, 0xc, 0xf, false);  // row_bcast:31 -> rows 2, 3
    {
      unsigned w = colbits;
      int pos = qn + incl - cntl;
      while (w != 0u) {
        const int k = __builtin_ctz(w);
        w &= w - 1u;
        queue[pos++] = (unsigned short)(((rbase + k) << 6) | lane);
      }
    }
    qn += __builtin_amdgcn_readlane(incl, 63);
    __syncthreads();
    const bool last = rend >= nrows;
    int done = 0;
    while (qn - done >= 64 || (last && done < qn)) {
      const int q = done + lane;
      if (q < qn) {
        const int e = queue[q];
        const int r = e >> 6, jl = e & 63;
        const int i = i0 + r, jj = c * 64 + jl;
        const OBox A = ob[i];
        const OBox B = ob[jj];
        const bool low = jj < i;  // diagonal block only: the lane's box precedes the row box -> it goes first
        const OBox F = low ? B : A, S = low ? A : B;
        const bool hit = iou_bev<64>(F, S, vs, lane) > thresh;
        if (hit) atomicOr(&words[r], 1ull << jl);
      }
      done += 64;
    }
    if (!last && done > 0) {  // carry the < 64 leftover candidates to the front of the queue
      const int rem = qn - done;
      unsigned short v = 0;
      if (lane < rem) v = queue[done + lane];
      __syncthreads();
      if (lane < rem) queue[lane] = v;
      qn = rem;
    }
    __syncthreads();
  }
  if (lane < nrows) {
    const int i = i0 + lane;
    const unsigned long long word = words[lane];
    if (rb == c) {
      const int il = i & 63;
      const unsigned long long below = (1ull << il) - 1ull;
      mask[(size_t)i * a.cbs + c] = word & ~(below | (1ull << il));
      colm[i] = word & below;
    } else {
      mask[(size_t)i * a.cbs + c] = word;
    }
  }
}

__global__ __launch_bounds__(64) void nms_mask_compact_kernel(const NmsArgs a, const OBox* __restrict__ ob_,
                                                              unsigned long long* __restrict__ mask_,
                                                              unsigned long long* __restrict__ colm_) {
  __shared__ CompactLds L;
  const int g = blockIdx.y;
  const int n = group_n(a, g);
  const int cb = (n + 63) >> 6;
  const int rows = a.rows;
  const int groups = 64 / rows;
  const unsigned pair = blockIdx.x / groups;
  if (pair >= (unsigned)(cb * (cb + 1) / 2)) return;  // grid is sized for `cap`
  int rb, c;
  pair_blocks(pair, cb, rb, c);
  const float thresh = a.thresh_dev != nullptr ? a.thresh_dev[g] : a.thresh;
  compact_pair(a, ob_ + (size_t)g * a.cap, mask_ + (size_t)g * a.cap * a.cbs, colm_ + (size_t)g * a.cap, n, rb, c,
               (int)(blockIdx.x % groups) * rows, rows, thresh, L);
}

// ---- Rotated mode, QUEUED (round 4): circle test and clipping as two kernels, every clipping pass full. -----------------------
// In the compacted kernel above every wave ends with one partly filled clipping pass (~1 % of a 64 x 64 block pair's 4096
// pairs survive the circle test: ~41 live lanes of 64) and the kernel lasts as long as its slowest waves (pairs with > 64
// survivors run two passes, the diagonal blocks evaluate every pair twice): profiles/r04_nms_pmc.txt.  Here
//   nms_circle_queue_kernel  one wave per block pair: zeroes the pair's mask words, runs ONLY the circle tests and appends the
//                            survivors (i << 16 | j, i < j) to a per-group queue in HBM (one wave-aggregated atomicAdd);
//                            a diagonal block queues every unordered pair ONCE (the compacted kernel evaluates it twice,
//                            as row i / lane j and as row j / lane i, with the same operand order and the same result);
//   nms_clip_queue_kernel    a fixed grid of waves walks the queue 64 entries at a time: lane l evaluates entry l with the
//                            full predicate and ORs its bit into mask[i][j / 64] — and, inside a diagonal block, into
//                            colm[j] as well (integer atomics: the words are the same whatever the order).
// Same predicate, same operand order (earlier box first), same bits.  A wave whose survivors do not fit the queue
// (128 entries per box over 64 shards; only pathological clouds get there) records its block pair instead and the clip kernel runs
// compact_pair() on it afterwards.
constexpr unsigned QUEUE_SENTINEL = 0xffffffffu;   // (65535, 65535): never a queued pair (i < j)
// The queue is cut into QUEUE_SHARDS sub-queues (block pair p appends to shard p % QUEUE_SHARDS), each with its own counter in
// its own 128-byte line: returning atomics on ONE word saturate at ~88 per us on this chip (MI355X_MICROARCH.md, "dequeue"), and
// the 2080 appends of an n = 4096 call through one counter cost 24 us — more than the clipping they were meant to feed.
constexpr unsigned QUEUE_SHARDS = 64;
constexpr unsigned CTL_STRIDE = 32;                                  // words: one 128-byte line per counter
constexpr unsigned CTL_WORDS = (QUEUE_SHARDS + 1) * CTL_STRIDE;      // per group: shard counters, then the overflow counter
__global__ __launch_bounds__(64) void nms_circle_queue_kernel(const NmsArgs a, const OBox* __restrict__ ob_,
                                                              unsigned long long* __restrict__ mask_,
                                                              unsigned long long* __restrict__ colm_, const QueueArgs q) {
  const int lane = threadIdx.x;
  const int g = blockIdx.y;
  const int n = group_n(a, g);
  const int cb = (n + 63) >> 6;
  const unsigned pair = blockIdx.x;
  if (pair >= (unsigned)(cb * (cb + 1) / 2)) return;  // grid is sized for `cap`
  int rb, c;
  pair_blocks(pair, cb, rb, c);
  const OBox* ob = ob_ + (size_t)g * a.cap;
  unsigned long long* mask = mask_ + (size_t)g * a.cap * a.cbs;
  const int i0 = rb * 64;
  const int nrows = min(64, n - i0);
  const int j = c * 64 + lane;
  const bool jv = j < n;
  float bcx = 0.0f, bcy = 0.0f, bext = 0.0f;
  if (jv) {
    const OBox& B = ob[j];
    bcx = B.cx;
    bcy = B.cy;
    bext = fabsf(B.x2 - B.x1) + fabsf(B.y2 - B.y1);
  }
  float rcx = 0.0f, rcy = 0.0f, rext = 0.0f;
  if (lane < nrows) {
    const OBox& R = ob[i0 + lane];
    rcx = R.cx;
    rcy = R.cy;
    rext = fabsf(R.x2 - R.x1) + fabsf(R.y2 - R.y1);
    mask[(size_t)(i0 + lane) * a.cbs + c] = 0ull;                 // the clip kernel ORs into these
    if (rb == c) {
      colm_[(size_t)g * a.cap + i0 + lane] = 0ull;
      if (q.lists != nullptr)   // the list scan's victim lists of this box: empty (the clip kernel appends)
        reinterpret_cast<uint2*>(list_counts(q, g))[i0 + lane] = make_uint2(0u, 0u);
    }
  }
  if (q.lists != nullptr && pair == 0 && lane == 0) *list_fail(q, a, g) = 0u;
  // circle tests, straight-line: lane l (column box j) against the 64 row boxes; bit r of `cand` = the pair (row i0 + r, column j)
  // survives.  On the diagonal block only the pairs with the column box AFTER the row box.  TWO ROWS PER INSTRUCTION (float2 ->
  // v_pk_add / v_pk_mul: the same IEEE operations per component as box_overlap's early-out, in its order), and the mask built by
  // shifting the compare's result in as a carry (w = w + w + carry: one instruction per row; rows descend so that row r ends in
  // bit r).  Left to itself the compiler packed the x / y components of ONE row and repacked between rows: 953 VALU instructions
  // per wave, more than the clipping kernel's 693 (profiles/r05_nms_pmc_counters.txt).
  typedef float f2 __attribute__((ext_vector_type(2)));
  const f2 bcx2 = {bcx, bcx}, bcy2 = {bcy, bcy}, bext2 = {bext, bext};
  auto shift_in = [](unsigned w, unsigned long long carry) -> unsigned {
    unsigned out;
    unsigned long long co;
    asm("v_addc_co_u32_e64 %0, %1, %2, %2, %3" : "=v"(out), "=s"(co) : "v"(w), "s"(carry));
    return out;
  };
  // the row boxes' (cx, cy, extent) go through LDS, laid out per PAIR of rows as the packed operands want them — [cx of row r + 1,
  // cx of row r, cy.., cy.., ext.., ext..] — and come back as broadcast reads (a uniform address: 2 LDS instructions per row pair
  // instead of 6 v_readlane, which are vector instructions: a quarter of the loop's)
  __shared__ __attribute__((aligned(16))) float srow[32][8];
  {
    float* const mypair = &srow[lane >> 1][1 - (lane & 1)];   // odd rows first
    mypair[0] = rcx;
    mypair[2] = rcy;
    mypair[4] = rext;
  }
  __syncthreads();
  unsigned wlo = 0u, whi = 0u;
#pragma unroll
  for (int r = 62; r >= 0; r -= 2) {   // rows r + 1 and r
    const float4 xy = *reinterpret_cast<const float4*>(&srow[r >> 1][0]);
    const float2 ex = *reinterpret_cast<const float2*>(&srow[r >> 1][4]);
    const f2 acx = {xy.x, xy.y}, acy = {xy.z, xy.w}, aext = {ex.x, ex.y};
    const f2 ddx = acx - bcx2, ddy = acy - bcy2;   // box_overlap's early-out, same operations (symmetric in the boxes)
    const f2 d2 = ddx * ddx + ddy * ddy;
    const f2 reach = 0.5f * (aext + bext2) + 1e-2f;
    const f2 lim = reach * reach * 1.0001f;
    const unsigned long long n1 = __ballot(!(d2.x > lim.x)), n0 = __ballot(!(d2.y > lim.y));
    if (r >= 32) {
      whi = shift_in(whi, n1);
      whi = shift_in(whi, n0);
    } else {
      wlo = shift_in(wlo, n1);
      wlo = shift_in(wlo, n0);
    }
  }
  unsigned long long cand = ((unsigned long long)whi << 32) | (unsigned long long)wlo;
  cand &= nrows >= 64 ? ~0ull : ((1ull << nrows) - 1ull);
  if (rb == c) cand &= (1ull << lane) - 1ull;        // rows r < lane only: i = i0 + r < j = i0 + lane
  if (!jv) cand = 0ull;
  const int cntl = __popcll(cand);
  int incl = cntl;
  incl += __builtin_amdgcn_update_dpp(0, incl, 0x111, 0xf, 0xf, true);   // row_shr:1 (out-of-row reads 0)
  incl += __builtin_amdgcn_update_dpp(0, incl, 0x112, 0xf, 0xf, true);   // row_shr:2
  incl += __builtin_amdgcn_update_dpp(0, incl, 0x114, 0xf, 0xf, true);   // row_shr:4
  incl += __builtin_amdgcn_update_dpp(0, incl, 0x118, 0xf, 0xf, true);   // row_shr:8
  incl += __builtin_amdgcn_update_dpp(0, incl, 0x142, 0xa, 0xf, false);  // row_bcast:15 -> rows 1, 3
  incl += __builtin_amdgcn_update_dpp(0, incl, 0x143, 0xc, 0xf, false);  // row_bcast:31 -> rows 2, 3
  const int total = __builtin_amdgcn_readlane(incl, 63);
  unsigned* const ctl = q.ctl + (size_t)g * CTL_WORDS;
  unsigned* const novf = ctl + QUEUE_SHARDS * CTL_STRIDE;
  const float thresh = a.thresh_dev != nullptr ? a.thresh_dev[g] : a.thresh;
  if (!(thresh >= 0.0f)) {   // uniform: a negative or NaN threshold makes EVERY valid pair a candidate (IoU +0 > thresh):
    if (lane == 0) {   // compact_pair's all-pairs case
      const unsigned k = atomicAdd(novf, 1u);
      if (k < q.npairs) q.ovl[(size_t)g * q.npairs + k] = pair;
    }
    return;
  }
  if (total == 0) return;   // uniform
  const unsigned shard = pair % QUEUE_SHARDS;
  unsigned base = 0u;
  if (lane == 0) base = atomicAdd(&ctl[shard * CTL_STRIDE], (unsigned)total);
  base = (unsigned)__builtin_amdgcn_readfirstlane((int)base);
  unsigned* const queue = q.queue + ((size_t)g * QUEUE_SHARDS + shard) * q.scap;
  unsigned pos = base + (unsigned)(incl - cntl);
  if (base + (unsigned)total > q.scap) {   // uniform: does not fit -> this block pair goes to the overflow list,
    if (lane == 0) {   // what was reserved is voided
      const unsigned k = atomicAdd(novf, 1u);
      if (k < q.npairs) q.ovl[(size_t)g * q.npairs + k] = pair;
    }
    for (int k = 0; k < cntl; ++k, ++pos)
      if (pos < q.scap) queue[pos] = QUEUE_SENTINEL;
    return;
  }
  while (cand != 0ull) {
    const int r = __builtin_ctzll(cand);
    cand &= cand - 1ull;
    queue[pos++] = ((unsigned)(i0 + r) << 16) | (unsigned)j;
  }
}

__global__ __launch_bounds__(64) void nms_clip_queue_kernel(const NmsArgs a, const OBox* __restrict__ ob_,
                                                            unsigned long long* __restrict__ mask_,
                                                            unsigned long long* __restrict__ colm_, const QueueArgs q) {
  __shared__ CompactLds L;
  const int lane = threadIdx.x;
  const int g = blockIdx.y;
  const int n = group_n(a, g);
  if (n == 0) return;
  const OBox* ob = ob_ + (size_t)g * a.cap;
  unsigned long long* mask = mask_ + (size_t)g * a.cap * a.cbs;
  unsigned long long* colm = colm_ + (size_t)g * a.cap;
  const float thresh = a.thresh_dev != nullptr ? a.thresh_dev[g] : a.thresh;
  const unsigned* const ctl = q.ctl + (size_t)g * CTL_WORDS;
  // wave w serves shard w % QUEUE_SHARDS (the grid is a multiple of QUEUE_SHARDS waves), every (grid / QUEUE_SHARDS)-th chunk of it
  const unsigned shard = blockIdx.x % QUEUE_SHARDS, per_shard = gridDim.x / QUEUE_SHARDS;
  // device-scope atomic loads: the counters were produced by the atomics of the previous kernel; a plain (scalar-cache) load of
  // a word that the same graph's previous replay also read is not guaranteed to be refetched
  const unsigned reserved = __hip_atomic_load(&ctl[shard * CTL_STRIDE], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const unsigned count = reserved < q.scap ? reserved : q.scap;
  const unsigned* const queue = q.queue + ((size_t)g * QUEUE_SHARDS + shard) * q.scap;
  for (unsigned b0 = (blockIdx.x / QUEUE_SHARDS) * 64u; b0 < count; b0 += per_shard * 64u) {   // uniform bounds
    const unsigned e = (b0 + lane < count) ? queue[b0 + lane] : QUEUE_SENTINEL;
    const int ei = (int)(e >> 16), ej = (int)(e & 0xffffu);
    if (e != QUEUE_SENTINEL && ei < ej && ej < n) {   // (the bounds cannot fail for an entry this call queued: they fence off garbage)
      const int i = ei, j = ej;   // i < j: the earlier box goes first, as in the greedy order
      const OBox A = ob[i];
      const OBox B = ob[j];
      if (iou_bev<64>(A, B, L.vs, lane) > thresh) {
        atomicOr(&mask[(size_t)i * a.cbs + (j >> 6)], 1ull << (j & 63));
        if ((i >> 6) == (j >> 6)) {
          atomicOr(&colm[j], 1ull << (i & 63));
        } else if (q.lists != nullptr) {   // i suppresses j of a later block: one more entry of i's near or far victim list
          const bool far = (j >> 6) - (i >> 6) > LIST_K;
          list_put(q, a, g, i, j, far, atomicAdd(&list_counts(q, g)[2 * i + (far ? 1 : 0)], 1u));
        }
      }
    }
  }
  unsigned novf = __hip_atomic_load(&ctl[QUEUE_SHARDS * CTL_STRIDE], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (novf == 0u) return;
  if (q.lists != nullptr && lane == 0) *list_fail(q, a, g) = 1u;   // pairs served by compact_pair() below write mask words only: no lists
  const int cb = (n + 63) >> 6;
  const unsigned npairs_now = (unsigned)(cb * (cb + 1) / 2);
  novf = novf < npairs_now ? novf : npairs_now;
  for (unsigned k = blockIdx.x; k < novf; k += gridDim.x) {   // block pairs that did not fit the queue: the compacted form
    const unsigned op = q.ovl[(size_t)g * q.npairs + k];
    if (op >= npairs_now) continue;   // (cannot happen for an entry this call recorded)
    int rb, c;
    pair_blocks(op, cb, rb, c);
    __syncthreads();
    compact_pair(a, ob, mask, colm, n, rb, c, 0, 64, thresh, L);
  }
}

#ifdef SCAN_PROFILE
#define SCAN_STAMP(k) do { if (lane == 0) dbg[(size_t)c * 16 + (k)] = clock64(); } while (0)
#define SCAN_STAMP_SYNC(k) do { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); SCAN_STAMP(k); } while (0)
#else
#define SCAN_STAMP(k) do { } while (0)
#define SCAN_STAMP_SYNC(k) do { } while (0)
#endif

// wave-wide OR on the DPP network (row_shr 1/2/4/8 inside each row of 16, row_bcast 15 / 31 across rows; lane 63 holds
// the result): replaces up to 64 same-address ds_or_b64, which the LDS serialises.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ unsigned int dpp_or(unsigned int v) {
  return v | (unsigned int)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, 0xf, true);
}
__device__ __forceinline__ unsigned int wave_or_u32(unsigned int v) {
  v = dpp_or<0x111, 0xf>(v);
  v = dpp_or<0x112, 0xf>(v);
  v = dpp_or<0x114, 0xf>(v);
  v = dpp_or<0x118, 0xf>(v);
  v = dpp_or<0x142, 0xa>(v);
  v = dpp_or<0x143, 0xc>(v);
  return (unsigned int)__builtin_amdgcn_readlane((int)v, 63);
}
__device__ __forceinline__ unsigned long long wave_or_u64(unsigned long long v) {
  // the two halves interleaved: every DPP step has to wait for the VALU result before it (two wait states); with two independent
  // chains one half's step fills the other's wait
  unsigned int lo = (unsigned int)v, hi = (unsigned int)(v >> 32);
  lo = dpp_or<0x111, 0xf>(lo); hi = dpp_or<0x111, 0xf>(hi);
  lo = dpp_or<0x112, 0xf>(lo); hi = dpp_or<0x112, 0xf>(hi);
  lo = dpp_or<0x114, 0xf>(lo); hi = dpp_or<0x114, 0xf>(hi);
  lo = dpp_or<0x118, 0xf>(lo); hi = dpp_or<0x118, 0xf>(hi);
  lo = dpp_or<0x142, 0xa>(lo); hi = dpp_or<0x142, 0xa>(hi);
  lo = dpp_or<0x143, 0xc>(lo); hi = dpp_or<0x143, 0xc>(hi);
  return ((unsigned long long)(unsigned int)__builtin_amdgcn_readlane((int)hi, 63) << 32) |
         (unsigned long long)(unsigned int)__builtin_amdgcn_readlane((int)lo, 63);
}
// ---- greedy scan: one workgroup, phase-shifted waves, one LDS-only barrier per 64-box block ("interval") -------------
// Measured with the cycle-stamp build (tools/scan_profile.py): a global-memory round trip from this CU is ~2700 cycles,
// a resolved block ~1200.  So no wave may load and use a value inside one interval:
//   wave 0 (resolver) reads what the NEXT block waits for — colm[64t+l] and the first "urgent" word mask[64t+l][t+1] — from an
//     LDS ring that other waves filled one interval earlier.  It solves the block wave-parallel (kept = alive; kept' = alive &
//     ~ballot(col & kept) until stable: the unique solution of the triangular system the greedy order defines), publishes the
//     kept word and the compacted lane list, and carries the OR of the kept lanes' first urgent word to the next block in
//     registers (a full DPP reduction; round 5).
//   field waves (3, one per phase) fetch the resolver's inputs for block t0+3 (five fields per box), and in the interval in which
//     their loads fly run the SCRIBE step of block t0: kept ids to `keep`, urgent words 2 and 3 OR-ed into remv[t0+2], remv[t0+3],
//     the running count (round 5: until then all of that sat on the resolver's critical path).
//   waves 1..12 (3 groups x SCAN_GW row waves, group j phase-shifted by j intervals) run super-iterations of three intervals:
//     interval t0      ISSUE  : loads of the mask rows kept in block t0-1 (words >= t0+3; the group's waves split the rows);
//     interval t0+1    nothing (the loads are in flight across two barriers; straight-line code inside ONE loop
//                               iteration, so the compiler waits for them only at their first use);
//     interval t0+2    CONSUME: OR the rows into remv (ds_or_b64).
//   Block b's rows therefore reach remv[w >= b+4] during interval b+3, one barrier before block b+4 is resolved; words
//   b+1..b+3 are covered by the urgent words.  Every wave executes exactly cb barriers.
//   What bounds it (profiles/r05_nms_pmc.txt): a wave issues one instruction per ~10 cycles here (4 waves per SIMD, dependent
//   scalar/vector chains), and an interval lasts as long as its longest instruction stream: the row waves' ISSUE (~90
//   instructions for 4 rows), then the resolver (~60) and the CONSUME (~50).
// History: one scalar readlane step per kept box + load->use inside the interval: 1.2-3.8 us per block.
constexpr int SCAN_GW = 4;                          // row waves per propagate group
constexpr int SCAN_U = 16;                          // rows in flight per row wave: SCAN_GW x SCAN_U = 64 = every box of a block
constexpr int SCAN_T = 64 * (1 + 3 * SCAN_GW + 3);  // 1024 threads: resolver, 3 x 4 row waves, 3 field waves
constexpr int SCAN_NU = 3;                      // urgent words per box
constexpr int SCAN_RING = 4;

__device__ __forceinline__ unsigned int lds_offset(const void* p) {   // byte offset of a __shared__ object (ds_* address operand)
  return (unsigned int)(unsigned long long)(const __attribute__((address_space(3))) void*)p;
}

__device__ __forceinline__ void lds_barrier() {  // orders LDS only: global loads stay in flight across it
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

__device__ __forceinline__ unsigned long long uniform_u64(unsigned long long v) {
  // (the builtin returns a signed int: go through unsigned before widening)
  return ((unsigned long long)(unsigned int)__builtin_amdgcn_readfirstlane((int)(v >> 32)) << 32) |
         (unsigned long long)(unsigned int)__builtin_amdgcn_readfirstlane((int)(unsigned int)v);
}

// U rows x CH 64-word chunks in flight per propagate lane (registers: 2*U*CH VGPRs); rows / chunks beyond that are OR-ed
// in synchronously during the issue interval (correct, slower; only very dense keeps or n > 64*(64*CH+4)).
// Two-level form (win.gremv != nullptr; n > 8448): the box range is cut into super-blocks of SCAN_SB 64-box blocks.  One
// launch of this kernel resolves ONE super-block [c_begin, c_end): its removed-set starts from the global words gremv
// (what earlier super-blocks suppressed), rows are propagated only to words inside the super-block (<= SCAN_SB words per
// row: short loads, one chunk), the kept words of its blocks go to gkept, the running keep count lives in num_keep.
// nms_propagate_kernel then ORs the kept rows into gremv for all words right of the super-block with the whole chip.
struct ScanWindow {
  int c_begin, c_end;               // blocks; c_end is clamped to the group's block count
  unsigned long long* gremv;        // (G, cbs) global removed-set, nullptr = single-level scan over all blocks
  unsigned long long* gkept;        // (G, cbs) kept word per block
  long long* gcount;                // (G) running keep count between the launches of a two-level scan
};
constexpr int SCAN_SB = 64;         // blocks per super-block (4096 boxes): rows inside it fit the one-chunk scan variant

template <int U, int CH>
__device__ __forceinline__ void nms_scan_body(const NmsArgs& a, const unsigned long long* __restrict__ mask_,
                                              const unsigned long long* __restrict__ colm_,
                                              long long* __restrict__ keep_, long long* __restrict__ num_keep,
                                              long long* __restrict__ dbg, const ScanWindow& win) {
  extern __shared__ __attribute__((aligned(16))) unsigned long long remv[];  // cbs words
  __shared__ unsigned long long skept[4];
  __shared__ int scount;        // boxes kept before the block the next scribe step handles (handed from field wave to field wave)
  __shared__ int klist[4][64];  // lane indices of the boxes kept in a block, compacted (k-th kept box -> lane)
  __shared__ unsigned long long rin[SCAN_RING][2 + SCAN_NU][64];  // [slot][col, urgent 1..3, id][lane]
  const int g = blockIdx.x;  // one workgroup per group
  const int n = group_n(a, g);
  const int cb_all = (n + 63) >> 6;
  const bool windowed = win.gremv != nullptr;
  const int c_begin = windowed ? win.c_begin : 0;
  const int cb = windowed ? min(win.c_end, cb_all) : cb_all;   // every "< cb" below means "inside this launch's range"
  const size_t cbs = (size_t)a.cbs;
  unsigned long long* gremv = windowed ? win.gremv + (size_t)g * cbs : nullptr;
  unsigned long long* gkept = windowed ? win.gkept + (size_t)g * cbs : nullptr;
  if (windowed && c_begin >= cb_all) {                          // uniform: this group ends before the super-block
    // an EMPTY group (cb_all == 0) never reaches a resolver: the first launch records its count here, as the
    // single-level scan does (callers allocate num_keep uninitialised)
    if (c_begin == 0 && threadIdx.x == 0) num_keep[g] = 0;
    return;
  }
  const long long* order = a.order != nullptr ? a.order + (size_t)g * a.cap : nullptr;
  const unsigned long long* mask = mask_ + (size_t)g * a.cap * cbs;
  const unsigned long long* colm = colm_ + (size_t)g * a.cap;
  long long* keep = keep_ + (size_t)g * a.cap;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  for (int w = c_begin + tid; w < cb; w += SCAN_T) remv[w] = (windowed && c_begin > 0) ? gremv[w] : 0ull;
  if (windowed && c_begin == 0)   // the first super-block opens the global removed-set for everything right of it
    for (int w = cb + tid; w < cb_all; w += SCAN_T) gremv[w] = 0ull;
  if (tid < 4) skept[tid] = 0ull;
  // (windowed: the running count travels from launch to launch in the workspace word gcount[g]; num_keep[g] is written ONCE, by
  //  the launch that resolves the group's last block — a caller may point num_keep at pinned host memory and poll it)
  if (tid == 0) scount = (windowed && c_begin > 0) ? (int)win.gcount[g] : 0;
  lds_barrier();
  const int NB = cb - c_begin;  // intervals = barriers every wave executes in the main phase

  // resolver inputs of block B for this lane (0 where the box or the word does not exist)
  // field f of block B for this lane: 0 = column word, 1..3 = urgent words mask[i][B+f], 4 = box id.  (Contiguous
  // copies of the urgent words were tried: +7 us in the mask kernel at n = 4096, nothing gained here.)
  auto load_field = [&](int B, int f) -> unsigned long long {
    const int i = B * 64 + lane;
    const bool ok = B < cb && i < n;
    if (f == 0) return ok ? colm[i] : 0ull;
    if (f <= SCAN_NU) return (ok && B + f < cb) ? mask[(size_t)i * cbs + B + f] : 0ull;
    return (unsigned long long)((ok && order != nullptr) ? order[i] : (long long)i);
  };

  if (wave == 0) {
    // ---------------------------------------------------------------- resolver
    for (int B = c_begin; B < c_begin + 3; ++B) {  // the first three blocks: nobody runs ahead of them
#pragma unroll
      for (int f = 0; f < 2 + SCAN_NU; ++f) rin[B & (SCAN_RING - 1)][f][lane] = load_field(B, f);
    }
    // Round 5: the resolver keeps ONLY what the next block waits for.  Per block: the column word and the FIRST urgent word from
    // the ring, remv[c] from LDS, the fixed point, the kept word + compacted lane list published for the row waves, and the OR
    // of the kept lanes' first urgent word carried to the next block IN REGISTERS (a full DPP reduction: no LDS atomic and no
    // LDS round trip between two blocks).  Everything else a kept block owes — the kept ids to `keep`, the urgent words 2 and 3
    // into remv[c+2], remv[c+3], the running count — is done ONE INTERVAL LATER by the field wave that idles in that interval
    // (`scribe` below): off the critical path.  Until then the resolver's own stream was the scan's critical path at clustered
    // scenes (stamps, profiles/r05_nms_pmc.txt: lds 180 | solve 176 | ids + urgent ORs + lists 580 | barrier 116 of 1052 cycles).
    unsigned long long carry = 0ull;   // kept boxes of the previous block -> removed lanes of this one
    for (int c = c_begin; c < cb; ++c) {
      SCAN_STAMP(0);
      const int slot = c & (SCAN_RING - 1);
      const unsigned long long col = rin[slot][0][lane];
      unsigned long long urg1 = rin[slot][1][lane];
      const unsigned int clo = (unsigned int)col, chi = (unsigned int)(col >> 32);
      unsigned long long cur = uniform_u64(remv[c]) | carry;
      const int nvalid = min(64, n - c * 64);
      if (nvalid < 64) cur |= ~0ull << nvalid;
      const unsigned long long alive = ~cur;
      unsigned long long kept = alive;
      SCAN_STAMP(1);
      for (;;) {  // <= 65 rounds; the fixed point is the greedy keep set of the block
        const bool sup = ((clo & (unsigned int)kept) | (chi & (unsigned int)(kept >> 32))) != 0u;
        const unsigned long long nk = alive & ~__ballot(sup);
        if (nk == kept) break;
        kept = nk;
      }
      SCAN_STAMP(2);
      const bool mine = (kept >> lane) & 1ull;
      if (mine) klist[c & 3][__builtin_popcountll(kept & ((1ull << lane) - 1ull))] = lane;
      if (lane == 0) skept[c & 3] = kept;
      urg1 = mine ? urg1 : 0ull;
      carry = (c + 1 < cb) ? wave_or_u64(urg1) : 0ull;   // (uniform bound)
      SCAN_STAMP(3);
      lds_barrier();
      SCAN_STAMP(5);
    }
    if (NB == 0 && lane == 0) num_keep[g] = 0;   // an empty group: no block, no scribe step (a windowed launch returned above)
  } else {
    // ---------------------------------------------------------------- propagate / loader groups
    // Round 4: the resolver's inputs (five fields per box of block t + 3) are fetched by three FIELD waves of their own, one
    // per phase; the nine row waves only spread kept rows.  Until then rank 0 / 1 / 2 of a group also loaded two / two / one
    // field, and the group's issue interval (~250 dependent instructions at the 5-6 cycles a lone wave pays each) was as long
    // as the whole interval — the scan's critical stream together with the resolver (profiles/r04_nms_pmc.txt).
    const bool field_wave = wave > 3 * SCAN_GW;
    const int grp = field_wave ? wave - 1 - 3 * SCAN_GW : (wave - 1) / SCAN_GW;
    const int rank = field_wave ? 0 : (wave - 1) - grp * SCAN_GW;
    const int lead = min(grp, NB);
    const int S = (NB - lead) / 3;
    const int trail = NB - lead - 3 * S;
    for (int q = 0; q < lead; ++q) lds_barrier();
    if (field_wave) {
      // scribe: what block c owes beyond the resolver's critical path (see there), run by a field wave during the interval AFTER
      // block c was resolved — the one of its three intervals in which it used to wait for its loads.  The ring still holds the
      // block's fields (slot c & 3 is rewritten three intervals later), skept / klist are the resolver's, the count of boxes
      // kept so far travels from scribe to scribe through `scount`.
      auto scribe = [&](int c) {
        const int slot = c & (SCAN_RING - 1);
        const unsigned long long kept = uniform_u64(skept[c & 3]);
        unsigned long long urg[SCAN_NU - 1];
#pragma unroll
        for (int k = 0; k < SCAN_NU - 1; ++k) urg[k] = rin[slot][2 + k][lane];
        const long long id = (long long)rin[slot][1 + SCAN_NU][lane];
        const int count = __builtin_amdgcn_readfirstlane(scount);
        const bool mine = (kept >> lane) & 1ull;
        if (mine)  // with `order` the kept indices come out already mapped to the caller's box numbering
          keep[count + __builtin_popcountll(kept & ((1ull << lane) - 1ull))] = id;
        // the kept lanes' urgent words 2.. : OR-reduced inside every QUAD of lanes on the DPP network, then lanes 3, 7, ... 63 OR
        // their quad's totals into remv[c+2..] with ds_or_b64 (16 same-address LDS atomics per word; written out because an
        // atomicOr() here is rewritten into a readlane loop over the active lanes plus a scalar round trip)
#pragma unroll
        for (int k = 0; k < SCAN_NU - 1; ++k) urg[k] = mine ? urg[k] : 0ull;
        {
          unsigned int h[2 * (SCAN_NU - 1)];
#pragma unroll
          for (int k = 0; k < SCAN_NU - 1; ++k) {
            h[2 * k] = (unsigned int)urg[k];
            h[2 * k + 1] = (unsigned int)(urg[k] >> 32);
          }
#pragma unroll
          for (int k = 0; k < 2 * (SCAN_NU - 1); ++k) h[k] = dpp_or<0x111, 0xf>(h[k]);   // row_shr:1
#pragma unroll
          for (int k = 0; k < 2 * (SCAN_NU - 1); ++k) h[k] = dpp_or<0x112, 0xf>(h[k]);   // row_shr:2 -> lane 4q+3 holds quad q
#pragma unroll
          for (int k = 0; k < SCAN_NU - 1; ++k) urg[k] = ((unsigned long long)h[2 * k + 1] << 32) | h[2 * k];
        }
        if ((lane & 3) == 3) {
#pragma unroll
          for (int k = 0; k < SCAN_NU - 1; ++k)
            if (c + 2 + k < cb)   // (uniform bound)
              asm volatile("ds_or_b64 %0, %1" ::"v"(lds_offset(&remv[c + 2 + k])), "v"(urg[k]) : "memory");
        }
        if (lane == 0) {
          const int total = count + __builtin_popcountll(kept);
          scount = total;
          if (windowed) gkept[c] = kept;
          if (c == cb - 1) {
            if (windowed) win.gcount[g] = total;
            if (cb == cb_all) num_keep[g] = total;
          }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the ds_or_b64 above are inline asm: the compiler does not count them
      };
      for (int s2 = 0; s2 < S; ++s2) {
        const int t0 = c_begin + grp + 3 * s2;
        // ---- interval t0: issue the loads of block t0 + 3's inputs (constant field ids: a run-time id cost ~430 cycles per field)
        unsigned long long in[2 + SCAN_NU];
#pragma unroll
        for (int f = 0; f < 2 + SCAN_NU; ++f) in[f] = load_field(t0 + 3, f);
        lds_barrier();
        // ---- interval t0+1: the loads fly; block t0 was resolved in the interval before: its scribe step
        scribe(t0);
        lds_barrier();
        // ---- interval t0+2: into the ring (first USE of the loaded registers pinned here, see the row waves)
#pragma unroll
        for (int f = 0; f < 2 + SCAN_NU; ++f) asm volatile("" : "+v"(in[f]));
        if (t0 + 3 < cb) {
          const int slot = (t0 + 3) & (SCAN_RING - 1);
#pragma unroll
          for (int f = 0; f < 2 + SCAN_NU; ++f) rin[slot][f][lane] = in[f];
        }
        lds_barrier();
      }
      // the trailing intervals of this wave (no block left to fetch): interval tq = lead + 3 S + q; its scribe step falls on q == 1
      const int tq = c_begin + lead + 3 * S;
      for (int q = 0; q < trail; ++q) {
        if (q == 1) scribe(tq);
        lds_barrier();
      }
      // the LAST block was resolved in the last interval: its scribe step comes after the last barrier, from the field wave
      // whose turn it would be (block cb - 1 belongs to the phase of group (NB - 1) % 3)
      if (NB > 0 && grp == (NB - 1) % 3) scribe(cb - 1);
      return;
    }

    for (int s2 = 0; s2 < S; ++s2) {
      const int t0 = c_begin + grp + 3 * s2;
      [[maybe_unused]] const int c = t0;  // (SCAN_STAMP index)
      if (wave == 1) SCAN_STAMP(8);
      // ---- interval t0: issue
      const int bk = t0 - 1;             // block whose kept rows this group spreads
      const int first = t0 + SCAN_NU;    // = bk + 1 + SCAN_NU: first word not covered by the urgent words
      // both LDS reads of the interval issued together, unconditionally (one round trip instead of two back to back: the kept
      // word used to be read under the bounds test and waited for before the lane list was even requested; ~120 of the issue
      // interval's ~650 cycles).  bk & 3 is a valid slot even for bk = c_begin - 1; its stale content is masked right below.
      // (written out: left to the compiler the first read is waited for — its value feeds scalar code — before the second is issued)
      unsigned long long kbv;
      int myl;
      asm volatile("ds_read_b64 %0, %2\n\tds_read_b32 %1, %3\n\ts_waitcnt lgkmcnt(0)"
                   : "=&v"(kbv), "=&v"(myl)
                   : "v"(lds_offset(&skept[bk & 3])), "v"(lds_offset(&klist[bk & 3][(rank + SCAN_GW * lane) & 63]))
                   : "memory");
      unsigned long long kb = uniform_u64(kbv);
      if (!(bk >= c_begin && first < cb)) kb = 0ull;   // (uniform)
      // this wave's share: every SCAN_GW-th kept box, read from the compacted list the resolver left in LDS: lane u
      // fetches the row of slot u, the slots then cost a v_readlane + multiply + load each (a lone wave pays ~5 cycles
      // per instruction: walking the kept bits with ffbl / and / compare cost more than the memory round trip)
      const int cnt = __builtin_popcountll(kb);
      const int m = cnt > rank ? (cnt - rank + SCAN_GW - 1) / SCAN_GW : 0;  // rows of this wave (uniform)
      // (the lane-list read above is unconditional: a lane beyond m reads a stale or foreign slot that no readlane below ever
      //  selects; predicating the read on lane < m made it wait for the kept word's own LDS round trip first)
      if (wave == 1) SCAN_STAMP_SYNC(13);
      const unsigned long long* blk = mask + (size_t)(max(bk, c_begin) * 64) * cbs;
      // Loads are unconditional per lane: the word index is clamped into the row (w < cb is the same for every row of a
      // chunk, so the surplus lanes are masked ONCE, at consume time) — a per-row lane predicate cost ~100 cycles per
      // row in exec-mask handling.  (Leaving the registers of absent row pairs unwritten and guarding their use at consume time
      // was tried in round 4: the compiler then copies every loaded value at the end of its conditional block — a use right
      // behind the load, one memory round trip per pair: 500 cycles each.  The zero fill below is the cheap form.)
      unsigned long long v[U][CH];
      unsigned int wcl[CH];
#pragma unroll
      for (int ch = 0; ch < CH; ++ch) wcl[ch] = (unsigned int)min(first + ch * 64 + lane, cb - 1);
      static_assert(U % 2 == 0 && U * SCAN_GW >= 64, "rows are issued in pairs; a group's waves cover a whole block");
      const int mlast = max(m - 1, 0);
      const unsigned int myrow = (unsigned int)myl * (unsigned int)cbs;   // word offset of this lane's row inside the block (< 64 * 1024)
#pragma unroll
      for (int u = 0; u < U; u += 2) {  // pairs: half the uniform branches; an odd tail re-loads its last row (OR is idempotent).
        // (Fours were tried in round 5: n = 4096 clustered 52.4 -> 51.6 us, but the dense scenes lose more — 64.4 -> 65.6 us,
        //  n = 9000 147.0 -> 149.9 us: three clamped row indices per group instead of one per pair.)
#pragma unroll
        for (int ch = 0; ch < CH; ++ch) v[u][ch] = v[u + 1][ch] = 0ull;
        if (u < m) {  // uniform
          // row base as a UNIFORM pointer (scalar registers) + the lane's word as the vector offset: one readlane, one 64-bit
          // shift-add and the load per row (the row offset is multiplied out once per lane above, not once per row on the scalar
          // unit; adding it to the lane's word first made the whole address vector arithmetic: three VALU instructions per row)
          const unsigned long long* const r0 = blk + (unsigned int)__builtin_amdgcn_readlane((int)myrow, u);
          const unsigned long long* const r1 = blk + (unsigned int)__builtin_amdgcn_readlane((int)myrow, min(u + 1, mlast));
#pragma unroll
          for (int ch = 0; ch < CH; ++ch) {
            v[u][ch] = r0[wcl[ch]];
            v[u + 1][ch] = r1[wcl[ch]];
          }
        }
      }
      if (wave == 1) SCAN_STAMP(14);
      if (wave == 1) SCAN_STAMP(15);
      if (m > U || (m > 0 && first + 64 * CH < cb)) {  // overflow: finish it now, synchronously (rare)
        for (int w0 = first; w0 < cb; w0 += 64) {
          const int w = w0 + lane;
          unsigned long long acc = 0ull;
          for (int u = (w0 - first) < 64 * CH ? U : 0; u < m; ++u) {
            const unsigned int off = (unsigned int)__builtin_amdgcn_readlane((int)myrow, u);
            if (w < cb) acc |= blk[off + (unsigned int)w];
          }
          if (acc) atomicOr(&remv[w], acc);
        }
      }
      if (wave == 1) SCAN_STAMP(10);
      lds_barrier();
      // ---- interval t0+1: the loads fly
      lds_barrier();
      if (wave == 1) SCAN_STAMP(11);
      // ---- interval t0+2: consume.  The empty asm pins the first USE of every loaded register here: without it the
      // scheduler hoists the (pure VALU) OR tree above the two barriers and has to wait for the loads before them.
#pragma unroll
      for (int u = 0; u < U; ++u)
#pragma unroll
        for (int ch = 0; ch < CH; ++ch) asm volatile("" : "+v"(v[u][ch]));
#pragma unroll
      for (int ch = 0; ch < CH; ++ch) {
        unsigned long long acc = 0ull;
#pragma unroll
        for (int u = 0; u < U; ++u) acc |= v[u][ch];
        const int w = first + ch * 64 + lane;
        if (w < cb && acc) atomicOr(&remv[w], acc);  // ds_or_b64: waves merge into the same words
      }
      if (wave == 1) SCAN_STAMP(9);
      lds_barrier();
      if (wave == 1) SCAN_STAMP(12);
    }
    for (int q = 0; q < trail; ++q) lds_barrier();
  }
}

// the classic scan as a kernel of its own (single-level, or one super-block of the two-level form)
template <int U, int CH>
__global__ __launch_bounds__(SCAN_T) void nms_scan_kernel(const NmsArgs a, const unsigned long long* __restrict__ mask_,
                                                          const unsigned long long* __restrict__ colm_,
                                                          long long* __restrict__ keep_, long long* __restrict__ num_keep,
                                                          long long* __restrict__ dbg, const ScanWindow win) {
  nms_scan_body<U, CH>(a, mask_, colm_, keep_, num_keep, dbg, win);
}

// ---- LIST scan (round 5): the greedy scan on per-box VICTIM LISTS and one state BYTE per box in LDS ----------------------------------
// The classic scan above keeps the removed-set as bit words and PUSHES whole 512-byte mask rows of every kept box through twelve row
// waves; its interval is an instruction stream of ~90 (row waves) / ~60 (resolver) instructions, and a wave issues one instruction
// per ~8 cycles.  Here the clip kernel, which finds every pair (i < j, IoU > thr) anyway, also appends j to box i's near or far
// VICTIM LIST when j lies in a later 64-block (in-block pairs stay in colm), and the scan is:
//   state byte of box j (stb[j], LDS): 0 alive so far | 0x01 a kept earlier box suppresses it | after its block was resolved:
//                                      0x80 kept, 0x02 not kept;
//   block c, resolver wave:  alive = (stb == 0) per lane -> in-block fixed point over colm (skipped when no box of the block has an
//                            in-block candidate) -> kept; own byte := 0x80 / 0x02; kept lanes write 0x01 to their NEAR victims
//                            (blocks c + 1 .. c + LIST_K) — the LDS addresses of those bytes sit in a ring that helper waves filled
//                            long before, four per instruction pair;
//   helper wave of block c:  after the block is resolved, writes 0x01 to the FAR victims (blocks > c + LIST_K) of its kept boxes and
//                            sets fdone[c]; the resolver looks at fdone[c - LIST_K - 1] before it reads block c's bytes.
// What bounds it is the RESOLVER's instruction stream (~30 instructions per block on the usual path, one LDS round trip — its own
// state bytes — on the dependent chain), so everything that can be prepared is prepared by the helpers.  The resolver works in
// groups of four blocks, the body instantiated four times with the slot offsets as instruction offsets, and with two register sets
// (block c + 2's fields are fetched while block c is worked on).  No barrier in the loop: the workgroup synchronises through LDS
// words (one CU's LDS executes every wave's accesses in issue order, so "data, then flag" by the writer and "flag, then data" by the
// reader is enough; compiler fences keep the statements in that order):
//   rflag[slot] = (ring generation + 1) << 8 | (some column word non-zero) << 7 | near chunks (0..4),
//                 written by a helper AFTER the slot's fields; re-read by the resolver only if the block is not there yet;
//   stb[64 t]     polled by helper waves (s_sleep) for "block t resolved";   fdone[]  as above.
// Twelve helper waves (those that do not share the resolver's SIMD: waves w, w + 4, w + 8, w + 12 sit on one SIMD — HW_ID), wave g
// serving blocks t = g, g + 12, ...: issue the loads of block t's far list (as many uint4 as the block's longest far list needs: the
// counts were loaded one iteration earlier) and of block t + 16's near list / column word / id, wait for block t, far victims,
// fdone, count the kept boxes since its last block (the running count is the helper's own business), put block t + 16 into the
// ring slot block t just vacated (same wave, same iteration: no other ordering needed), write block t's kept ids.  The prologue
// fills the ring with all sixteen waves in one memory round trip.  Every polling loop is bounded (a bug must not hang the GPU): the
// scan is then marked failed and num_keep = -1.
// Two things this kernel is sensitive to, both measured (profiles/r05_nms_pmc.txt): (a) CODE SIZE — every launch starts with a cold
// instruction cache; a first build (resolver unrolled over all 16 slots, list loops unrolled: 61 KB) spent 2000-4000 cycles per
// block in its first pass; (b) the BYTES the helpers load — with one 128-byte list per box loaded twice per block the resolver ran at
// half speed although it never waited for a helper: hence near / far lists split by the clip kernel and far loads sized by count.
// Same greedy decisions by construction.  A full list or an overflowed block pair sets *lfail in the clip kernel: the workgroup then
// runs the CLASSIC scan instead (same launch: no second kernel).
// History (n = 9000, thr 0.7, scan kernels only): classic two-level ~100 us; pull formulation (suppressor lists, kept bits gathered)
// 66.8 us, 44.4 us once its field waves no longer kept loaded fields in SCRATCH memory (a select between two uint4 objects), 23.5 us
// with kept bytes, no barrier and an unrolled resolver — for thresholds >= 0.5 only; this push formulation serves every threshold.
constexpr int LIST_RING = 16;
constexpr int LIST_HW = 12;
constexpr int LIST_SPIN_MAX = 1 << 22;
#ifndef LIST_POLL_SLEEP
#define LIST_POLL_SLEEP 2
#endif

// volatile accesses that stay LDS instructions (a volatile access through a generic pointer becomes a flat_load / flat_store with an
// immediate wait)
typedef __attribute__((address_space(3))) unsigned int lds_u32;
typedef __attribute__((address_space(3))) unsigned char lds_u8;
__device__ __forceinline__ unsigned int lds_peek(const unsigned int* p) { return *(const volatile lds_u32*)p; }
__device__ __forceinline__ void lds_poke(unsigned int* p, unsigned int v) { *(volatile lds_u32*)p = v; }
// the byte at an LDS ADDRESS held in a register (ds_write_b8 vaddr, v: no base to add — the instruction's 16-bit offset field cannot
// reach an array the compiler placed beyond 64 KB)
__device__ __forceinline__ void lds_mark_at(unsigned int addr) { *(lds_u8*)(size_t)addr = 1; }
#define COMPILER_FENCE() asm volatile("" ::: "memory")
__device__ __forceinline__ int wave_max_i32(int m) {   // uniform result
  m = max(m, __builtin_amdgcn_update_dpp(0, m, 0x111, 0xf, 0xf, true));
  m = max(m, __builtin_amdgcn_update_dpp(0, m, 0x112, 0xf, 0xf, true));
  m = max(m, __builtin_amdgcn_update_dpp(0, m, 0x114, 0xf, 0xf, true));
  m = max(m, __builtin_amdgcn_update_dpp(0, m, 0x118, 0xf, 0xf, true));
  m = max(m, __builtin_amdgcn_update_dpp(0, m, 0x142, 0xa, 0xf, false));
  m = max(m, __builtin_amdgcn_update_dpp(0, m, 0x143, 0xc, 0xf, false));
  return __builtin_amdgcn_readlane(m, 63);
}

struct RingFields {      // what a ring slot is made of, as loaded
  uint4 n0, n1;          // the near list: sixteen 16-bit ids
  uint2 cnt;             // near / far count
  unsigned long long col;
  long long id;
};

// returns false — before anything was written — when the clip kernel's failure word says that the lists are unusable
__device__ __forceinline__ bool nms_list_body(const NmsArgs& a, const unsigned long long* __restrict__ colm_,
                                              const unsigned short* __restrict__ lists_, const unsigned* __restrict__ lcnt_,
                                              unsigned lblock,
                                              long long* __restrict__ keep_, long long* __restrict__ num_keep_,
                                              [[maybe_unused]] long long* __restrict__ dbg) {
  constexpr int SB = (int)LIST_MAX_N + 384;
  __shared__ __attribute__((aligned(16))) unsigned char stb[SB];   // state byte per box; [LIST_DUMMY + 4 lane] are scratch
  __shared__ unsigned int rent[LIST_RING][LIST_NEAR][64];    // [slot][k][lane]: LDS address of the state byte of the lane's k-th near victim
  __shared__ unsigned long long rcol[LIST_RING][64], rid[LIST_RING][64];
  __shared__ unsigned int rflag[LIST_RING];
  __shared__ unsigned int fdone[256 + LIST_K + 1 + 7];       // [b + LIST_K + 1] != 0: the far victims of block b's kept boxes are marked
  __shared__ unsigned int failed;                            // a polling loop gave up: the result is void (num_keep = -1)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = blockIdx.x;  // one workgroup per group
  const int n = group_n(a, g);
  const int cb = (n + 63) >> 6;
  const long long* order = a.order != nullptr ? a.order + (size_t)g * a.cap : nullptr;
  const unsigned long long* const colm = colm_ + (size_t)g * a.cap;
  const unsigned short* const lists = lists_ + (size_t)g * a.cap * (LIST_NEAR + LIST_FAR);
  const unsigned* const lcnt = lcnt_ + (size_t)g * lblock;   // the group's counters, then (last 64 words of the block) its failure word
  const unsigned* const lfail = lcnt + (lblock - 64u);
  long long* const keep = keep_ + (size_t)g * a.cap;
  long long* const num_keep = num_keep_ + g;
  const unsigned short* const flists = lists + (size_t)a.cap * LIST_NEAR;
  const uint2* const cnt2 = reinterpret_cast<const uint2*>(lcnt);
  const unsigned int sb0 = (unsigned int)(size_t)(lds_u8*)stb;   // LDS address of stb[0]: the ring holds ADDRESSES of state bytes
  const unsigned int mydummy = (unsigned int)LIST_DUMMY + 4u * (unsigned int)lane;   // (same-address byte writes of many lanes would be serialised)
  for (int w = tid; w < SB / 4; w += SCAN_T) reinterpret_cast<unsigned int*>(stb)[w] = 0u;
  if (tid < LIST_RING) rflag[tid] = 0u;
  if (tid < 256 + LIST_K + 1) fdone[tid] = tid <= LIST_K ? 1u : 0u;   // (nothing to wait for before block LIST_K + 1)
  if (tid == 0) failed = 0u;
  // The resolver works in groups of four blocks: blocks cb .. cbp - 1 are PADDING, entered into the ring like real ones; the state
  // bytes of everything past box n - 1 start as "suppressed".
  const int cbp = (cb + 3) & ~3;

  // ---- helper-side pieces.  Nothing may touch a loaded value before its consumer — not even a select: a use makes the compiler wait
  // for the load where the use stands (rows are read from a clamped index and masked where they are consumed).
  auto load_ring = [&](int B) -> RingFields {
    RingFields f;
    const int j = min(B * 64 + lane, n - 1);   // n >= 1 here
    const uint4* const l4 = reinterpret_cast<const uint4*>(lists + (size_t)j * LIST_NEAR);
    f.n0 = l4[0];
    f.n1 = l4[1];
    f.cnt = cnt2[j];
    f.col = colm[j];
    f.id = order != nullptr ? order[j] : (long long)j;
    return f;
  };
  auto store_ring = [&](int B, const RingFields f) {   // whole wave; (B < cbp is the caller's business)
    const int slot = B & (LIST_RING - 1);
    const bool ok = B * 64 + lane < n;
    const int cnt = ok ? (int)min(f.cnt.x, (unsigned)LIST_NEAR) : 0;
    unsigned int* const row0 = &rent[slot][0][lane];
    const unsigned int dummy = sb0 + mydummy;
    const int chunks = (wave_max_i32(cnt) + 3) >> 2;        // rows the resolver will look at
    auto put = [&](int k, unsigned int e) { row0[k * 64] = k < cnt ? sb0 + e : dummy; };
    auto put8 = [&](int k, const uint4 q) {
      put(k + 0, q.x & 0xffffu); put(k + 1, q.x >> 16); put(k + 2, q.y & 0xffffu); put(k + 3, q.y >> 16);
      put(k + 4, q.z & 0xffffu); put(k + 5, q.z >> 16); put(k + 6, q.w & 0xffffu); put(k + 7, q.w >> 16);
    };
    if (chunks > 0) put8(0, f.n0);
    if (chunks > 2) put8(8, f.n1);
    const unsigned long long col = ok ? f.col : 0ull;
    rcol[slot][lane] = col;
    rid[slot][lane] = (unsigned long long)f.id;
    const unsigned int hascol = __ballot(col != 0ull) != 0ull ? 0x80u : 0u;
    COMPILER_FENCE();                            // the flag goes last
    if (lane == 63) lds_poke(&rflag[slot], ((unsigned int)((B >> 4) + 1) << 8) | hascol | (unsigned int)chunks);
  };

  // prologue: sixteen waves, sixteen blocks, ONE memory round trip — the failure word travels with the first blocks' fields
  // (checked before them it is a round trip of its own; measured: no difference in the kernel's time, kept for the shorter chain)
  const unsigned int fail = __hip_atomic_load(lfail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  RingFields f0 = {};
  if (wave < cbp) f0 = load_ring(wave);
  lds_barrier();                                 // (the zero fill above, before anything else is written)
  if (fail != 0u) return false;                  // uniform over the workgroup
#ifdef SCAN_PROFILE
  if (lane == 0) dbg[(size_t)(2 * cb + 16) * 16 + wave] = __builtin_amdgcn_s_getreg(63492);   // HW_ID of every wave
#endif
  for (int j = n + tid; j < cbp * 64; j += SCAN_T) stb[j] = 1;   // boxes past the end: suppressed from the start
  if (wave < cbp) store_ring(wave, f0);
  lds_barrier();                                 // the only barriers of this scan
  if (cb == 0) {
    if (tid == 0) num_keep[0] = 0;
    return true;
  }

  if (wave == 0) {
    // ---------------------------------------------------------------- resolver
    const unsigned int* const rl = &rent[0][0][lane];
    const unsigned long long* const rc = &rcol[0][lane];
    struct Near { unsigned int x, y, z, w; };
    // entries 4 chunk .. 4 chunk + 3 of the lane's near list in the slot at `r` (rows 256 bytes apart)
    auto ring4 = [&](const unsigned int* r, int chunk) -> Near { return Near{r[chunk * 256], r[chunk * 256 + 64], r[chunk * 256 + 128], r[chunk * 256 + 192]}; };
    auto mark4 = [&](const Near& e) { lds_mark_at(e.x); lds_mark_at(e.y); lds_mark_at(e.z); lds_mark_at(e.w); };
    constexpr int SLOT_DW = LIST_NEAR * 64;   // dwords per ring slot
    // two register sets, blocks of even / odd index: block c's fields are fetched while block c - 1 waits for its state bytes, so
    // that neither the flag nor the entries are waited for
    unsigned long long colA = rc[0], colB = 0ull;
    Near l0A = ring4(rl, 0), l1A = ring4(rl, 1), l0B = {}, l1B = {};
    unsigned int nflagA = rflag[0], nflagB = 0u;         // (block 0 is in the ring since the prologue; set B is fetched during block 0)
    unsigned int fdA = 1u, fdB = 1u;                     // (block 0 waits for nobody)
    for (int c0 = 0; c0 < cbp; c0 += 4) {
      unsigned char* const kw = stb + c0 * 64 + lane;
      const unsigned int* const fdp = fdone + c0;
      const int s0 = c0 & (LIST_RING - 1), s4 = (c0 + 4) & (LIST_RING - 1);   // slots of blocks c0 and c0 + 4
      // ring addresses of the group's own slots and of the next group's (blocks c0 + 2 .. c0 + 5 are fetched here): computed once
      // per group, so that every access of a block is base register + instruction offset
      const unsigned int* const rl0 = rl + s0 * SLOT_DW;
      const unsigned int* const rl4 = rl + s4 * SLOT_DW;
      const unsigned long long* const rc0 = rc + s0 * 64;
      const unsigned long long* const rc4 = rc + s4 * 64;
      const unsigned int* const rf0 = rflag + s0;
      const unsigned int* const rf4 = rflag + s4;
      const unsigned int gen1 = (unsigned int)(c0 >> 4) + 1u;   // what the flag of every block of this group carries
      auto block = [&](auto U) {
        constexpr int u = decltype(U)::value;
        [[maybe_unused]] const int c = c0 + u;
        SCAN_STAMP(0);
#ifdef SCAN_PROFILE
        if (lane == 0) dbg[(size_t)c * 16 + 6] = dbg[(size_t)c * 16 + 7] = 0;
#endif
        const unsigned int* const crl = rl0 + u * SLOT_DW;          // this block's slot
        // the slot fetched in this block: block c + 1
        const unsigned int* const frl = u < 3 ? rl0 + (u + 1) * SLOT_DW : rl4;
        const unsigned long long* const frc = u < 3 ? rc0 + (u + 1) * 64 : rc4;
        const unsigned int* const frf = u < 3 ? rf0 + (u + 1) : rf4;
        Near& l0 = (u & 1) ? l0B : l0A;
        Near& l1 = (u & 1) ? l1B : l1A;
        unsigned long long& col = (u & 1) ? colB : colA;
        unsigned int& nflag = (u & 1) ? nflagB : nflagA;
        unsigned int& fd = (u & 1) ? fdB : fdA;
        // 1. nothing of block c may be read before the far victims of block c - LIST_K - 1 are marked (flag fetched a block ago)
        if (__builtin_expect(__builtin_amdgcn_readfirstlane((int)fd) == 0, 0)) {
          bool got = false;
          for (int spins = 0; spins < LIST_SPIN_MAX && lds_peek(&failed) == 0u; ++spins) {
            __builtin_amdgcn_s_sleep(1);
#ifdef SCAN_PROFILE
            if (lane == 0) dbg[(size_t)c * 16 + 7] = spins + 1;
#endif
            if (__builtin_amdgcn_readfirstlane((int)lds_peek(&fdp[u])) != 0) {
              got = true;
              break;
            }
          }
          if (!got) lds_poke(&failed, 1u);
        }
        COMPILER_FENCE();
        // 2. the block's state bytes: THE round trip of the block.  Everything that does not depend on it is issued in its shadow:
        //    block c + 1's fields into the other register set (its last user, block c - 1, is done), this block's ring flag check
        const unsigned char state = kw[u * 64];
        COMPILER_FENCE();
        {
          Near& l0n = (u & 1) ? l0A : l0B;
          Near& l1n = (u & 1) ? l1A : l1B;
          unsigned long long& coln = (u & 1) ? colA : colB;
          unsigned int& nflagn = (u & 1) ? nflagA : nflagB;
          unsigned int& fdn = (u & 1) ? fdA : fdB;
          nflagn = lds_peek(frf);          // flag first, then the fields it vouches for
          COMPILER_FENCE();
          l0n = ring4(frl, 0);
          l1n = ring4(frl, 1);
          coln = frc[0];
          fdn = lds_peek(&fdp[u + 1]);
          COMPILER_FENCE();
        }
        unsigned int flag = (unsigned int)__builtin_amdgcn_readfirstlane((int)nflag);
        if (__builtin_expect((flag >> 8) != gen1, 0)) {
          // the block is not in the ring yet (never in steady state): re-read flag and fields.  Gives up after LIST_SPIN_MAX polls,
          // marks the scan failed and goes on with a harmless flag; once failed, no more waiting.
          flag = gen1 << 8;
          bool got = false;
          for (int spins = 0; spins < LIST_SPIN_MAX && lds_peek(&failed) == 0u; ++spins) {
            __builtin_amdgcn_s_sleep(1);
            COMPILER_FENCE();
#ifdef SCAN_PROFILE
            if (lane == 0) dbg[(size_t)c * 16 + 6] = spins + 1;
#endif
            const unsigned int fl = (unsigned int)__builtin_amdgcn_readfirstlane((int)lds_peek(rf0 + u));
            COMPILER_FENCE();
            l0 = ring4(crl, 0);
            l1 = ring4(crl, 1);
            col = rc0[u * 64];
            if ((fl >> 8) == gen1) {
              flag = fl;
              got = true;
              break;
            }
          }
          if (!got) lds_poke(&failed, 1u);
        }
        unsigned long long kept = __ballot(state == 0);   // nobody kept so far suppresses the lane's box
        SCAN_STAMP(1);
        if (__builtin_expect((flag & 0x80u) != 0u, 0)) {   // some box of the block has an earlier box of the block on its column word
          const unsigned long long alive = kept;
          const unsigned int clo = (unsigned int)col, chi = (unsigned int)(col >> 32);
          if (__ballot(((clo & (unsigned int)alive) | (chi & (unsigned int)(alive >> 32))) != 0u) != 0ull) {
            for (;;) {  // <= 65 rounds; the fixed point is the greedy keep set of the block
              const bool sup = ((clo & (unsigned int)kept) | (chi & (unsigned int)(kept >> 32))) != 0u;
              const unsigned long long nk = alive & ~__ballot(sup);
              if (nk == kept) break;
              kept = nk;
            }
          }
        }
        SCAN_STAMP(2);
        const bool mine = __builtin_amdgcn_inverse_ballot_w64(kept);
        // near victims first (read back by THIS wave for later blocks — LDS runs a wave's accesses in order), the block's own state
        // bytes LAST: they are what the helper waves poll, and a helper that sees the block resolved will refill this block's ring
        // slot — from which entries 8..15 (rare) are still being read here
        if ((flag & 7u) != 0u && mine) {
          mark4(l0);
          if ((flag & 6u) != 0u) {         // more than one 4-entry chunk (the count is 0..4)
            mark4(l1);
            if ((flag & 7u) > 2u) {
              mark4(ring4(crl, 2));
              if ((flag & 7u) > 3u) mark4(ring4(crl, 3));
            }
          }
        }
        COMPILER_FENCE();
        kw[u * 64] = mine ? 0x80 : 0x02;
        COMPILER_FENCE();
        SCAN_STAMP(3);
      };
      block(std::integral_constant<int, 0>{});
      block(std::integral_constant<int, 1>{});
      block(std::integral_constant<int, 2>{});
      block(std::integral_constant<int, 3>{});
    }
    if (lds_peek(&failed) != 0u && lane == 0) num_keep[0] = -1;
    return true;
  }
  // ------------------------------------------------------------------ helper waves: every wave that is not on the resolver's SIMD
  if ((wave & 3) == 0) return true;
  const int hw = wave - 1 - (wave >> 2);
  int base = 0;                                   // kept boxes before block t
  auto kept_word = [&](int blk) -> unsigned long long { return __ballot(stb[blk * 64 + lane] == 0x80); };
  unsigned int fcnt_next = hw < cb ? cnt2[min(hw * 64 + lane, n - 1)].y : 0u;   // far count of this wave's next block, one iteration ahead
  for (int t = hw; t < cb; t += LIST_HW) {
#ifdef SCAN_PROFILE
#define HSTAMP(k) do { if (lane == 0) dbg[(size_t)(cb + 8 + t) * 16 + (k)] = clock64(); } while (0)
#else
#define HSTAMP(k) do { } while (0)
#endif
    HSTAMP(0);
    const bool more = t + LIST_RING < cbp;
    // block t's far lists: as many uint4 as its longest one needs (the counts were loaded an iteration ago; lanes past the end have
    // a clamped index and are never kept)
    const int fcnt = (int)min(fcnt_next, (unsigned)LIST_FAR);
    const int nq = (wave_max_i32(fcnt) + 7) >> 3;
    uint4 q[LIST_FAR / 8];
    {
      const uint4* const l4 = reinterpret_cast<const uint4*>(flists + (size_t)min(t * 64 + lane, n - 1) * LIST_FAR);
#pragma unroll
      for (int k = 0; k < LIST_FAR / 8; ++k) q[k] = k < nq ? l4[k] : make_uint4(0u, 0u, 0u, 0u);
    }
    RingFields fr = {};                           // block t + 16's fields (ring), in flight while the resolver works up to t
    if (more) fr = load_ring(t + LIST_RING);
    if (t + LIST_HW < cb) fcnt_next = cnt2[min((t + LIST_HW) * 64 + lane, n - 1)].y;
    int spins = 0;
    while ((__builtin_amdgcn_readfirstlane((int)*(const volatile lds_u8*)(size_t)(sb0 + (unsigned int)t * 64u)) & 0x82) == 0) {
      if (++spins > LIST_SPIN_MAX) {   // a helper gives up: the scan is void — say so (nobody may be left to notice otherwise:
        lds_poke(&failed, 1u);         // the blocks after this one have no waiter) and report it in the count
        if (lane == 0) num_keep[0] = -1;
        return true;
      }
      __builtin_amdgcn_s_sleep(LIST_POLL_SLEEP);
    }
    COMPILER_FENCE();
    HSTAMP(1);
    const unsigned long long kept = kept_word(t);
    const bool mine = (kept >> lane) & 1ull;
    {   // far victims of the kept boxes: due before the resolver reaches block t + LIST_K + 1.  ROLLED, the list rotating through
        // q[0] (register arrays cannot be indexed): code size matters here (see the header)
      if (mine) {
#pragma unroll 1
        for (int k = 0; k < nq; ++k) {
          const uint4 v = q[0];
          const int kb = k * 8;
          auto mark = [&](int i, unsigned int e) { stb[kb + i < fcnt ? e : mydummy] = 1; };
          mark(0, v.x & 0xffffu); mark(1, v.x >> 16); mark(2, v.y & 0xffffu); mark(3, v.y >> 16);
          mark(4, v.z & 0xffffu); mark(5, v.z >> 16); mark(6, v.w & 0xffffu); mark(7, v.w >> 16);
#pragma unroll
          for (int r = 0; r + 1 < LIST_FAR / 8; ++r) q[r] = q[r + 1];
        }
      }
      COMPILER_FENCE();
      if (lane == 0) lds_poke(&fdone[t + LIST_K + 1], 1u);
      COMPILER_FENCE();
    }
    // scribe step of block t; its global store goes last (loads and stores share one in-order counter)
    HSTAMP(2);
    // the kept boxes of the blocks since this wave's last one (all resolved before t).  Straight-line — eleven reads in flight at once,
    // a clamped index and a masked count for the first iteration: as a loop it was eleven LDS round trips in a row (1400-1900 cycles,
    // the longest phase of a helper and most of the kernel's tail after the resolver's last block)
#pragma unroll
    for (int i = 1; i < LIST_HW; ++i) {
      const int cntb = __builtin_popcountll(kept_word(max(t - i, 0)));
      base += t - i >= 0 ? cntb : 0;
    }
    const long long id = (long long)rid[t & (LIST_RING - 1)][lane];
    HSTAMP(3);
    if (more) {
      COMPILER_FENCE();                           // (the id above is read before the slot is overwritten)
      asm volatile("" : "+v"(fr.col), "+v"(fr.id));   // first use of the loaded fields pinned here
      store_ring(t + LIST_RING, fr);
      COMPILER_FENCE();
    }
    HSTAMP(4);
    if (mine) keep[base + __builtin_popcountll(kept & ((1ull << lane) - 1ull))] = id;
    base += __builtin_popcountll(kept);
    if (t == cb - 1 && lane == 0 && lds_peek(&failed) == 0u) num_keep[0] = base;
  }
  return true;
}

// ONE launch for a call that may take the list scan: the failure word the clip kernel left decides (uniform) between the list scan
// (thirteen waves; the others leave after the prologue) and the classic single-level scan — beyond two chunks per row its <.., 2> form ORs the
// rest in synchronously: correct, slower than the two-level form, and only ever run as a fallback here.
template <int CH>
__global__ __launch_bounds__(SCAN_T) void nms_list_or_scan_kernel(const NmsArgs a, const unsigned long long* __restrict__ mask,
                                                                  const unsigned long long* __restrict__ colm,
                                                                  const unsigned short* __restrict__ lists,
                                                                  const unsigned* __restrict__ lcnt, unsigned lblock,
                                                                  long long* __restrict__ keep, long long* __restrict__ num_keep,
                                                                  long long* __restrict__ dbg, const ScanWindow win) {
  if (!nms_list_body(a, colm, lists, lcnt, lblock, keep, num_keep, dbg)) nms_scan_body<SCAN_U, CH>(a, mask, colm, keep, num_keep, dbg, win);
}

// Second level of the two-level scan: after super-block [c_begin, c_end) has been resolved, every box it KEPT suppresses
// boxes further right; those mask rows are OR-ed into the global removed-set by the whole chip instead of by the one scan
// workgroup.  One wave per (64-box row block of the super-block, 64-word chunk right of it): lane = word, the wave walks
// the kept boxes of its block (independent 512-byte row loads), one atomicOr (integer: deterministic) per word.
__global__ __launch_bounds__(256) void nms_propagate_kernel(const NmsArgs a, const unsigned long long* __restrict__ mask_,
                                                            const ScanWindow win, int wchunks) {
  const int g = blockIdx.y;
  const int n = group_n(a, g);
  const int cb = (n + 63) >> 6;
  const int c_end = win.c_end;
  if (c_end >= cb) return;                                     // nothing right of the super-block in this group
  const size_t cbs = (size_t)a.cbs;
  const int rb = win.c_begin + (int)(blockIdx.x / wchunks), wc = (int)(blockIdx.x % wchunks);
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int w = c_end + wc * 64 + lane;
  if (c_end + wc * 64 >= cb) return;                           // uniform
  const unsigned long long kept = win.gkept[(size_t)g * cbs + rb];   // uniform
  // the four waves of the workgroup share the block's kept rows round-robin (k-th kept row -> wave k % 4)
  unsigned long long mine = 0ull;
  int k = 0;
  for (unsigned long long t = kept; t != 0ull; t &= t - 1ull, ++k)
    if ((k & 3) == wave) mine |= t & (~t + 1ull);
  if (mine == 0ull) return;
  const unsigned long long* rows = mask_ + ((size_t)g * a.cap + (size_t)rb * 64) * cbs;
  const unsigned int wcl = (unsigned int)min(w, cb - 1);
  unsigned long long acc = 0ull;
  while (mine != 0ull) {   // four independent 512-byte row loads per round trip (a duplicate row is harmless: OR)
    int idx[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      idx[u] = mine != 0ull ? __builtin_ctzll(mine) : idx[u > 0 ? u - 1 : 0];
      mine &= mine - (mine != 0ull ? 1ull : 0ull);
    }
    const unsigned long long v0 = rows[(size_t)idx[0] * cbs + wcl], v1 = rows[(size_t)idx[1] * cbs + wcl];
    const unsigned long long v2 = rows[(size_t)idx[2] * cbs + wcl], v3 = rows[(size_t)idx[3] * cbs + wcl];
    acc |= (v0 | v1) | (v2 | v3);
  }
  if (w < cb && acc != 0ull) atomicOr(&win.gremv[(size_t)g * cbs + w], acc);
}

// pairwise IoU matrices ------------------------------------------------------------------
constexpr int IOU_T = 256;
__global__ __launch_bounds__(IOU_T) void riou_xyxyr_kernel(const float* __restrict__ a, long long na,
                                                           const float* __restrict__ b, long long nb,
                                                           float* __restrict__ out) {
  __shared__ VertexScratch<IOU_T> vs;
  const long long idx = (long long)blockIdx.x * IOU_T + threadIdx.x;
  if (idx >= na * nb) return;
  const long long i = idx / nb, j = idx - i * nb;
  float ra[5], rb[5];
#pragma unroll
  for (int k = 0; k < 5; ++k) {
    ra[k] = a[i * 5 + k];
    rb[k] = b[j * 5 + k];
  }
  OBox A, B;
  obox_make(ra, A);
  obox_make(rb, B);
  out[idx] = iou_bev<IOU_T>(A, B, vs, threadIdx.x);
}

constexpr int EVAL_T = 128;
// affinity.cpp:8-81
template <bool IS3D>
__global__ __launch_bounds__(EVAL_T) void riou_eval_kernel(const float* __restrict__ det, long long nd,
                                                           const float* __restrict__ gt, long long ng, float z_offset,
                                                           float* __restrict__ out) {
  __shared__ HullScratch<EVAL_T> hs;
  const long long idx = (long long)blockIdx.x * EVAL_T + threadIdx.x;
  if (idx >= nd * ng) return;
  const long long di = idx / ng, gi = idx - di * ng;
  float d[7], g[7];
#pragma unroll
  for (int k = 0; k < 7; ++k) {
    d[k] = det[di * 7 + k];
    g[k] = gt[gi * 7 + k];
  }
  out[idx] = eval_iou<IS3D, EVAL_T>(d, g, z_offset, hs, threadIdx.x);
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

static size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

}  // namespace rbox

using namespace rbox;

static const int64_t RNMS_MAX_N = 65536;  // 1024 mask words per row; removed-set = 8 KiB of LDS; queue entries hold two 16-bit box indices
static const int64_t QUEUE_MIN_N = 768;   // below: the compacted one-kernel form (3 us faster at 128-256 boxes, equal at 512-768,
                                          // 1 us slower at 1000, 4 at 4096, 34 at 9000: sweep in profiles/r04_nms_queue_ab.txt)

extern "C" {

// workspace of G groups of up to `cap` boxes:
//   OBox records | mask (cap x cbs words) | colm (cap words) | gremv, gkept (cbs words each: two-level scan) |
//   candidate queue of the queued mask form (QUEUE_PER_BOX entries per box) | its control words | overflowed block pairs
constexpr size_t QUEUE_PER_BOX = 128;
struct WsLayout {
  size_t mask, colm, gremv, queue, qctl, ovl, lists, lcnt, total;
  unsigned scap, npairs, lblock;
};
static WsLayout ws_layout(size_t G, size_t cap) {
  const size_t cb = (cap + 63) / 64;
  WsLayout L;
  L.mask = align_up(G * cap * sizeof(OBox), 256);
  L.colm = L.mask + align_up(G * cap * cb * sizeof(unsigned long long), 256);
  L.gremv = L.colm + align_up(G * cap * sizeof(unsigned long long), 256);
  L.queue = L.gremv + align_up((2 * G * cb + G) * sizeof(unsigned long long), 256);   // gremv | gkept | one running count per group
  L.scap = (unsigned)((cap * QUEUE_PER_BOX + QUEUE_SHARDS - 1) / QUEUE_SHARDS);   // entries per shard
  L.npairs = (unsigned)(cb * (cb + 1) / 2);
  L.qctl = L.queue + align_up(G * QUEUE_SHARDS * L.scap * sizeof(unsigned), 256);
  L.ovl = L.qctl + align_up(G * CTL_WORDS * sizeof(unsigned), 256);
  // victim lists of the list scan (groups of at most LIST_MAX_N boxes only): ids, then per group one block of `lblock` words: the
  // (cap, 2) counters followed by the group's failure word in a 256-byte tail (group 0's — with one group: THE — failure word is
  // found 256 bytes before the end of the workspace: tests read it).  One zero fill per group clears counters and failure word.
  const bool wl = cap <= LIST_MAX_N;
  L.lists = L.ovl + align_up(G * L.npairs * sizeof(unsigned), 256);
  L.lcnt = L.lists + align_up(wl ? G * cap * (LIST_NEAR + LIST_FAR) * sizeof(unsigned short) : 0, 256);
  L.lblock = wl ? (unsigned)(align_up(cap * 2 * sizeof(unsigned), 256) / sizeof(unsigned) + 64) : 64u;
  L.total = L.lcnt + G * (size_t)L.lblock * sizeof(unsigned);
  return L;
}

size_t rnms_workspace_bytes(int64_t n) {
  if (n <= 0) return 16;
  return ws_layout(1, (size_t)n).total;
}

size_t rnms_batched_workspace_bytes(int32_t groups, int64_t cap) {
  if (groups <= 0 || cap <= 0) return 16;
  return ws_layout((size_t)groups, (size_t)cap).total;
}

// shared by the single and the batched entry points: G groups of up to `cap` boxes
static int rnms_launch(int mode, const float* boxes, const int64_t* order, const int32_t* counts, int32_t G, int64_t cap,
                       float thresh, double thresh_d, const float* thresh_dev, int64_t* keep, int64_t* num_keep,
                       void* workspace, void* stream, bool prepped = false, bool ctl_zeroed = false) {
  hipStream_t s = (hipStream_t)stream;
  if (cap > RNMS_MAX_N) return GD3D_E_TOOLARGE;
  if (G > 65535) return GD3D_E_TOOLARGE;
  NmsArgs a;
  a.boxes = boxes;
  a.order = (const long long*)order;
  a.counts = (const int*)counts;
  a.thresh_dev = thresh_dev;
  a.n = counts == nullptr ? (int)cap : 0;
  a.cap = (int)cap;
  a.cbs = ((int)cap + 63) / 64;
  a.thresh = thresh;
  a.thresh_d = thresh_d;
  const WsLayout W = ws_layout((size_t)G, (size_t)cap);
  OBox* ob = (OBox*)workspace;
  unsigned long long* mask = (unsigned long long*)((char*)workspace + W.mask);
  unsigned long long* colm = (unsigned long long*)((char*)workspace + W.colm);  // per box: the earlier boxes of its own 64-block that suppress it
  const long long pairs = (long long)a.cbs * (a.cbs + 1) / 2;
  int rows;
  if (mode == MODE_ROT) {
    // compacted kernel: 8..64 rows per wave, as many as keep >= 256 waves in the grid (measured, mask kernel alone:
    // n = 1000: 25 us at 8 rows, 21 at 32, 28 at 64; n = 4096: 47 at 16, 37 at 32, 29 at 64; n = 9000: 229 at 8, 97 at 64)
    rows = 64;
    while (rows > 8 && pairs * G * (64 / rows) < 256) rows /= 2;
  } else {
    rows = 1;
    while (rows < 8 && pairs * G * 64 / (rows * 2) >= 16384) rows *= 2;  // keep >= ~16 K waves in the grid
  }
  a.rows = rows;
  if (pairs * (64 / rows) > 0x7fffffffLL) return GD3D_E_TOOLARGE;
  const dim3 mgrid((unsigned)(pairs * (64 / rows)), (unsigned)G);
  // queued form (circle tests and clipping as two kernels): from QUEUE_MIN_N boxes on, with a threshold every group shares and
  // that is a plain non-negative number (a negative or NaN threshold makes EVERY pair a candidate: the compacted kernel's case)
  static const long long queue_min_n = [] {
    const char* e = getenv("RNMS_QUEUE_MIN_N");   // measurement override (tests/perf/nms_time.py A/B)
    return e != nullptr ? atoll(e) : (long long)QUEUE_MIN_N;
  }();
  const bool queued = mode == MODE_ROT && cap >= queue_min_n && pairs <= 0x7fffffffLL && (thresh_dev != nullptr || thresh >= 0.0f);   // (per-group device thresholds are checked in the kernel)
  // list scan (victim lists and state bytes instead of mask-row propagation): groups of QUEUE_MIN_N .. LIST_MAX_N boxes, rotated and
  // axis-aligned boxes;
  // a full list or an overflowed block pair (a negative / NaN device threshold included) falls back to the classic scan on the
  // device, per group
  static const float list_min_thr = [] {
    const char* e = getenv("RNMS_LIST_MIN_THR");   // measurement / test override (a value > 1 switches the list scan off)
    return e != nullptr ? (float)atof(e) : 0.0f;
  }();
  // (not for circle NMS: its mask kernel is so cheap that building the lists — 15.8 -> 21.9 us at n = 4096 — and clearing the
  // counters in a launch of their own — 4.4 us — cost what the list scan saves, 29.5 -> 18.0 us; axis-aligned: 62 -> 55 us)
  const bool lists_wanted = list_min_thr <= 1.0f && cap >= queue_min_n && cap <= (int64_t)LIST_MAX_N && mode != MODE_CIRCLE &&
                            (mode != MODE_ROT || thresh_dev != nullptr || thresh >= list_min_thr);
  bool use_lists = false;
  QueueArgs ql;   // the list part alone: what the axis-aligned / circle mask kernel takes
  ql.queue = ql.ctl = ql.ovl = nullptr;
  ql.scap = ql.npairs = 0u;
  ql.lists = nullptr;
  ql.lcnt = nullptr;
  ql.lblock = W.lblock;
  if (mode != MODE_ROT && lists_wanted) {
    use_lists = true;
    ql.lists = (unsigned short*)((char*)workspace + W.lists);
    ql.lcnt = (unsigned*)((char*)workspace + W.lcnt);
    // counters and failure words start at zero: the scored paths' rank_place_kernel cleared them (`ctl_zeroed`), otherwise a fill
    // kernel in the stream (a kernel, not a memset node)
    if (!ctl_zeroed) hipLaunchKernelGGL(zero_words_kernel, dim3(1, (unsigned)G), dim3(256), 0, s, ql.lcnt, (int)W.lblock);
  }
  if (mode == MODE_ROT && queued) {
    QueueArgs q;
    q.queue = (unsigned*)((char*)workspace + W.queue);
    q.ctl = (unsigned*)((char*)workspace + W.qctl);
    q.ovl = (unsigned*)((char*)workspace + W.ovl);
    q.scap = W.scap;
    q.npairs = W.npairs;
    use_lists = lists_wanted;
    q.lists = use_lists ? (unsigned short*)((char*)workspace + W.lists) : nullptr;
    q.lcnt = use_lists ? (unsigned*)((char*)workspace + W.lcnt) : nullptr;
    q.lblock = W.lblock;
    // the control words start at zero: cleared by whichever prep kernel ran (this one, or the scored paths' rank_place_kernel:
    // `ctl_zeroed`); only a caller that prepared the records itself pays a fill in the stream (4.4 us in the trace)
    const int zero_n = (int)CTL_WORDS;   // per group
    if (!prepped)
      hipLaunchKernelGGL(obox_prep_kernel, dim3(((unsigned)cap + 255) / 256, (unsigned)G), dim3(256), 0, s, a, ob, q.ctl, zero_n);
    else if (!ctl_zeroed) {
      hipLaunchKernelGGL(zero_words_kernel, dim3(1, (unsigned)G), dim3(256), 0, s, q.ctl, zero_n);   // (a kernel, not a memset node)
    }
    hipLaunchKernelGGL(nms_circle_queue_kernel, dim3((unsigned)pairs, (unsigned)G), dim3(64), 0, s, a, (const OBox*)ob, mask, colm, q);
    // clipping waves: a multiple of the shard count, about one per block pair, at most 4096 and at least 8 per shard: with few
    // block pairs few shards are in use, and one wave per shard walked its ~90 entries in two passes one after the other (n = 256:
    // 25 us for 900 candidates; waves that find their shard empty leave after one load).  The kernel is as long as a wave's passes
    // (memory round trips + one clipping pass each, vector units 27 % busy): more waves with one pass each beat 2048 waves with
    // three to four (n = 9000: 20.3 -> 18.0 us at 64 per shard; 128: the same — LDS holds ten waves per CU).
    long long per = (pairs + QUEUE_SHARDS - 1) / QUEUE_SHARDS;
    per = per < 8 ? 8 : (per > 64 ? 64 : per);
    hipLaunchKernelGGL(nms_clip_queue_kernel, dim3((unsigned)(per * QUEUE_SHARDS), (unsigned)G), dim3(64), 0, s, a, (const OBox*)ob, mask, colm, q);
  } else if (mode == MODE_ROT) {
    if (!prepped)
      hipLaunchKernelGGL(obox_prep_kernel, dim3(((unsigned)cap + 255) / 256, (unsigned)G), dim3(256), 0, s, a, ob, (unsigned*)nullptr, 0);
    hipLaunchKernelGGL(nms_mask_compact_kernel, mgrid, dim3(64), 0, s, a, (const OBox*)ob, mask, colm);
  } else if (mode == MODE_NORMAL) {
    hipLaunchKernelGGL((nms_mask_kernel<MODE_NORMAL>), mgrid, dim3(64), 0, s, a, (const OBox*)ob, mask, colm, ql);
  } else {
    hipLaunchKernelGGL((nms_mask_kernel<MODE_CIRCLE>), mgrid, dim3(64), 0, s, a, (const OBox*)ob, mask, colm, ql);
  }
  const dim3 sgrid((unsigned)G), sblk(SCAN_T);
  const size_t slds = (size_t)a.cbs * sizeof(unsigned long long);
  ScanWindow win;
  win.c_begin = 0;
  win.c_end = a.cbs;
  win.gremv = win.gkept = nullptr;
  win.gcount = nullptr;
  if (use_lists) {   // ONE launch: list scan, or — decided on the device from the clip kernel's failure word — the classic one
    const unsigned short* const lists = (const unsigned short*)((char*)workspace + W.lists);
    const unsigned* const lcnt = (const unsigned*)((char*)workspace + W.lcnt);
    if (a.cbs <= 64 + 1 + SCAN_NU)
      hipLaunchKernelGGL((nms_list_or_scan_kernel<1>), sgrid, sblk, slds, s, a, (const unsigned long long*)mask,
                         (const unsigned long long*)colm, lists, lcnt, W.lblock, (long long*)keep, (long long*)num_keep, (long long*)ob, win);
    else
      hipLaunchKernelGGL((nms_list_or_scan_kernel<2>), sgrid, sblk, slds, s, a, (const unsigned long long*)mask,
                         (const unsigned long long*)colm, lists, lcnt, W.lblock, (long long*)keep, (long long*)num_keep, (long long*)ob, win);
    return (int)hipGetLastError();
  }
  // n <= 8448: one launch resolves everything.  Beyond that the single workgroup's row propagation (three 64-word chunks
  // per kept row, one CU's miss bandwidth) dominates and the two-level form wins: r02, kernels of rnms_bev, single ->
  // two-level: n = 9000 339 -> 237 us (72 % kept), 208 -> 207 (25 % kept), 400 -> 237 (79 % kept); n = 16384 1052 -> 476 us;
  // it loses below (n = 6000: 117 -> 128 us: five launches instead of one) — profiles/r02_nms_scan_levels.txt.
  if (a.cbs <= 128 + 1 + SCAN_NU) {
    if (a.cbs <= 64 + 1 + SCAN_NU)  // one 64-word chunk right of any block (n <= 4352): 16 rows x 1 chunk in flight per row wave
      hipLaunchKernelGGL((nms_scan_kernel<SCAN_U, 1>), sgrid, sblk, slds, s, a, (const unsigned long long*)mask,
                         (const unsigned long long*)colm, (long long*)keep, (long long*)num_keep, (long long*)ob, win);
    else
      hipLaunchKernelGGL((nms_scan_kernel<SCAN_U, 2>), sgrid, sblk, slds, s, a, (const unsigned long long*)mask,
                         (const unsigned long long*)colm, (long long*)keep, (long long*)num_keep, (long long*)ob, win);
    return (int)hipGetLastError();
  }
  // two-level scan: super-blocks of SCAN_SB blocks resolved one after the other by the scan workgroup (rows stay inside
  // the super-block: one chunk), the rows of the kept boxes spread to everything right of it by nms_propagate_kernel
  win.gremv = (unsigned long long*)((char*)workspace + W.gremv);
  win.gkept = win.gremv + (size_t)G * a.cbs;
  win.gcount = (long long*)(win.gkept + (size_t)G * a.cbs);
  for (int c0 = 0; c0 < a.cbs; c0 += SCAN_SB) {
    win.c_begin = c0;
    win.c_end = c0 + SCAN_SB < a.cbs ? c0 + SCAN_SB : a.cbs;
    hipLaunchKernelGGL((nms_scan_kernel<SCAN_U, 1>), sgrid, sblk, slds, s, a, (const unsigned long long*)mask,
                       (const unsigned long long*)colm, (long long*)keep, (long long*)num_keep, (long long*)ob, win);
    if (win.c_end < a.cbs) {
      const int wchunks = (a.cbs - win.c_end + 63) / 64;
      hipLaunchKernelGGL(nms_propagate_kernel, dim3((unsigned)((win.c_end - c0) * wchunks), (unsigned)G), dim3(256), 0, s, a,
                         (const unsigned long long*)mask, win, wchunks);
    }
  }
  return (int)hipGetLastError();
}

static int rnms_impl(int mode, const float* boxes, const int64_t* order, int64_t n, float thresh, double thresh_d,
                     int64_t* keep, int64_t* num_keep, void* workspace, void* stream) {
  if (n < 0 || num_keep == nullptr) return GD3D_E_BADARG;
  if (n == 0) return fill_words(num_keep, sizeof(int64_t), 0u, (hipStream_t)stream);
  if (boxes == nullptr || keep == nullptr || workspace == nullptr) return GD3D_E_BADARG;
  return rnms_launch(mode, boxes, order, nullptr, 1, n, thresh, thresh_d, nullptr, keep, num_keep, workspace, stream);
}

int rnms_batched(int32_t mode, const float* boxes, const int64_t* order, const int32_t* counts, int32_t groups, int64_t cap,
                 const float* thresh, int64_t* keep, int64_t* num_keep, void* workspace, void* stream) {
  if (mode < MODE_ROT || mode > MODE_CIRCLE || groups < 0 || cap < 0) return GD3D_E_BADARG;
  if (groups == 0) return 0;
  if (num_keep == nullptr) return GD3D_E_BADARG;
  if (cap == 0) return fill_words(num_keep, sizeof(int64_t) * (size_t)groups, 0u, (hipStream_t)stream);
  if (boxes == nullptr || order == nullptr || counts == nullptr || thresh == nullptr || keep == nullptr ||
      workspace == nullptr)
    return GD3D_E_BADARG;
  return rnms_launch(mode, boxes, order, counts, groups, cap, 0.0f, 0.0, thresh, keep, num_keep, workspace, stream);
}

int rnms_batched_prepared(const float* boxes, const int64_t* order, const int32_t* counts, int32_t groups, int64_t cap,
                          const float* thresh, int64_t* keep, int64_t* num_keep, void* workspace, void* stream) {
  if (groups < 0 || cap < 0) return GD3D_E_BADARG;
  if (groups == 0) return 0;
  if (num_keep == nullptr) return GD3D_E_BADARG;
  if (cap == 0) return fill_words(num_keep, sizeof(int64_t) * (size_t)groups, 0u, (hipStream_t)stream);
  if (boxes == nullptr || order == nullptr || counts == nullptr || thresh == nullptr || keep == nullptr ||
      workspace == nullptr)
    return GD3D_E_BADARG;
  return rnms_launch(MODE_ROT, boxes, order, counts, groups, cap, 0.0f, 0.0, thresh, keep, num_keep, workspace, stream,
                     /*prepped=*/true);
}

int rnms_circle_ordered(const float* xy, const int64_t* order, int64_t n, double thresh, int64_t* keep, int64_t* num_keep,
                        void* workspace, void* stream) {
  return rnms_impl(MODE_CIRCLE, xy, order, n, (float)thresh, thresh, keep, num_keep, workspace, stream);
}

int rnms_bev(const float* boxes_sorted, int64_t n, float thresh, int64_t* keep, int64_t* num_keep, void* workspace,
             void* stream) {
  return rnms_impl(MODE_ROT, boxes_sorted, nullptr, n, thresh, 0.0, keep, num_keep, workspace, stream);
}

int rnms_scored_max_n(void) { return RANK_MAX; }

size_t rnms_scored_workspace_bytes(int64_t n_all, int64_t n_keep) {
  if (n_all < 1) n_all = 1;
  if (n_keep < 1) n_keep = 1;
  return align_up(rnms_workspace_bytes(n_keep), 256) + align_up((size_t)n_keep * sizeof(int64_t), 256);   // NMS workspace | order
}

int rnms_scored(int32_t normal, const float* boxes, const float* scores, int64_t n_all, int64_t pre_max, float thresh,
                int64_t* keep, int64_t* num_keep, void* workspace, void* stream) {
  if (n_all < 0 || num_keep == nullptr) return GD3D_E_BADARG;
  if (n_all > RANK_MAX) return GD3D_E_TOOLARGE;
  const int64_t n = (pre_max >= 0 && pre_max < n_all) ? pre_max : n_all;
  hipStream_t s = (hipStream_t)stream;
  if (n == 0) return fill_words(num_keep, sizeof(int64_t), 0u, s);
  if (boxes == nullptr || scores == nullptr || keep == nullptr || workspace == nullptr) return GD3D_E_BADARG;
  long long* order = (long long*)((char*)workspace + align_up(rnms_workspace_bytes(n), 256));
  const dim3 sg((unsigned)((n_all + 15) / 16));   // one 16-wave workgroup per 16 boxes: counts their ranks and places them
  unsigned* const qctl = (unsigned*)((char*)workspace + ws_layout(1, (size_t)n).qctl);   // control words of the queued mask form
  const WsLayout W1 = ws_layout(1, (size_t)n);
  if (normal)   // (axis-aligned: no queue; the list scan's counters and failure word are cleared instead)
    hipLaunchKernelGGL((rank_place_kernel<false>), sg, dim3(1024), 0, s, boxes, scores, (const unsigned char*)nullptr,
                       (const int*)nullptr, (int)n_all, (int)n, order, (OBox*)workspace, (int*)nullptr, 0,
                       (unsigned*)((char*)workspace + W1.lcnt), (int)W1.lblock);
  else
    hipLaunchKernelGGL((rank_place_kernel<true>), sg, dim3(1024), 0, s, boxes, scores, (const unsigned char*)nullptr,
                       (const int*)nullptr, (int)n_all, (int)n, order, (OBox*)workspace, (int*)nullptr, 0, qctl, (int)CTL_WORDS);
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) return (int)e;
  return rnms_launch(normal ? MODE_NORMAL : MODE_ROT, boxes, (const int64_t*)order, nullptr, 1, n, thresh, 0.0, nullptr, keep,
                     num_keep, workspace, stream, /*prepped=*/true, /*ctl_zeroed=*/true);
}

size_t rnms_batched_scored_workspace_bytes(int32_t groups, int64_t n, int64_t cap) {
  if (groups < 1) groups = 1;
  if (n < 1) n = 1;
  if (cap < 1) cap = 1;
  return align_up(rnms_batched_workspace_bytes(groups, cap), 256) + align_up((size_t)groups * cap * sizeof(int64_t), 256) +
         align_up((size_t)groups * sizeof(int), 256);   // NMS workspace | order | counts
}

static int batched_scored_impl(int32_t mode, const float* boxes, const float* scores, const uint8_t* valid, const int32_t* seg,
                               int32_t groups, int64_t n, int64_t pre_max, const float* thresh, int64_t* keep,
                               int64_t* num_keep, void* workspace, void* stream, int gps = 0) {
  if (mode < MODE_ROT || mode > MODE_CIRCLE || groups < 0 || n < 0) return GD3D_E_BADARG;
  if (groups == 0) return 0;
  if (num_keep == nullptr) return GD3D_E_BADARG;
  if (n > RANK_MAX || groups > 65535) return GD3D_E_TOOLARGE;
  const int64_t cap = (pre_max >= 0 && pre_max < n) ? pre_max : n;
  hipStream_t s = (hipStream_t)stream;
  if (cap == 0) return fill_words(num_keep, sizeof(int64_t) * (size_t)groups, 0u, s);
  if (boxes == nullptr || scores == nullptr || thresh == nullptr || keep == nullptr || workspace == nullptr) return GD3D_E_BADARG;
  char* p = (char*)workspace + align_up(rnms_batched_workspace_bytes(groups, cap), 256);
  long long* order = (long long*)p;
  p += align_up((size_t)groups * cap * sizeof(int64_t), 256);
  int* counts = (int*)p;
  hipError_t e;
  const dim3 sg((unsigned)((n + 15) / 16), (unsigned)groups);
  static const unsigned masked_threads = [] {
    const char* e = getenv("RNMS_RANK_MASKED_THREADS");   // measurement override (tools/nms_batched_ab.sh)
    const unsigned v = e != nullptr ? (unsigned)atoi(e) : 256u;
    return (v == 64u || v == 128u || v == 256u || v == 512u || v == 1024u) ? v : 256u;
  }();
  const dim3 sb(valid != nullptr && seg == nullptr ? masked_threads : 1024u);   // masked dense form: four waves per workgroup
  unsigned* const qctl = (unsigned*)((char*)workspace + ws_layout((size_t)groups, (size_t)cap).qctl);
  if (mode == MODE_ROT)
    hipLaunchKernelGGL((rank_place_kernel<true>), sg, sb, 0, s, boxes, scores, (const unsigned char*)valid, (const int*)seg,
                       (int)n, (int)cap, order, (OBox*)workspace, counts, gps, qctl, (int)CTL_WORDS);
  else {
    const WsLayout WG = ws_layout((size_t)groups, (size_t)cap);
    hipLaunchKernelGGL((rank_place_kernel<false>), sg, sb, 0, s, boxes, scores, (const unsigned char*)valid, (const int*)seg,
                       (int)n, (int)cap, order, (OBox*)workspace, counts, gps, (unsigned*)((char*)workspace + WG.lcnt), (int)WG.lblock);
  }
  e = hipGetLastError();
  if (e != hipSuccess) return (int)e;
  return rnms_launch(mode, boxes, (const int64_t*)order, (const int32_t*)counts, groups, cap, 0.0f, 0.0, thresh, keep, num_keep,
                     workspace, stream, /*prepped=*/mode == MODE_ROT, /*ctl_zeroed=*/true);
}

int rnms_batched_scored(int32_t mode, const float* boxes, const float* scores, const uint8_t* valid, int32_t groups, int64_t n,
                        int64_t pre_max, const float* thresh, int64_t* keep, int64_t* num_keep, void* workspace, void* stream) {
  return batched_scored_impl(mode, boxes, scores, valid, nullptr, groups, n, pre_max, thresh, keep, num_keep, workspace, stream);
}

int rnms_batched_scored_sets(int32_t mode, const float* boxes, const float* scores, const uint8_t* valid, int32_t sets,
                             int32_t groups_per_set, int64_t n, int64_t pre_max, const float* thresh, int64_t* keep, int64_t* num_keep,
                             void* workspace, void* stream) {
  if (sets < 0 || groups_per_set < 1 || (int64_t)sets * groups_per_set > 65535 || (int64_t)sets * n > 0x7fffffffLL) return GD3D_E_BADARG;
  return batched_scored_impl(mode, boxes, scores, valid, nullptr, sets * groups_per_set, n, pre_max, thresh, keep, num_keep, workspace,
                             stream, groups_per_set);
}

int rnms_segmented_scored(int32_t mode, const float* boxes, const float* scores, const int32_t* seg, int32_t groups,
                          int64_t max_seg, int64_t pre_max, const float* thresh, int64_t* keep, int64_t* num_keep,
                          void* workspace, void* stream) {
  if (groups > 0 && seg == nullptr) return GD3D_E_BADARG;
  return batched_scored_impl(mode, boxes, scores, nullptr, seg, groups, max_seg, pre_max, thresh, keep, num_keep, workspace,
                             stream);
}

int rnms_bev_ordered(const float* boxes, const int64_t* order, int64_t n, float thresh, int64_t* keep, int64_t* num_keep,
                     void* workspace, void* stream) {
  if (n > 0 && order == nullptr) return GD3D_E_BADARG;
  return rnms_impl(MODE_ROT, boxes, order, n, thresh, 0.0, keep, num_keep, workspace, stream);
}

int rnms_normal_bev_ordered(const float* boxes, const int64_t* order, int64_t n, float thresh, int64_t* keep,
                            int64_t* num_keep, void* workspace, void* stream) {
  if (n > 0 && order == nullptr) return GD3D_E_BADARG;
  return rnms_impl(MODE_NORMAL, boxes, order, n, thresh, 0.0, keep, num_keep, workspace, stream);
}

int rnms_normal_bev(const float* boxes_sorted, int64_t n, float thresh, int64_t* keep, int64_t* num_keep,
                    void* workspace, void* stream) {
  return rnms_impl(MODE_NORMAL, boxes_sorted, nullptr, n, thresh, 0.0, keep, num_keep, workspace, stream);
}

int riou_bev_xyxyr(const float* a, int64_t na, const float* b, int64_t nb, float* iou, void* stream) {
  if (na < 0 || nb < 0) return GD3D_E_BADARG;
  if (na == 0 || nb == 0) return 0;
  if (a == nullptr || b == nullptr || iou == nullptr) return GD3D_E_BADARG;
  const long long tot = (long long)na * nb;
  const long long blocks = (tot + IOU_T - 1) / IOU_T;
  if (blocks > 0x7fffffffLL) return GD3D_E_TOOLARGE;
  hipLaunchKernelGGL(riou_xyxyr_kernel, dim3((unsigned)blocks), dim3(IOU_T), 0, (hipStream_t)stream, a, (long long)na,
                     b, (long long)nb, iou);
  return (int)hipGetLastError();
}

static int riou_eval_impl(bool is3d, const float* det, int64_t nd, const float* gt, int64_t ng, float z_offset,
                          float* iou, void* stream) {
  if (nd < 0 || ng < 0) return GD3D_E_BADARG;
  if (nd == 0 || ng == 0) return 0;
  if (det == nullptr || gt == nullptr || iou == nullptr) return GD3D_E_BADARG;
  const long long tot = (long long)nd * ng;
  const long long blocks = (tot + EVAL_T - 1) / EVAL_T;
  if (blocks > 0x7fffffffLL) return GD3D_E_TOOLARGE;
  if (is3d)
    hipLaunchKernelGGL((riou_eval_kernel<true>), dim3((unsigned)blocks), dim3(EVAL_T), 0, (hipStream_t)stream, det,
                       (long long)nd, gt, (long long)ng, z_offset, iou);
  else
    hipLaunchKernelGGL((riou_eval_kernel<false>), dim3((unsigned)blocks), dim3(EVAL_T), 0, (hipStream_t)stream, det,
                       (long long)nd, gt, (long long)ng, z_offset, iou);
  return (int)hipGetLastError();
}

int riou_eval_bev(const float* det, int64_t nd, const float* gt, int64_t ng, float* iou, void* stream) {
  return riou_eval_impl(false, det, nd, gt, ng, 0.5f, iou, stream);
}

int riou_eval_3d(const float* det, int64_t nd, const float* gt, int64_t ng, float z_offset, float* iou, void* stream) {
  return riou_eval_impl(true, det, nd, gt, ng, z_offset, iou, stream);
}

}  // extern "C"
