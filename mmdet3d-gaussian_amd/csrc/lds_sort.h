// lds_sort.h — descending bitonic sort of 64-bit entries in LDS by one 1024-thread workgroup (gfx950), shared by
// center_infer.hip (candidate cells as (key, ~index)) and center_targets.hip (boxes as inverted (task, sample, class, index) keys).
// Layout: one pad entry after every 8 (PH): a thread that owns 8 consecutive entries (64 bytes) would otherwise share its two
// LDS banks with 31 other lanes of its wave.  The LDS block that holds the list must be 8-byte aligned — declare a dynamic
// `extern __shared__` block with __attribute__((aligned(16))): static __shared__ words in the same kernel can push it to an odd
// multiple of 4, and every 64-bit LDS access then runs as a misaligned one (measured: 2.5x slower, tools/sort_probe.hip).
#pragma once
#include <hip/hip_runtime.h>

namespace ldssort {

constexpr int T = 1024;

__device__ __forceinline__ int PH(int i) { return i + (i >> 3); }

// Descending bitonic sort of list[0 .. M) in LDS (padded with zeros, which lie below every real entry), register blocked:
// the compare-exchange steps j = k/2 .. 1 of a merge are taken three at a time — a thread loads the 8 entries that differ
// in those three index bits, runs the three steps in registers and stores them back, so the list crosses the LDS once per
// three steps instead of once per step (one entry pair per thread and step moved 4 LDS instructions per compare-exchange:
// 36 us for 1024 entries).
template <int C>
__device__ __forceinline__ void bitonic_chunk(unsigned long long* list, int P, int k, int b) {
  constexpr int E = 1 << C;
  const unsigned low = (1u << b) - 1u;
  for (int t = threadIdx.x; t < (P >> C); t += T) {
    const int base = (int)((((unsigned)t & ~low) << C) | ((unsigned)t & low));
    const bool desc = (base & k) == 0;
    // entries as (high, low) words: a 64-bit compare is a quarter-rate instruction here, three 32-bit compares are not.
    // Entries are distinct (the index is part of them), so "a > b" is "not a < b": one comparison serves both directions
    // (equal entries exist only as zero padding, where a swap changes nothing).
    unsigned xh[E], xl[E];
#pragma unroll
    for (int m = 0; m < E; ++m) {
      const unsigned long long v = list[PH(base + (m << b))];
      xh[m] = (unsigned)(v >> 32);
      xl[m] = (unsigned)v;
    }
#pragma unroll
    for (int s2 = C - 1; s2 >= 0; --s2) {
#pragma unroll
      for (int m = 0; m < E; ++m) {
        if ((m & (1 << s2)) == 0) {
          const int n = m | (1 << s2);
          const bool lt = (xh[m] < xh[n]) | ((xh[m] == xh[n]) & (xl[m] < xl[n]));
          const bool sw = lt == desc;
          const unsigned h0 = xh[m], l0 = xl[m];
          xh[m] = sw ? xh[n] : h0;
          xl[m] = sw ? xl[n] : l0;
          xh[n] = sw ? h0 : xh[n];
          xl[n] = sw ? l0 : xl[n];
        }
      }
    }
#pragma unroll
    for (int m = 0; m < E; ++m) list[PH(base + (m << b))] = ((unsigned long long)xh[m] << 32) | (unsigned long long)xl[m];
  }
}

// list[0 .. M) unordered -> list[0 .. P) descending, P = the power of two >= max(M, 8); entries M .. P are set to zero
__device__ __forceinline__ void bitonic_desc(unsigned long long* list, int M) {
  const int tid = threadIdx.x;
  int P = 8;
  while (P < M) P <<= 1;
  for (int i = M + tid; i < P; i += T) list[PH(i)] = 0ull;
  __syncthreads();
  int bitsk = 1;
  for (int k = 2; k <= P; k <<= 1, ++bitsk) {
    int top = bitsk;                         // index bits [0, top) still to be merged for this k
    while (top > 0) {
      const int c = top >= 3 ? 3 : top;
      const int b = top - c;
      if (c == 3) bitonic_chunk<3>(list, P, k, b);
      else if (c == 2) bitonic_chunk<2>(list, P, k, b);
      else bitonic_chunk<1>(list, P, k, b);
      __syncthreads();
      top = b;
    }
  }
}

}  // namespace ldssort
