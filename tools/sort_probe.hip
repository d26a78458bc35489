// sort_probe.hip — stand-alone timing of the LDS bitonic sort of csrc/center_infer.hip ((key, ~index) entries, one 1024-thread
// workgroup, register-blocked compare-exchange chunks) on gfx950: full sort / barriers only / no barriers, per list size, and the
// same sort after the chip has idled, after sparse one-workgroup launches and right after a chip-filling burst.
// Development aid (tools/build_probes.py builds it); what it settled is in DESIGN.md §3.7:
//   * a pass costs ~0.45 us whatever the bank conflicts and the compare width: the chain LDS read -> compare -> write -> barrier;
//   * the chip's load history does not change it (no idle-clock effect on a single workgroup);
//   * the same code ran 2.5x slower inside select_kernel until the dynamic LDS block was declared 16-byte aligned: static
//     __shared__ words had pushed it to an offset that is no multiple of 8 and every 64-bit LDS access was a misaligned one.
#include <hip/hip_runtime.h>
#include <unistd.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <vector>

constexpr int T = 1024;
typedef unsigned long long u64;
__device__ __forceinline__ int PH(int i) { return i + (i >> 3); }

template <int C, int MODE>
__device__ __forceinline__ void chunk(u64* list, int P, int k, int b) {
  constexpr int E = 1 << C;
  const unsigned low = (1u << b) - 1u;
  for (int t = threadIdx.x; t < (P >> C); t += T) {
    const int base = (int)((((unsigned)t & ~low) << C) | ((unsigned)t & low));
    const bool desc = (base & k) == 0;
    u64 x[E];
#pragma unroll
    for (int m = 0; m < E; ++m) x[m] = list[PH(base + (m << b))];
    if (MODE != 1) {
#pragma unroll
      for (int s2 = C - 1; s2 >= 0; --s2) {
#pragma unroll
        for (int m = 0; m < E; ++m) {
          if ((m & (1 << s2)) == 0) {
            const u64 lo = x[m], hi = x[m | (1 << s2)];
            const bool sw = desc ? lo < hi : lo > hi;
            x[m] = sw ? hi : lo;
            x[m | (1 << s2)] = sw ? lo : hi;
          }
        }
      }
    }
#pragma unroll
    for (int m = 0; m < E; ++m) list[PH(base + (m << b))] = x[m];
  }
}

// MODE 0: full sort; 1: barriers only (the compiler drops the load / store pairs); 2: no barriers (wrong result, timing only)
template <int MODE>
__device__ __forceinline__ void bitonic(u64* list, int P) {
  int bitsk = 1;
  for (int k = 2; k <= P; k <<= 1, ++bitsk) {
    int top = bitsk;
    while (top > 0) {
      const int c = top >= 3 ? 3 : top;
      const int b = top - c;
      if (c == 3) chunk<3, MODE>(list, P, k, b);
      else if (c == 2) chunk<2, MODE>(list, P, k, b);
      else chunk<1, MODE>(list, P, k, b);
      if (MODE != 2) __syncthreads();
      top = b;
    }
  }
}

template <int MODE>
__global__ __launch_bounds__(T) void probe(const u64* in, u64* out, int P, long long* ticks, int reps) {
  extern __shared__ __attribute__((aligned(16))) u64 lds[];
  long long best = 1ll << 60;
  for (int r = 0; r < reps; ++r) {
    for (int i = threadIdx.x; i < P; i += T) lds[PH(i)] = in[i];
    __syncthreads();
    const long long t0 = (long long)wall_clock64();      // 100 MHz
    bitonic<MODE>(lds, P);
    const long long t1 = (long long)wall_clock64();
    best = t1 - t0 < best ? t1 - t0 : best;
    __syncthreads();
  }
  for (int i = threadIdx.x; i < P; i += T) out[i] = lds[PH(i)];
  if (threadIdx.x == 0) ticks[0] = best;
}

__global__ __launch_bounds__(T) void barrier_probe(long long* ticks, int n) {
  const long long t0 = (long long)wall_clock64();
  for (int i = 0; i < n; ++i) __syncthreads();
  const long long t1 = (long long)wall_clock64();
  if (threadIdx.x == 0) ticks[0] = t1 - t0;
}

#define CK(x) do { if ((x) != hipSuccess) { printf("HIP error at line %d\n", __LINE__); return 1; } } while (0)

int main() {
  u64 *din, *dout;
  long long* dt;
  CK(hipMalloc(&din, 8192 * 8));
  CK(hipMalloc(&dout, 8192 * 8));
  CK(hipMalloc(&dt, 64));
  std::vector<u64> h(8192), o(8192);
  srand(1);
  for (auto& v : h) v = ((u64)rand() << 33) ^ ((u64)rand() << 11) ^ (u64)rand();
  CK(hipMemcpy(din, h.data(), 8192 * 8, hipMemcpyHostToDevice));
  const size_t lds = (8192 + 1024) * 8;
  CK(hipFuncSetAttribute((const void*)probe<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  CK(hipFuncSetAttribute((const void*)probe<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  CK(hipFuncSetAttribute((const void*)probe<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  for (int warm = 0; warm < 200; ++warm) hipLaunchKernelGGL(probe<0>, dim3(64), dim3(T), lds, 0, din, dout, 2048, dt, 5);
  CK(hipDeviceSynchronize());
  for (int P : {512, 1024, 2048, 4096, 8192}) {
    long long t[3];
    for (int mode = 0; mode < 3; ++mode) {
      if (mode == 0) hipLaunchKernelGGL(probe<0>, dim3(1), dim3(T), lds, 0, din, dout, P, dt, 20);
      if (mode == 1) hipLaunchKernelGGL(probe<1>, dim3(1), dim3(T), lds, 0, din, dout, P, dt, 20);
      if (mode == 2) hipLaunchKernelGGL(probe<2>, dim3(1), dim3(T), lds, 0, din, dout, P, dt, 20);
      CK(hipDeviceSynchronize());
      CK(hipMemcpy(&t[mode], dt, 8, hipMemcpyDeviceToHost));
      if (mode == 0) {
        CK(hipMemcpy(o.data(), dout, P * 8, hipMemcpyDeviceToHost));
        std::vector<u64> ref(h.begin(), h.begin() + P);
        std::sort(ref.begin(), ref.end(), std::greater<u64>());
        if (!std::equal(ref.begin(), ref.end(), o.begin())) printf("  !! P=%d not sorted\n", P);
      }
    }
    int passes = 0;
    for (int m = 1; (1 << m) <= P; ++m) passes += (m + 2) / 3;
    printf("P=%5d  passes %3d  full %7.2f us   barriers only %7.2f us   no barriers %7.2f us\n", P, passes, t[0] / 100.0,
           t[1] / 100.0, t[2] / 100.0);
  }
  for (int round = 0; round < 3; ++round) {
    if (round == 0) usleep(300000);
    if (round == 1)
      for (int i = 0; i < 300; ++i) {
        hipLaunchKernelGGL(probe<0>, dim3(1), dim3(T), lds, 0, din, dout, 1024, dt, 1);
        CK(hipDeviceSynchronize());
        usleep(100);
      }
    if (round == 2) {
      for (int warm = 0; warm < 300; ++warm) hipLaunchKernelGGL(probe<0>, dim3(256), dim3(T), lds, 0, din, dout, 2048, dt, 5);
      CK(hipDeviceSynchronize());
    }
    hipLaunchKernelGGL(probe<0>, dim3(1), dim3(T), lds, 0, din, dout, 1024, dt, 1);
    CK(hipDeviceSynchronize());
    long long t;
    CK(hipMemcpy(&t, dt, 8, hipMemcpyDeviceToHost));
    printf("P=1024 single sort %s: %.2f us\n",
           round == 0 ? "after 0.3 s idle" : round == 1 ? "after 300 sparse one-workgroup launches" : "right after a chip-filling burst", t / 100.0);
  }
  for (int n : {10, 100}) {
    hipLaunchKernelGGL(barrier_probe, dim3(1), dim3(T), 0, 0, dt, n);
    CK(hipDeviceSynchronize());
    long long t;
    CK(hipMemcpy(&t, dt, 8, hipMemcpyDeviceToHost));
    printf("%d empty barriers of 16 waves: %.2f us\n", n, t / 100.0);
  }
  return 0;
}
