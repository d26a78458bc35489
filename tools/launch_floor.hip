// launch_floor.hip — what does a TINY kernel cost on this chip when it follows a streaming kernel on the same stream?
// (round 6: is the second stage of the loss sum — 5.3 us as a 1024-thread pass over 39 063 partials — reducible, or is it at the
//  floor of "a dependent kernel that loads something"?)   Durations are the dispatch's own begin/end timestamps
// (hipExtLaunchKernel events), i.e. what rocprofv3 --kernel-trace reports.
//   hipcc --offload-arch=gfx950 -O3 -o launch_floor tools/launch_floor.hip && ./launch_floor
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef float v4f __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %d at %s:%d\n", (int)e, __FILE__, __LINE__); return 1; } } while (0)

__global__ __launch_bounds__(256) void k_stream(const v4f* __restrict__ a, v4f* __restrict__ c, long long n, long long* slots) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i < n) __builtin_nontemporal_store(__builtin_nontemporal_load(a + i), c + i);
  if (slots != nullptr && threadIdx.x < 2)   // the accumulator form's two no-return integer atomics per workgroup
    __hip_atomic_fetch_add(slots + threadIdx.x * 64 + (blockIdx.x & 63), (long long)blockIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__global__ void k_empty() {}
__global__ void k_store(float* out) { if (threadIdx.x == 0) *out = 1.0f; }
__global__ void k_load1(const float* in, float* out) { if (threadIdx.x == 0) *out = *in + 1.0f; }
__global__ __launch_bounds__(64) void k_slots(long long* slots, float* out) {   // the shape of reduce_slots_kernel
  long long v[12];
#pragma unroll
  for (int r = 0; r < 12; ++r) v[r] = slots[r * 64 + threadIdx.x];
#pragma unroll
  for (int r = 0; r < 12; ++r) slots[r * 64 + threadIdx.x] = 0;
  long long s = 0;
#pragma unroll
  for (int r = 0; r < 12; ++r) s += v[r];
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) s += __shfl_down(s, off, 64);
  if (threadIdx.x == 0) *out = (float)s;
}
__global__ __launch_bounds__(256) void k_exit(const float* g) { if (g[0] == 1.0f) return; __builtin_trap(); }   // grad_finish's early exit
__global__ __launch_bounds__(1024) void k_partials(const float* __restrict__ p, long long nb, float* out) {   // reduce_partials' shape
  __shared__ double sd[1024];
  double acc = 0;
  const v4f* p4 = (const v4f*)p;
  v4f v[12];
#pragma unroll
  for (int u = 0; u < 12; ++u) { const long long i = (long long)u * 1024 + threadIdx.x; v[u] = i < nb / 4 ? p4[i] : (v4f){0, 0, 0, 0}; }
#pragma unroll
  for (int u = 0; u < 12; ++u) acc += ((double)v[u].x + (double)v[u].y) + ((double)v[u].z + (double)v[u].w);
  sd[threadIdx.x] = acc;
  __syncthreads();
  if (threadIdx.x < 64) {
    double s = 0;
    for (int k = 0; k < 16; ++k) s += sd[threadIdx.x * 16 + k];
    for (int off = 32; off >= 1; off >>= 1) s += __shfl_down(s, off, 64);
    if (threadIdx.x == 0) *out = (float)s;
  }
}

int main() {
  const long long n = 17500000;   // 280 MB per buffer
  v4f *a, *c; float *small, *parts; long long* slots;
  CK(hipMalloc(&a, n * 16)); CK(hipMalloc(&c, n * 16)); CK(hipMalloc(&small, 4096)); CK(hipMalloc(&slots, 12 * 64 * 8)); CK(hipMalloc(&parts, 39064 * 4));
  CK(hipMemset(a, 0, n * 16)); CK(hipMemset(slots, 0, 12 * 64 * 8)); CK(hipMemset(parts, 0, 39064 * 4));
  float one = 1.0f; CK(hipMemcpy(small, &one, 4, hipMemcpyHostToDevice));
  hipStream_t s; CK(hipStreamCreate(&s));
  const unsigned blocks = (unsigned)((n + 255) / 256);
  struct Row { const char* name; int kind; bool after_stream; bool atomics; };
  const Row rows[] = {
    {"empty kernel, idle stream", 0, false, false}, {"empty kernel, after a 560 MB streaming kernel", 0, true, false},
    {"one store, after streaming", 1, true, false}, {"one load + one store, after streaming", 2, true, false},
    {"reduce_slots shape (1 wave: 12 x 8 B loads per lane, re-zero, fold), after streaming WITH the two atomics per workgroup", 3, true, true},
    {"reduce_slots shape, after streaming without atomics (slots cold)", 3, true, false},
    {"reduce_partials shape (1024 threads, 39 063 floats), after streaming", 5, true, false},
    {"grad_finish early exit (256 x 256 threads, one scalar load), after streaming", 4, true, false},
    {"the streaming kernel itself, no atomics", 6, false, false}, {"the streaming kernel itself, two no-return int64 atomics per workgroup", 6, false, true},
  };
  for (const Row& r : rows) {
    std::vector<float> ms;
    for (int it = 0; it < 60; ++it) {
      hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
      if (r.kind == 6) {
        hipExtLaunchKernelGGL(k_stream, dim3(blocks), dim3(256), 0, s, e0, e1, 0, (const v4f*)a, c, n, r.atomics ? slots : (long long*)nullptr);
      } else {
        if (r.after_stream) hipLaunchKernelGGL(k_stream, dim3(blocks), dim3(256), 0, s, (const v4f*)a, c, n, r.atomics ? slots : (long long*)nullptr);
        switch (r.kind) {
          case 0: hipExtLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, s, e0, e1, 0); break;
          case 1: hipExtLaunchKernelGGL(k_store, dim3(1), dim3(64), 0, s, e0, e1, 0, small + 8); break;
          case 2: hipExtLaunchKernelGGL(k_load1, dim3(1), dim3(64), 0, s, e0, e1, 0, (const float*)small, small + 8); break;
          case 3: hipExtLaunchKernelGGL(k_slots, dim3(1), dim3(64), 0, s, e0, e1, 0, slots, small + 8); break;
          case 4: hipExtLaunchKernelGGL(k_exit, dim3(256), dim3(256), 0, s, e0, e1, 0, (const float*)small); break;
          default: hipExtLaunchKernelGGL(k_partials, dim3(1), dim3(1024), 0, s, e0, e1, 0, (const float*)parts, 39063LL, small + 8); break;
        }
      }
      CK(hipStreamSynchronize(s));
      float t; CK(hipEventElapsedTime(&t, e0, e1));
      if (it >= 10) ms.push_back(t);
      CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
    }
    std::sort(ms.begin(), ms.end());
    double mean = 0; for (float t : ms) mean += t; mean /= ms.size();
    printf("%-120s  mean %7.2f us  median %7.2f  min %7.2f\n", r.name, mean * 1e3, ms[ms.size() / 2] * 1e3, ms[0] * 1e3);
  }
  return 0;
}
