// pair_variants.hip — does HOW the two read streams are read change what a cross-class pair costs?  (round 6, DESIGN.md 5.3:
// separately allocated buffers fall into two classes; z = x + y is 5-8 % slower when x and y belong to different ones.)
// Allocates ten 280 MB buffers, finds the fastest and the slowest (x, y) pair with the product's access mode (nontemporal 16-byte
// loads and stores, 64-thread workgroups), and — when the draw has two classes — times these variants on both pairs:
//   nt      : nontemporal loads (what the fused kernel's LDS-DMA pieces use)        plain : plain loads
//   sc1     : loads with the sc1 bit (bypass L1, device-coherent)                     split : thread loads x, WAITS, then loads y
//   block   : a workgroup reads 7 KiB of x, then 7 KiB of y (the fused kernel's tile shape), 256 threads
//   hipcc --offload-arch=gfx950 -O3 -o pair_variants tools/pair_variants.hip && ./pair_variants
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <algorithm>
#include <cstdio>
#include <vector>
typedef float v4f __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %d at %s:%d\n", (int)e, __FILE__, __LINE__); return 1; } } while (0)
enum { NT = 0, PLAIN = 1, SC1 = 2, SPLIT = 3 };
template <int MODE>
__global__ __launch_bounds__(64) void k_add(const v4f* __restrict__ x, const v4f* __restrict__ y, v4f* __restrict__ z, long long nv) {
  const long long i = (long long)blockIdx.x * 64 + threadIdx.x;
  if (i >= nv) return;
  v4f p, q;
  if (MODE == NT) { p = __builtin_nontemporal_load(x + i); q = __builtin_nontemporal_load(y + i); }
  else if (MODE == PLAIN) { p = x[i]; q = y[i]; }
  else if (MODE == SC1) {
    asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(p) : "v"(x + i) : "memory");
    asm volatile("global_load_dwordx4 %0, %1, off sc1\n s_waitcnt vmcnt(0)" : "=v"(q) : "v"(y + i) : "memory");
  } else {
    p = __builtin_nontemporal_load(x + i);
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(p)::"memory");
    q = __builtin_nontemporal_load(y + i);
  }
  __builtin_nontemporal_store(p + q, z + i);
}
__global__ __launch_bounds__(256) void k_block(const v4f* __restrict__ x, const v4f* __restrict__ y, v4f* __restrict__ z, long long ntiles) {
  const long long base = (long long)blockIdx.x * 448;
  const int t = threadIdx.x;
  v4f p0 = __builtin_nontemporal_load(x + base + t), p1 = {0, 0, 0, 0};
  if (t < 192) p1 = __builtin_nontemporal_load(x + base + 256 + t);
  asm volatile("s_waitcnt vmcnt(0)" : "+v"(p0), "+v"(p1)::"memory");
  v4f q0 = __builtin_nontemporal_load(y + base + t), q1 = {0, 0, 0, 0};
  if (t < 192) q1 = __builtin_nontemporal_load(y + base + 256 + t);
  __builtin_nontemporal_store(p0 + q0, z + base + t);
  if (t < 192) __builtin_nontemporal_store(p1 + q1, z + base + 256 + t);
}
static hipStream_t s;
template <typename F>
static float timed(F&& launch, int reps) {
  std::vector<float> ms;
  for (int it = 0; it < reps; ++it) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    launch(e0, e1);
    (void)hipStreamSynchronize(s);
    float t; (void)hipEventElapsedTime(&t, e0, e1); ms.push_back(t * 1e3f);
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  }
  std::sort(ms.begin(), ms.end());
  return ms[ms.size() / 2];
}
int main() {
  const long long nv = 17500000;   // 16-byte vectors per buffer: 280 MB; a multiple of 448
  const int NB = 10;
  v4f* b[NB]; v4f* z;
  for (int i = 0; i < NB; ++i) { CK(hipMalloc(&b[i], nv * 16)); CK(hipMemset(b[i], 0, nv * 16)); }
  CK(hipMalloc(&z, nv * 16)); CK(hipMemset(z, 0, nv * 16));
  CK(hipStreamCreate(&s));
  const unsigned g64 = (unsigned)((nv + 63) / 64);
  auto run = [&](int mode, int i, int j, int reps) {
    return timed([&](hipEvent_t e0, hipEvent_t e1) {
      switch (mode) {
        case NT: hipExtLaunchKernelGGL((k_add<NT>), dim3(g64), dim3(64), 0, s, e0, e1, 0, (const v4f*)b[i], (const v4f*)b[j], z, nv); break;
        case PLAIN: hipExtLaunchKernelGGL((k_add<PLAIN>), dim3(g64), dim3(64), 0, s, e0, e1, 0, (const v4f*)b[i], (const v4f*)b[j], z, nv); break;
        case SC1: hipExtLaunchKernelGGL((k_add<SC1>), dim3(g64), dim3(64), 0, s, e0, e1, 0, (const v4f*)b[i], (const v4f*)b[j], z, nv); break;
        case SPLIT: hipExtLaunchKernelGGL((k_add<SPLIT>), dim3(g64), dim3(64), 0, s, e0, e1, 0, (const v4f*)b[i], (const v4f*)b[j], z, nv); break;
        default: hipExtLaunchKernelGGL(k_block, dim3((unsigned)(nv / 448)), dim3(256), 0, s, e0, e1, 0, (const v4f*)b[i], (const v4f*)b[j], z, nv / 448); break;
      }
    }, reps);
  };
  run(NT, 0, 1, 20);
  float best = 1e9f, worst = 0; int bi = 0, bj = 1, wi = 0, wj = 1;
  for (int i = 0; i < NB; ++i)
    for (int j = 0; j < NB; ++j) {
      if (i == j) continue;
      const float t = run(NT, i, j, 3);
      if (t < best) { best = t; bi = i; bj = j; }
      if (t > worst) { worst = t; wi = i; wj = j; }
    }
  printf("search: fastest (%d,%d) %.1f us, slowest (%d,%d) %.1f us\n", bi, bj, best, wi, wj, worst);
  if (worst - best < 5.0f) { printf("single-class draw: nothing to compare in this process\n"); return 0; }
  const char* names[] = {"nt loads", "plain loads", "sc1 loads", "split: x, wait, then y", "block: 7 KiB of x, wait, 7 KiB of y"};
  for (int m = 0; m < 5; ++m) {
    const float f = run(m, bi, bj, 9), w = run(m, wi, wj, 9);
    printf("%-38s fast pair %7.1f us   slow pair %7.1f us   penalty %+5.1f us (%+.1f %%)\n", names[m], f, w, w - f, 100.0f * (w - f) / f);
  }
  return 0;
}
