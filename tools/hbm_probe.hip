// hbm_probe.hip — measured HBM ceilings for the access mix of the fused loss kernel
// (2 streams read : 1 stream written, 16 B per lane), to put roofline.frac in context.
//   hipcc --offload-arch=gfx950 -O3 -o hbm_probe tools/hbm_probe.hip && ./hbm_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float v4f __attribute__((ext_vector_type(4)));
#define float4 v4f
#define make_float4(a,b,c,d) ((v4f){a,b,c,d})
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %d at %s:%d\n", (int)e, __FILE__, __LINE__); return 1; } } while (0)

template <bool NT>
__global__ __launch_bounds__(256) void k_copy(const float4* __restrict__ a, float4* __restrict__ c, long long n) {
  const long long stride = (long long)gridDim.x * 256;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
    float4 v = NT ? __builtin_nontemporal_load(a + i) : a[i];
    if (NT) __builtin_nontemporal_store(v, c + i); else c[i] = v;
  }
}
template <bool NT>
__global__ __launch_bounds__(256) void k_add(const float4* __restrict__ a, const float4* __restrict__ b, float4* __restrict__ c, long long n) {
  const long long stride = (long long)gridDim.x * 256;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
    float4 x = NT ? __builtin_nontemporal_load(a + i) : a[i];
    float4 y = NT ? __builtin_nontemporal_load(b + i) : b[i];
    float4 v = make_float4(x.x + y.x, x.y + y.y, x.z + y.z, x.w + y.w);
    if (NT) __builtin_nontemporal_store(v, c + i); else c[i] = v;
  }
}
// 7-KiB tile per block per stream (448 float4), like the fused kernel: thread t moves vec t and t+256 (<448)
template <bool NT>
__global__ __launch_bounds__(256) void k_add_tile(const float4* __restrict__ a, const float4* __restrict__ b, float4* __restrict__ c, long long ntiles) {
  for (long long t = blockIdx.x; t < ntiles; t += gridDim.x) {
    const long long base = t * 448;
    const int i0 = threadIdx.x, i1 = threadIdx.x + 256;
    float4 x0 = a[base + i0], y0 = b[base + i0];
    float4 x1 = make_float4(0, 0, 0, 0), y1 = x1;
    if (i1 < 448) { x1 = a[base + i1]; y1 = b[base + i1]; }
    float4 v0 = make_float4(x0.x + y0.x, x0.y + y0.y, x0.z + y0.z, x0.w + y0.w);
    float4 v1 = make_float4(x1.x + y1.x, x1.y + y1.y, x1.z + y1.z, x1.w + y1.w);
    if (NT) { __builtin_nontemporal_store(v0, c + base + i0); if (i1 < 448) __builtin_nontemporal_store(v1, c + base + i1); }
    else { c[base + i0] = v0; if (i1 < 448) c[base + i1] = v1; }
  }
}

int main() {
  const long long n = 17500000;  // float4s per buffer = 280 MB (the fused kernel's per-tensor footprint at 10 M pairs)
  float4 *a, *b, *c;
  CK(hipMalloc(&a, n * 16)); CK(hipMalloc(&b, n * 16)); CK(hipMalloc(&c, n * 16));
  CK(hipMemset(a, 1, n * 16)); CK(hipMemset(b, 1, n * 16)); CK(hipMemset(c, 0, n * 16));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int iters = 30;
  auto run = [&](const char* name, auto launch, double bytes) {
    for (int i = 0; i < 5; ++i) launch();
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < iters; ++i) launch();
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-34s %8.1f us  %7.1f GB/s\n", name, ms / iters * 1e3, bytes / (ms / iters * 1e-3) / 1e9);
  };
  const long long full = (n + 255) / 256;
  for (long long g : {2048LL, 4096LL, 8192LL, 16384LL, full}) {
    char nm[64];
    snprintf(nm, 64, "copy plain grid=%lld", g); run(nm, [&] { k_copy<false><<<g, 256>>>(a, c, n); }, 2.0 * n * 16);
    snprintf(nm, 64, "copy nt    grid=%lld", g); run(nm, [&] { k_copy<true><<<g, 256>>>(a, c, n); }, 2.0 * n * 16);
    snprintf(nm, 64, "add2R1W plain grid=%lld", g); run(nm, [&] { k_add<false><<<g, 256>>>(a, b, c, n); }, 3.0 * n * 16);
    snprintf(nm, 64, "add2R1W nt    grid=%lld", g); run(nm, [&] { k_add<true><<<g, 256>>>(a, b, c, n); }, 3.0 * n * 16);
  }
  const long long ntiles = n / 448;
  for (long long g : {2048LL, 8192LL, ntiles}) {
    char nm[64];
    snprintf(nm, 64, "add tile448 plain grid=%lld", g); run(nm, [&] { k_add_tile<false><<<g, 256>>>(a, b, c, ntiles); }, 3.0 * ntiles * 448 * 16);
    snprintf(nm, 64, "add tile448 nt-st grid=%lld", g); run(nm, [&] { k_add_tile<true><<<g, 256>>>(a, b, c, ntiles); }, 3.0 * ntiles * 448 * 16);
  }
  return 0;
}
