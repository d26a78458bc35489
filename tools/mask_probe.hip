// mask_probe.hip — cycle stamps of ONE wave of nms_mask_compact_kernel inside a real rotated-NMS call (n boxes in
// clusters of jittered duplicates, random score order), to see where a wave's ~20 us go.
//   stamps: 0 start | 1 column/row summaries loaded | then per 16-row chunk: 2+2c circle tests + queue, 3+2c drain passes
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -I include -o tools/mask_probe tools/mask_probe.hip && ./tools/mask_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
__device__ long long mask_stamps[16];
__device__ int mask_probe_block = 0;
__device__ int mask_cands;
#define MASK_NOTE(v) do { if (lane == 0 && blockIdx.y == 0 && (int)blockIdx.x == mask_probe_block) mask_cands += (v); } while (0)
#define MASK_STAMP_SYNC(k) do { __builtin_amdgcn_s_waitcnt(0); MASK_STAMP(k); } while (0)
#define MASK_STAMP(k) do { if (lane == 0 && blockIdx.y == 0 && (int)blockIdx.x == mask_probe_block) mask_stamps[k] = clock64(); } while (0)
#include "../mmdet3d-gaussian_amd/csrc/rbox.hip"
#define CK(x) do { hipError_t e = (hipError_t)(x); if (e != hipSuccess) { printf("error %d at %s:%d\n", (int)e, __FILE__, __LINE__); return 1; } } while (0)

int main() {
  for (int n : {1000, 4096}) {
    std::vector<float> hb((size_t)n * 5);
    unsigned s = 777u + n;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (float)(s >> 8) / 16777216.0f; };
    const int nc = n / 8;
    std::vector<float> cx(nc), cy(nc), cr(nc);
    for (int c = 0; c < nc; ++c) { cx[c] = rnd() * 150.f - 75.f; cy[c] = rnd() * 150.f - 75.f; cr[c] = rnd() * 6.28f - 3.14f; }
    for (int i = 0; i < n; ++i) {
      const int c = (int)(rnd() * nc) % nc;
      const float x = cx[c] + (rnd() - 0.5f), y = cy[c] + (rnd() - 0.5f), w = 4.7f * (0.9f + 0.2f * rnd()), h = 2.1f * (0.9f + 0.2f * rnd());
      hb[i * 5 + 0] = x - w / 2; hb[i * 5 + 1] = y - h / 2; hb[i * 5 + 2] = x + w / 2; hb[i * 5 + 3] = y + h / 2;
      hb[i * 5 + 4] = cr[c] + 0.2f * (rnd() - 0.5f);
    }
    float* db; long long *keep, *num; void* ws;
    CK(hipMalloc(&db, hb.size() * 4)); CK(hipMalloc(&keep, n * 8)); CK(hipMalloc(&num, 8)); CK(hipMalloc(&ws, rnms_workspace_bytes(n)));
    CK(hipMemcpy(db, hb.data(), hb.size() * 4, hipMemcpyHostToDevice));
    const int cb = (n + 63) / 64;
    for (int blk : {0, cb / 2, cb * (cb + 1) / 2 - 1}) {   // first (diagonal), an off-diagonal, the last block pair
      CK(hipMemcpyToSymbol(HIP_SYMBOL(mask_probe_block), &blk, sizeof(int)));
      const int zero = 0;
      CK(hipMemcpyToSymbol(HIP_SYMBOL(mask_cands), &zero, sizeof(int)));
      for (int rep = 0; rep < 3; ++rep) CK(rnms_bev(db, n, 0.25f, (int64_t*)keep, (int64_t*)num, ws, nullptr));
      CK(hipDeviceSynchronize());
      long long st[16];
      CK(hipMemcpyFromSymbol(st, HIP_SYMBOL(mask_stamps), sizeof(st)));
      printf("n=%d workgroup %d: load %lld |", n, blk, st[1] - st[0]);
      long long prev = st[1];
      for (int k = 2; k < 10 && st[k] > prev; ++k) { printf(" %s %lld", (k & 1) ? "drain" : "circle", st[k] - prev); prev = st[k]; }
      printf("   last drain pass: loads %lld, predicate %lld, atomicOr %lld\n", st[11] - st[10], st[12] - st[11], st[13] - st[12]);
      int cands;
      CK(hipMemcpyFromSymbol(&cands, HIP_SYMBOL(mask_cands), sizeof(int)));
      printf(" | total %lld cycles, %d candidates\n", prev - st[0], cands / 3);
    }
    long long kept; CK(hipMemcpy(&kept, num, 8, hipMemcpyDeviceToHost));
    printf("n=%d kept %lld\n", n, kept);
    hipFree(db); hipFree(keep); hipFree(num); hipFree(ws);
  }
  return 0;
}
