// clip_probe.hip — cycle accounting of ONE polygon-clipping pass of the rotated-NMS predicate (rbox_device.h box_overlap)
// for a lone wave with 64 live lanes, as nms_mask_compact_kernel runs it.  Stamps (lane 0, s_memtime):
//   0 start | 1 after the circle early-out | 2 after 16 segment intersections | 3 after 8 corner-in-box tests
//   4 after the angles | 5 after the sort | 6 after the fan area | 7 after the IoU division
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o tools/clip_probe tools/clip_probe.hip && ./tools/clip_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
__device__ long long rb_stamps[8];
#define RB_STAMP(k) do { if (t == 0) rb_stamps[k] = clock64(); } while (0)
#include "../mmdet3d-gaussian_amd/csrc/rbox_device.h"
using namespace rbox;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %d at %s:%d\n", (int)e, __FILE__, __LINE__); return 1; } } while (0)

__global__ __launch_bounds__(64) void probe(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out,
                                             int reps) {
  __shared__ VertexScratch<64> vs;
  const int t = threadIdx.x;
  const size_t p = (size_t)blockIdx.x * 64 + t;
  OBox A, B;
  obox_make(a + p * 5, A);
  obox_make(b + p * 5, B);
  float acc = 0.0f;
  for (int r = 0; r < reps; ++r) {
    if (t == 0) rb_stamps[0] = clock64();
    acc += iou_bev<64>(A, B, vs, t);
    if (t == 0) rb_stamps[7] = clock64();
  }
  out[p] = acc;
}

int run(int mode) {
  const int waves = 4096, n = waves * 64;
  std::vector<float> ha(n * 5), hb(n * 5);
  unsigned s = 12345u;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (float)(s >> 8) / 16777216.0f; };
  for (int i = 0; i < n; ++i) {  // car-sized boxes, the second one a jittered copy: 8 intersection vertices typically
    const float x = rnd() * 100.f, y = rnd() * 100.f, w = 4.5f + rnd() * 0.5f, h = 2.f + rnd() * 0.2f, r = rnd() * 6.28f - 3.14f;
    ha[i * 5 + 0] = x - w / 2; ha[i * 5 + 1] = y - h / 2; ha[i * 5 + 2] = x + w / 2; ha[i * 5 + 3] = y + h / 2; ha[i * 5 + 4] = r;
    // mode 0: jittered copies (coherent lanes, 8 vertices each); mode 1: what a queue of circle-test survivors looks
    // like — shifts up to +-2.5 m, any relative yaw, a third of the pairs not overlapping at all
    const float sh = mode ? 5.0f : 0.6f, ro = mode ? 3.0f : 0.3f;
    const float dx = rnd() * sh - sh / 2, dy = rnd() * sh - sh / 2, dr = rnd() * ro - ro / 2;
    hb[i * 5 + 0] = x + dx - w / 2; hb[i * 5 + 1] = y + dy - h / 2; hb[i * 5 + 2] = x + dx + w / 2; hb[i * 5 + 3] = y + dy + h / 2;
    hb[i * 5 + 4] = r + dr;
  }
  float *da, *db, *dout;
  CK(hipMalloc(&da, n * 5 * 4)); CK(hipMalloc(&db, n * 5 * 4)); CK(hipMalloc(&dout, n * 4));
  CK(hipMemcpy(da, ha.data(), n * 5 * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(db, hb.data(), n * 5 * 4, hipMemcpyHostToDevice));
  // (1) one lone wave: phase stamps of the last of 4 repetitions (warm instruction cache)
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, da, db, dout, 4);
  CK(hipDeviceSynchronize());
  long long st[8];
  CK(hipMemcpyFromSymbol(st, HIP_SYMBOL(rb_stamps), sizeof(st)));
  const char* names[8] = {"", "circle early-out", "16 segment intersections", "8 corner-in-box tests", "angles (atan2)", "bubble sort",
                          "fan area", "IoU division"};
  printf("%s pairs — lone wave, 64 live lanes, s_memtime ticks per phase (total %lld):\n", mode ? "diverse" : "jittered-copy", st[7] - st[0]);
  for (int k = 1; k < 8; ++k) printf("  %-26s %8lld\n", names[k], st[k] - st[k - 1]);
  // (2) wall time of one pass: a single wave vs enough waves to fill the chip
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int g : {1, 256, 1024, 4096}) {
    for (int reps : {1, 9}) {
      hipLaunchKernelGGL(probe, dim3(g), dim3(64), 0, 0, da, db, dout, reps);
      CK(hipEventRecord(e0));
      for (int it = 0; it < 20; ++it) hipLaunchKernelGGL(probe, dim3(g), dim3(64), 0, 0, da, db, dout, reps);
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      printf("grid %5d waves x %d pass(es): %8.2f us per launch\n", g, reps, ms / 20 * 1e3);
    }
  }
  CK(hipFree(da)); CK(hipFree(db)); CK(hipFree(dout));
  return 0;
}

int main() { return run(0) || run(1); }
