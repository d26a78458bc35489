// hbm_probe2.hip — where do the last 3 % between the fused loss kernel and a flat copy go?
// 2 streams read : 1 written over 3 x 280 MB (the fused kernel's traffic at 10 M pairs), nontemporal everywhere.
//   flat<V,T>   : T threads, V float4 per thread, block chunk = V*T vectors (no LDS)           -> the ceiling
//   tile<BAR>   : the fused kernel's data path without its math: 7-KiB tiles (448 vectors) per tensor brought in by
//                 14 LDS-DMA pieces, (barrier), each thread reads "its rows" (7 dwords x 2), writes 7 dwords back to
//                 LDS, (barrier), 448 16-byte stores.  BAR=0 drops both workgroup barriers (wave-local waits only;
//                 the data is then garbage, only the time matters).
//   hipcc --offload-arch=gfx950 -O3 -o tools/hbm_probe2 tools/hbm_probe2.hip && tools/hbm_probe2
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v4f __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_ptr_t;
typedef const __attribute__((address_space(1))) void gbl_cptr_t;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %d at %s:%d\n", (int)e, __FILE__, __LINE__); return 1; } } while (0)

template <int V, int T>
__global__ __launch_bounds__(T) void flat(const v4f* __restrict__ a, const v4f* __restrict__ b, v4f* __restrict__ c, long long n) {
  const long long base = (long long)blockIdx.x * (V * T) + threadIdx.x;
  v4f x[V], y[V];
#pragma unroll
  for (int k = 0; k < V; ++k) {
    const long long i = base + k * T;
    if (i < n) { x[k] = __builtin_nontemporal_load(a + i); y[k] = __builtin_nontemporal_load(b + i); }
  }
#pragma unroll
  for (int k = 0; k < V; ++k) {
    const long long i = base + k * T;
    if (i < n) __builtin_nontemporal_store(x[k] + y[k], c + i);
  }
}

template <bool BAR, int VALU>
__global__ __launch_bounds__(256) void tile(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ c, long long ntiles) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* sp = smem; float* st = smem + 1792;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const long long t = blockIdx.x;
  if (t >= ntiles) return;
  const float* ga = a + t * 1792; const float* gb = b + t * 1792; float* gc = c + t * 1792;
#pragma unroll
  for (int j0 = 0; j0 < 14; j0 += 4) {
    const int j = j0 + wave;
    if (j < 7) __builtin_amdgcn_global_load_lds((gbl_cptr_t*)(ga + j * 256 + lane * 4), (lds_ptr_t*)(sp + j * 256), 16, 0, 2);
    else if (j < 14) __builtin_amdgcn_global_load_lds((gbl_cptr_t*)(gb + (j - 7) * 256 + lane * 4), (lds_ptr_t*)(st + (j - 7) * 256), 16, 0, 2);
  }
  if (BAR) __syncthreads(); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  float r[7];
#pragma unroll
  for (int k = 0; k < 7; ++k) r[k] = sp[tid * 7 + k] + st[tid * 7 + k];
  if (VALU >= 0) {
#pragma unroll
    for (int it = 0; it < VALU; ++it) {
#pragma unroll
      for (int k = 0; k < 7; ++k) r[k] = __builtin_fmaf(r[k], 1.0000001f, 1e-9f);
    }
  } else {   // one dependent chain of (-VALU % 1000) FMAs per thread; below -1000: every 13th step is a v_rcp_f32;
             // below -2000: a non-VALU issue slot (s_nop) after 4 of every 5 steps as well — the real kernels spend 43 % of
             // their issue slots on scalar work, branches and waits (profiles/r02_pmc_issue_mix_by_cap.txt)
    const int len = (-VALU) % 1000;
    float x = r[0];
#pragma unroll
    for (int it = 0; it < len; ++it) {
      x = __builtin_fmaf(x, 1.0000001f, r[1 + it % 6]);
      if (VALU < -1000 && it % 13 == 12) x = __builtin_amdgcn_rcpf(x);
      if (VALU < -2000 && it % 5 != 4) asm volatile("s_nop 0");
    }
    r[0] = x;
  }
#pragma unroll
  for (int k = 0; k < 7; ++k) sp[tid * 7 + k] = r[k];
  if (BAR) __syncthreads(); else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  const v4f v0 = reinterpret_cast<const v4f*>(sp)[tid];
  __builtin_nontemporal_store(v0, reinterpret_cast<v4f*>(gc) + tid);
  if (tid < 192) {
    const v4f v1 = reinterpret_cast<const v4f*>(sp)[tid + 256];
    __builtin_nontemporal_store(v1, reinterpret_cast<v4f*>(gc) + tid + 256);
  }
}

// flat copy whose loads go through LDS-DMA: each wave DMAs 1 KiB of a and of b into its own LDS slots, waits, reads them
// back (ds_read_b128), adds, stores.  Separates "LDS-DMA instead of register loads" from "7-KiB transposing tiles".
__global__ __launch_bounds__(256) void flat_dma(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ c, long long nv) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const long long v0 = ((long long)blockIdx.x * 4 + wave) * 64;       // first float4 of this wave
  if (v0 + 64 > nv) return;
  float* sa = smem + wave * 512; float* sb = sa + 256;
  __builtin_amdgcn_global_load_lds((gbl_cptr_t*)(a + (v0 + lane) * 4), (lds_ptr_t*)sa, 16, 0, 2);
  __builtin_amdgcn_global_load_lds((gbl_cptr_t*)(b + (v0 + lane) * 4), (lds_ptr_t*)sb, 16, 0, 2);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const v4f x = reinterpret_cast<const v4f*>(sa)[lane], y = reinterpret_cast<const v4f*>(sb)[lane];
  __builtin_nontemporal_store(x + y, reinterpret_cast<v4f*>(c) + v0 + lane);
}

// the 7-KiB tile structure with REGISTER loads (nontemporal global_load_dwordx4 -> ds_write_b128) instead of LDS-DMA
__global__ __launch_bounds__(256) void tile_reg(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ c, long long ntiles) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* sp = smem; float* st = smem + 1792;
  const int tid = threadIdx.x;
  const long long t = blockIdx.x;
  if (t >= ntiles) return;
  const v4f* ga = reinterpret_cast<const v4f*>(a + t * 1792); const v4f* gb = reinterpret_cast<const v4f*>(b + t * 1792);
  float* gc = c + t * 1792;
  v4f x0 = __builtin_nontemporal_load(ga + tid), y0 = __builtin_nontemporal_load(gb + tid), x1 = x0, y1 = y0;
  if (tid < 192) { x1 = __builtin_nontemporal_load(ga + tid + 256); y1 = __builtin_nontemporal_load(gb + tid + 256); }
  reinterpret_cast<v4f*>(sp)[tid] = x0; reinterpret_cast<v4f*>(st)[tid] = y0;
  if (tid < 192) { reinterpret_cast<v4f*>(sp)[tid + 256] = x1; reinterpret_cast<v4f*>(st)[tid + 256] = y1; }
  __syncthreads();
  float r[7];
#pragma unroll
  for (int k = 0; k < 7; ++k) r[k] = sp[tid * 7 + k] + st[tid * 7 + k];
#pragma unroll
  for (int k = 0; k < 7; ++k) sp[tid * 7 + k] = r[k];
  __syncthreads();
  __builtin_nontemporal_store(reinterpret_cast<const v4f*>(sp)[tid], reinterpret_cast<v4f*>(gc) + tid);
  if (tid < 192) __builtin_nontemporal_store(reinterpret_cast<const v4f*>(sp)[tid + 256], reinterpret_cast<v4f*>(gc) + tid + 256);
}

// Wave-owned sub-tiles: every wave moves and owns 64 rows (1792 B per tensor = 2 x 768-B dwordx3 DMA pieces + one 256-B
// dword piece, all 64 lanes active), no workgroup barrier at all; NW waves per workgroup (NW = 1: 64-thread workgroups).
template <int NW, int VALU>
__global__ __launch_bounds__(64 * NW) void tilew(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ c, long long nsub) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const long long t = (long long)blockIdx.x * NW + wave;
  if (t >= nsub) return;
  float* sp = smem + wave * 896; float* st = sp + 448;
  const float* ga = a + t * 448; const float* gb = b + t * 448; float* gc = c + t * 448;
  __builtin_amdgcn_global_load_lds((gbl_cptr_t*)(ga + lane * 3), (lds_ptr_t*)sp, 12, 0, 2);
  __builtin_amdgcn_global_load_lds((gbl_cptr_t*)(gb + lane * 3), (lds_ptr_t*)st, 12, 0, 2);
  __builtin_amdgcn_global_load_lds((gbl_cptr_t*)(ga + 192 + lane * 3), (lds_ptr_t*)(sp + 192), 12, 0, 2);
  __builtin_amdgcn_global_load_lds((gbl_cptr_t*)(gb + 192 + lane * 3), (lds_ptr_t*)(st + 192), 12, 0, 2);
  __builtin_amdgcn_global_load_lds((gbl_cptr_t*)(ga + 384 + lane), (lds_ptr_t*)(sp + 384), 4, 0, 2);
  __builtin_amdgcn_global_load_lds((gbl_cptr_t*)(gb + 384 + lane), (lds_ptr_t*)(st + 384), 4, 0, 2);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  float r[7];
#pragma unroll
  for (int k = 0; k < 7; ++k) r[k] = sp[lane * 7 + k] + st[lane * 7 + k];
#pragma unroll
  for (int it = 0; it < VALU; ++it) {
#pragma unroll
    for (int k = 0; k < 7; ++k) r[k] = __builtin_fmaf(r[k], 1.0000001f, 1e-9f);
  }
#pragma unroll
  for (int k = 0; k < 7; ++k) sp[lane * 7 + k] = r[k];
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // in-order LDS pipe of one wave: the writes above are visible below
  const v4f v0 = reinterpret_cast<const v4f*>(sp)[lane];
  __builtin_nontemporal_store(v0, reinterpret_cast<v4f*>(gc) + lane);
  if (lane < 48) {
    const v4f v1 = reinterpret_cast<const v4f*>(sp)[lane + 64];
    __builtin_nontemporal_store(v1, reinterpret_cast<v4f*>(gc) + lane + 64);
  }
}

int main() {
  const long long n = 17500000;  // float4 per buffer = 280 MB
  v4f *a, *b, *c;
  CK(hipMalloc(&a, n * 16)); CK(hipMalloc(&b, n * 16)); CK(hipMalloc(&c, n * 16));
  CK(hipMemset(a, 1, n * 16)); CK(hipMemset(b, 1, n * 16)); CK(hipMemset(c, 0, n * 16));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int iters = 30;
  const double bytes = 3.0 * n * 16;
  auto run = [&](const char* name, auto launch) {
    float best = 1e9f, sum = 0;
    for (int r = 0; r < 3; ++r) {
      for (int i = 0; i < 5; ++i) launch();
      hipDeviceSynchronize();
      hipEventRecord(e0);
      for (int i = 0; i < iters; ++i) launch();
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      ms /= iters; sum += ms; if (ms < best) best = ms;
    }
    printf("%-44s best %7.1f us  mean %7.1f us  %7.1f GB/s\n", name, best * 1e3, sum / 3 * 1e3, bytes / (best * 1e-3) / 1e9);
  };
#define FLAT(V, T) run("flat V=" #V " T=" #T, [&] { flat<V, T><<<(unsigned)((n + (V) * (T) - 1) / ((V) * (T))), T>>>(a, b, c, n); })
  FLAT(1, 256); FLAT(2, 256); FLAT(4, 256); FLAT(1, 512); FLAT(2, 512); FLAT(1, 1024); FLAT(1, 128); FLAT(1, 64); FLAT(2, 64); FLAT(4, 64);
  const long long ntiles = n / 448;
#define TILE(B, VL) run("tile448 LDS-DMA bar=" #B " valu=" #VL "x7fma", [&] { tile<B, VL><<<(unsigned)ntiles, 256, 2 * 1792 * 4>>>((const float*)a, (const float*)b, (float*)c, ntiles); })
  TILE(true, 0); TILE(false, 0); TILE(true, 16); TILE(true, 32); TILE(true, 48); TILE(false, 32);
  const long long nsub = n / 112;   // 64-row sub-tiles (112 float4 each)
#define TILEW(NW, VL) run("wave-owned 64-row subtiles waves/WG=" #NW " valu=" #VL "x7fma", [&] { tilew<NW, VL><<<(unsigned)((nsub + (NW) - 1) / (NW)), 64 * (NW), (NW) * 896 * 4>>>((const float*)a, (const float*)b, (float*)c, nsub); })
  TILEW(1, 0); TILEW(1, 32); TILEW(1, 48); TILEW(2, 0); TILEW(2, 32); TILEW(4, 0); TILEW(4, 32); TILEW(8, 32); TILEW(16, 32);
  FLAT(1, 256); FLAT(1, 64); TILE(true, 32); TILEW(1, 32); TILEW(2, 32);
  run("flat T=256 loads via LDS-DMA (1 KiB per wave)", [&] { flat_dma<<<(unsigned)(n / 256), 256, 4 * 512 * 4>>>((const float*)a, (const float*)b, (float*)c, n); });
  run("tile448 register loads + ds_write_b128", [&] { tile_reg<<<(unsigned)ntiles, 256, 2 * 1792 * 4>>>((const float*)a, (const float*)b, (float*)c, ntiles); });
  // occupancy cap by LDS size: workgroups per CU = min(8, 160 KiB / lds)
  for (int lds : {14336, 20480, 23400, 27300, 32768, 40960, 54600}) {
    char nm[96]; snprintf(nm, 96, "tile448 LDS-DMA bar valu=32  lds=%d B (%d WG/CU)", lds, (163840 / lds) < 8 ? (163840 / lds) : 8);
    run(nm, [&] { tile<true, 32><<<(unsigned)ntiles, 256, lds>>>((const float*)a, (const float*)b, (float*)c, ntiles); });
  }
  // VALU count under the occupancy cap: independent FMAs (7 chains) vs the same count as ONE dependent chain per thread
  for (int lds : {14336, 27300, 32768}) {
    char nm[96];
    snprintf(nm, 96, "tile448 valu=48x7 indep   lds=%d", lds); run(nm, [&] { tile<true, 48><<<(unsigned)ntiles, 256, lds>>>((const float*)a, (const float*)b, (float*)c, ntiles); });
    snprintf(nm, 96, "tile448 valu=64x7 indep   lds=%d", lds); run(nm, [&] { tile<true, 64><<<(unsigned)ntiles, 256, lds>>>((const float*)a, (const float*)b, (float*)c, ntiles); });
    snprintf(nm, 96, "tile448 270 dependent fma lds=%d", lds); run(nm, [&] { tile<true, -270><<<(unsigned)ntiles, 256, lds>>>((const float*)a, (const float*)b, (float*)c, ntiles); });
    snprintf(nm, 96, "tile448 270 dep, 20 rcp   lds=%d", lds); run(nm, [&] { tile<true, -1270><<<(unsigned)ntiles, 256, lds>>>((const float*)a, (const float*)b, (float*)c, ntiles); });
    snprintf(nm, 96, "tile448 270 dep, 20 rcp, 216 s_nop lds=%d", lds); run(nm, [&] { tile<true, -2270><<<(unsigned)ntiles, 256, lds>>>((const float*)a, (const float*)b, (float*)c, ntiles); });
  }
  FLAT(1, 256);
  return 0;
}
