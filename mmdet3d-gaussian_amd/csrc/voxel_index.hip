// voxel_index.hip — from point coordinates to everything the dynamic scatter ops need, in one stream-ordered call for gfx950
// (SURVEY.md §8f-4).  Replaces what the reference does per SAMPLE with masked_fill + unique_dim
//   /root/reference/mmdet3d_gaussian/ops/voxel/scatter.py:101-117, ops/voxel/src/scatter_points_cuda.cu:221-251
// and what this package did in round 2 with ~20 ATen launches (mixed-radix key, torch.sort, unique_consecutive, cumsum,
// index writes).  Here:
//   extent_kernel  per-column extents of the batch (integer atomicMax: deterministic)
//   key_kernel     one int64 key per point, mixed radix over those extents, -1 for a point with a negative coordinate
//                  (dropped, scatter_points_cuda.cu:236-246)
//   radix sort     (key, point id) pairs, stable: ascending keys = lexicographically sorted unique rows = the reference's
//                  unique_dim order, samples in batch order; ascending point id inside a voxel = the grouping the atomic-free
//                  reduce / backward kernels of voxel_scatter.hip walk.  The sort is rocPRIM's device radix sort (the one
//                  library primitive on this path; torch.sort runs the same one).
//   head / scan / finish kernels: run heads -> voxel ids (inclusive scan), point -> voxel map, segment starts, counts,
//                  decoded voxel coordinates, and the two numbers the host needs to size its views: voxels and dropped points.
// Outputs are caller-allocated at their upper bounds (a voxel per point); nothing is read back inside the call.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

#include "../../include/gd3d.h"

namespace voxidx {

constexpr int MAX_DIM = 8;
constexpr int T = 256;

struct Layout {   // carve of the caller's workspace (all offsets multiples of 256 bytes)
  size_t ext, key_in, key_out, val_in, heads, vids, sort_tmp, scan_tmp, total;
  size_t sort_bytes, scan_bytes;
};

static size_t up256(size_t x) { return (x + 255) & ~(size_t)255; }

static hipError_t layout(int64_t n, Layout& L) {
  L.sort_bytes = 0;
  L.scan_bytes = 0;
  hipError_t e = rocprim::radix_sort_pairs(nullptr, L.sort_bytes, (const long long*)nullptr, (long long*)nullptr,
                                           (const int*)nullptr, (int*)nullptr, (size_t)n, 0, 64, (hipStream_t)0);
  if (e != hipSuccess) return e;
  e = rocprim::inclusive_scan(nullptr, L.scan_bytes, (const int*)nullptr, (int*)nullptr, (size_t)n, rocprim::plus<int>(),
                              (hipStream_t)0);
  if (e != hipSuccess) return e;
  size_t o = 0;
  L.ext = o; o += up256(sizeof(int) * MAX_DIM);
  L.key_in = o; o += up256(sizeof(long long) * (size_t)n);
  L.key_out = o; o += up256(sizeof(long long) * (size_t)n);
  L.val_in = o; o += up256(sizeof(int) * (size_t)n);
  L.heads = o; o += up256(sizeof(int) * (size_t)n);
  L.vids = o; o += up256(sizeof(int) * (size_t)n);
  L.sort_tmp = o; o += up256(L.sort_bytes);
  L.scan_tmp = o; o += up256(L.scan_bytes);
  L.total = o;
  return hipSuccess;
}

__global__ __launch_bounds__(T) void extent_kernel(const int* __restrict__ coors, long long n, int ndim, int* __restrict__ ext) {
  __shared__ int smax[MAX_DIM];
  if (threadIdx.x < MAX_DIM) smax[threadIdx.x] = 0;
  __syncthreads();
  int m[MAX_DIM];
#pragma unroll
  for (int d = 0; d < MAX_DIM; ++d) m[d] = 0;
  for (long long i = (long long)blockIdx.x * T + threadIdx.x; i < n; i += (long long)gridDim.x * T) {
#pragma unroll
    for (int d = 0; d < MAX_DIM; ++d)
      if (d < ndim) m[d] = max(m[d], coors[i * ndim + d]);
  }
#pragma unroll
  for (int d = 0; d < MAX_DIM; ++d) {
    if (d < ndim) {
      int v = m[d];
#pragma unroll
      for (int off = 32; off >= 1; off >>= 1) v = max(v, __shfl_down(v, off, 64));
      if ((threadIdx.x & 63) == 0) atomicMax(&smax[d], v);
    }
  }
  __syncthreads();
  if ((int)threadIdx.x < ndim) atomicMax(&ext[threadIdx.x], smax[threadIdx.x] + 1);   // extent = max(coordinate, 0) + 1
}

struct Radix {
  long long stride[MAX_DIM];
  long long ext[MAX_DIM];
};

__device__ __forceinline__ void radix_of(const int* __restrict__ ext, int ndim, Radix& r) {
  long long s = 1;
#pragma unroll
  for (int d = MAX_DIM - 1; d >= 0; --d) {
    if (d < ndim) {
      r.ext[d] = ext[d];
      r.stride[d] = s;
      s *= r.ext[d];
    }
  }
}

__global__ __launch_bounds__(T) void key_kernel(const int* __restrict__ coors, long long n, int ndim, const int* __restrict__ ext,
                                                long long* __restrict__ keys, int* __restrict__ vals) {
  const long long i = (long long)blockIdx.x * T + threadIdx.x;
  if (i >= n) return;
  Radix r;
  radix_of(ext, ndim, r);
  long long key = 0;
  bool neg = false;
#pragma unroll
  for (int d = 0; d < MAX_DIM; ++d) {
    if (d < ndim) {
      const int c = coors[i * ndim + d];
      neg |= c < 0;
      key += (long long)c * r.stride[d];
    }
  }
  keys[i] = neg ? -1LL : key;
  vals[i] = (int)i;
}

__global__ __launch_bounds__(T) void head_kernel(const long long* __restrict__ skey, long long n, int* __restrict__ heads) {
  const long long j = (long long)blockIdx.x * T + threadIdx.x;
  if (j >= n) return;
  heads[j] = (j == 0 || skey[j] != skey[j - 1]) ? 1 : 0;
}

// sorted position j: voxel id = (#heads up to j) - 1 - (is there a dropped bucket); dropped points get -1
__global__ __launch_bounds__(T) void finish_kernel(const long long* __restrict__ skey, const int* __restrict__ order,
                                                   const int* __restrict__ heads, const int* __restrict__ scan, long long n,
                                                   int ndim, const int* __restrict__ ext, int* __restrict__ pmap,
                                                   int* __restrict__ seg, int* __restrict__ voxel_coors,
                                                   long long* __restrict__ num) {
  const long long j = (long long)blockIdx.x * T + threadIdx.x;
  if (j >= n) return;
  const int has_drop = skey[0] < 0 ? 1 : 0;
  const long long key = skey[j];
  const int vid = key < 0 ? -1 : scan[j] - 1 - has_drop;
  pmap[order[j]] = vid;
  if (heads[j] && key >= 0) {
    seg[vid] = (int)j;
    Radix r;
    radix_of(ext, ndim, r);
#pragma unroll
    for (int d = 0; d < MAX_DIM; ++d)
      if (d < ndim) voxel_coors[(long long)vid * ndim + d] = (int)((key / r.stride[d]) % r.ext[d]);
  }
  if (j == n - 1) {
    const long long v = (long long)scan[j] - has_drop;   // keys are sorted: the last key is >= 0 unless every point is dropped
    // the mixed-radix key must fit 63 bits: with up to 8 columns of int32 extents the product can wrap, and wrapped keys
    // merge voxels or pose as dropped points.  Report it instead: num[0] = -1 (the host raises).
    long long prod = 1;
    bool wrapped = false;
    for (int d = 0; d < ndim; ++d) wrapped |= __builtin_mul_overflow(prod, (long long)ext[d], &prod);
    num[0] = wrapped ? -1LL : v;
    seg[v] = (int)n;
  }
  if (j == 0) {
    // dropped points = position of the first head with key >= 0; with no voxel at all it is n
    // (written by whoever finds it: the thread of that head, below; default here for the all-dropped batch)
    if (skey[n - 1] < 0) num[1] = n;
  }
  if (heads[j] && key >= 0 && (j == 0 || skey[j - 1] < 0)) num[1] = j;
}

__global__ __launch_bounds__(T) void counts_kernel(const int* __restrict__ seg, const long long* __restrict__ num,
                                                   long long n, int* __restrict__ counts) {
  const long long v = (long long)blockIdx.x * T + threadIdx.x;
  if (v < num[0] && v < n) counts[v] = seg[v + 1] - seg[v];
}

}  // namespace voxidx

using namespace voxidx;

extern "C" {

size_t vox_index_workspace_bytes(int64_t n, int32_t ndim) {
  if (n <= 0 || ndim <= 0 || ndim > MAX_DIM) return 256;
  Layout L;
  if (layout(n, L) != hipSuccess) return 0;
  return L.total;
}

int vox_index_build(const int32_t* coors, int64_t n, int32_t ndim, void* workspace, int32_t* point2voxel_map,
                    int32_t* order, int32_t* seg, int32_t* counts, int32_t* voxel_coors, int64_t* num, void* stream) {
  if (n < 0 || ndim <= 0 || ndim > MAX_DIM || num == nullptr) return GD3D_E_BADARG;
  if (n > 0x7fffffffLL) return GD3D_E_TOOLARGE;
  hipStream_t s = (hipStream_t)stream;
  if (n == 0) return (int)hipMemsetAsync(num, 0, 2 * sizeof(int64_t), s);
  if (coors == nullptr || workspace == nullptr || point2voxel_map == nullptr || order == nullptr || seg == nullptr ||
      counts == nullptr || voxel_coors == nullptr)
    return GD3D_E_BADARG;
  if (((uintptr_t)workspace & 255) != 0) return GD3D_E_BADARG;
  Layout L;
  hipError_t e = layout(n, L);
  if (e != hipSuccess) return (int)e;
  char* w = (char*)workspace;
  int* ext = (int*)(w + L.ext);
  long long* key_in = (long long*)(w + L.key_in);
  long long* key_out = (long long*)(w + L.key_out);
  int* val_in = (int*)(w + L.val_in);
  int* heads = (int*)(w + L.heads);
  int* vids = (int*)(w + L.vids);
  e = hipMemsetAsync(ext, 0, sizeof(int) * MAX_DIM, s);
  if (e != hipSuccess) return (int)e;
  e = hipMemsetAsync(num, 0, 2 * sizeof(int64_t), s);
  if (e != hipSuccess) return (int)e;
  const unsigned blocks = (unsigned)((n + T - 1) / T);
  const unsigned rblocks = blocks < 1024u ? blocks : 1024u;
  hipLaunchKernelGGL(extent_kernel, dim3(rblocks), dim3(T), 0, s, (const int*)coors, (long long)n, (int)ndim, ext);
  hipLaunchKernelGGL(key_kernel, dim3(blocks), dim3(T), 0, s, (const int*)coors, (long long)n, (int)ndim, (const int*)ext, key_in,
                     val_in);
  size_t sb = L.sort_bytes;
  e = rocprim::radix_sort_pairs((void*)(w + L.sort_tmp), sb, (const long long*)key_in, key_out, (const int*)val_in, (int*)order,
                                (size_t)n, 0, 64, s);
  if (e != hipSuccess) return (int)e;
  hipLaunchKernelGGL(head_kernel, dim3(blocks), dim3(T), 0, s, (const long long*)key_out, (long long)n, heads);
  size_t cb = L.scan_bytes;
  e = rocprim::inclusive_scan((void*)(w + L.scan_tmp), cb, (const int*)heads, vids, (size_t)n, rocprim::plus<int>(), s);
  if (e != hipSuccess) return (int)e;
  hipLaunchKernelGGL(finish_kernel, dim3(blocks), dim3(T), 0, s, (const long long*)key_out, (const int*)order, (const int*)heads,
                     (const int*)vids, (long long)n, (int)ndim, (const int*)ext, (int*)point2voxel_map, (int*)seg,
                     (int*)voxel_coors, (long long*)num);
  hipLaunchKernelGGL(counts_kernel, dim3(blocks), dim3(T), 0, s, (const int*)seg, (const long long*)num, (long long)n, (int*)counts);
  return (int)hipGetLastError();
}

}  // extern "C"
