// center_targets.hip — CenterPoint target assignment on the device, for gfx950 (include/gd3d.h, ABI 4): the producer of the
// heat maps, `anno_boxes` and `pos_inds` that the head's loss slice (gd3d_center_head_loss) consumes.
//
// The reference does it per sample, per task and per box in Python, with a host copy of every box size and a numpy Gaussian per box
//   /root/reference/mmdet3d_gaussian/models/dense_heads/gd_centerpoint_head.py:65-81 (get_targets), :83-156 (get_targets_single)
//   + mmdet3d's gaussian_radius / draw_heatmap_gaussian (third party, absent: restated from the published 0.x text).
// Here: two launches for all samples and tasks.
//   assign_kernel (one 1024-thread workgroup, up to 8192 boxes): per box the task and class of its label, the cell
//     x = trunc((x - pc0) / vs0 / osf) — `.long()` truncates toward zero, so a centre up to one cell left of the range lands
//     in cell 0, as in the reference —, validity (:124-126), the Gaussian radius in the reference's fp32 operation order
//     (0-dim tensor arithmetic of gaussian_radius, then max(min_radius, int(r))); an LDS bitonic sort of the key
//     (task, sample, class in task, box index) gives the reference's output order (tasks; samples in batch order; inside a
//     sample the classes of the task in turn, boxes of a class in index order, :97-113); then every valid box writes its
//     row of anno_boxes / pos_inds ([batch, x, y], :71-80) and a draw record; task_start[t] = first row of task t.
//   draw_kernel (one 256-thread workgroup per valid box): the (2r+1)^2 window of exp(-(dx^2+dy^2) / (2 sigma^2)), sigma =
//     (2r+1)/6, evaluated in fp64 and rounded to fp32 as numpy does, values below eps zeroed, clipped at the map border,
//     merged with integer atomicMax on the float bits (values are positive: the order of the bits is the order of the
//     floats, and max is commutative: deterministic).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/gd3d_extras.h"
#include "lds_sort.h"

namespace ctargets {

using ldssort::PH;

constexpr int T = 1024;
constexpr int MAX_BOXES = 8192;
constexpr int MAXT = CENTER_TARGETS_MAX_TASKS;
constexpr int MAXB = CENTER_TARGETS_MAX_BATCH;

struct Draw {          // one valid box, in output order
  int plane;           // float offset of its heat-map plane (task, sample, class) in the flat heat-map buffer
  int x, y, radius;
};

struct Args {
  const float* boxes;        // (total, cols)
  const long long* labels;   // (total)
  int total, cols, B, H, W, Tn, bottom_center;
  int sample_start[MAXB + 1];
  int class_start[MAXT + 1];   // task t owns labels [class_start[t], class_start[t + 1])
  int heat_start[MAXT + 1];    // float offset of task t's (B, C_t, H, W) block
  float pc0, pc1, vs0, vs1, osf;
  int min_radius;
  double overlap;            // the python float of train_cfg['gaussian_overlap'], as the reference's arithmetic sees it
  float* anno;               // (total, cols) rows in output order
  long long* pos;            // (total, 3) [batch, x, y]
  long long* task_start;     // (Tn + 1)
  Draw* draw;                // (total)
  float* heat;
};

// gaussian_radius((height, width) = (length, width), min_overlap) on float32 scalars, operation by operation (python numbers
// enter as float32 scalars after their own float64 arithmetic); this file is compiled with -ffp-contract=off
__device__ __forceinline__ float gaussian_radius(float height, float width, double mo) {
  const float b1 = height + width;
  const float c1 = width * height * (float)(1.0 - mo) / (float)(1.0 + mo);
  const float sq1 = sqrtf(b1 * b1 - 4.0f * c1);
  const float r1 = (b1 + sq1) / 2.0f;
  const float b2 = 2.0f * (height + width);
  const float c2 = (float)(1.0 - mo) * width * height;
  const float sq2 = sqrtf(b2 * b2 - 16.0f * c2);
  const float r2 = (b2 + sq2) / 2.0f;
  const float a3x4 = (float)(4.0 * (4.0 * mo));
  const float b3 = (float)(-2.0 * mo) * (height + width);
  const float c3 = (float)(mo - 1.0) * width * height;
  const float sq3 = sqrtf(b3 * b3 - a3x4 * c3);
  const float r3 = (b3 + sq3) / 2.0f;
  float m = r1;            // python's min(r1, r2, r3): a later value replaces the running one only if it compares smaller
  if (r2 < m) m = r2;
  if (r3 < m) m = r3;
  return m;
}

struct Cell {
  int x, y, radius, t, c, b;
  bool valid;
};

__device__ __forceinline__ Cell cell_of(const Args& a, int i) {
  Cell r;
  const float* row = a.boxes + (size_t)i * a.cols;
  const long long lab = a.labels[i];
  int t = -1;
  for (int q = 0; q < a.Tn; ++q)
    if (lab >= a.class_start[q] && lab < a.class_start[q + 1]) t = q;
  int b = 0;
  for (int q = 1; q < a.B; ++q) b += i >= a.sample_start[q] ? 1 : 0;
  const float width = row[3] / a.vs0 / a.osf;
  const float length = row[4] / a.vs1 / a.osf;
  const float fx = (row[0] - a.pc0) / a.vs0 / a.osf;
  const float fy = (row[1] - a.pc1) / a.vs1 / a.osf;
  // .long() of a float tensor truncates toward zero; values outside int64 are undefined there: treated as invalid here
  const bool finite = fabsf(fx) < 1.0e9f && fabsf(fy) < 1.0e9f;
  const int x = finite ? (int)fx : -1, y = finite ? (int)fy : -1;
  r.valid = t >= 0 && finite && width > 0.0f && length > 0.0f && x >= 0 && x < a.W && y >= 0 && y < a.H;
  r.x = x;
  r.y = y;
  r.t = t;
  r.b = b;
  r.c = t >= 0 ? (int)(lab - a.class_start[t]) : 0;
  r.radius = 0;
  if (r.valid) {
    const float rad = gaussian_radius(length, width, a.overlap);
    const int ri = (int)rad;              // int(radius): truncation
    r.radius = ri > a.min_radius ? ri : a.min_radius;
  }
  return r;
}

__global__ __launch_bounds__(T) void assign_kernel(const Args a) {
  extern __shared__ __attribute__((aligned(16))) unsigned long long keys[];   // padded list of P entries
  __shared__ int s_start[MAXT + 2];
  __shared__ int s_nvalid;
  const int tid = threadIdx.x;
  int P = 8;
  while (P < a.total) P <<= 1;
  if (tid <= a.Tn + 1) s_start[tid] = -1;
  if (tid == 0) s_nvalid = 0;
  __syncthreads();
  // key: (task, sample, class in task, index), INVERTED so that the descending sort yields the ascending order; an invalid
  // box gets 0, the padding value: both sink to the end
  int nv = 0;
  for (int i = tid; i < P; i += T) {
    unsigned long long k = 0ull;
    if (i < a.total) {
      const Cell c = cell_of(a, i);
      if (c.valid) {
        const unsigned long long asc = ((unsigned long long)((c.t * a.B + c.b) * 64 + c.c) << 32) | (unsigned long long)(unsigned)i;
        k = ~asc;
        ++nv;
      }
    }
    keys[PH(i)] = k;
  }
  if (nv) atomicAdd(&s_nvalid, nv);
  __syncthreads();
  ldssort::bitonic_desc(keys, P);          // the list is already padded to P entries
  const int nvalid = s_nvalid;
  // heads of the tasks' runs
  for (int r = tid; r < nvalid; r += T) {
    const unsigned long long asc = ~keys[PH(r)];
    const int t = (int)(asc >> 32) / 64 / a.B;
    const int tp = r > 0 ? (int)((~keys[PH(r - 1)]) >> 32) / 64 / a.B : -1;
    if (t != tp) s_start[t] = r;
  }
  __syncthreads();
  if (tid == 0) {             // a task without boxes starts where the next one does
    s_start[a.Tn] = nvalid;
    for (int t = a.Tn - 1; t >= 0; --t)
      if (s_start[t] < 0) s_start[t] = s_start[t + 1];
    for (int t = 0; t <= a.Tn; ++t) a.task_start[t] = s_start[t];
  }
  for (int r = tid; r < nvalid; r += T) {
    const unsigned long long asc = ~keys[PH(r)];
    const int i = (int)(unsigned)asc;
    const Cell c = cell_of(a, i);
    const float* row = a.boxes + (size_t)i * a.cols;
    float* out = a.anno + (size_t)r * a.cols;
    for (int j = 0; j < a.cols; ++j) out[j] = row[j];
    if (a.bottom_center) out[2] = row[2] + row[5] * 0.5f;      // gravity_center of a bottom-centred box (:85-87)
    a.pos[(size_t)r * 3] = c.b;
    a.pos[(size_t)r * 3 + 1] = c.x;
    a.pos[(size_t)r * 3 + 2] = c.y;
    Draw d;
    const int ct = a.class_start[c.t + 1] - a.class_start[c.t];
    d.plane = a.heat_start[c.t] + (c.b * ct + c.c) * a.H * a.W;
    d.x = c.x;
    d.y = c.y;
    d.radius = c.radius;
    a.draw[r] = d;
  }
}

__global__ __launch_bounds__(256) void draw_kernel(const Args a) {
  const long long nvalid = a.task_start[a.Tn];
  if ((long long)blockIdx.x >= nvalid) return;
  const Draw d = a.draw[blockIdx.x];
  const int radius = d.radius, diameter = 2 * radius + 1;
  const double sigma = (double)diameter / 6.0;
  const double denom = 2.0 * sigma * sigma;
  const int left = d.x < radius ? d.x : radius, right = (a.W - d.x) < (radius + 1) ? (a.W - d.x) : (radius + 1);
  const int top = d.y < radius ? d.y : radius, bottom = (a.H - d.y) < (radius + 1) ? (a.H - d.y) : (radius + 1);
  const int w = left + right, h = top + bottom;
  int* plane = (int*)(a.heat + d.plane);
  for (int q = threadIdx.x; q < w * h; q += 256) {
    const int yy = q / w, xx = q - yy * w;
    const int dx = xx - left, dy = yy - top;
    const double g = exp(-((double)(dx * dx) + (double)(dy * dy)) / denom);
    if (g < 2.220446049250313e-16) continue;          // h[h < eps * h.max()] = 0 (the centre value is 1)
    const float v = (float)g;
    atomicMax(&plane[(d.y + dy) * a.W + (d.x + dx)], __float_as_int(v));
  }
}

static size_t up256(size_t x) { return (x + 255) & ~(size_t)255; }

}  // namespace ctargets

using namespace ctargets;

extern "C" {

int center_targets_max_boxes(void) { return MAX_BOXES; }

size_t center_targets_workspace_bytes(int64_t total) {
  if (total < 1) total = 1;
  return up256(sizeof(Draw) * (size_t)total);
}

int center_targets_build(const center_targets_desc* d, const float* boxes, const int64_t* labels, void* workspace, float* heatmaps,
                         float* anno_boxes, int64_t* pos_inds, int64_t* task_start, void* stream) {
  if (d == nullptr || task_start == nullptr) return GD3D_E_BADARG;
  if (d->num_tasks < 1 || d->num_tasks > MAXT || d->batch < 1 || d->batch > MAXB || d->height < 1 || d->width < 1) return GD3D_E_BADARG;
  if (d->total < 0 || d->box_cols < 7) return GD3D_E_BADARG;
  if (d->total > MAX_BOXES) return GD3D_E_TOOLARGE;
  hipStream_t s = (hipStream_t)stream;
  if (d->total == 0) return (int)hipMemsetAsync(task_start, 0, sizeof(int64_t) * (size_t)(d->num_tasks + 1), s);
  if (boxes == nullptr || labels == nullptr || workspace == nullptr || heatmaps == nullptr || anno_boxes == nullptr ||
      pos_inds == nullptr || ((uintptr_t)workspace & 255) != 0)
    return GD3D_E_BADARG;
  Args a = {};
  a.boxes = boxes;
  a.labels = (const long long*)labels;
  a.total = d->total;
  a.cols = d->box_cols;
  a.B = d->batch;
  a.H = d->height;
  a.W = d->width;
  a.Tn = d->num_tasks;
  a.bottom_center = d->bottom_center;
  int64_t heat = 0;
  int cls = 0;
  for (int t = 0; t < d->num_tasks; ++t) {
    if (d->classes[t] < 1 || d->classes[t] > 64) return GD3D_E_BADARG;
    a.class_start[t] = cls;
    a.heat_start[t] = (int)heat;
    cls += d->classes[t];
    heat += (int64_t)d->batch * d->classes[t] * d->height * d->width;
    if (heat >= 0x7fffffffLL) return GD3D_E_TOOLARGE;
  }
  a.class_start[d->num_tasks] = cls;
  a.heat_start[d->num_tasks] = (int)heat;
  for (int b = 0; b <= d->batch; ++b) {
    a.sample_start[b] = d->sample_start[b];
    if (b > 0 && d->sample_start[b] < d->sample_start[b - 1]) return GD3D_E_BADARG;
  }
  if (d->sample_start[0] != 0 || d->sample_start[d->batch] != d->total) return GD3D_E_BADARG;
  a.pc0 = d->pc_range[0];
  a.pc1 = d->pc_range[1];
  a.vs0 = d->voxel_size[0];
  a.vs1 = d->voxel_size[1];
  a.osf = d->out_size_factor;
  a.overlap = d->gaussian_overlap;
  a.min_radius = d->min_radius;
  a.anno = anno_boxes;
  a.pos = (long long*)pos_inds;
  a.task_start = (long long*)task_start;
  a.draw = (Draw*)workspace;
  a.heat = heatmaps;
  int P = 8;
  while (P < d->total) P <<= 1;
  const size_t lds = sizeof(unsigned long long) * (size_t)(P + P / 8 + 8);
  static bool attr_set[64] = {};
  int devid = 0;
  if (hipGetDevice(&devid) != hipSuccess) return GD3D_E_BADARG;
  if (lds > 65536 && (devid < 0 || devid >= 64 || !attr_set[devid])) {
    const hipError_t e = hipFuncSetAttribute((const void*)assign_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                             (int)(sizeof(unsigned long long) * (MAX_BOXES + MAX_BOXES / 8 + 8)));
    if (e != hipSuccess) return (int)e;
    if (devid >= 0 && devid < 64) attr_set[devid] = true;
  }
  hipLaunchKernelGGL(assign_kernel, dim3(1), dim3(T), lds, s, a);
  hipLaunchKernelGGL(draw_kernel, dim3((unsigned)d->total), dim3(256), 0, s, a);
  return (int)hipGetLastError();
}

}  // extern "C"
