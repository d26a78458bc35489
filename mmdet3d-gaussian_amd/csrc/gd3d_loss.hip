// gd3d_loss.hip — fused forward+gradient kernels of the Gaussian-distance losses for gfx950
// and their C-ABI entry points (include/gd3d.h).
//
// Data layout in HBM: pred / target / grad_* are (N,7) fp32 row-major (28-byte rows, exactly what
// the reference's bbox coders hand to GDLoss.forward, gaussian_distance_loss.py:280-310);
// loss / row_weight are (N,) fp32.
//
// Kernel shape (HBM-bound: 88 algorithmic bytes per pair, ~300 VALU ops per pair):
//   * one workgroup = 256 threads = one tile of 256 pairs = 7168 contiguous bytes per tensor
//     = exactly 7 wave-wide 1-KiB LDS-DMA pieces (global_load_lds_dwordx4): 14 pieces bring the
//     pred and target tiles into LDS with no VGPR round trip and fully coalesced 16-B lanes;
//   * thread i then reads row i (7 dwords at stride 7: coprime with the 32 banks, conflict-free),
//     does all the 2x2 algebra in registers, and writes its 7 gradient dwords back into ITS OWN
//     LDS row (no barrier needed for that hand-back);
//   * after one barrier the tile leaves as 448 coalesced 16-B stores; the per-pair loss as one
//     coalesced dword store; the tile's loss sum goes wave-shuffle -> LDS -> one fp32 partial per
//     block, and a second tiny kernel adds the partials in a fixed order in fp64 (deterministic,
//     no float atomics).  Finishing the sum INSIDE this kernel (sc1-stored partial, drained, agent-scope arrival
//     ticket per 64-tile group, last arriver adds) was built and measured in round 2: 516 us per 3-loss step against
//     421 us with the separate launch (profiles/r02_ticket_ab.txt) — the drain + ticket round trip keeps one wave of
//     every workgroup resident ~30 % longer; removed.
//   * tiles that are partial (the last one) or whose base pointers are not 16-B aligned take a
//     guarded scalar load/store path around the same compute code.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdlib>

#include "gd3d_device.h"

namespace gd3d {

#ifndef GD_TILE
#define GD_TILE 256                  // pairs (= threads) per workgroup; multiples of 256 give whole 1-KiB DMA pieces.
#endif                               // r01 A/B inside bench.py under rocprofv3 (tools/ab_rocprof.sh): 512 halves the
                                     // partials (reduce stage 6.9 -> 4.9 us) and is neutral for the fused kernel once
                                     // it is below ~300 VALU/pair, but gave only +0.5 % step throughput with ~1 us
                                     // longer event-bracketed kernels (noise level); 1024 is 4 % slower.  256 ships.
constexpr int TILE = GD_TILE;        // pairs per workgroup
constexpr int TILE_F = TILE * 7;     // floats per tensor tile (1792)
constexpr int TILE_V4 = TILE_F / 4;  // 16-byte vectors per tensor tile (448)
constexpr int NPIECE = TILE_F / 256; // 1-KiB LDS-DMA pieces per tensor tile (7)
constexpr int NWAVE = TILE / 64;     // waves per workgroup (4)
constexpr int HEAD_T = 256;          // threads per workgroup of the head-level gather kernel

typedef __attribute__((address_space(3))) void lds_ptr_t;
typedef const __attribute__((address_space(1))) void gbl_cptr_t;

// ---- tuning switches (A/B-tested with tools/build_variants.py + tools/kernel_time.py) ----
#ifndef GD_NT_LOAD
#define GD_NT_LOAD 1   // LDS-DMA loads with the nt cache policy: every input byte is read exactly once
                       // (measured r01, 10 M pairs: nt loads + nt stores 129 us vs 151 us plain; a persistent
                       //  double-buffered grid-stride variant was 143-160 us and was dropped)
#endif
// Occupancy cap.  The fused launch requests AT LEAST this much dynamic LDS, i.e. floor(160 KiB / bytes) workgroups per CU
// instead of the 8 that its own 14.4 KiB would admit: fewer tiles in flight per CU stream better (tools/hbm_probe2, the
// kernel's data path without its math: 8 WG/CU 131.0 us, 6: 130.2, 5: 126.7, 4: 126.8, 3: 162; flat copy 127.5), but
// fewer waves hide less VALU latency, so the best cap depends on the loss.  Measured per loss at FIXED buffer placement
// (tools/lds_fixed_placement.py, profiles/r02_lds_fixed_placement.txt; 10 M pairs, us per launch, good / bad placement):
//   WG/CU   gwd3d (217 VALU/pair)   kld3d (270)      bd3d (296)
//     8     134.1 / 137.6           134.2 / 139.5    134.5 / 139.1
//     7     133.1 / 137.1           131.2 / 138.8    133.0 / 138.3
//     6     131.7 / 136.4           129.6 / 138.0    130.1 / 137.6
//     5     131.4 / 134.7           135.8 / 137.6    140.0 / 140.9
//     4     132.2 / 134.6           136.0 / 136.8    139.9 / 140.3
#ifndef GD_MIN_LDS
#define GD_MIN_LDS 27300      // 6 workgroups per CU
#endif
#ifndef GD_MIN_LDS_GWD
#define GD_MIN_LDS_GWD 32768  // 5
#endif
#ifndef GD_MIN_LDS_W7
#define GD_MIN_LDS_W7 32768   // 5, launches with (N,7) weights (any loss)
#endif
#ifndef GD_MIN_LDS_KLD
#define GD_MIN_LDS_KLD GD_MIN_LDS
#endif
#ifndef GD_MIN_LDS_BD
#define GD_MIN_LDS_BD GD_MIN_LDS
#endif
#ifndef GD_PLAIN_ALL
#define GD_PLAIN_ALL 0   // 1: the option-free instantiation for every loss type (gwd3d has it regardless), see launch_one
#endif
#ifndef GD_NT_STORE
#define GD_NT_STORE 1  // nontemporal 16-B gradient stores: written once, never re-read by this kernel
#endif
constexpr int DMA_AUX = GD_NT_LOAD ? 2 : 0;

typedef float v4f __attribute__((ext_vector_type(4)));

GD_DEV void store_v4(float* dst, const float* src_lds, int idx) {
  const v4f v = reinterpret_cast<const v4f*>(src_lds)[idx];
  if (GD_NT_STORE) __builtin_nontemporal_store(v, reinterpret_cast<v4f*>(dst) + idx);
  else reinterpret_cast<v4f*>(dst)[idx] = v;
}

// 14 LDS-DMA pieces of 1 KiB bring one 256-pair tile of pred and target into LDS; wave w issues pieces w, w+4, w+8, w+12.
// The two tiles are adjacent in LDS (st = sp + TILE_F = sp + 7 pieces), so piece j of the 14 lands at sp + j * 256 and only
// its SOURCE depends on which tensor it belongs to — a scalar select, no branch.  (Written as a loop over both tensors
// with per-piece if / else-if, the compiler built a tree of ~60 scalar compares and branches in front of the first load:
// issue slots on every workgroup's critical path, profiles/r02_pmc_issue_mix_by_cap.txt.)
// (an optional third tile, the (256,7) weights, sits behind the wave sums and keeps its own loop)
GD_DEV void issue_tile_dma(const float* gpred, const float* gtarget, float* sp, float* st, int wave, int lane,
                           const float* gw7, float* sw7) {
  static_assert(TILE_F == NPIECE * 256, "pred and target tiles must be whole pieces for the adjacent-piece addressing");
  (void)st;
  const float* const t_shifted = gtarget - NPIECE * 256;   // so that piece j >= NPIECE reads target piece j - NPIECE
#pragma unroll
  for (int k = 0; k * NWAVE < 2 * NPIECE; ++k) {
    const int j = wave + k * NWAVE;  // wave-uniform
    if ((k + 1) * NWAVE <= 2 * NPIECE || j < 2 * NPIECE) {   // only the last round needs the runtime test
      const float* src = (j < NPIECE ? gpred : t_shifted) + j * 256 + lane * 4;
      __builtin_amdgcn_global_load_lds((gbl_cptr_t*)src, (lds_ptr_t*)(sp + j * 256), 16, 0, DMA_AUX);
    }
  }
  if (gw7 != nullptr) {
#pragma unroll
    for (int k = 0; k * NWAVE < NPIECE; ++k) {
      const int j = wave + k * NWAVE;
      if ((k + 1) * NWAVE <= NPIECE || j < NPIECE)
        __builtin_amdgcn_global_load_lds((gbl_cptr_t*)(gw7 + j * 256 + lane * 4), (lds_ptr_t*)(sw7 + j * 256), 16, 0, DMA_AUX);
    }
  }
}

struct LossArgs {
  const float* pred;
  const float* target;
  const float* w;    // nullable: (N,) row weights
  const float* w7;   // nullable: (N,7) weights, row mean taken in the kernel (GDLoss.forward :295-296)
  float* loss;       // nullable
  float* gp;         // nullable
  float* gt;         // nullable (only read when the kernel is instantiated with GT)
  float* partials;   // nullable
  // GDLoss.forward's early-out when no weight entry is > 0 (gaussian_distance_loss.py:290-292), resolved on the device:
  // with wsel the block also leaves sum(pred * weight7) in partials[nbp + b] and "any weight > 0" in partials[2 nbp + b]
  int wsel;
  long long nbp;     // partial-array stride (number of tiles rounded up to 4)
  // single-launch form for small problems (<= FIN_MAX_TILES tiles): the workgroup whose arrival ticket comes last adds
  // the partials itself (fixed order, fp64) and writes the result — no reduce launch.  fin: device int32, 0 at launch,
  // left 0; fin_out: the loss sum; fin_any: the any-positive flag of a selecting call (nullable).
  int* fin;
  float* fin_out;
  int* fin_any;
  long long n;
  float scale, alpha, ia2, tau;
  float c0, c1, c2;
  int vec_ok;        // all (N,7) pointers 16-byte aligned
  // bbox-coder decode fused into the prologue (include/gd3d.h gd3d_prologue)
  int pro;
  int norm_bbox;
  const float* aux;
  float osf, vs0, vs1, pc0, pc1;
};

// Decode the encoded rows in registers and remember what the chain rule needs.
//   ANCHOR_DELTA: j = (diag, diag, ha, w, l, h, 1) with the z/h cross term handled in encode_grad()
struct DecodeJac {
  float j[7];
};

GD_DEV void decode_anchor(const float (&enc)[7], const float (&an)[7], float (&dec)[7], DecodeJac& J) {
  const float diag = fsqrt(fmaf(an[4], an[4], an[3] * an[3]));
  const float w = expf(enc[3]) * an[3], l = expf(enc[4]) * an[4], h = expf(enc[5]) * an[5];
  dec[0] = fmaf(enc[0], diag, an[0]);
  dec[1] = fmaf(enc[1], diag, an[1]);
  dec[2] = fmaf(enc[2], an[5], an[2] + an[5] * 0.5f) - h * 0.5f;
  dec[3] = w;
  dec[4] = l;
  dec[5] = h;
  dec[6] = enc[6] + an[6];
  J.j[0] = diag; J.j[1] = diag; J.j[2] = an[5]; J.j[3] = w; J.j[4] = l; J.j[5] = h; J.j[6] = 1.0f;
}

GD_DEV void decode_center(const float (&enc)[7], float loc0, float loc1, const LossArgs& a, float (&dec)[7],
                          DecodeJac& J) {
  dec[0] = (enc[0] + loc0) * a.osf * a.vs0 + a.pc0;
  dec[1] = (enc[1] + loc1) * a.osf * a.vs1 + a.pc1;
  dec[2] = enc[2];
#pragma unroll
  for (int k = 3; k < 6; ++k) {
    dec[k] = a.norm_bbox ? expf(enc[k]) : enc[k];
    J.j[k] = a.norm_bbox ? dec[k] : 1.0f;
  }
  dec[6] = enc[6];
  J.j[0] = a.osf * a.vs0; J.j[1] = a.osf * a.vs1; J.j[2] = 1.0f; J.j[6] = 1.0f;
}

// gradient wrt the decoded row -> gradient wrt the encoded row (in place)
GD_DEV void encode_grad(float (&g)[7], const DecodeJac& J, bool anchor_kind) {
  const float gz = g[2];
#pragma unroll
  for (int k = 0; k < 7; ++k) g[k] *= J.j[k];
  if (anchor_kind) g[5] = fmaf(-0.5f * gz, J.j[5], g[5]);  // z = ... - h/2 with h = exp(ht)*ha
}

// wave64 sum with DPP adds (no LDS crossbar): inclusive scan inside each 16-lane row (row_shr 1,2,4,8 with
// zero fill), then row_bcast:15 / row_bcast:31 carry the row totals; lane 63 holds the total.  Fixed order.
template <int CTRL, int ROW_MASK>
GD_DEV float dpp_add(float v) {
  const int moved = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xf, true);
  return v + __builtin_bit_cast(float, moved);
}
GD_DEV float wave_sum(float v) {
  v = dpp_add<0x111, 0xf>(v);  // row_shr:1
  v = dpp_add<0x112, 0xf>(v);  // row_shr:2
  v = dpp_add<0x114, 0xf>(v);  // row_shr:4
  v = dpp_add<0x118, 0xf>(v);  // row_shr:8
  v = dpp_add<0x142, 0xa>(v);  // row_bcast:15 into rows 1 and 3
  v = dpp_add<0x143, 0xc>(v);  // row_bcast:31 into rows 2 and 3
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

// PLAIN: the launcher has checked that none of the options is in use (no weights, no selection, no prologue, no per-pair
// loss output, no single-launch finish); folding them to constants here removes their uniform tests and the kernarg loads
// behind them from every wave's issue stream (profiles/r02_pmc_issue_mix_by_cap.txt: issue slots are what the streaming
// launch has too many of).  Same code otherwise.
template <int LOSS, int FUN, bool FLAG, bool GT, bool PLAIN = false>
__global__ __launch_bounds__(TILE) void fused_kernel(const LossArgs a_in) {
  LossArgs a = a_in;
  if (PLAIN) {
    a.w = nullptr;
    a.w7 = nullptr;
    a.wsel = 0;
    a.pro = GD3D_PRO_NONE;
    a.aux = nullptr;
    a.loss = nullptr;
    a.fin = nullptr;
    a.fin_out = nullptr;
    a.fin_any = nullptr;
  }
  // dynamic LDS (16-byte aligned base, every carve offset a multiple of 16): two tiles + 32 floats of per-wave sums
  // (loss | pred*weight | any weight > 0), plus a third tile only when (N,7) weights are given, so that the common
  // launch keeps 8 workgroups per CU
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* const sp = smem;
  float* const st = smem + TILE_F;
  float* const swave = smem + 2 * TILE_F;
  float* const sw7 = smem + 2 * TILE_F + 32;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // provably wave-uniform: scalar branches below
  const long long base = (long long)blockIdx.x * TILE;
  const long long rows_left = a.n - base;
  const bool full = rows_left >= TILE;
  const bool fast = full && a.vec_ok;  // workgroup-uniform
  const bool valid = tid < rows_left;
  const float* gpred = a.pred + base * 7;
  const float* gtarget = a.target + base * 7;

  float wi = 1.0f;
  if (a.w != nullptr && valid) wi = a.w[base + tid];

  const float* gw7 = a.w7 != nullptr ? a.w7 + base * 7 : nullptr;
  if (fast) {
    issue_tile_dma(gpred, gtarget, sp, st, wave, lane, gw7, sw7);
  } else {
    const long long fl = (rows_left < TILE ? rows_left : TILE) * 7;
    for (int i = tid; i < TILE_F; i += TILE) {
      sp[i] = i < fl ? gpred[i] : 1.0f;
      st[i] = i < fl ? gtarget[i] : 1.0f;
      if (gw7 != nullptr) sw7[i] = i < fl ? gw7[i] : 0.0f;
    }
  }
  __syncthreads();  // s_waitcnt vmcnt(0) + barrier: every wave's pieces have landed

  if (a.w7 != nullptr) {  // weight.mean(dim=-1): sum of the 7 entries in index order, then / 7
    float sum = sw7[tid * 7];
#pragma unroll
    for (int k = 1; k < 7; ++k) sum += sw7[tid * 7 + k];
    wi = sum / 7.0f;
  }

  float pv[7], tv[7];
#pragma unroll
  for (int k = 0; k < 7; ++k) {
    pv[k] = sp[tid * 7 + k];
    tv[k] = st[tid * 7 + k];
  }
  const float c[3] = {a.c0, a.c1, a.c2};
  const float f = a.scale * wi;
  float g1[7], g2[7];
  DecodeJac Jp, Jt;
  if (a.pro != GD3D_PRO_NONE) {  // workgroup-uniform; head-level calls are small (P <~ 1e4), aux is read directly
    if (a.pro == GD3D_PRO_ANCHOR_DELTA) {
      float an[7], dp[7], dt[7];
#pragma unroll
      for (int k = 0; k < 7; ++k) an[k] = valid ? a.aux[(base + tid) * 7 + k] : 1.0f;
      decode_anchor(pv, an, dp, Jp);
      decode_anchor(tv, an, dt, Jt);
#pragma unroll
      for (int k = 0; k < 7; ++k) {
        pv[k] = dp[k];
        tv[k] = dt[k];
      }
    } else {
      float dp[7];
      const float l0 = valid ? a.aux[(base + tid) * 2] : 0.0f, l1 = valid ? a.aux[(base + tid) * 2 + 1] : 0.0f;
      decode_center(pv, l0, l1, a, dp, Jp);
#pragma unroll
      for (int k = 0; k < 7; ++k) {
        pv[k] = dp[k];
        Jt.j[k] = 1.0f;
      }
    }
  }
  float alt = 0.0f;      // wsel: this row's share of (pred * weight).sum() (ref :292; pred = the DECODED row)
  bool anyp = false;     //       and of torch.any(weight > 0) (ref :290; a NaN weight is not > 0)
  if (a.wsel && valid) { // uniform && per-thread
#pragma unroll
    for (int k = 0; k < 7; ++k) {
      const float wk = sw7[tid * 7 + k];
      anyp |= wk > 0.0f;
      alt = fmaf(pv[k], wk, alt);
    }
  }
#ifdef GD_FAKE_MATH   // experiment builds only: the tile mechanics without the closed forms (what is the floor of this kernel shape?)
  float L = 0.0f;
#pragma unroll
  for (int k = 0; k < 7; ++k) {
    g1[k] = f * (pv[k] - tv[k]);
    g2[k] = -g1[k];
    L = fmaf(g1[k], g1[k], L);
  }
#else
  const float L = pair_loss<LOSS, FUN, FLAG, GT>(pv, tv, c, a.alpha, a.ia2, a.tau, f, g1, g2);
#endif
  const float fl = valid ? f * L : 0.0f;
  if (a.pro != GD3D_PRO_NONE) {
    encode_grad(g1, Jp, a.pro == GD3D_PRO_ANCHOR_DELTA);
    if (GT) encode_grad(g2, Jt, a.pro == GD3D_PRO_ANCHOR_DELTA);
  }

  if (a.loss != nullptr && valid) a.loss[base + tid] = fl;

  if (a.gp != nullptr) {
#pragma unroll
    for (int k = 0; k < 7; ++k) sp[tid * 7 + k] = g1[k];  // own row: no hazard with other threads
  }
  if (GT) {
#pragma unroll
    for (int k = 0; k < 7; ++k) st[tid * 7 + k] = g2[k];
  }

  float bsum = 0.0f;
  if (a.partials != nullptr) {
    const float ws = wave_sum(fl);  // uniform (readlane 63)
    if (lane == 0) swave[wave] = ws;
    if (a.wsel) {                   // uniform
      const float wa = wave_sum(alt);
      const bool any_w = __builtin_amdgcn_ballot_w64(anyp) != 0ull;
      if (lane == 0) {
        swave[NWAVE + wave] = wa;
        swave[2 * NWAVE + wave] = any_w ? 1.0f : 0.0f;
      }
    }
  }
  __syncthreads();
  if (a.partials != nullptr && tid == 0) {
#pragma unroll
    for (int w4 = 0; w4 < NWAVE; w4 += 4) bsum += (swave[w4] + swave[w4 + 1]) + (swave[w4 + 2] + swave[w4 + 3]);
    float asum = 0.0f, fany = 0.0f;
    if (a.wsel) {
#pragma unroll
      for (int w4 = 0; w4 < NWAVE; ++w4) {
        asum += swave[NWAVE + w4];
        fany += swave[2 * NWAVE + w4];
      }
    }
    if (a.fin == nullptr) {
      a.partials[blockIdx.x] = bsum;
      if (a.wsel) {
        a.partials[a.nbp + blockIdx.x] = asum;
        a.partials[2 * a.nbp + blockIdx.x] = fany;   // > 0: some weight of this tile is > 0
      }
    } else {   // handed to the last workgroup: write-through (sc1) stores, see the finish below
      __hip_atomic_store(a.partials + blockIdx.x, bsum, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (a.wsel) {
        __hip_atomic_store(a.partials + a.nbp + blockIdx.x, asum, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(a.partials + 2 * a.nbp + blockIdx.x, fany, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
  }

  if (fast) {
    if (a.gp != nullptr) {
      store_v4(a.gp + base * 7, sp, tid);
      if (tid < TILE_V4 - TILE) store_v4(a.gp + base * 7, sp, tid + TILE);  // 7/4 vectors per thread
    }
    if (GT) {
      store_v4(a.gt + base * 7, st, tid);
      if (tid < TILE_V4 - TILE) store_v4(a.gt + base * 7, st, tid + TILE);
    }
  } else {
    const long long fl7 = (rows_left < TILE ? rows_left : TILE) * 7;
    if (a.gp != nullptr)
      for (int i = tid; i < fl7; i += TILE) a.gp[base * 7 + i] = sp[i];
    if (GT)
      for (int i = tid; i < fl7; i += TILE) a.gt[base * 7 + i] = st[i];
  }

  // Single-launch finish (small grids only; the same hand-off cost 20 % in the streaming regime and is not used there,
  // profiles/r02_ticket_ab.txt).  Form: MI355X_MICROARCH.md, valid forms, row 1 — every handed-off word was stored sc1 by
  // lane 0 of its workgroup; that wave drains vmcnt, then takes an agent-scope ticket; the wave whose add returned last
  // loads the words with sc1 loads, one per lane, and adds them in lane order in fp64.  The ticket returns to 0.
  if (a.fin != nullptr && wave == 0) {   // uniform
    const int nb = (int)gridDim.x;       // <= 64: one partial per lane
    int t = 0;
    if (lane == 0) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      t = __hip_atomic_fetch_add(a.fin, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    t = __builtin_amdgcn_readfirstlane(t);
    if (t == nb - 1) {
      double v[3] = {0.0, 0.0, 0.0};
      const int terms = a.wsel ? 3 : 1;
      for (int k = 0; k < terms; ++k) {
        if (lane < nb)
          v[k] = (double)__hip_atomic_load(a.partials + (long long)k * a.nbp + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) v[k] += __shfl_down(v[k], off, 64);
      }
      if (lane == 0) {
        const bool any = !a.wsel || v[2] > 0.0;
        *a.fin_out = (float)(any ? v[0] : v[1]);
        if (a.fin_any != nullptr) *a.fin_any = any ? 1 : 0;
        __hip_atomic_store(a.fin, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------------
// Anchor-head slice with the gather fused in (gd_anchor3d_head.py:95-141): one thread per POSITIVE.
// It reads its 7 encoded predictions straight out of the NCHW head output (B, A*7, H, W) — no permute/reshape copy,
// no index kernels — its target / weight rows from the (M,7) arrays and its anchor from the per-sample anchor list,
// decodes both boxes (DeltaXYZWLHR), evaluates the loss and scatters the chained gradient back into the NCHW
// gradient (pre-zeroed by the caller).  P is O(1e2..1e4): latency-bound, so no LDS tiling.
struct HeadArgs {
  const float* bbox_pred;      // (B, A*7, H, W)
  const float* bbox_targets;   // (M,7), M = B*H*W*A, row m = ((b*H + h)*W + w)*A + a
  const float* bbox_weights;   // (M,7) nullable
  const float* anchors;        // (H*W*A, 7) anchors of one sample
  const long long* pos_inds;   // (P) positive rows, or NULL: dense mode, thread m tests labels[m] itself
  const long long* labels;     // dense mode: (M) class labels; positive iff 0 <= label < num_classes
  int num_classes;
  float* grad_bbox_pred;       // (B, A*7, H, W), zero-filled by the caller; nullable
  float* partials;
  long long P;
  int A, H, W;
  float dw[7];                 // train_cfg['decode_weight'] (all 1 when weights are given without it)
  float scale, alpha, ia2, tau, c0, c1, c2;
  // encoded-box SmoothL1 term of loss_single (gd_anchor3d_head.py:152-159), added to the same sum / gradient
  int dw_on;                   // GD term weighted by mean_k(bbox_weights * dw); else unweighted
  int sl1;                     // 0: off
  int sl1_cw;                  // element weight = bbox_weights * cw (train_cfg['code_weight']); else 1
  int sin_diff;                // diff_rad_by_sin: add_sin_difference on the yaw column
  float beta, sl1_scale;       // SmoothL1Loss.beta (0 = L1Loss), loss_weight / avg_factor
  float cw[7];
  // device-resident normaliser (ABI 4): when avg_dev != NULL the two scales are w_gd / *avg_dev and w_sl1 / *avg_dev, divided in
  // double and rounded once, as the host does with a host-side avg_factor
  const float* avg_dev;
  double w_gd, w_sl1;
};

template <int LOSS, int FUN, bool FLAG>
__global__ __launch_bounds__(HEAD_T) void head_anchor_kernel(const HeadArgs a) {
  __shared__ float swave[HEAD_T / 64];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const long long i = (long long)blockIdx.x * HEAD_T + tid;
  bool valid = i < a.P;
  long long m = i;
  if (valid) {
    if (a.pos_inds != nullptr) {
      m = a.pos_inds[i];
    } else {  // dense mode: no nonzero()/compaction/host sync upstream; non-positives leave here
      const long long lab = a.labels[i];
      valid = lab >= 0 && lab < a.num_classes;
    }
  }
  float fl = 0.0f;
  if (valid) {
    const long long hwa = (long long)a.H * a.W * a.A;
    const long long b = m / hwa, r = m - b * hwa;
    const int an_i = (int)(r % a.A);
    const long long hw = r / a.A;                                  // h*W + w
    const long long plane = (long long)a.H * a.W;
    const float* pbase = a.bbox_pred + ((b * a.A + an_i) * 7) * plane + hw;   // + k*plane per channel
    float pe[7], te[7], an[7], pv[7], tv[7], wrow[7];
    float wi = 1.0f;
#pragma unroll
    for (int k = 0; k < 7; ++k) {
      pe[k] = pbase[k * plane];
      te[k] = a.bbox_targets[m * 7 + k];
      an[k] = a.anchors[r * 7 + k];
      wrow[k] = a.bbox_weights != nullptr ? a.bbox_weights[m * 7 + k] : 1.0f;
    }
    if (a.dw_on) {
      float sum = wrow[0] * a.dw[0];
#pragma unroll
      for (int k = 1; k < 7; ++k) sum += wrow[k] * a.dw[k];
      wi = sum / 7.0f;
    }
    DecodeJac Jp, Jt;
    decode_anchor(pe, an, pv, Jp);
    decode_anchor(te, an, tv, Jt);
    const float c[3] = {a.c0, a.c1, a.c2};
    float gd_scale = a.scale, sl1_scale = a.sl1_scale;
    if (a.avg_dev != nullptr) {
      const double avg = (double)*a.avg_dev;
      gd_scale = (float)(a.w_gd / avg);
      sl1_scale = (float)(a.w_sl1 / avg);
    }
    const float f = gd_scale * wi;
    float g1[7], g2[7];
    const float L = pair_loss<LOSS, FUN, FLAG, false>(pv, tv, c, a.alpha, a.ia2, a.tau, f, g1, g2);
    fl = f * L;
    float gs[7] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (a.sl1) {  // uniform.  mmdet smooth_l1_loss on the ENCODED rows, weight (P,7), sum / avg_factor
      float d[7], j6 = 1.0f;
#pragma unroll
      for (int k = 0; k < 6; ++k) d[k] = pe[k] - te[k];
      if (a.sin_diff) {  // add_sin_difference: sin(p)cos(t) vs cos(p)sin(t); both sides depend on the prediction
        float sp6, cp6, st6, ct6;
        sincos_f(pe[6], sp6, cp6);
        sincos_f(te[6], st6, ct6);
        d[6] = sp6 * ct6 - cp6 * st6;
        j6 = cp6 * ct6 + sp6 * st6;
      } else {
        d[6] = pe[6] - te[6];
      }
      float ls = 0.0f;
#pragma unroll
      for (int k = 0; k < 7; ++k) {
        const float ad = fabsf(d[k]);
        const bool quad = ad < a.beta;
        const float l = quad ? 0.5f * ad * ad / a.beta : ad - 0.5f * a.beta;
        const float sg = d[k] > 0.0f ? 1.0f : (d[k] < 0.0f ? -1.0f : 0.0f);   // torch abs'(0) = 0
        const float g = quad ? d[k] / a.beta : sg;
        const float w = a.sl1_cw ? wrow[k] * a.cw[k] : 1.0f;
        ls += l * w;
        gs[k] = g * w * sl1_scale * (k == 6 ? j6 : 1.0f);
      }
      fl += sl1_scale * ls;
    }
    if (a.grad_bbox_pred != nullptr) {
      encode_grad(g1, Jp, true);
#pragma unroll
      for (int k = 0; k < 7; ++k) g1[k] += gs[k];
      float* gbase = a.grad_bbox_pred + ((b * a.A + an_i) * 7) * plane + hw;
#pragma unroll
      for (int k = 0; k < 7; ++k) gbase[k * plane] = g1[k];
    }
  }
  if (a.partials != nullptr) {
    const float ws = wave_sum(fl);
    if (lane == 0) swave[wave] = ws;
    __syncthreads();
    if (tid == 0) a.partials[blockIdx.x] = (swave[0] + swave[1]) + (swave[2] + swave[3]);
  }
}

template <int LOSS, int FUN>
static void launch_head(bool flag, unsigned grid, hipStream_t s, const HeadArgs& a) {
  if (flag) hipLaunchKernelGGL((head_anchor_kernel<LOSS, FUN, true>), dim3(grid), dim3(HEAD_T), 0, s, a);
  else hipLaunchKernelGGL((head_anchor_kernel<LOSS, FUN, false>), dim3(grid), dim3(HEAD_T), 0, s, a);
}

template <int LOSS>
static void launch_head_fun(int fun, bool flag, unsigned grid, hipStream_t s, const HeadArgs& a) {
  if (fun == GD3D_FUN_LOG1P) launch_head<LOSS, GD3D_FUN_LOG1P>(flag, grid, s, a);
  else launch_head<LOSS, GD3D_FUN_NONE>(flag, grid, s, a);
}

// ------------------------------------------------------------------------------------------------------
// CenterGDHead regression losses of ALL tasks in one launch (gd_centerpoint_head.py:402-441):
//   per task: pred = cat(reg, height, dim, yaw, dir[, vel])[b, :, y, x] gathered at the positives (:416-420, the cat and
//   the gather never materialise: a thread reads its 9/11 values straight from the NCHW head maps), pred_gd =
//   coder.decode(locs, pred)[:7] (:422-423), target = coder.encode(anno) (:409-411: [anno[:7], sin yaw, cos yaw, vel]),
//   loss_gd = GDLoss(pred_gd, target[:7], avg_factor) (:433-434), loss_l1 = L1Loss(pred[7:], target[7:], code_weights,
//   avg_factor) (:426-432).  Two objects of a task may share a cell (the reference's index backward accumulates), so the
//   gradient takes two steps, deterministic and without float atomics: this kernel stages every object's 11 gradient
//   values and counts the objects per cell (integer atomics); center_accum_kernel then writes single-object cells
//   directly and lets the lowest-index object of a shared cell add the cell's contributions in ascending object order.
// blockIdx.y = task; partials[(task * 2 + term) * pstride + block], term 0 = l1, 1 = gd.
constexpr int CENTER_MAX_TASKS = 8;
struct CenterTask {
  const float* maps[6];  // reg(2) height(1) dim(3) yaw(1) dir(2) vel(2); reg / vel nullable
  float* grads[6];       // nullable
  const long long* pos_ind;
  const float* anno;
  int* count;            // (B*H*W) objects per cell, zero-filled by the caller; nullptr = no gradient wanted
  int* keys;             // (n) workspace: cell index of object i, -1 = not live
  float* og;             // (n, 11) workspace: object i's gradient contributions, map order reg|height|dim|yaw|dir|vel
  long long n;
  int B, H, W, anno_cols;
  float gd_scale, l1_scale;
  // device-resident form (nullable): the task's rows are [rows_dev[0], rows_dev[1]) of pos_ind / anno (n is then the
  // capacity the grid was sized for) and the scales are weight / max(*avg_dev, 1): nothing about the task's size or its
  // normaliser has to pass through the host
  const long long* rows_dev;
  const float* avg_dev;
  double gd_weight, l1_weight;
};
struct CenterDyn {
  long long row0, n;
  float gd_scale, l1_scale;
};
GD_DEV CenterDyn center_dyn(const CenterTask& T) {
  CenterDyn d;
  d.row0 = 0;
  d.n = T.n;
  d.gd_scale = T.gd_scale;
  d.l1_scale = T.l1_scale;
  if (T.rows_dev != nullptr) {
    d.row0 = T.rows_dev[0];
    long long m = T.rows_dev[1] - d.row0;
    m = m < 0 ? 0 : m;
    d.n = m < T.n ? m : T.n;
  }
  if (T.avg_dev != nullptr) {     // the host form divides two Python floats and rounds once: the same here
    const double avg = (double)fmaxf(*T.avg_dev, 1.0f);
    d.gd_scale = (float)(T.gd_weight / avg);
    d.l1_scale = (float)(T.l1_weight / avg);
  }
  return d;
}
struct CenterArgs {
  CenterTask t[CENTER_MAX_TASKS];
  int num_tasks, n_l1, norm_bbox;
  float osf, vs0, vs1, pc0, pc1;
  float alpha, ia2, tau, c0, c1, c2;
  float cw[4];
  float* partials;
  long long pstride;
  long long max_n;       // keys rows are max_n long: entries past a task's own n are set to -1 (not live)
};

template <int LOSS, int FUN, bool FLAG>
__global__ __launch_bounds__(HEAD_T) void head_center_kernel(const CenterArgs a) {
  __shared__ float swave[2][HEAD_T / 64];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ti = blockIdx.y;
  const CenterTask& T = a.t[ti];
  const CenterDyn D = center_dyn(T);
  const long long i = (long long)blockIdx.x * HEAD_T + tid;
  if ((long long)blockIdx.x * HEAD_T >= D.n) {  // uniform: this task has fewer positives than the largest one
    if (T.count != nullptr && i < a.max_n) T.keys[i] = -1;   // the rest of its key row: not live (the sorted finish reads whole rows)
    return;
  }
  float fgd = 0.0f, fl1 = 0.0f;
  bool live = i < D.n;
  int key = -1;
  long long b = 0, x = 0, y = 0;
  const long long row = D.row0 + i;
  if (live) {
    b = T.pos_ind[row * 3];
    x = T.pos_ind[row * 3 + 1];
    y = T.pos_ind[row * 3 + 2];
    if (b < 0 || b >= T.B || x < 0 || x >= T.W || y < 0 || y >= T.H) {
      // an index outside the head map (the reference would fault in its gather): no memory is touched for it and both
      // losses of the task come out NaN, so the error is loud without a host-side range check (= a sync per task)
      live = false;
      fgd = fl1 = __builtin_nanf("");
    }
  }
  if (live) {
    const long long plane = (long long)T.H * T.W;
    const long long off = y * T.W + x;
    // channel k of head h at this cell: maps[h][(b * ch_h + k) * plane + off]
    float enc[7];
    enc[0] = T.maps[0] != nullptr ? T.maps[0][(b * 2 + 0) * plane + off] : 0.5f;  // no 'reg' head: 0.5 (:377-378)
    enc[1] = T.maps[0] != nullptr ? T.maps[0][(b * 2 + 1) * plane + off] : 0.5f;
    enc[2] = T.maps[1][b * plane + off];
#pragma unroll
    for (int k = 0; k < 3; ++k) enc[3 + k] = T.maps[2][(b * 3 + k) * plane + off];
    enc[6] = T.maps[3][b * plane + off];
    float tv[7];
#pragma unroll
    for (int k = 0; k < 7; ++k) tv[k] = T.anno[row * T.anno_cols + k];
    // decode (centerpoint_bbox_yaw_coders.py:18-31, correct_yaw=False)
    float pv[7], jac[7];
    pv[0] = (enc[0] + (float)x) * a.osf * a.vs0 + a.pc0;
    pv[1] = (enc[1] + (float)y) * a.osf * a.vs1 + a.pc1;
    pv[2] = enc[2];
#pragma unroll
    for (int k = 3; k < 6; ++k) {
      pv[k] = a.norm_bbox ? expf(enc[k]) : enc[k];
      jac[k] = a.norm_bbox ? pv[k] : 1.0f;
    }
    pv[6] = enc[6];
    jac[0] = a.osf * a.vs0; jac[1] = a.osf * a.vs1; jac[2] = 1.0f; jac[6] = 1.0f;
    const float c[3] = {a.c0, a.c1, a.c2};
    float g1[7], g2[7];
    const float L = pair_loss<LOSS, FUN, FLAG, false>(pv, tv, c, a.alpha, a.ia2, a.tau, D.gd_scale, g1, g2);
    fgd = D.gd_scale * L;
    // L1 on the remaining channels: dir (sin, cos) and velocity
    float sy, cy;
    sincos_f(tv[6], sy, cy);
    float gl1[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (k < a.n_l1) {
        const float p = k < 2 ? T.maps[4][(b * 2 + k) * plane + off] : T.maps[5][(b * 2 + (k - 2)) * plane + off];
        const float t = k == 0 ? sy : (k == 1 ? cy : T.anno[row * T.anno_cols + 7 + (k - 2)]);
        const float d = p - t;
        fl1 += fabsf(d) * a.cw[k];
        gl1[k] = (d > 0.0f ? 1.0f : (d < 0.0f ? -1.0f : 0.0f)) * a.cw[k] * D.l1_scale;  // torch abs'(0) = 0
      }
    }
    fl1 *= D.l1_scale;
    // stage: GD gradient -> reg / height / dim / yaw slots, L1 gradient -> dir / vel slots
    if (T.count != nullptr) {
      float* o = T.og + i * 11;
#pragma unroll
      for (int k = 0; k < 7; ++k) o[k] = g1[k] * jac[k];
#pragma unroll
      for (int k = 0; k < 4; ++k) o[7 + k] = gl1[k];
      key = (int)(b * plane + off);
      atomicAdd(&T.count[key], 1);
    }
  }
  if (T.count != nullptr && i < a.max_n) T.keys[i] = key;   // -1 for rows that are not live and past the task's n
  const float w0 = wave_sum(fl1), w1 = wave_sum(fgd);
  if (lane == 0) {
    swave[0][wave] = w0;
    swave[1][wave] = w1;
  }
  __syncthreads();
  if (tid < 2)
    a.partials[((long long)ti * 2 + tid) * a.pstride + blockIdx.x] =
        (swave[tid][0] + swave[tid][1]) + (swave[tid][2] + swave[tid][3]);
}

// Second step of the CenterGDHead launch pair.  grid = (pstride, tasks), same geometry as head_center_kernel.
//  (1) gradient: thread i owns object i.  count[cell] == 1: its 11 staged values go straight to the maps.  Shared cells:
//      one such lane at a time, the wave scans the task's keys in ascending object order (64 per step, ballot); the lane
//      is the cell's OWNER iff the first match is itself, and then lanes 0..10 add the matching objects' staged values
//      in that order and write the cell.  Sums are in ascending object index whatever the launch geometry.
//  (2) block (0, task): fixed-order fp64 sum of the task's loss partials -> losses[task * 2 + {l1, gd}].
GD_DEV float* center_slot(const CenterTask& T, int k, long long b, long long off, long long plane, int n_l1) {
  // slot k of the staged row -> address in the gradient maps (nullptr: that map wants no gradient)
  if (k < 2) return T.grads[0] != nullptr ? T.grads[0] + (b * 2 + k) * plane + off : nullptr;
  if (k == 2) return T.grads[1] != nullptr ? T.grads[1] + b * plane + off : nullptr;
  if (k < 6) return T.grads[2] != nullptr ? T.grads[2] + (b * 3 + (k - 3)) * plane + off : nullptr;
  if (k == 6) return T.grads[3] != nullptr ? T.grads[3] + b * plane + off : nullptr;
  if (k < 9) return (T.grads[4] != nullptr && n_l1 >= 2) ? T.grads[4] + (b * 2 + (k - 7)) * plane + off : nullptr;
  return (T.grads[5] != nullptr && n_l1 > 2) ? T.grads[5] + (b * 2 + (k - 9)) * plane + off : nullptr;
}

// block (0, task): fixed-order fp64 sum of the task's loss partials -> losses[task * 2 + {l1, gd}]
GD_DEV void center_loss_sums(const CenterArgs& a, int ti, long long n, double* sd, float* __restrict__ losses) {
  const int tid = threadIdx.x;
  const long long nb = (n + HEAD_T - 1) / HEAD_T;
  for (int term = 0; term < 2; ++term) {
    const float* p = a.partials + ((long long)ti * 2 + term) * a.pstride;
    double acc = 0.0;
    for (long long k = tid; k < nb; k += HEAD_T) acc += (double)p[k];
    __syncthreads();
    sd[tid] = acc;
    __syncthreads();
#pragma unroll
    for (int s2 = HEAD_T / 2; s2 > 0; s2 >>= 1) {
      if (tid < s2) sd[tid] += sd[tid + s2];
      __syncthreads();
    }
    if (tid == 0) losses[ti * 2 + term] = (float)sd[0];
  }
}

__global__ __launch_bounds__(HEAD_T) void center_accum_kernel(const CenterArgs a, float* __restrict__ losses) {
  __shared__ double sd[HEAD_T];
  const int tid = threadIdx.x, lane = tid & 63;
  const int ti = blockIdx.y;
  const CenterTask& T = a.t[ti];
  const long long Tn = center_dyn(T).n;
  const long long plane = (long long)T.H * T.W;
  if (T.count != nullptr && (long long)blockIdx.x * HEAD_T < Tn) {  // uniform
    const long long i = (long long)blockIdx.x * HEAD_T + tid;
    const int key = i < Tn ? T.keys[i] : -1;
    const int c = key >= 0 ? T.count[key] : 0;
    if (c == 1) {
      const long long b = key / plane, off = key - b * plane;
#pragma unroll
      for (int k = 0; k < 11; ++k) {
        float* dst = center_slot(T, k, b, off, plane, a.n_l1);
        if (dst != nullptr) *dst = T.og[i * 11 + k];
      }
    }
    unsigned long long dup = __builtin_amdgcn_ballot_w64(c > 1);
    while (dup != 0ull) {                                             // wave-uniform
      const int L = __builtin_ctzll(dup);
      dup &= dup - 1;
      const int kL = __builtin_amdgcn_readlane(key, L);
      const long long iL = i - lane + L;
      float acc = 0.0f;
      bool owner = true, first = true;
      for (long long j0 = 0; j0 < Tn && owner; j0 += 64) {
        const long long j = j0 + lane;
        unsigned long long m = __builtin_amdgcn_ballot_w64(j < Tn && T.keys[j] == kL);
        while (m != 0ull) {
          const long long jj = j0 + __builtin_ctzll(m);
          m &= m - 1;
          if (first) {
            first = false;
            if (jj != iL) {                                           // an earlier object owns this cell
              owner = false;
              break;
            }
          }
          if (lane < 11) acc += T.og[jj * 11 + lane];
        }
      }
      if (owner && lane < 11) {
        const long long b = kL / plane, off = kL - b * plane;
        float* dst = center_slot(T, lane, b, off, plane, a.n_l1);
        if (dst != nullptr) *dst = acc;
      }
    }
  }
  if (blockIdx.x != 0) return;
  center_loss_sums(a, ti, Tn, sd, losses);
}

// The same second step when the caller hands in, per task, the positions of the key row sorted by key (STABLE: objects of
// one cell stay in ascending index; order is (tasks, max_n) int64 — torch.sort(keys, dim=1, stable=True) on the rows that
// head_center_kernel left).  Thread s owns sorted position s: the first entry of a run of equal keys adds the run's staged
// rows in that order and writes the cell.  O(n) whatever the number of objects per cell: the scan form above costs
// O(shared-cell objects x n / 64) wave steps, fine for a detection batch (n = 4000, a few shared cells) and quadratic when
// tens of thousands of objects fall into few cells.
__global__ __launch_bounds__(HEAD_T) void center_accum_sorted_kernel(const CenterArgs a, float* __restrict__ losses,
                                                                     const long long* __restrict__ order) {
  __shared__ double sd[HEAD_T];
  const int tid = threadIdx.x;
  const int ti = blockIdx.y;
  const CenterTask& T = a.t[ti];
  const long long plane = (long long)T.H * T.W;
  const long long s = (long long)blockIdx.x * HEAD_T + tid;
  if (T.count != nullptr && s < a.max_n) {
    const long long* ord = order + (long long)ti * a.max_n;
    const long long i = ord[s];
    const int key = (i >= 0 && i < a.max_n) ? T.keys[i] : -1;
    const long long ip = s > 0 ? ord[s - 1] : -1;
    const int prev = (ip >= 0 && ip < a.max_n) ? T.keys[ip] : -2;   // (a malformed order must not read outside the row)
    if (key >= 0 && key != prev) {  // run start: this thread owns the cell
      float acc[11];
#pragma unroll
      for (int k = 0; k < 11; ++k) acc[k] = T.og[i * 11 + k];
      for (long long e = s + 1; e < a.max_n; ++e) {
        const long long j = ord[e];
        if (j < 0 || j >= a.max_n || T.keys[j] != key) break;
#pragma unroll
        for (int k = 0; k < 11; ++k) acc[k] += T.og[j * 11 + k];
      }
      const long long b = key / plane, off = key - b * plane;
#pragma unroll
      for (int k = 0; k < 11; ++k) {
        float* dst = center_slot(T, k, b, off, plane, a.n_l1);
        if (dst != nullptr) *dst = acc[k];
      }
    }
  }
  if (blockIdx.x != 0) return;
  center_loss_sums(a, ti, center_dyn(T).n, sd, losses);
}

// backward of the same call when the upstream gradient is not all ones: grads of task t are scaled by
// gout[t*2 + 1] (reg / height / dim / yaw: the GD term) or gout[t*2] (dir / vel: the L1 term); a (task, map) slice whose
// factor is exactly 1 exits after one scalar load.  blockIdx.y = task * 6 + map.
__global__ __launch_bounds__(256) void center_scale_kernel(const CenterArgs a, const float* __restrict__ gout) {
  const int ti = blockIdx.y / 6, m = blockIdx.y - ti * 6;
  const CenterTask& T = a.t[ti];
  float* gmap = T.grads[m];
  if (gmap == nullptr) return;
  const float gs = gout[ti * 2 + (m < 4 ? 1 : 0)];
  if (gs == 1.0f) return;
  const int ch = (m == 0 || m >= 4) ? 2 : (m == 2 ? 3 : 1);
  const long long nflt = (long long)T.B * ch * T.H * T.W;
  for (long long k = (long long)blockIdx.x * 256 + threadIdx.x; k < nflt; k += (long long)gridDim.x * 256) gmap[k] *= gs;
}


// Clears (or fills) small or large device buffers from a KERNEL.  Not hipMemsetAsync: inside a captured hipGraph a memset node was
// found not to be reliably ordered against the kernels around it on this ROCm (profiles/r04_nms_queue_ab.txt) —
// rule of this library: no memset nodes in paths a caller may capture.
__global__ __launch_bounds__(256) void fill_words_kernel(unsigned* __restrict__ p, long long nwords, unsigned value) {
  const long long stride = (long long)gridDim.x * 256;
  long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if ((((uintptr_t)p) & 15) == 0) {
    uint4* p4 = reinterpret_cast<uint4*>(p);
    const long long nv = nwords >> 2;
    const uint4 v4 = make_uint4(value, value, value, value);
    for (long long k = i; k < nv; k += stride) p4[k] = v4;
    for (long long k = (nv << 2) + i; k < nwords; k += stride) p[k] = value;
    return;
  }
  for (; i < nwords; i += stride) p[i] = value;
}
static int fill_words(void* p, size_t bytes, unsigned value, hipStream_t s) {   // bytes: a multiple of 4
  const long long nwords = (long long)(bytes / 4);
  if (nwords == 0) return 0;
  long long blocks = (nwords / 4 + 255) / 256;
  blocks = blocks < 1 ? 1 : (blocks > 4096 ? 4096 : blocks);
  hipLaunchKernelGGL(fill_words_kernel, dim3((unsigned)blocks), dim3(256), 0, s, (unsigned*)p, nwords, value);
  return (int)hipGetLastError();
}

// second stage: fixed-order fp64 sum of the per-block partials -> one fp32.  One workgroup; the kernel is pure latency
// (39 K floats at 10 M pairs): every thread issues ALL its 16-byte loads of a 48 K-partial round before the first add
// (one memory round trip at 10 M pairs), then one barrier: thread t of wave 0 adds 16 consecutive per-thread sums from
// LDS and the wave finishes with 6 shuffle steps.  Deterministic (fixed order), independent of the launch geometry.
__global__ __launch_bounds__(1024) void reduce_partials_kernel(const float* __restrict__ partials, long long nb,
                                                               float* __restrict__ out) {
  __shared__ double sd[1024];
  constexpr int RU = 12;
  const int tid = threadIdx.x;
  const long long nv = nb >> 2;  // whole float4s (the workspace is 16-byte aligned)
  const float4* p4 = reinterpret_cast<const float4*>(partials);
  double acc = 0.0;
  for (long long i0 = 0; i0 < nv; i0 += 1024 * RU) {
    float4 v[RU];
#pragma unroll
    for (int u = 0; u < RU; ++u) {
      const long long i = i0 + (long long)u * 1024 + tid;
      v[u] = i < nv ? p4[i] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int u = 0; u < RU; ++u) acc += ((double)v[u].x + (double)v[u].y) + ((double)v[u].z + (double)v[u].w);
  }
  const long long tail = (nv << 2) + tid;
  if (tail < nb) acc += (double)partials[tail];
  sd[tid] = acc;
  __syncthreads();
  if (tid < 64) {
    double s2 = 0.0;
#pragma unroll
    for (int k = 0; k < 16; ++k) s2 += sd[tid * 16 + k];
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) s2 += __shfl_down(s2, off, 64);
    if (tid == 0) *out = (float)s2;
  }
}

// grad[i,:] *= g  (scalar g: the whole grid exits after one scalar load when g == 1)
__global__ __launch_bounds__(1024) void scale_rows_kernel(float* __restrict__ grad, const float* __restrict__ g,
                                                         int per_row, long long nflt) {
  float gs = 1.0f;
  if (!per_row) {
    gs = g[0];
    if (gs == 1.0f) return;  // uniform across the grid
  }
  const long long stride = (long long)gridDim.x * blockDim.x;
  const long long i0 = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (!per_row && (((uintptr_t)grad & 15) == 0)) {
    v4f* g4 = reinterpret_cast<v4f*>(grad);
    const long long nv = nflt >> 2;
    for (long long i = i0; i < nv; i += stride) g4[i] = g4[i] * gs;
    for (long long i = (nv << 2) + i0; i < nflt; i += stride) grad[i] *= gs;
    return;
  }
  for (long long i = i0; i < nflt; i += stride) grad[i] *= per_row ? g[i / 7] : gs;
}

// The same second stage for a weighted call that must honour GDLoss.forward's early-out (ref :290-292) without a host
// sync: besides the loss partials the fused kernel left sum(pred * weight) and "some weight > 0" per tile.  One launch
// adds all three in a fixed order and SELECTS on the device:
//   *out = any weight > 0 ? scale * sum_i w_i L_i : sum(pred * weight)          *any_pos = that predicate (for backward)
__global__ __launch_bounds__(1024) void reduce_select_kernel(const float* __restrict__ partials, long long nb, long long nbp,
                                                             float* __restrict__ out, int* __restrict__ any_pos) {
  __shared__ double sd[3][1024];
  const int tid = threadIdx.x;
  double acc[3] = {0.0, 0.0, 0.0};
  for (long long i = tid; i < nb; i += 1024) {
#pragma unroll
    for (int k = 0; k < 3; ++k) acc[k] += (double)partials[k * nbp + i];
  }
#pragma unroll
  for (int k = 0; k < 3; ++k) sd[k][tid] = acc[k];
  __syncthreads();
  if (tid < 64) {
    double s3[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      double s2 = 0.0;
      for (int j = 0; j < 16; ++j) s2 += sd[k][tid * 16 + j];
#pragma unroll
      for (int off = 32; off >= 1; off >>= 1) s2 += __shfl_down(s2, off, 64);
      s3[k] = s2;
    }
    if (tid == 0) {
      const bool any = s3[2] > 0.0;
      *out = (float)(any ? s3[0] : s3[1]);
      if (any_pos != nullptr) *any_pos = any ? 1 : 0;
    }
  }
}

// Autograd backward of a reduced call (replaces scale_rows_kernel there):
//   any weight > 0 (or no selection):  grad *= g            (the whole grid leaves after two scalar loads when g == 1)
//   otherwise (ref :290-292 took `(pred * weight).sum()`):  grad_pred = g * d(sum(dec(pred) * weight7)) / d pred,
//                                                           grad_target = 0
// One thread per row on the rare path; it is not a bandwidth path.
struct FinishArgs {
  float* gp;             // (n,7) nullable
  float* gt;             // (n,7) nullable
  const float* g;        // device scalar: upstream gradient
  const int* any_pos;    // device flag written by reduce_select_kernel; nullptr = no selection
  const float* w7;       // (n,7) weights      (selection only)
  const float* pred;     // (n,7) encoded rows (selection with a prologue only)
  const float* aux;
  long long n;
  int pro, norm_bbox;
  float osf, vs0, vs1;
};

constexpr int FINISH_T = 256;   // one wave per SIMD and CU on a 256-workgroup grid: the common case is the g == 1 early exit, whose cost
constexpr int FINISH_U = 8;     // is the dispatch of the grid's waves (4096 waves: 4.4 us, 1024 waves: see profiles/r03); a real
                                // scaling pass keeps FINISH_U 16-byte loads in flight per lane instead (32 KiB per CU)
__global__ __launch_bounds__(FINISH_T) void grad_finish_kernel(const FinishArgs a) {
  const float gs = a.g[0];
  const bool normal = a.any_pos == nullptr || a.any_pos[0] != 0;
  if (normal && gs == 1.0f) return;  // uniform across the grid
  const long long stride = (long long)gridDim.x * blockDim.x;
  const long long i0 = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long nflt = a.n * 7;
  if (normal) {
    for (int which = 0; which < 2; ++which) {
      float* grad = which == 0 ? a.gp : a.gt;
      if (grad == nullptr) continue;
      if (((uintptr_t)grad & 15) == 0) {
        v4f* g4 = reinterpret_cast<v4f*>(grad);
        const long long nv = nflt >> 2;
        for (long long i = i0; i < nv; i += FINISH_U * stride) {
          v4f v[FINISH_U];
#pragma unroll
          for (int u = 0; u < FINISH_U; ++u)
            if (i + u * stride < nv) v[u] = g4[i + u * stride];
#pragma unroll
          for (int u = 0; u < FINISH_U; ++u)
            if (i + u * stride < nv) g4[i + u * stride] = v[u] * gs;
        }
        for (long long i = (nv << 2) + i0; i < nflt; i += stride) grad[i] *= gs;
      } else {
        for (long long i = i0; i < nflt; i += stride) grad[i] *= gs;
      }
    }
    return;
  }
  for (long long i = i0; i < a.n; i += stride) {
    float w[7];
#pragma unroll
    for (int k = 0; k < 7; ++k) w[k] = a.w7[i * 7 + k];
    if (a.pro == GD3D_PRO_ANCHOR_DELTA) {
      float enc[7], an[7], dec[7];
      DecodeJac J;
#pragma unroll
      for (int k = 0; k < 7; ++k) {
        enc[k] = a.pred[i * 7 + k];
        an[k] = a.aux[i * 7 + k];
      }
      decode_anchor(enc, an, dec, J);
      encode_grad(w, J, true);
    } else if (a.pro == GD3D_PRO_CENTER) {
      w[0] *= a.osf * a.vs0;
      w[1] *= a.osf * a.vs1;
      if (a.norm_bbox) {
#pragma unroll
        for (int k = 3; k < 6; ++k) w[k] *= expf(a.pred[i * 7 + k]);
      }
    }
#pragma unroll
    for (int k = 0; k < 7; ++k) {
      if (a.gp != nullptr) a.gp[i * 7 + k] = gs * w[k];
      if (a.gt != nullptr) a.gt[i * 7 + k] = 0.0f;
    }
  }
}

// HBM ceiling probe for the access mix of the fused kernel: c = a + b over float4 vectors, nontemporal loads and
// stores, one vector per thread, full grid of 64-THREAD workgroups — the fastest and the most repeatable of the shapes
// tools/hbm_probe2.hip measures (126.1-129 us over five processes for 3 x 280 MB; 256-thread workgroups: 126.8-133).
// bench.py runs it on the fused kernel's OWN three buffers right after the timed region, so that
// `roofline.copy_ceiling_GBps` is the ceiling of that box and of that buffer placement.
constexpr int PROBE_T = 64;
__global__ __launch_bounds__(PROBE_T) void probe_add_kernel(const v4f* __restrict__ x, const v4f* __restrict__ y,
                                                            v4f* __restrict__ z, long long nv) {
  const long long i = (long long)blockIdx.x * PROBE_T + threadIdx.x;
  if (i < nv) {
    const v4f p = __builtin_nontemporal_load(x + i);
    const v4f q = __builtin_nontemporal_load(y + i);
    __builtin_nontemporal_store(p + q, z + i);
  }
}

struct Geometry {
  unsigned tgrid;  // one workgroup per 256-pair tile
  // profiling only (gd3d_loss_fused_timed): events bound to THIS dispatch, see launch_one
  hipEvent_t ev_start = nullptr, ev_stop = nullptr;
};

template <int LOSS, int FUN, bool FLAG, bool GT>
static void launch_one(const Geometry& g, hipStream_t s, const LossArgs& a) {
  size_t lds = (size_t)(2 * TILE_F + 32 + (a.w7 != nullptr ? TILE_F : 0)) * sizeof(float);
  constexpr int min_lds = LOSS == GD3D_GWD3D ? GD_MIN_LDS_GWD : (LOSS == GD3D_KLD3D ? GD_MIN_LDS_KLD :
                          (LOSS == GD3D_BD3D ? GD_MIN_LDS_BD : GD_MIN_LDS));
  if (lds < (size_t)min_lds) lds = (size_t)min_lds;   // occupancy cap (see GD_MIN_LDS above)
  // with (N,7) weights a workgroup keeps three tiles in flight: 5 per CU for every loss (10 M pairs, fixed placement,
  // 7/6/5/4/3 per CU: kld3d 185.5/185.4/179.8/180.9/202.7 us, bd3d 185.8/184.7/178.0/177.9/204.9, gwd3d 185.0/185.2/182.3/182.7/201.4)
  if (a.w7 != nullptr && lds < (size_t)GD_MIN_LDS_W7) lds = (size_t)GD_MIN_LDS_W7;
#ifdef GD_LDS_ENV   // experiment builds only (tools/lds_fixed_placement.py): the cap is re-read from the environment per launch
  if (const char* e = getenv("GD3D_MIN_LDS")) {
    const size_t base = (size_t)(2 * TILE_F + 32 + (a.w7 != nullptr ? TILE_F : 0)) * sizeof(float);
    const size_t want = (size_t)atoi(e);
    lds = want > base ? want : base;
  }
#endif
  // The option-free instantiation, for calls that use none of the options — built and used for gwd3d only, without a
  // target gradient.  Same buffers, 4 sets, us per 10 M pairs against the general instantiation (tools/variant_probe.py,
  // profiles/r02_plain_and_dma_issue_ab.txt): gwd3d -2.5 -0.7 -1.8 -0.5 (issue units per wave 424 -> 340), but kld3d
  // +1.0 +1.5 +1.2 +1.0 and bd3d +0.9 +0.7 +1.1 +1.9 at their cap of 6 workgroups per CU although their streams shrink
  // as well (483 -> 415, 519 -> 433): with less to issue they keep more bytes in flight, which is what the cap exists to
  // limit, and at 5 per CU they lose more than they gain on a fast box (133.0 vs 130.3 us).  gwd3d is the slowest of
  // the three on fast boxes, i.e. the kernel roofline.frac is computed from.
  constexpr bool HAS_PLAIN = LOSS == GD3D_GWD3D || (GD_PLAIN_ALL && (LOSS == GD3D_KLD3D || LOSS == GD3D_BD3D));
  const bool plain = HAS_PLAIN && !GT && a.w == nullptr && a.w7 == nullptr && !a.wsel && a.pro == GD3D_PRO_NONE &&
                     a.loss == nullptr && a.fin == nullptr;
  if (g.ev_start != nullptr || g.ev_stop != nullptr) {
    // hipExtLaunchKernel binds the two events to the begin / end timestamps of this dispatch packet itself: no marker
    // packets enter the stream, and hipEventElapsedTime(start, stop) is the kernel's execution time as rocprofv3 reports
    // it (events recorded AROUND a launch add the ~3 us of two barrier packets to every bracket).
    if (plain)
      hipExtLaunchKernelGGL((fused_kernel<LOSS, FUN, FLAG, false, HAS_PLAIN>), dim3(g.tgrid), dim3(TILE),
                            (std::uint32_t)lds, s, g.ev_start, g.ev_stop, 0u, a);
    else
      hipExtLaunchKernelGGL((fused_kernel<LOSS, FUN, FLAG, GT>), dim3(g.tgrid), dim3(TILE), (std::uint32_t)lds, s,
                            g.ev_start, g.ev_stop, 0u, a);
    return;
  }
  if (plain)
    hipLaunchKernelGGL((fused_kernel<LOSS, FUN, FLAG, false, HAS_PLAIN>), dim3(g.tgrid), dim3(TILE), lds, s, a);
  else
    hipLaunchKernelGGL((fused_kernel<LOSS, FUN, FLAG, GT>), dim3(g.tgrid), dim3(TILE), lds, s, a);
}

template <int LOSS, int FUN>
static hipError_t launch_flag_gt(bool flag, bool gt, const Geometry& grid, hipStream_t s, const LossArgs& a) {
  switch ((flag ? 2 : 0) | (gt ? 1 : 0)) {
    case 0: launch_one<LOSS, FUN, false, false>(grid, s, a); break;
    case 1: launch_one<LOSS, FUN, false, true>(grid, s, a); break;
    case 2: launch_one<LOSS, FUN, true, false>(grid, s, a); break;
    default: launch_one<LOSS, FUN, true, true>(grid, s, a); break;
  }
  return hipGetLastError();
}

template <int LOSS>
static hipError_t launch_fun(int fun, bool flag, bool gt, const Geometry& grid, hipStream_t s, const LossArgs& a) {
  if (fun == GD3D_FUN_LOG1P) return launch_flag_gt<LOSS, GD3D_FUN_LOG1P>(flag, gt, grid, s, a);
  return launch_flag_gt<LOSS, GD3D_FUN_NONE>(flag, gt, grid, s, a);
}

static hipError_t launch_kfiou(int fun, bool gt, const Geometry& grid, hipStream_t s, const LossArgs& a) {
  // `sqrt` is accepted and ignored by kfiou3d_loss (ref :228), so FLAG is pinned to false
  switch (fun) {
    case GD3D_FUN_EXPM1: return launch_flag_gt<GD3D_KFIOU3D, GD3D_FUN_EXPM1>(false, gt, grid, s, a);
    case GD3D_FUN_NLOG: return launch_flag_gt<GD3D_KFIOU3D, GD3D_FUN_NLOG>(false, gt, grid, s, a);
    default: return launch_flag_gt<GD3D_KFIOU3D, GD3D_FUN_NONE>(false, gt, grid, s, a);
  }
}

}  // namespace gd3d

using namespace gd3d;

extern "C" {

// workspace layout (all offsets multiples of 16 bytes), nb = tiles of 256 rows, nbp = nb rounded up to 4, ng = 64-tile groups:
//   float  partials[3][nbp]   loss | sum(pred * weight7) | any weight > 0   (rows 1, 2 only for gd3d_loss_fused_select)
static inline int64_t ws_nbp(int64_t n) { return (((n + 255) / 256) + 3) & ~(int64_t)3; }

size_t gd3d_loss_workspace_bytes(int64_t n) {
  if (n <= 0) return 64;
  return (size_t)(12 * ws_nbp(n) + 16);
}

int gd3d_loss_fused(const gd3d_params* p, const float* pred, const float* target, const float* row_weight,
                    int64_t n, float scale, float* loss, float* loss_sum, float* grad_pred, float* grad_target,
                    void* workspace, void* stream) {
  return gd3d_loss_fused_w7(p, pred, target, row_weight, nullptr, n, scale, loss, loss_sum, grad_pred, grad_target,
                            workspace, stream);
}

int gd3d_loss_fused_w7(const gd3d_params* p, const float* pred, const float* target, const float* row_weight,
                       const float* weight7, int64_t n, float scale, float* loss, float* loss_sum, float* grad_pred,
                       float* grad_target, void* workspace, void* stream) {
  return gd3d_loss_fused_decoded(p, nullptr, pred, target, row_weight, weight7, n, scale, loss, loss_sum, grad_pred,
                                 grad_target, workspace, stream);
}

static int loss_launch(const gd3d_params* p, const gd3d_prologue* pro, const float* pred, const float* target,
                       const float* row_weight, const float* weight7, int64_t n, float scale, float* loss,
                       float* loss_sum, float* grad_pred, float* grad_target, void* workspace, void* stream,
                       void* start_event, void* stop_event, int32_t* any_positive, bool select, int32_t* ticket = nullptr);

int gd3d_loss_fused_decoded(const gd3d_params* p, const gd3d_prologue* pro, const float* pred, const float* target,
                            const float* row_weight, const float* weight7, int64_t n, float scale, float* loss,
                            float* loss_sum, float* grad_pred, float* grad_target, void* workspace, void* stream) {
  return loss_launch(p, pro, pred, target, row_weight, weight7, n, scale, loss, loss_sum, grad_pred, grad_target,
                     workspace, stream, nullptr, nullptr, nullptr, false);
}

int gd3d_loss_fused_timed(const gd3d_params* p, const gd3d_prologue* pro, const float* pred, const float* target,
                          const float* row_weight, const float* weight7, int64_t n, float scale, float* loss,
                          float* loss_sum, float* grad_pred, float* grad_target, void* workspace, void* stream,
                          void* start_event, void* stop_event) {
  return loss_launch(p, pro, pred, target, row_weight, weight7, n, scale, loss, loss_sum, grad_pred, grad_target,
                     workspace, stream, start_event, stop_event, nullptr, false);
}

int gd3d_loss_fused_select(const gd3d_params* p, const gd3d_prologue* pro, const float* pred, const float* target,
                           const float* weight7, int64_t n, float scale, float* loss_sum, int32_t* any_positive,
                           float* grad_pred, float* grad_target, void* workspace, void* stream, void* start_event,
                           void* stop_event) {
  if ((n > 0 && weight7 == nullptr) || loss_sum == nullptr || any_positive == nullptr || workspace == nullptr)
    return GD3D_E_BADARG;   // (an empty weight array has no address)
  return loss_launch(p, pro, pred, target, nullptr, weight7, n, scale, nullptr, loss_sum, grad_pred, grad_target,
                     workspace, stream, start_event, stop_event, any_positive, true);
}

int gd3d_loss_fused_one_launch(const gd3d_params* p, const gd3d_prologue* pro, const float* pred, const float* target,
                               const float* row_weight, const float* weight7, int64_t n, float scale, float* loss_sum,
                               int32_t* any_positive, float* grad_pred, float* grad_target, void* workspace,
                               int32_t* ticket, void* stream) {
  if (loss_sum == nullptr || workspace == nullptr || ticket == nullptr || n > gd3d_one_launch_max_n()) return GD3D_E_BADARG;
  if (any_positive != nullptr && (row_weight != nullptr || (n > 0 && weight7 == nullptr))) return GD3D_E_BADARG;
  return loss_launch(p, pro, pred, target, row_weight, weight7, n, scale, nullptr, loss_sum, grad_pred, grad_target,
                     workspace, stream, nullptr, nullptr, any_positive, any_positive != nullptr, ticket);
}

int64_t gd3d_one_launch_max_n(void) { return 64 * (int64_t)TILE; }

static int loss_launch(const gd3d_params* p, const gd3d_prologue* pro, const float* pred, const float* target,
                       const float* row_weight, const float* weight7, int64_t n, float scale, float* loss,
                       float* loss_sum, float* grad_pred, float* grad_target, void* workspace, void* stream,
                       void* start_event, void* stop_event, int32_t* any_positive, bool select, int32_t* ticket) {
  if (row_weight != nullptr && weight7 != nullptr) return GD3D_E_BADARG;
  if (pro != nullptr && pro->kind != GD3D_PRO_NONE) {
    if (pro->kind != GD3D_PRO_ANCHOR_DELTA && pro->kind != GD3D_PRO_CENTER) return GD3D_E_BADARG;
    if (n > 0 && pro->aux == nullptr) return GD3D_E_BADARG;
  }
  if (p == nullptr || n < 0) return GD3D_E_BADARG;
  if (n > 0 && (pred == nullptr || target == nullptr)) return GD3D_E_BADARG;
  if (p->loss_type < 0 || p->loss_type >= GD3D_NUM_LOSS_TYPES) return GD3D_E_BADARG;
  // fun domain per loss type: GDLoss.__init__ asserts (gaussian_distance_loss.py:267-270)
  if (p->loss_type == GD3D_KFIOU3D) {
    if (p->fun != GD3D_FUN_NONE && p->fun != GD3D_FUN_EXPM1 && p->fun != GD3D_FUN_NLOG) return GD3D_E_BADARG;
  } else if (p->fun != GD3D_FUN_NONE && p->fun != GD3D_FUN_LOG1P) {
    return GD3D_E_BADARG;
  }
  if (loss_sum != nullptr && workspace == nullptr) return GD3D_E_BADARG;
  if (workspace != nullptr && ((uintptr_t)workspace & 15) != 0) return GD3D_E_BADARG;
  hipStream_t s = (hipStream_t)stream;
  const int64_t nb = (n + TILE - 1) / TILE;
  if (nb > 0x7fffffffLL) return GD3D_E_TOOLARGE;
  if (n == 0) {
    // no rows: sum 0; torch.any() of an empty weight is False, so a selecting call takes the early-out, whose value is 0 too
    if (any_positive != nullptr) {
      const int e0 = fill_words(any_positive, sizeof(int32_t), 0u, s);
      if (e0 != 0) return e0;
    }
    if (loss_sum != nullptr) return fill_words(loss_sum, sizeof(float), 0u, s);
    return 0;
  }
  if (loss_sum == nullptr && loss == nullptr && grad_pred == nullptr && grad_target == nullptr && workspace == nullptr)
    return 0;  // nothing requested
  LossArgs a;
  a.wsel = select ? 1 : 0;
  a.nbp = ws_nbp(n);
  a.fin = (int*)ticket;
  a.fin_out = loss_sum;
  a.fin_any = (int*)any_positive;

  a.pred = pred;
  a.target = target;
  a.w = row_weight;
  a.w7 = weight7;
  a.loss = loss;
  a.gp = grad_pred;
  a.gt = grad_target;
  a.partials = (float*)workspace;  // per-workgroup partial sums are produced whenever a workspace is given
  a.n = n;
  a.scale = scale;
  a.alpha = p->alpha;
  a.ia2 = gd3d_inv_alpha2(p->alpha);
  a.tau = p->tau;
  a.c0 = p->center_offset[0];
  a.c1 = p->center_offset[1];
  a.c2 = p->center_offset[2];
  a.pro = (pro != nullptr) ? pro->kind : GD3D_PRO_NONE;
  a.norm_bbox = (pro != nullptr) ? pro->norm_bbox : 0;
  a.aux = (pro != nullptr) ? pro->aux : nullptr;
  a.osf = (pro != nullptr) ? pro->out_size_factor : 1.0f;
  a.vs0 = (pro != nullptr) ? pro->voxel_size[0] : 1.0f;
  a.vs1 = (pro != nullptr) ? pro->voxel_size[1] : 1.0f;
  a.pc0 = (pro != nullptr) ? pro->pc_range[0] : 0.0f;
  a.pc1 = (pro != nullptr) ? pro->pc_range[1] : 0.0f;
  const uintptr_t bits = (uintptr_t)pred | (uintptr_t)target | (uintptr_t)grad_pred | (uintptr_t)grad_target |
                         (uintptr_t)weight7;
  a.vec_ok = (bits & 15) == 0;
  const bool gt = grad_target != nullptr;
  const bool flag = p->flag != 0;
  Geometry grid;
  grid.tgrid = (unsigned)nb;
  grid.ev_start = (hipEvent_t)start_event;
  grid.ev_stop = (hipEvent_t)stop_event;
  hipError_t e;
  switch (p->loss_type) {
    case GD3D_GWD3D: e = launch_fun<GD3D_GWD3D>(p->fun, flag, gt, grid, s, a); break;
    case GD3D_KLD3D: e = launch_fun<GD3D_KLD3D>(p->fun, flag, gt, grid, s, a); break;
    case GD3D_BD3D: e = launch_fun<GD3D_BD3D>(p->fun, flag, gt, grid, s, a); break;
    case GD3D_JD3D: e = launch_fun<GD3D_JD3D>(p->fun, flag, gt, grid, s, a); break;
    case GD3D_KLD3D_SYMMAX: e = launch_fun<GD3D_KLD3D_SYMMAX>(p->fun, flag, gt, grid, s, a); break;
    case GD3D_KLD3D_SYMMIN: e = launch_fun<GD3D_KLD3D_SYMMIN>(p->fun, flag, gt, grid, s, a); break;
    default: e = launch_kfiou(p->fun, gt, grid, s, a); break;
  }
  if (e != hipSuccess) return (int)e;
  if (ticket != nullptr) return 0;   // the last workgroup of the fused kernel wrote the result
  if (select) {
    hipLaunchKernelGGL(reduce_select_kernel, dim3(1), dim3(1024), 0, s, (const float*)workspace, (long long)nb,
                       (long long)ws_nbp(n), loss_sum, (int*)any_positive);
    return (int)hipGetLastError();
  }
  if (loss_sum != nullptr) return gd3d_loss_reduce(workspace, n, loss_sum, stream);
  return 0;
}

int gd3d_grad_finish(float* grad_pred, float* grad_target, const float* g, int64_t n, const int32_t* any_positive,
                     const float* weight7, const float* pred, const gd3d_prologue* pro, void* stream) {
  if (n < 0 || g == nullptr) return GD3D_E_BADARG;
  if (n == 0 || (grad_pred == nullptr && grad_target == nullptr)) return 0;
  FinishArgs a;
  a.gp = grad_pred;
  a.gt = grad_target;
  a.g = g;
  a.any_pos = (const int*)any_positive;
  a.w7 = weight7;
  a.pred = pred;
  a.n = n;
  a.pro = (pro != nullptr) ? pro->kind : GD3D_PRO_NONE;
  a.norm_bbox = (pro != nullptr) ? pro->norm_bbox : 0;
  a.aux = (pro != nullptr) ? pro->aux : nullptr;
  a.osf = (pro != nullptr) ? pro->out_size_factor : 1.0f;
  a.vs0 = (pro != nullptr) ? pro->voxel_size[0] : 1.0f;
  a.vs1 = (pro != nullptr) ? pro->voxel_size[1] : 1.0f;
  if (any_positive != nullptr) {
    if (weight7 == nullptr) return GD3D_E_BADARG;
    if (a.pro != GD3D_PRO_NONE && pred == nullptr) return GD3D_E_BADARG;
    if (a.pro == GD3D_PRO_ANCHOR_DELTA && a.aux == nullptr) return GD3D_E_BADARG;
  }
  // few waves: the common case is the g == 1 early exit, whose cost is the dispatch itself
  long long blocks = ((long long)n * 7 + 4 * FINISH_T - 1) / (4 * FINISH_T);
  if (blocks > 256) blocks = 256;
  hipLaunchKernelGGL(grad_finish_kernel, dim3((unsigned)blocks), dim3(FINISH_T), 0, (hipStream_t)stream, a);
  return (int)hipGetLastError();
}

int gd3d_probe_stream(const float* x, const float* y, float* z, int64_t n_floats, void* stream, void* start_event,
                      void* stop_event) {
  if (n_floats < 0 || (n_floats & 3) != 0) return GD3D_E_BADARG;
  if (n_floats == 0) return 0;
  if (x == nullptr || y == nullptr || z == nullptr) return GD3D_E_BADARG;
  if ((((uintptr_t)x | (uintptr_t)y | (uintptr_t)z) & 15) != 0) return GD3D_E_BADARG;
  const long long nv = n_floats >> 2;
  const long long nb = (nv + PROBE_T - 1) / PROBE_T;
  if (nb > 0x7fffffffLL) return GD3D_E_TOOLARGE;
  hipExtLaunchKernelGGL(probe_add_kernel, dim3((unsigned)nb), dim3(PROBE_T), 0u, (hipStream_t)stream, (hipEvent_t)start_event,
                        (hipEvent_t)stop_event, 0u, (const v4f*)x, (const v4f*)y, (v4f*)z, nv);
  return (int)hipGetLastError();
}

int gd3d_loss_reduce(const void* workspace, int64_t n, float* loss_sum, void* stream) {
  if (n < 0 || loss_sum == nullptr) return GD3D_E_BADARG;
  hipStream_t s = (hipStream_t)stream;
  if (n == 0) return fill_words(loss_sum, sizeof(float), 0u, s);
  if (workspace == nullptr) return GD3D_E_BADARG;
  const long long nparts = (n + TILE - 1) / TILE;
  hipLaunchKernelGGL(reduce_partials_kernel, dim3(1), dim3(1024), 0, s, (const float*)workspace, nparts, loss_sum);
  return (int)hipGetLastError();
}

static int anchor_head_impl(const gd3d_params* p, const gd3d_smooth_l1* sl1, const float* bbox_pred, int32_t B, int32_t A,
                            int32_t H, int32_t W, const float* bbox_targets, const float* bbox_weights,
                            const float* decode_weight,
                            const float* anchors, const int64_t* pos_inds, const int64_t* labels, int32_t num_classes,
                            int64_t P, float scale, float* loss_sum, float* grad_bbox_pred, void* workspace,
                            void* stream, const float* avg_dev = nullptr, double w_gd = 0.0, double w_sl1 = 0.0) {
  if (p == nullptr || P < 0 || B <= 0 || A <= 0 || H <= 0 || W <= 0) return GD3D_E_BADARG;
  if (p->loss_type < 0 || p->loss_type >= GD3D_NUM_LOSS_TYPES) return GD3D_E_BADARG;
  if (p->loss_type == GD3D_KFIOU3D) {
    if (p->fun != GD3D_FUN_NONE && p->fun != GD3D_FUN_EXPM1 && p->fun != GD3D_FUN_NLOG) return GD3D_E_BADARG;
  } else if (p->fun != GD3D_FUN_NONE && p->fun != GD3D_FUN_LOG1P) {
    return GD3D_E_BADARG;
  }
  hipStream_t s = (hipStream_t)stream;
  if (P == 0) {
    if (loss_sum != nullptr) return fill_words(loss_sum, sizeof(float), 0u, s);
    return 0;
  }
  if (bbox_pred == nullptr || bbox_targets == nullptr || anchors == nullptr) return GD3D_E_BADARG;
  if (pos_inds == nullptr && labels == nullptr) return GD3D_E_BADARG;
  if (loss_sum != nullptr && workspace == nullptr) return GD3D_E_BADARG;
  HeadArgs a;
  a.bbox_pred = bbox_pred;
  a.bbox_targets = bbox_targets;
  a.bbox_weights = bbox_weights;
  a.anchors = anchors;
  a.pos_inds = (const long long*)pos_inds;
  a.labels = (const long long*)labels;
  a.num_classes = num_classes;
  a.grad_bbox_pred = grad_bbox_pred;
  a.partials = (float*)workspace;
  a.P = P;
  a.A = A;
  a.H = H;
  a.W = W;
  for (int k = 0; k < 7; ++k) a.dw[k] = decode_weight != nullptr ? decode_weight[k] : 1.0f;  // HOST array of 7
  a.dw_on = bbox_weights != nullptr && (decode_weight != nullptr || sl1 == nullptr);
  a.sl1 = 0;
  a.sl1_cw = a.sin_diff = 0;
  a.beta = a.sl1_scale = 0.0f;
  for (int k = 0; k < 7; ++k) a.cw[k] = 1.0f;
  if (sl1 != nullptr) {
    if (!(sl1->beta >= 0.0f)) return GD3D_E_BADARG;
    if (sl1->has_code_weight && bbox_weights == nullptr) return GD3D_E_BADARG;
    a.sl1 = 1;
    a.sl1_cw = sl1->has_code_weight != 0;
    a.sin_diff = sl1->diff_rad_by_sin != 0;
    a.beta = sl1->beta;
    a.sl1_scale = sl1->scale;
    for (int k = 0; k < 7; ++k) a.cw[k] = sl1->code_weight[k];
  }
  a.scale = scale;
  a.avg_dev = avg_dev;
  a.w_gd = w_gd;
  a.w_sl1 = w_sl1;
  a.alpha = p->alpha;
  a.ia2 = gd3d_inv_alpha2(p->alpha);
  a.tau = p->tau;
  a.c0 = p->center_offset[0];
  a.c1 = p->center_offset[1];
  a.c2 = p->center_offset[2];
  const long long nb = (P + HEAD_T - 1) / HEAD_T;
  if (nb > 0x7fffffffLL) return GD3D_E_TOOLARGE;
  const bool flag = p->flag != 0;
  const unsigned grid = (unsigned)nb;
  switch (p->loss_type) {
    case GD3D_GWD3D: launch_head_fun<GD3D_GWD3D>(p->fun, flag, grid, s, a); break;
    case GD3D_KLD3D: launch_head_fun<GD3D_KLD3D>(p->fun, flag, grid, s, a); break;
    case GD3D_BD3D: launch_head_fun<GD3D_BD3D>(p->fun, flag, grid, s, a); break;
    case GD3D_JD3D: launch_head_fun<GD3D_JD3D>(p->fun, flag, grid, s, a); break;
    case GD3D_KLD3D_SYMMAX: launch_head_fun<GD3D_KLD3D_SYMMAX>(p->fun, flag, grid, s, a); break;
    case GD3D_KLD3D_SYMMIN: launch_head_fun<GD3D_KLD3D_SYMMIN>(p->fun, flag, grid, s, a); break;
    default:
      if (p->fun == GD3D_FUN_EXPM1) launch_head<GD3D_KFIOU3D, GD3D_FUN_EXPM1>(false, grid, s, a);
      else if (p->fun == GD3D_FUN_NLOG) launch_head<GD3D_KFIOU3D, GD3D_FUN_NLOG>(false, grid, s, a);
      else launch_head<GD3D_KFIOU3D, GD3D_FUN_NONE>(false, grid, s, a);
      break;
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return (int)e;
  if (loss_sum != nullptr) {
    hipLaunchKernelGGL(reduce_partials_kernel, dim3(1), dim3(1024), 0, s, (const float*)workspace, nb, loss_sum);
    return (int)hipGetLastError();
  }
  return 0;
}

int gd3d_anchor_head_loss(const gd3d_params* p, const float* bbox_pred, int32_t B, int32_t A, int32_t H, int32_t W,
                          const float* bbox_targets, const float* bbox_weights, const float* decode_weight,
                          const float* anchors, const int64_t* pos_inds, int64_t P, float scale, float* loss_sum,
                          float* grad_bbox_pred, void* workspace, void* stream) {
  if (P > 0 && pos_inds == nullptr) return GD3D_E_BADARG;
  return anchor_head_impl(p, nullptr, bbox_pred, B, A, H, W, bbox_targets, bbox_weights, decode_weight, anchors, pos_inds,
                          nullptr, 0, P, scale, loss_sum, grad_bbox_pred, workspace, stream);
}

int gd3d_anchor_head_bbox_loss(const gd3d_params* p, const gd3d_smooth_l1* sl1, const float* bbox_pred, int32_t B,
                               int32_t A, int32_t H, int32_t W, const float* bbox_targets, const float* bbox_weights,
                               const float* decode_weight, const float* anchors, const int64_t* pos_inds, int64_t P,
                               const int64_t* labels, int32_t num_classes, float scale, float* loss_sum,
                               float* grad_bbox_pred, void* workspace, void* stream) {
  if (B <= 0 || A <= 0 || H <= 0 || W <= 0) return GD3D_E_BADARG;
  if ((pos_inds != nullptr) == (labels != nullptr)) return GD3D_E_BADARG;  // exactly one way to name the positives
  if (labels != nullptr) P = (int64_t)B * A * H * W;
  return anchor_head_impl(p, sl1, bbox_pred, B, A, H, W, bbox_targets, bbox_weights, decode_weight, anchors, pos_inds,
                          labels, num_classes, P, scale, loss_sum, grad_bbox_pred, workspace, stream);
}

int gd3d_anchor_head_bbox_loss_dyn(const gd3d_params* p, const gd3d_smooth_l1* sl1, const float* bbox_pred, int32_t B,
                                   int32_t A, int32_t H, int32_t W, const float* bbox_targets, const float* bbox_weights,
                                   const float* decode_weight, const float* anchors, const int64_t* labels,
                                   int32_t num_classes, double gd_weight, double sl1_weight, const float* avg_dev,
                                   float* loss_sum, float* grad_bbox_pred, void* workspace, void* stream) {
  if (B <= 0 || A <= 0 || H <= 0 || W <= 0 || labels == nullptr || avg_dev == nullptr) return GD3D_E_BADARG;
  return anchor_head_impl(p, sl1, bbox_pred, B, A, H, W, bbox_targets, bbox_weights, decode_weight, anchors, nullptr, labels,
                          num_classes, (int64_t)B * A * H * W, 0.0f, loss_sum, grad_bbox_pred, workspace, stream, avg_dev,
                          gd_weight, sl1_weight);
}

int gd3d_anchor_head_loss_dense(const gd3d_params* p, const float* bbox_pred, int32_t B, int32_t A, int32_t H, int32_t W,
                                const float* bbox_targets, const float* bbox_weights, const float* decode_weight,
                                const float* anchors, const int64_t* labels, int32_t num_classes, float scale,
                                float* loss_sum, float* grad_bbox_pred, void* workspace, void* stream) {
  if (B <= 0 || A <= 0 || H <= 0 || W <= 0 || labels == nullptr) return GD3D_E_BADARG;
  const int64_t M = (int64_t)B * A * H * W;
  return anchor_head_impl(p, nullptr, bbox_pred, B, A, H, W, bbox_targets, bbox_weights, decode_weight, anchors, nullptr,
                          labels, num_classes, M, scale, loss_sum, grad_bbox_pred, workspace, stream);
}

static size_t center_partial_bytes(int32_t num_tasks, int64_t max_n) {
  const int64_t nb = (max_n + HEAD_T - 1) / HEAD_T;
  return (size_t)((2 * (int64_t)num_tasks * nb * 4 + 15) / 16 * 16);
}

static int center_fill(const gd3d_params* p, const gd3d_prologue* coder, const gd3d_center_task* tasks, int32_t num_tasks,
                       const float* code_weights, int32_t n_l1, void* workspace, CenterArgs& a, long long& max_n) {
  if (p == nullptr || coder == nullptr || tasks == nullptr || num_tasks <= 0 || num_tasks > CENTER_MAX_TASKS)
    return GD3D_E_BADARG;
  if (n_l1 != 0 && n_l1 != 2 && n_l1 != 4) return GD3D_E_BADARG;
  if (n_l1 > 0 && code_weights == nullptr) return GD3D_E_BADARG;
  a.num_tasks = num_tasks;
  a.n_l1 = n_l1;
  a.norm_bbox = coder->norm_bbox;
  a.osf = coder->out_size_factor;
  a.vs0 = coder->voxel_size[0];
  a.vs1 = coder->voxel_size[1];
  a.pc0 = coder->pc_range[0];
  a.pc1 = coder->pc_range[1];
  a.alpha = p->alpha;
  a.ia2 = gd3d_inv_alpha2(p->alpha);
  a.tau = p->tau;
  a.c0 = p->center_offset[0];
  a.c1 = p->center_offset[1];
  a.c2 = p->center_offset[2];
  for (int k = 0; k < 4; ++k) a.cw[k] = k < n_l1 ? code_weights[k] : 0.0f;
  max_n = 0;
  for (int t = 0; t < num_tasks; ++t) {
    const gd3d_center_task& s = tasks[t];
    if (s.n < 0 || s.B <= 0 || s.H <= 0 || s.W <= 0) return GD3D_E_BADARG;
    if (s.n > 0) {
      if (s.pos_ind == nullptr || s.anno == nullptr || s.anno_cols < 7 + (n_l1 > 2 ? 2 : 0)) return GD3D_E_BADARG;
      for (int m = 1; m <= 3; ++m)
        if (s.maps[m] == nullptr) return GD3D_E_BADARG;
      if (n_l1 >= 2 && s.maps[4] == nullptr) return GD3D_E_BADARG;
      if (n_l1 == 4 && s.maps[5] == nullptr) return GD3D_E_BADARG;
    }
    CenterTask& d = a.t[t];
    for (int m = 0; m < 6; ++m) {
      d.maps[m] = s.maps[m];
      d.grads[m] = s.grads[m];
    }
    d.pos_ind = (const long long*)s.pos_ind;
    d.anno = s.anno;
    bool wants = false;
    for (int m = 0; m < 6; ++m) wants |= s.grads[m] != nullptr;
    if (wants && s.cell_count == nullptr) return GD3D_E_BADARG;
    d.count = wants ? (int*)s.cell_count : nullptr;
    d.keys = nullptr;
    d.og = nullptr;
    d.n = s.n;
    d.B = s.B;
    d.H = s.H;
    d.W = s.W;
    d.anno_cols = s.anno_cols;
    d.gd_scale = s.gd_scale;
    d.l1_scale = s.l1_scale;
    d.rows_dev = (const long long*)s.rows_dev;
    d.avg_dev = s.avg_dev;
    d.gd_weight = s.gd_weight;
    d.l1_weight = s.l1_weight;
    if (s.n > max_n) max_n = s.n;
  }
  a.partials = (float*)workspace;
  a.pstride = (max_n + HEAD_T - 1) / HEAD_T;
  a.max_n = max_n;
  // workspace: partials (2 * tasks * pstride floats, padded to 16 B) | per task: keys (max_n int32) | og (max_n * 11 fp32)
  if (workspace != nullptr) {
    char* base = (char*)workspace + center_partial_bytes(num_tasks, max_n);
    for (int t = 0; t < num_tasks; ++t) {
      a.t[t].keys = (int*)(base + (size_t)t * 48 * (size_t)max_n);
      a.t[t].og = (float*)(base + (size_t)t * 48 * (size_t)max_n + 4 * (size_t)max_n);
    }
  }
  return 0;
}

size_t gd3d_center_head_workspace_bytes(int32_t num_tasks, int64_t max_n) {
  if (num_tasks <= 0 || max_n <= 0) return 16;
  return center_partial_bytes(num_tasks, max_n) + (size_t)num_tasks * 48 * (size_t)max_n;
}

static int center_stage(const gd3d_params* p, const gd3d_prologue* coder, const gd3d_center_task* tasks, int32_t num_tasks,
                        const float* code_weights, int32_t n_l1, float* losses, void* workspace, void* stream, CenterArgs& a,
                        long long& max_n, bool launch) {
  const int rc = center_fill(p, coder, tasks, num_tasks, code_weights, n_l1, workspace, a, max_n);
  if (rc != 0) return rc;
  if (losses == nullptr) return GD3D_E_BADARG;
  if (p->loss_type < 0 || p->loss_type >= GD3D_NUM_LOSS_TYPES) return GD3D_E_BADARG;
  if (p->loss_type == GD3D_KFIOU3D) {
    if (p->fun != GD3D_FUN_NONE && p->fun != GD3D_FUN_EXPM1 && p->fun != GD3D_FUN_NLOG) return GD3D_E_BADARG;
  } else if (p->fun != GD3D_FUN_NONE && p->fun != GD3D_FUN_LOG1P) {
    return GD3D_E_BADARG;
  }
  hipStream_t s = (hipStream_t)stream;
  if (max_n == 0) return launch ? fill_words(losses, sizeof(float) * 2 * (size_t)num_tasks, 0u, s) : 0;
  if (workspace == nullptr) return GD3D_E_BADARG;
  if (a.pstride > 0x7fffffffLL) return GD3D_E_TOOLARGE;
  if (!launch) return 0;
  const dim3 grid((unsigned)a.pstride, (unsigned)num_tasks);
  const bool flag = p->flag != 0;
#define GD3D_CENTER_LAUNCH(LT, FN, FL) hipLaunchKernelGGL((head_center_kernel<LT, FN, FL>), grid, dim3(HEAD_T), 0, s, a)
#define GD3D_CENTER_FUN(LT)                                                   \
  if (p->fun == GD3D_FUN_LOG1P) {                                             \
    if (flag) GD3D_CENTER_LAUNCH(LT, GD3D_FUN_LOG1P, true);                   \
    else GD3D_CENTER_LAUNCH(LT, GD3D_FUN_LOG1P, false);                       \
  } else {                                                                    \
    if (flag) GD3D_CENTER_LAUNCH(LT, GD3D_FUN_NONE, true);                    \
    else GD3D_CENTER_LAUNCH(LT, GD3D_FUN_NONE, false);                        \
  }
  switch (p->loss_type) {
    case GD3D_GWD3D: GD3D_CENTER_FUN(GD3D_GWD3D) break;
    case GD3D_KLD3D: GD3D_CENTER_FUN(GD3D_KLD3D) break;
    case GD3D_BD3D: GD3D_CENTER_FUN(GD3D_BD3D) break;
    case GD3D_JD3D: GD3D_CENTER_FUN(GD3D_JD3D) break;
    case GD3D_KLD3D_SYMMAX: GD3D_CENTER_FUN(GD3D_KLD3D_SYMMAX) break;
    case GD3D_KLD3D_SYMMIN: GD3D_CENTER_FUN(GD3D_KLD3D_SYMMIN) break;
    default:
      if (p->fun == GD3D_FUN_EXPM1) GD3D_CENTER_LAUNCH(GD3D_KFIOU3D, GD3D_FUN_EXPM1, false);
      else if (p->fun == GD3D_FUN_NLOG) GD3D_CENTER_LAUNCH(GD3D_KFIOU3D, GD3D_FUN_NLOG, false);
      else GD3D_CENTER_LAUNCH(GD3D_KFIOU3D, GD3D_FUN_NONE, false);
      break;
  }
#undef GD3D_CENTER_FUN
#undef GD3D_CENTER_LAUNCH
  return (int)hipGetLastError();
}

static int center_finish(const CenterArgs& a, int32_t num_tasks, long long max_n, float* losses, const int64_t* order,
                         void* stream) {
  if (max_n == 0) return 0;   // the stage call already zeroed the losses
  const dim3 grid((unsigned)a.pstride, (unsigned)num_tasks);
  // second step: gradient accumulation (deterministic) + the loss sums.  Tasks without positives: their partial slices
  // are never written; the sum reads nb = 0 entries -> 0
  if (order != nullptr)
    hipLaunchKernelGGL(center_accum_sorted_kernel, grid, dim3(HEAD_T), 0, (hipStream_t)stream, a, losses, (const long long*)order);
  else
    hipLaunchKernelGGL(center_accum_kernel, grid, dim3(HEAD_T), 0, (hipStream_t)stream, a, losses);
  return (int)hipGetLastError();
}

int gd3d_center_head_loss(const gd3d_params* p, const gd3d_prologue* coder, const gd3d_center_task* tasks, int32_t num_tasks,
                          const float* code_weights, int32_t n_l1, float* losses, void* workspace, void* stream) {
  CenterArgs a;
  long long max_n = 0;
  const int rc = center_stage(p, coder, tasks, num_tasks, code_weights, n_l1, losses, workspace, stream, a, max_n, true);
  if (rc != 0) return rc;
  return center_finish(a, num_tasks, max_n, losses, nullptr, stream);
}

int gd3d_center_head_stage(const gd3d_params* p, const gd3d_prologue* coder, const gd3d_center_task* tasks, int32_t num_tasks,
                           const float* code_weights, int32_t n_l1, float* losses, void* workspace, void* stream) {
  CenterArgs a;
  long long max_n = 0;
  return center_stage(p, coder, tasks, num_tasks, code_weights, n_l1, losses, workspace, stream, a, max_n, true);
}

int gd3d_center_head_finish(const gd3d_params* p, const gd3d_prologue* coder, const gd3d_center_task* tasks, int32_t num_tasks,
                            const float* code_weights, int32_t n_l1, float* losses, void* workspace, const int64_t* order,
                            void* stream) {
  CenterArgs a;
  long long max_n = 0;
  const int rc = center_stage(p, coder, tasks, num_tasks, code_weights, n_l1, losses, workspace, stream, a, max_n, false);
  if (rc != 0) return rc;
  return center_finish(a, num_tasks, max_n, losses, order, stream);
}

int gd3d_center_head_keys(int32_t num_tasks, int64_t max_n, int64_t* byte_offset, int64_t* byte_stride) {
  if (num_tasks <= 0 || max_n < 0 || byte_offset == nullptr || byte_stride == nullptr) return GD3D_E_BADARG;
  *byte_offset = (int64_t)center_partial_bytes(num_tasks, max_n);
  *byte_stride = 48 * max_n;
  return 0;
}

int gd3d_center_head_scale(const gd3d_center_task* tasks, int32_t num_tasks, const float* grad_losses, void* stream) {
  if (tasks == nullptr || num_tasks <= 0 || num_tasks > CENTER_MAX_TASKS || grad_losses == nullptr) return GD3D_E_BADARG;
  CenterArgs a;
  a.num_tasks = num_tasks;
  for (int t = 0; t < num_tasks; ++t) {
    for (int m = 0; m < 6; ++m) {
      a.t[t].maps[m] = nullptr;
      a.t[t].grads[m] = tasks[t].grads[m];
    }
    if (tasks[t].B <= 0 || tasks[t].H <= 0 || tasks[t].W <= 0) return GD3D_E_BADARG;
    a.t[t].B = tasks[t].B;
    a.t[t].H = tasks[t].H;
    a.t[t].W = tasks[t].W;
    a.t[t].n = 0;
    a.t[t].count = nullptr;
    a.t[t].keys = nullptr;
    a.t[t].og = nullptr;
  }
  hipLaunchKernelGGL(center_scale_kernel, dim3(64, 6 * (unsigned)num_tasks), dim3(256), 0, (hipStream_t)stream, a,
                     grad_losses);
  return (int)hipGetLastError();
}

int gd3d_prof_event_create(void** event) {
  if (event == nullptr) return GD3D_E_BADARG;
  hipEvent_t e = nullptr;
  const hipError_t rc = hipEventCreate(&e);
  *event = (void*)e;
  return (int)rc;
}

int gd3d_prof_event_destroy(void* event) {
  if (event == nullptr) return 0;
  return (int)hipEventDestroy((hipEvent_t)event);
}

int gd3d_prof_event_elapsed_ms(void* start_event, void* stop_event, float* ms) {
  if (start_event == nullptr || stop_event == nullptr || ms == nullptr) return GD3D_E_BADARG;
  return (int)hipEventElapsedTime(ms, (hipEvent_t)start_event, (hipEvent_t)stop_event);
}

int gd3d_scale_rows(float* grad, const float* g, int per_row, int64_t n, void* stream) {
  if (n < 0 || (n > 0 && (grad == nullptr || g == nullptr))) return GD3D_E_BADARG;
  if (n == 0) return 0;
  const long long nflt = (long long)n * 7;
  // grid-stride with FEW, LARGE workgroups: the common case is the g == 1 early exit, whose cost is the dispatch of the
  // workgroups (1024 x 256 threads: 4.5 us; 256 x 1024 threads: the same 262 144 lanes for a real scaling pass)
  long long blocks = (nflt + 1023) / 1024;
  if (blocks > 256) blocks = 256;
  hipLaunchKernelGGL(scale_rows_kernel, dim3((unsigned)blocks), dim3(1024), 0, (hipStream_t)stream, grad, g, per_row,
                     nflt);
  return (int)hipGetLastError();
}

int gd3d_abi_version(const char** arch) {
  if (arch != nullptr) *arch = "gfx950";
  return GD3D_ABI_VERSION;
}

}  // extern "C"
