// common.hpp -- shared device helpers for the gfx950 kernels (wave64, fp32 MFMA 16x16x4).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/g2v.h"

// Device address of the persistent rollouts' fault latch (dec_persist.hip; NULL if it cannot be resolved).  The kernels that
// COMMIT a training step's results to the model state -- clip + Adam, the EMA codebook update, the BatchNorm running statistics --
// read it and leave the state untouched when a fault is latched: the step's gradients / statistics are garbage then, and the host
// (which reads the latch at its next sync point) can simply repeat the step on the per-step kernels.
const unsigned* g2v_internal_persist_fault_ptr();

// The three implementation switches (include/g2v.h: g2v_ctx) of the context bound to the CALLING THREAD, or of the process's
// default context when the thread has bound none (misc.hip owns both).  Everything that used to read a file-scope `static int`
// reads these.
struct G2vOptions {
  int persist = 1;            // G2V_OPT_PERSISTENT   0..3   (dec_rollout.hip)
  int gru_cluster = 1;        // G2V_OPT_GRU_CLUSTER  0 / 1  (gru.hip)
  int smallm_max_rows = 1024; // G2V_OPT_SMALLM_ROWS  >= 0   (linear.hip)
  int gru_resident_rows = 1025; // G2V_OPT_GRU_RESIDENT_ROWS  batch rows from which g2v_gru_seq_fwd / _bwd keep W_hh in the CU; 0 = never (gru.hip)
  int gru_resident_bwd = 1;     // G2V_OPT_GRU_RESIDENT_BWD   0 / 1: the BPTT too (a resident launch leaves no room on a CU for a co-running kernel)
};
G2vOptions& g2v_internal_options();

namespace g2v {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int WAVE = 64;
constexpr int TILE = 16;  // MFMA 16x16x4 tile edge

void set_error(const char* fmt, ...);

#define G2V_REQUIRE(cond, msg)                                  \
  do {                                                          \
    if (!(cond)) {                                              \
      g2v::set_error("%s: %s", __func__, msg);                  \
      return G2V_ERR_ARG;                                       \
    }                                                           \
  } while (0)

#define G2V_CHECK_LAUNCH()                                                          \
  do {                                                                              \
    hipError_t e_ = hipGetLastError();                                              \
    if (e_ != hipSuccess) {                                                         \
      g2v::set_error("%s: launch failed: %s", __func__, hipGetErrorString(e_));     \
      return G2V_ERR_LAUNCH;                                                        \
    }                                                                               \
  } while (0)

static inline int cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }
static inline int round_up(int a, int b) { return (a + b - 1) / b * b; }

// ---------------------------------------------------------------------------------------------
// v_mfma_f32_16x16x4_f32:  D(16x16) = A(16x4) * B(4x16) + C.   Lane l supplies
//   a = A[i = l & 15][k = l >> 4],  b = B[k = l >> 4][j = l & 15];
// it receives D[i = 4*(l >> 4) + r][j = l & 15] in element r of the accumulator.
// Numerics: bit-for-bit a k-ordered chain of fp32 fmas (no wider accumulation).
// ---------------------------------------------------------------------------------------------
// Workgroup barrier that orders LDS traffic ONLY.  `__syncthreads()` also drains the vector-memory counter
// (s_waitcnt vmcnt(0)): every barrier would then wait for this wave's outstanding global loads AND stores
// (an HBM round trip, ~2-3k cycles), serialising the prefetches and the saved-for-backward stores with the MFMAs.
// Use it wherever the barrier protects LDS data only (never for global-memory hand-offs between threads).
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// Four consecutive k-values (k..k+3) of one row of a row-major global matrix, zero-filled where
// !valid or k >= K.  `p_row` points at the row start.
__device__ __forceinline__ float4 ldg_frag(const float* __restrict__ p_row, bool valid, int k, int K, bool vec_ok) {
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  if (valid) {
    const float* p = p_row + k;
    if (vec_ok && k + 3 < K) {
      v = *reinterpret_cast<const float4*>(p);
    } else {
      if (k < K) v.x = p[0];
      if (k + 1 < K) v.y = p[1];
      if (k + 2 < K) v.z = p[2];
      if (k + 3 < K) v.w = p[3];
    }
  }
  return v;
}

__host__ __device__ __forceinline__ bool ptr_vec_ok(const void* p, int64_t ld) {
  return ((reinterpret_cast<uintptr_t>(p) & 15) == 0) && ((ld & 3) == 0);
}

// One wave accumulates NT 16x16 tiles over a K-deep contraction:
//   acc[t][r] += sum_k W[n0 + t*nstride + 4q + r][k] * X[j][k]      (q = lane>>4, j = lane&15)
// W: global row-major [.. ][ldw] (weights; L2 resident), rows valid while (feature-in-tile index) < nvalid.
// Xs: LDS [16][ldx], zero padded to a multiple of 16 columns >= K.
// Each lane fetches float4s along k for both operands (k = k0 + 4q + e), i.e. MFMA e of a group
// contracts the k-set {k0 + 4q' + e : q' = 0..3}; any pairing is valid as long as A and B agree.
template <int NT>
__device__ __forceinline__ void wave_gemm(f32x4 (&acc)[NT], const float* __restrict__ W, int64_t ldw, bool wvec,
                                          int n0, int nstride, int nvalid, int K,
                                          const float* Xs, int ldx, int lane) {
  const int i = lane & 15, q = lane >> 4;
  const int Kp = (K + 15) & ~15;
  const bool valid = i < nvalid;
  const float* wrow[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) wrow[t] = W + (int64_t)(n0 + t * nstride + (valid ? i : 0)) * ldw;
  if (wvec) {
    // 16-byte-aligned rows (K % 4 == 0): branch-free fragment loads (clamped address + select), four k-steps per chunk,
    // the NEXT chunk's 4*NT loads in flight while the current chunk's 16*NT MFMAs run (one exposed L2 round trip per
    // 4 k-steps instead of one per k-step).
    constexpr int CH = 4;
    const int nks = Kp >> 4, nch = (nks + CH - 1) / CH;
    float4 wa[CH][NT], wb[CH][NT];
    auto loadc = [&](int c, float4 (&w)[CH][NT]) {
#pragma unroll
      for (int j = 0; j < CH; ++j) {
        const int k = 16 * (c * CH + j) + 4 * q;
        const bool ok = valid && (k < K);
        const int kk = ok ? k : 0;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          const float4 v = *reinterpret_cast<const float4*>(wrow[t] + kk);
          w[j][t] = ok ? v : make_float4(0.f, 0.f, 0.f, 0.f);
        }
      }
    };
    auto mmac = [&](int c, const float4 (&w)[CH][NT]) {
#pragma unroll
      for (int j = 0; j < CH; ++j) {
        const int ks = c * CH + j;
        if (ks < nks) {
          const float4 xb = *reinterpret_cast<const float4*>(Xs + i * ldx + 16 * ks + 4 * q);
#pragma unroll
          for (int t = 0; t < NT; ++t) {
            acc[t] = mfma16(w[j][t].x, xb.x, acc[t]);
            acc[t] = mfma16(w[j][t].y, xb.y, acc[t]);
            acc[t] = mfma16(w[j][t].z, xb.z, acc[t]);
            acc[t] = mfma16(w[j][t].w, xb.w, acc[t]);
          }
        }
      }
    };
    loadc(0, wa);
    for (int c = 0; c < nch; c += 2) {
      loadc(c + 1, wb);                    // past the end: clamped addresses, zero fragments
      __builtin_amdgcn_sched_barrier(0);
      mmac(c, wa);
      loadc(c + 2, wa);
      __builtin_amdgcn_sched_barrier(0);
      if (c + 1 < nch) mmac(c + 1, wb);
    }
    return;
  }
  for (int k0 = 0; k0 < Kp; k0 += 16) {
    const float4 xb = *reinterpret_cast<const float4*>(Xs + i * ldx + k0 + 4 * q);
    float4 wa[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) wa[t] = ldg_frag(wrow[t], valid, k0 + 4 * q, K, wvec);
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      acc[t] = mfma16(wa[t].x, xb.x, acc[t]);
      acc[t] = mfma16(wa[t].y, xb.y, acc[t]);
      acc[t] = mfma16(wa[t].z, xb.z, acc[t]);
      acc[t] = mfma16(wa[t].w, xb.w, acc[t]);
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Packed ("fragment-major") weights.  A weight matrix used as the MFMA A operand is re-laid-out once per
// call so that the 64 lanes of a wave read one fully coalesced, 16-byte-aligned 1 KiB block per
// (16-feature tile, 16-deep k-step), with zero padding baked in (no bounds checks, no alignment cases):
//   P[((tile * KS + s) * 64 + lane) * 4 + e] = W[row(tile, lane & 15)][16 s + 4 (lane >> 4) + e]
// Gate matrices (3H x H) are packed gate-major: tile = g * ntile_g + f covers rows g*H + 16 f + i.
// ---------------------------------------------------------------------------------------------
struct PackDesc {
  const float* src;
  float* dst;
  int rows_per_group, ngroups, group_row_stride, K, ld, transposed;   // transposed: elem(r,c) = src[c*ld + r]
  int ntile_alloc;   // tiles written per group (>= ceil(rows/16); extra tiles are zero) ; 0 = exactly ceil(rows/16)
};
static inline int pack_ntile_g(int rows_per_group) { return (rows_per_group + 15) >> 4; }
static inline int pack_ks(int K) { return (K + 15) >> 4; }
static inline size_t pack_floats(int rows_per_group, int ngroups, int K) {
  return (size_t)ngroups * pack_ntile_g(rows_per_group) * pack_ks(K) * 256;
}
constexpr int MAX_PACK = 8;
struct PackBatch {
  PackDesc d[MAX_PACK];
  int n;
};
__device__ __forceinline__ void pack_one(const PackDesc& d) {
  const int ntg = d.ntile_alloc > 0 ? d.ntile_alloc : ((d.rows_per_group + 15) >> 4), KS = (d.K + 15) >> 4;
  const int nblocks = d.ngroups * ntg * KS;            // one 256-float block per (tile, kstep) = 64 lanes x 4
  for (int blk = blockIdx.x; blk < nblocks; blk += gridDim.x) {
    const int s = blk % KS, tile = blk / KS;
    const int g = tile / ntg, f = tile - g * ntg;
    const int lane = threadIdx.x >> 2, e = threadIdx.x & 3;
    const int i = lane & 15, q = lane >> 4;
    const int rg = 16 * f + i, c = 16 * s + 4 * q + e;
    float v = 0.f;
    if (rg < d.rows_per_group && c < d.K) {
      const int r = g * d.group_row_stride + rg;
      v = d.transposed ? d.src[(int64_t)c * d.ld + r] : d.src[(int64_t)r * d.ld + c];
    }
    d.dst[(int64_t)blk * 256 + threadIdx.x] = v;
  }
}
static __global__ __launch_bounds__(256) void pack_kernel(PackBatch pb) { pack_one(pb.d[blockIdx.y]); }
// The packs of SEVERAL launches' workspaces plus the zero-fill of up to two regions (16-byte aligned, multiples of 16 bytes:
// the persistent rollouts' exchange records) as ONE launch: g2v_train_step_prepare.
constexpr int MAX_PACK_L = 16, MAX_FILL = 2;
struct PackBatchL {
  PackDesc d[MAX_PACK_L];
  int n, nfill;
  float4* fill_dst[MAX_FILL];
  int64_t fill_n4[MAX_FILL];
};
static __global__ __launch_bounds__(256) void pack_fill_kernel(PackBatchL pb) {
  if ((int)blockIdx.y < pb.n) {
    pack_one(pb.d[blockIdx.y]);
    return;
  }
  const int k = blockIdx.y - pb.n;
  float4* dst = pb.fill_dst[k];
  const int64_t n4 = pb.fill_n4[k];
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n4; e += (int64_t)gridDim.x * 256)
    dst[e] = make_float4(0.f, 0.f, 0.f, 0.f);
}
static inline void launch_pack_fill(const PackBatchL& pb, hipStream_t st) {
  hipLaunchKernelGGL(pack_fill_kernel, dim3(64, pb.n + pb.nfill), dim3(256), 0, st, pb);
}
static inline void launch_pack(const PackBatch& pb, hipStream_t st) {
  hipLaunchKernelGGL(pack_kernel, dim3(64, pb.n), dim3(256), 0, st, pb);
}

// One wave accumulates NT tiles from PACKED weights: tile index of accumulator t is tile0 + t * tile_stride.
// KS_T > 0: compile-time k-steps, every fragment load is issued before the first MFMA (registers: NT*KS_T*4).
// KS_T == 0: run-time k-steps with a one-step register prefetch.
template <int NT, int KS_T>
__device__ __forceinline__ void wave_gemm_p(f32x4 (&acc)[NT], const float* __restrict__ P, int KS, int tile0,
                                            int tile_stride, const float* Xs, int ldx, int lane) {
  const int i = lane & 15, q = lane >> 4;
  const float* xrow = Xs + i * ldx + 4 * q;
  if constexpr (KS_T > 0) {
    float4 wa[NT][KS_T];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int s = 0; s < KS_T; ++s)
        wa[t][s] = *reinterpret_cast<const float4*>(P + ((int64_t)((tile0 + t * tile_stride) * KS_T + s) * 64 + lane) * 4);
    __builtin_amdgcn_sched_barrier(0);   // keep every fragment load ahead of the MFMAs (hipcc otherwise re-rolls them)
#pragma unroll
    for (int s = 0; s < KS_T; ++s) {
      const float4 xb = *reinterpret_cast<const float4*>(xrow + 16 * s);
#pragma unroll
      for (int t = 0; t < NT; ++t) acc[t] = mfma16(wa[t][s].x, xb.x, acc[t]);
#pragma unroll
      for (int t = 0; t < NT; ++t) acc[t] = mfma16(wa[t][s].y, xb.y, acc[t]);
#pragma unroll
      for (int t = 0; t < NT; ++t) acc[t] = mfma16(wa[t][s].z, xb.z, acc[t]);
#pragma unroll
      for (int t = 0; t < NT; ++t) acc[t] = mfma16(wa[t][s].w, xb.w, acc[t]);
    }
  } else {
    const float* pt[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) pt[t] = P + ((int64_t)(tile0 + t * tile_stride) * KS * 64 + lane) * 4;
    // PD k-steps of fragments in flight: step s is multiplied from ring slot s % PD, which is then refilled with step
    // s + PD - a fragment has PD x 4 NT MFMAs (>= 1.5k cycles for NT = 3) to arrive instead of one k-step's 384, and the
    // refills go out one k-step at a time between the MFMAs.
    constexpr int PD = 8;
    float4 ring[PD][NT];
#pragma unroll
    for (int j = 0; j < PD; ++j)
#pragma unroll
      for (int t = 0; t < NT; ++t) ring[j][t] = *reinterpret_cast<const float4*>(pt[t] + (int64_t)min(j, KS - 1) * 256);
    // (round 4) the activation fragment of k-step s+1 is read from LDS BEFORE the MFMAs of k-step s: behind the sched_barrier
    // that pins the ring refills, the read used to be issued only after the previous k-step's MFMAs and its ~130-cycle LDS round
    // trip stood in front of every group of 4 NT MFMAs
    float4 xn = *reinterpret_cast<const float4*>(xrow);
    auto kstep = [&](int s, int j) {
        {
          const float4 xb = xn;
          xn = *reinterpret_cast<const float4*>(xrow + 16 * min(s + 1, KS - 1));
#pragma unroll
          for (int t = 0; t < NT; ++t) acc[t] = mfma16(ring[j][t].x, xb.x, acc[t]);
#pragma unroll
          for (int t = 0; t < NT; ++t) acc[t] = mfma16(ring[j][t].y, xb.y, acc[t]);
#pragma unroll
          for (int t = 0; t < NT; ++t) acc[t] = mfma16(ring[j][t].z, xb.z, acc[t]);
#pragma unroll
          for (int t = 0; t < NT; ++t) acc[t] = mfma16(ring[j][t].w, xb.w, acc[t]);
          const int sn = min(s + PD, KS - 1);         // past the end: a valid, unused fragment
#pragma unroll
          for (int t = 0; t < NT; ++t) ring[j][t] = *reinterpret_cast<const float4*>(pt[t] + (int64_t)sn * 256);
          __builtin_amdgcn_sched_barrier(0);
        }
    };
    // (whole groups of PD k-steps branch-free, then the tail: see wave_gemm_p_rows)
    int s0 = 0;
    for (; s0 + PD <= KS; s0 += PD) {
#pragma unroll
      for (int j = 0; j < PD; ++j) kstep(s0 + j, j);
    }
    const int rem = KS - s0;
#pragma unroll
    for (int j = 0; j < PD; ++j)
      if (j < rem) kstep(s0 + j, j);
  }
}

// wave_gemm_p (run-time k-steps, PD fragments in flight) over NR 16-row tiles of the SAME activation matrix that share every
// weight fragment: acc[t][r] += tile t of P (A operand) x rows [16 r, 16 r + 16) of Xs.  A kernel that streams its weights from
// L2 every time step does 4 NT NR MFMAs per 1 KiB fragment instead of 4 NT: at 16 rows per workgroup the generic recurrent
// kernels were bound by the fragment stream (one KiB per 128 cycles and wave = the CU's whole L1 rate at two workgroups per CU).
template <int NT, int NR, int PD = 8, bool WHOLE_GROUPS = true>
__device__ __forceinline__ void wave_gemm_p_rows(f32x4 (&acc)[NT][NR], const float* __restrict__ P, int KS, int tile0,
                                                 int tile_stride, const float* Xs, int ldx, int lane) {
  const int i = lane & 15, q = lane >> 4;
  const float* xrow = Xs + i * ldx + 4 * q;
  const float* pt[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) pt[t] = P + ((int64_t)(tile0 + t * tile_stride) * KS * 64 + lane) * 4;
  float4 ring[PD][NT];
#pragma unroll
  for (int j = 0; j < PD; ++j)
#pragma unroll
    for (int t = 0; t < NT; ++t) ring[j][t] = *reinterpret_cast<const float4*>(pt[t] + (int64_t)min(j, KS - 1) * 256);
  float4 xn[NR];
#pragma unroll
  for (int r = 0; r < NR; ++r) xn[r] = *reinterpret_cast<const float4*>(xrow + r * 16 * ldx);
  auto kstep = [&](int s, int j, bool refill) {
    float4 xb[NR];
#pragma unroll
    for (int r = 0; r < NR; ++r) {          // (one k-step ahead: see wave_gemm_p)
      xb[r] = xn[r];
      xn[r] = *reinterpret_cast<const float4*>(xrow + r * 16 * ldx + 16 * min(s + 1, KS - 1));
    }
#pragma unroll
    for (int r = 0; r < NR; ++r)
#pragma unroll
      for (int t = 0; t < NT; ++t) acc[t][r] = mfma16(ring[j][t].x, xb[r].x, acc[t][r]);
#pragma unroll
    for (int r = 0; r < NR; ++r)
#pragma unroll
      for (int t = 0; t < NT; ++t) acc[t][r] = mfma16(ring[j][t].y, xb[r].y, acc[t][r]);
#pragma unroll
    for (int r = 0; r < NR; ++r)
#pragma unroll
      for (int t = 0; t < NT; ++t) acc[t][r] = mfma16(ring[j][t].z, xb[r].z, acc[t][r]);
#pragma unroll
    for (int r = 0; r < NR; ++r)
#pragma unroll
      for (int t = 0; t < NT; ++t) acc[t][r] = mfma16(ring[j][t].w, xb[r].w, acc[t][r]);
    if (refill) {
      const int sn = min(s + PD, KS - 1);         // past the end: a valid, unused fragment
#pragma unroll
      for (int t = 0; t < NT; ++t) ring[j][t] = *reinterpret_cast<const float4*>(pt[t] + (int64_t)sn * 256);
    }
    __builtin_amdgcn_sched_barrier(0);
  };
  // Whole groups of PD k-steps as a BRANCH-FREE body: every path into the loop header then carries the same queue of PD x NT
  // pending fragment loads in the same order, and hipcc's wait-count pass can wait for exactly the oldest slot
  // (s_waitcnt vmcnt((PD - 1) NT)).  With a per-k-step `if (s < KS)` inside the loop it fell back to vmcnt(0) at the head of every
  // group: all PD refills -- the newest issued a few cycles earlier -- had to land before the first MFMA of the group, i.e. one full
  // L2 round trip per PD k-steps stood in the stream (seen in the ISA of every generic recurrent kernel; round 4).
  // (WHOLE_GROUPS = false keeps the per-k-step test, and with it the vmcnt(0) at every group head: A/B only)
  if constexpr (!WHOLE_GROUPS) {
    for (int s0 = 0; s0 < KS; s0 += PD) {
#pragma unroll
      for (int j = 0; j < PD; ++j)
        if (s0 + j < KS) kstep(s0 + j, j, true);
    }
  } else {
    int s0 = 0;
    for (; s0 + PD <= KS; s0 += PD) {
#pragma unroll
      for (int j = 0; j < PD; ++j) kstep(s0 + j, j, true);
    }
    const int rem = KS - s0;
#pragma unroll
    for (int j = 0; j < PD; ++j)
      if (j < rem) kstep(s0 + j, j, false);
  }
}
// wave_gemm_p for run-time k-steps over TWO weight streams at once (a GRU cell's input-side and hidden-side products of one
// feature tile): one prologue and one tail per tile instead of two, 8 NT MFMAs between refills; PD k-steps of each stream in flight.
template <int NT, int PD = 4>
__device__ __forceinline__ void wave_gemm_p_dual(f32x4 (&acc1)[NT], const float* __restrict__ P1, const float* X1,
                                                 f32x4 (&acc2)[NT], const float* __restrict__ P2, const float* X2, int KS,
                                                 int tile0, int tile_stride, int ldx, int lane) {
  const int i = lane & 15, q = lane >> 4;
  const float* x1 = X1 + i * ldx + 4 * q;
  const float* x2 = X2 + i * ldx + 4 * q;
  int64_t off[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) off[t] = ((int64_t)(tile0 + t * tile_stride) * KS * 64 + lane) * 4;
  float4 r1[PD][NT], r2[PD][NT];
#pragma unroll
  for (int j = 0; j < PD; ++j)
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int64_t o = off[t] + (int64_t)min(j, KS - 1) * 256;
      r1[j][t] = *reinterpret_cast<const float4*>(P1 + o);
      r2[j][t] = *reinterpret_cast<const float4*>(P2 + o);
    }
  float4 an = *reinterpret_cast<const float4*>(x1), bn = *reinterpret_cast<const float4*>(x2);
  auto kstep = [&](int s, int j) {
      {
        const float4 a = an, b = bn;
        an = *reinterpret_cast<const float4*>(x1 + 16 * min(s + 1, KS - 1));
        bn = *reinterpret_cast<const float4*>(x2 + 16 * min(s + 1, KS - 1));
#pragma unroll
        for (int t = 0; t < NT; ++t) acc1[t] = mfma16(r1[j][t].x, a.x, acc1[t]);
#pragma unroll
        for (int t = 0; t < NT; ++t) acc2[t] = mfma16(r2[j][t].x, b.x, acc2[t]);
#pragma unroll
        for (int t = 0; t < NT; ++t) acc1[t] = mfma16(r1[j][t].y, a.y, acc1[t]);
#pragma unroll
        for (int t = 0; t < NT; ++t) acc2[t] = mfma16(r2[j][t].y, b.y, acc2[t]);
#pragma unroll
        for (int t = 0; t < NT; ++t) acc1[t] = mfma16(r1[j][t].z, a.z, acc1[t]);
#pragma unroll
        for (int t = 0; t < NT; ++t) acc2[t] = mfma16(r2[j][t].z, b.z, acc2[t]);
#pragma unroll
        for (int t = 0; t < NT; ++t) acc1[t] = mfma16(r1[j][t].w, a.w, acc1[t]);
#pragma unroll
        for (int t = 0; t < NT; ++t) acc2[t] = mfma16(r2[j][t].w, b.w, acc2[t]);
        const int64_t on = (int64_t)min(s + PD, KS - 1) * 256;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          r1[j][t] = *reinterpret_cast<const float4*>(P1 + off[t] + on);
          r2[j][t] = *reinterpret_cast<const float4*>(P2 + off[t] + on);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
  };
  // (whole groups of PD k-steps branch-free, then the tail: see wave_gemm_p_rows)
  int s0 = 0;
  for (; s0 + PD <= KS; s0 += PD) {
#pragma unroll
    for (int j = 0; j < PD; ++j) kstep(s0 + j, j);
  }
  const int rem = KS - s0;
#pragma unroll
  for (int j = 0; j < PD; ++j)
    if (j < rem) kstep(s0 + j, j);
}
template <int NT, int KS_T>
__device__ __forceinline__ void wave_gemm_p2(f32x4 (&acc1)[NT], const float* __restrict__ P1, const float* X1,
                                             f32x4 (&acc2)[NT], const float* __restrict__ P2, const float* X2,
                                             int tile0, int tile_stride, int ldx, int lane) {
  static_assert(KS_T > 0, "static k-steps only");
  const int i = lane & 15, q = lane >> 4;
  float4 w1[NT][KS_T], w2[NT][KS_T];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int s = 0; s < KS_T; ++s) {
      const int64_t off = ((int64_t)((tile0 + t * tile_stride) * KS_T + s) * 64 + lane) * 4;
      w1[t][s] = *reinterpret_cast<const float4*>(P1 + off);
      w2[t][s] = *reinterpret_cast<const float4*>(P2 + off);
    }
  __builtin_amdgcn_sched_barrier(0);   // all 2*NT*KS_T fragment loads are in flight before the first MFMA
  const float* x1 = X1 + i * ldx + 4 * q;
  const float* x2 = X2 + i * ldx + 4 * q;
#pragma unroll
  for (int s = 0; s < KS_T; ++s) {
    const float4 a = *reinterpret_cast<const float4*>(x1 + 16 * s);
    const float4 b = *reinterpret_cast<const float4*>(x2 + 16 * s);
#pragma unroll
    for (int t = 0; t < NT; ++t) acc1[t] = mfma16(w1[t][s].x, a.x, acc1[t]);
#pragma unroll
    for (int t = 0; t < NT; ++t) acc2[t] = mfma16(w2[t][s].x, b.x, acc2[t]);
#pragma unroll
    for (int t = 0; t < NT; ++t) acc1[t] = mfma16(w1[t][s].y, a.y, acc1[t]);
#pragma unroll
    for (int t = 0; t < NT; ++t) acc2[t] = mfma16(w2[t][s].y, b.y, acc2[t]);
#pragma unroll
    for (int t = 0; t < NT; ++t) acc1[t] = mfma16(w1[t][s].z, a.z, acc1[t]);
#pragma unroll
    for (int t = 0; t < NT; ++t) acc2[t] = mfma16(w2[t][s].z, b.z, acc2[t]);
#pragma unroll
    for (int t = 0; t < NT; ++t) acc1[t] = mfma16(w1[t][s].w, a.w, acc1[t]);
#pragma unroll
    for (int t = 0; t < NT; ++t) acc2[t] = mfma16(w2[t][s].w, b.w, acc2[t]);
  }
}

// ---- register-resident weight fragments --------------------------------------------------------------------------
// frag_load issues every 16-byte fragment load of NT tiles x KS_T k-steps of a packed matrix; frag_mma consumes them.
// Splitting the two lets a kernel issue fragment loads long before the product that needs them (vmcnt retires in
// order, so a later wait for these fragments does not wait for anything issued after them).
template <int NT, int KS_T>
struct WFrag {
  float4 w[NT][KS_T];
};
template <int NT, int KS_T>
__device__ __forceinline__ void frag_load(WFrag<NT, KS_T>& f, const float* __restrict__ P, int tile0, int tile_stride,
                                          int lane) {
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int s = 0; s < KS_T; ++s)
      f.w[t][s] = *reinterpret_cast<const float4*>(P + ((int64_t)((tile0 + t * tile_stride) * KS_T + s) * 64 + lane) * 4);
}
template <int NT, int KS_T>
__device__ __forceinline__ void frag_mma(f32x4 (&acc)[NT], const WFrag<NT, KS_T>& f, const float* Xs, int ldx, int lane) {
  const float* xrow = Xs + (lane & 15) * ldx + 4 * (lane >> 4);
#pragma unroll
  for (int s = 0; s < KS_T; ++s) {
    const float4 xb = *reinterpret_cast<const float4*>(xrow + 16 * s);
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = mfma16(f.w[t][s].x, xb.x, acc[t]);
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = mfma16(f.w[t][s].y, xb.y, acc[t]);
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = mfma16(f.w[t][s].z, xb.z, acc[t]);
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = mfma16(f.w[t][s].w, xb.w, acc[t]);
  }
}

// frag_mma with loads for LATER phases interleaved into the MFMA stream: `issue(k)`, k = 0..NL-1, is called once after
// every STRIDE MFMAs (pinned with sched_barrier); k is a compile-time constant at each call site after unrolling, so
// `issue` may switch on it freely.  A burst of loads in front of a product costs the wave ~80 cycles per load (the
// CU's vector-memory front end moves ~48 B/clk, shared by its 4 waves) during which its matrix pipe idles; spread between
// MFMAs (32 cycles each) the same loads are absorbed at the rate the front end accepts them.
template <int NL, int NT, int KS_T, class Issue>
__device__ __forceinline__ void frag_mma_issue(f32x4 (&acc)[NT], const WFrag<NT, KS_T>& f, const float* Xs, int ldx, int lane,
                                               Issue issue) {
  constexpr int NM = NT * KS_T * 4;
  constexpr int STRIDE = (NL > 0 && NM / (NL > 0 ? NL : 1) > 0) ? NM / (NL > 0 ? NL : 1) : 1;
  const float* xrow = Xs + (lane & 15) * ldx + 4 * (lane >> 4);
#pragma unroll
  for (int s = 0; s < KS_T; ++s) {
    const float4 xb4 = *reinterpret_cast<const float4*>(xrow + 16 * s);
    const float xb[4] = {xb4.x, xb4.y, xb4.z, xb4.w};
#pragma unroll
    for (int c = 0; c < 4; ++c) {
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const float wv[4] = {f.w[t][s].x, f.w[t][s].y, f.w[t][s].z, f.w[t][s].w};
        acc[t] = mfma16(wv[c], xb[c], acc[t]);
        const int m = (s * 4 + c) * NT + t;
        if ((m % STRIDE) == STRIDE - 1 && (m / STRIDE) < NL) {
          issue(m / STRIDE);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
  }
#pragma unroll
  for (int k = NM / STRIDE; k < NL; ++k) issue(k);     // more loads than MFMA slots: the rest in a burst
}
// the common case: the interleaved loads are the fragments of ONE later product
template <int NT, int KS_T, int NT2, int KS2>
__device__ __forceinline__ void frag_mma_pf(f32x4 (&acc)[NT], const WFrag<NT, KS_T>& f, const float* Xs, int ldx, int lane,
                                            WFrag<NT2, KS2>& nxt, const float* __restrict__ P2, int tile0_2, int tile_stride_2,
                                            bool do_load) {
  frag_mma_issue<NT2 * KS2>(acc, f, Xs, ldx, lane, [&](int k) {
    const int t2 = k / KS2, s2 = k % KS2;
    if (do_load)
      nxt.w[t2][s2] = *reinterpret_cast<const float4*>(P2 + ((int64_t)((tile0_2 + t2 * tile_stride_2) * KS2 + s2) * 64 + lane) * 4);
  });
}

// 16-byte store that is WRITE-THROUGH (sc1): the bytes leave the XCD's L2 while the kernel is still running instead of
// sitting dirty until the end-of-kernel write-back.  For arrays this launch only produces (saved-for-backward tensors,
// gradients consumed by a later GEMM): a launch that leaves B dirty bytes adds ~B / 6 TB/s to the kernel boundary
// (MI355X_MICROARCH.md, price list row "boundary"), which is ~3 us behind the 18 MB a decoder step writes.
// `wt == false` is a plain store (arrays the NEXT launch re-reads on the same XCD: states, carries, partial sums).
__device__ __forceinline__ void st4(float* p, const float4& v, bool wt) {
  if (wt) {
    const f32x4 t = {v.x, v.y, v.z, v.w};
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(t) : "memory");
  } else {
    *reinterpret_cast<float4*>(p) = v;
  }
}

// predicated 16-byte load (the address is not touched when !ok)
__device__ __forceinline__ float4 ld4_or_zero(const float* p, bool ok) {
  return ok ? *reinterpret_cast<const float4*>(p) : make_float4(0.f, 0.f, 0.f, 0.f);
}

// v_rcp_f32 / v_rsq_f32 (1 ulp) instead of the correctly rounded division: __frcp_rn and 1.0f / sqrtf() expand to the
// ten-instruction IEEE sequence (v_div_scale x2, v_rcp, 4 x v_fma, v_div_fmas, v_div_fixup), twelve of them per lane in a GRU
// cell epilogue that sits between two dependent MFMA phases of the recurrent kernels (~120 of its ~270 VALU instructions).
// The exponent next to it (__expf = v_exp_f32) is a 1-2 ulp approximation already.
__device__ __forceinline__ float sigmoidf_(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
__device__ __forceinline__ float tanhf_(float x) { return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __expf(2.0f * x)); }
// 1 / sqrt(var + eps) of BatchNorm1d (eps = 1e-5), forward and backward of the decoder kernels use the same function
__device__ __forceinline__ float bn_invstd_(float var) { return __builtin_amdgcn_rsqf(var + 1e-5f); }

// sum over the 16 lanes that share lane>>4 (i.e. over j = lane & 15)
// Sum over the 16 lanes of a DPP row (= the 16 batch rows of an MFMA tile), result in every lane.  Same pairwise tree
// as xor-shuffles 1, 2, 4, 8 (after the first two steps a quad is uniform, so the half-row / full-row mirrors pair the
// same partial sums as xor 4 / xor 8) but on the DPP path: four VALU ops instead of four LDS-crossbar round trips.
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float reduce16(float v) {
  v += dpp_mov<0xB1>(v);     // quad_perm [1,0,3,2]
  v += dpp_mov<0x4E>(v);     // quad_perm [2,3,0,1]
  v += dpp_mov<0x141>(v);    // row_half_mirror
  v += dpp_mov<0x140>(v);    // row_mirror
  return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// ---- custom_loss (train_eval/train_seq2seq.py:40-88) gradient, in the one form every kernel that produces it uses (misc.hip's
// custom_loss kernels and the persistent rollouts, which fold the loss in): bitwise identical results by construction.
//   d loss / d y[t,b,d] = c1 sign(y_t - tgt_t) + c2 sign(y_t - y_{t-1}) - c2 sign(y_{t+1} - y_t) - c3 y_t / ||y[:,b,d]||_2
// The three signs travel as one code byte: bits 0-1 sign(y_t - tgt_t) + 1, bits 2-3 sign(y_t - y_{t-1}) + 1 (1 at t = 0),
// bits 4-5 sign(y_{t+1} - y_t) + 1 (1 at t = T-1); bit 6 is free for the carrier (the rollouts keep the Dropout(0.95) flag there).
__device__ __forceinline__ int loss_sign_code(float x) { return (x > 0.f) ? 2 : ((x < 0.f) ? 0 : 1); }
__device__ __forceinline__ float loss_grad_const(float c1, float c2, int code) {
  const float s1 = (float)((code & 3) - 1), sa = (float)(((code >> 2) & 3) - 1), sb = (float)(((code >> 4) & 3) - 1);
  return __fsub_rn(__fadd_rn(__fmul_rn(c1, s1), __fmul_rn(c2, sa)), __fmul_rn(c2, sb));
}
// cn = c3 / ||y[:,b,d]||_2 (0 for a zero column), as one rounded product c3 * (1 / norm)
__device__ __forceinline__ float loss_col_coef(float c3, float sumsq, float& norm) {
  norm = sqrtf(sumsq);
  return __fmul_rn(c3, (norm > 0.f) ? 1.0f / norm : 0.f);
}
__device__ __forceinline__ float loss_grad(float gconst, float cn, float v) { return fmaf(-cn, v, gconst); }
// terms (5 floats: total, l1, cont, var, mse) from the four grand sums (sum |y - tgt|, sum |y_t - y_{t-1}|, sum of column norms,
// sum (y - tgt)^2)
__device__ __forceinline__ void loss_terms_write(float* __restrict__ terms, float s_l1, float s_cont, float s_norm, float s_sq,
                                                 float c1, float c2, float c3, float inv_n) {
  const float l1 = s_l1 * c1, cont = s_cont * c2, var = -s_norm * c3;
  terms[0] = l1 + cont + var;
  terms[1] = l1;
  terms[2] = cont;
  terms[3] = var;
  terms[4] = s_sq * inv_n;   // plain MSE (evaluate_testset's metric)
}

// Stage a [16][K] row tile into LDS [16][ldx] with zero padding up to Kp columns.
// rows >= nrows_valid are zero.  256 threads.
__device__ __forceinline__ void stage_rows(float* Xs, int ldx, int Kp, const float* __restrict__ src, int64_t ld,
                                           int nrows_valid, int K, int tid, int nthreads) {
  const int total = 16 * Kp;
  for (int e = tid; e < total; e += nthreads) {
    const int r = e / Kp, k = e - r * Kp;
    float v = 0.f;
    if (r < nrows_valid && k < K) v = src[(int64_t)r * ld + k];
    Xs[r * ldx + k] = v;
  }
}


// sum of `nsplit` slabs of n floats, in a fixed order (deterministic), optionally on top of `out`.
// 256 threads = 64 consecutive elements x 4 split-groups; each group keeps 4 independent accumulators so that
// 16 loads per element are in flight (the slabs were just written by another kernel: every load is an L2/MALL miss).
static __global__ __launch_bounds__(256) void slab_reduce_kernel(const float* __restrict__ slab, int nsplit, int64_t n,
                                                                 float* __restrict__ out, int accumulate) {
  __shared__ float red[4][64];
  const int col = threadIdx.x & 63, grp = threadIdx.x >> 6;
  const int64_t e = (int64_t)blockIdx.x * 64 + col;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (e < n) {
    int p = grp;
    for (; p + 12 < nsplit; p += 16) {
      s0 += slab[(int64_t)p * n + e];
      s1 += slab[(int64_t)(p + 4) * n + e];
      s2 += slab[(int64_t)(p + 8) * n + e];
      s3 += slab[(int64_t)(p + 12) * n + e];
    }
    for (; p < nsplit; p += 4) s0 += slab[(int64_t)p * n + e];
  }
  red[grp][col] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (grp == 0 && e < n) {
    const float t = (red[0][col] + red[1][col]) + (red[2][col] + red[3][col]);
    out[e] = accumulate ? out[e] + t : t;
  }
}
static inline void launch_slab_reduce(const float* slab, int nsplit, int64_t n, float* out, int accumulate, hipStream_t st) {
  hipLaunchKernelGGL(slab_reduce_kernel, dim3(cdiv(n, 64)), dim3(256), 0, st, slab, nsplit, n, out, accumulate);
}

// Column sums of a [nblk][ncol] array of per-workgroup partials into LDS `out[ncol]`, by all 256 threads, in a fixed
// order.  The partials were written by OTHER CUs in the previous launch, so every load is an L2 miss (~1500 cycles):
// the point is to have (almost) all of a thread's loads in flight at once.  ncol % 4 == 0: float4 columns, up to
// 8 row segments, 16 loads in flight per thread; else a scalar fallback.
// scratch: LDS, `scratch_floats` (>= 1024) floats: more of it = more row segments.  Ends with a __syncthreads(); out is valid for every thread afterwards.
template <int NTHR = 256>
__device__ __forceinline__ void reduce_partials(const float* __restrict__ part, int nblk, int ncol, float* out,
                                                float* scratch, int tid, int scratch_floats = 1024) {
  if ((ncol & 3) == 0 && ncol <= 1024) {
    const int nc4 = ncol >> 2;                       // float4 columns
    for (int c0 = 0; c0 < nc4; c0 += NTHR) {
      const int nc = min(NTHR, nc4 - c0);
      const int nseg = max(1, min(NTHR, scratch_floats >> 2) / nc);
      const int seg = tid / nc, col = tid - seg * nc;
      const int per = (nblk + nseg - 1) / nseg;
      float4 tot = make_float4(0.f, 0.f, 0.f, 0.f);
      if (seg < nseg) {
        const int kb = seg * per, ke = min(nblk, kb + per);
        const float4* p = reinterpret_cast<const float4*>(part) + c0 + col;
        float4 s[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) s[j] = make_float4(0.f, 0.f, 0.f, 0.f);
        int k = kb;
        for (; k + 16 <= ke; k += 16) {
          float4 v[16];
#pragma unroll
          for (int j = 0; j < 16; ++j) v[j] = p[(int64_t)(k + j) * nc4];
#pragma unroll
          for (int j = 0; j < 16; ++j) {
            s[j & 3].x += v[j].x; s[j & 3].y += v[j].y; s[j & 3].z += v[j].z; s[j & 3].w += v[j].w;
          }
        }
        for (; k < ke; ++k) {
          const float4 v = p[(int64_t)k * nc4];
          s[0].x += v.x; s[0].y += v.y; s[0].z += v.z; s[0].w += v.w;
        }
        tot.x = (s[0].x + s[1].x) + (s[2].x + s[3].x);
        tot.y = (s[0].y + s[1].y) + (s[2].y + s[3].y);
        tot.z = (s[0].z + s[1].z) + (s[2].z + s[3].z);
        tot.w = (s[0].w + s[1].w) + (s[2].w + s[3].w);
      }
      lds_barrier();
      if (seg < nseg) reinterpret_cast<float4*>(scratch)[seg * nc + col] = tot;
      lds_barrier();
      for (int e = tid; e < 4 * nc; e += NTHR) {
        float t = 0.f;
        for (int g = 0; g < nseg; ++g) t += scratch[g * 4 * nc + e];
        out[4 * c0 + e] = t;
      }
    }
    lds_barrier();
    return;
  }
  for (int c0 = 0; c0 < ncol; c0 += NTHR) {
    const int nc = min(NTHR, ncol - c0);
    const int nseg = max(1, min(NTHR, scratch_floats) / nc);
    const int seg = tid / nc, col = tid - seg * nc;
    const int per = (nblk + nseg - 1) / nseg;
    float tot = 0.f;
    if (seg < nseg) {
      const int kb = seg * per, ke = min(nblk, kb + per);
      const float* p = part + c0 + col;
      float s[16];
#pragma unroll
      for (int j = 0; j < 16; ++j) s[j] = 0.f;
      int k = kb;
      for (; k + 16 <= ke; k += 16) {
#pragma unroll
        for (int j = 0; j < 16; ++j) s[j] += p[(int64_t)(k + j) * ncol];
      }
      for (; k < ke; ++k) s[0] += p[(int64_t)k * ncol];
#pragma unroll
      for (int j = 0; j < 8; ++j) s[j] += s[j + 8];
      tot = ((s[0] + s[1]) + (s[2] + s[3])) + ((s[4] + s[5]) + (s[6] + s[7]));
    }
    lds_barrier();             // scratch may still be read from the previous chunk
    if (seg < nseg) scratch[seg * nc + col] = tot;
    lds_barrier();
    if (tid < nc) {
      float t = 0.f;
      for (int g = 0; g < nseg; ++g) t += scratch[g * nc + tid];
      out[c0 + tid] = t;
    }
  }
  lds_barrier();
}

// reduce_partials for ncol = 128 (32 float4 columns x 8 row segments, every thread active) with a hook: `between()`
// runs after the first 16 loads of every thread are in flight and before they are consumed, so independent work
// (e.g. the hidden-side GRU products of a decoder step) executes inside the L2 round trip of the partials.
// `between` may contain workgroup barriers: every thread calls it exactly once, from straight-line code.
template <class Between>
__device__ __forceinline__ void reduce_partials_128_hook(const float* __restrict__ part, int nblk, float* out, float* scratch,
                                                         int tid, Between between) {
  constexpr int nc4 = 32, nseg = 8;
  const int seg = tid >> 5, col = tid & 31;
  const int per = (nblk + nseg - 1) / nseg;
  const int kb = seg * per, ke = min(nblk, kb + per);
  const float4* p = reinterpret_cast<const float4*>(part) + col;
  float4 s[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) s[j] = make_float4(0.f, 0.f, 0.f, 0.f);
  float4 v[16], v2[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    const int k = kb + j;
    v[j] = p[(int64_t)(k < ke ? k : 0) * nc4];      // clamped address: no branch around the load
  }
  __builtin_amdgcn_sched_barrier(0);                // the loads stay ahead of the hook (hipcc would sink them below it)
  // The hook receives `issue2(j)`, j = 0..15: the second batch of 16 rows, to be requested one at a time from INSIDE the
  // hook's MFMA stream (a second burst after the hook would expose a full L2 round trip).  It must call each j once.
  between([&](int j) {
    const int k = kb + 16 + j;
    v2[j] = p[(int64_t)(k < ke ? k : 0) * nc4];
  });
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    const bool ok = kb + j < ke;
    s[j & 3].x += ok ? v[j].x : 0.f; s[j & 3].y += ok ? v[j].y : 0.f;
    s[j & 3].z += ok ? v[j].z : 0.f; s[j & 3].w += ok ? v[j].w : 0.f;
  }
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    const bool ok = kb + 16 + j < ke;
    s[j & 3].x += ok ? v2[j].x : 0.f; s[j & 3].y += ok ? v2[j].y : 0.f;
    s[j & 3].z += ok ? v2[j].z : 0.f; s[j & 3].w += ok ? v2[j].w : 0.f;
  }
  for (int k = kb + 32; k < ke; ++k) {              // nblk > 256: the remainder, one by one
    const float4 w = p[(int64_t)k * nc4];
    s[0].x += w.x; s[0].y += w.y; s[0].z += w.z; s[0].w += w.w;
  }
  float4 tot;
  tot.x = (s[0].x + s[1].x) + (s[2].x + s[3].x);
  tot.y = (s[0].y + s[1].y) + (s[2].y + s[3].y);
  tot.z = (s[0].z + s[1].z) + (s[2].z + s[3].z);
  tot.w = (s[0].w + s[1].w) + (s[2].w + s[3].w);
  lds_barrier();
  reinterpret_cast<float4*>(scratch)[seg * nc4 + col] = tot;
  lds_barrier();
  for (int e = tid; e < 4 * nc4; e += 256) {
    float t = 0.f;
    for (int g = 0; g < nseg; ++g) t += scratch[g * 4 * nc4 + e];
    out[e] = t;
  }
  lds_barrier();
}

static __global__ void transpose_kernel(const float* __restrict__ in, float* __restrict__ out, int rows, int cols) {
  __shared__ float tile[32][33];
  const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
  for (int j = threadIdx.y; j < 32; j += blockDim.y) {
    const int r = r0 + j, c = c0 + threadIdx.x;
    if (r < rows && c < cols) tile[j][threadIdx.x] = in[(int64_t)r * cols + c];
  }
  __syncthreads();
  for (int j = threadIdx.y; j < 32; j += blockDim.y) {
    const int c = c0 + j, r = r0 + threadIdx.x;
    if (r < rows && c < cols) out[(int64_t)c * rows + r] = tile[threadIdx.x][j];
  }
}
static inline void launch_transpose(const float* in, float* out, int rows, int cols, hipStream_t st) {
  hipLaunchKernelGGL(transpose_kernel, dim3(cdiv(cols, 32), cdiv(rows, 32)), dim3(32, 8), 0, st, in, out, rows, cols);
}

}  // namespace g2v
