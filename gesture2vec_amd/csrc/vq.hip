// vq.hip -- EMA vector quantiser kernels (K1..K5') for gfx950.
//
// Replaces VQ_Payam_EMA.forward and its autograd (model/Autoencoder_VQVAE_model.py:1217-1296),
// VQ_Payam (:1114-1173) and the code-assignment call sites (lmdb_data_loader.py:1274-1281,
// Clustering.py:151-157).
//
//   vq_assign_kernel   d = ||x||^2 + ||W||^2 - 2 x W^T on v_mfma_f32_16x16x4_f32 (exact fp32 fma
//                      chains, same algebraic form as :1234-1238 so that near-ties round alike),
//                      running argmin in registers (lowest index wins ties, = torch.argmin),
//                      q = W[idx], straight-through output z + (q - z), per-block SSE partial.
//                      16 rows per workgroup, the K codes are split over the 4 waves; the codebook
//                      (K*E*4 B, 256 KB at K=512,E=128) is streamed from L2 as MFMA A-fragments.
//   vq_stats_kernel    cnt / dw as a one-hot^T x flat contraction on MFMA, the one-hot generated
//                      from idx on the fly, split over rows into slabs that are summed in order
//                      (deterministic; no float atomics, no contention when the codebook collapses).
//   vq_ema_*           K4 (EMA + Laplace smoothing + codebook refresh) and the loss / perplexity scalars.
//   vq_bwd_kernel      K5' straight-through + commitment gradient.
#include "common.hpp"

#ifdef G2V_VQSTAMPS       // diagnostic build only (gpurun_tools/vqstamps.py): shader-clock stamps of four workgroups
__device__ unsigned long long g2v_vqstamps[4 * 32];
#define VSTAMP(k)                                                                                                        \
  do {                                                                                                                   \
    const int sb_ = blockIdx.x == 0 ? 0 : (blockIdx.x == 5 ? 1 : (blockIdx.x == 128 ? 2 : (blockIdx.x == 255 ? 3 : -1))); \
    if (threadIdx.x == 0 && sb_ >= 0) g2v_vqstamps[sb_ * 32 + (k)] = __builtin_amdgcn_s_memtime();                        \
  } while (0)
extern "C" int g2v_read_vqstamps(unsigned long long* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g2v_vqstamps), sizeof(unsigned long long) * 128);
}
#define VSTAMP_B(k)                                                                                                      \
  do {                                                                                                                   \
    const int sb_ = blockIdx.x == 0 ? 0 : (blockIdx.x == 5 ? 1 : (blockIdx.x == 128 ? 2 : (blockIdx.x == 255 ? 3 : -1))); \
    if (threadIdx.x == 256 && sb_ >= 0) g2v_vqstamps[sb_ * 32 + 16 + (k)] = __builtin_amdgcn_s_memtime();                  \
  } while (0)
#else
#define VSTAMP(k)
#define VSTAMP_B(k)
#endif

namespace g2v {

constexpr int VQ_ROWS = 16;
static inline bool ptr_aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

__global__ void code_sqnorm_kernel(const float* __restrict__ W, float* __restrict__ out, int K, int E) {
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
  if (wave >= K) return;
  const float* w = W + (int64_t)wave * E;
  float s = 0.f;
  for (int e = lane; e < E; e += 64) s += w[e] * w[e];
  s = wave_sum(s);
  if (lane == 0) out[wave] = s;
}

// torch.argmin semantics (the reference's :1259): lowest index among equal minima; a NaN distance counts as smaller than every
// number and the FIRST NaN wins.  A candidate therefore beats the incumbent when it is smaller, or when it is NaN and the
// incumbent is not; on a tie (equal values, or both NaN) the lower index stays.  No sentinel index ever leaves a kernel:
// the running pair starts at (+inf, code 0), so an all-+inf row resolves to code 0 exactly like torch.
__device__ __forceinline__ bool argmin_better(float d2, float d) { return d2 < d || (d2 != d2 && d == d); }
__device__ __forceinline__ void argmin_merge(float& d, int& k, float d2, int k2) {
  const bool tie = (d2 == d) || (d2 != d2 && d != d);
  if (argmin_better(d2, d) || (tie && k2 < k)) {
    d = d2;
    k = k2;
  }
}

__global__ __launch_bounds__(256) void vq_assign_kernel(const float* __restrict__ flat, const float* __restrict__ z,
                                                        const float* __restrict__ W, const float* __restrict__ wsq,
                                                        int64_t* __restrict__ idx_out, float* __restrict__ quant,
                                                        float* __restrict__ dist_min, float* __restrict__ sse_partial,
                                                        int N, int E, int K) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int Ep = (E + 15) & ~15, ldx = Ep + 4;
  float* Xs = smem;                       // [16][ldx]
  float* xx = Xs + VQ_ROWS * ldx;         // [16]
  float* wbest_d = xx + 16;               // [4][16]
  int* wbest_k = (int*)(wbest_d + 64);    // [4][16]
  int* best_k = wbest_k + 64;             // [16]
  float* red = (float*)(best_k + 16);     // [4]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r0 = blockIdx.x * VQ_ROWS;
  const int nrows = min(VQ_ROWS, N - r0);
  stage_rows(Xs, ldx, Ep, flat + (int64_t)r0 * E, E, nrows, E, tid, 256);
  __syncthreads();
  {  // ||x||^2 per row: 16 threads per row
    const int row = tid >> 4, part = tid & 15;
    float s = 0.f;
    for (int k = part; k < E; k += 16) s += Xs[row * ldx + k] * Xs[row * ldx + k];
    s = reduce16(s);
    if (part == 0) xx[row] = s;
  }
  __syncthreads();

  const int i = lane & 15, q = lane >> 4;
  const bool wvec = ptr_vec_ok(W, E);
  const float xr = xx[i];
  float bd = INFINITY;
  int bk = 0;
  const int ntile = (K + 15) >> 4;
  auto take = [&](const f32x4& acc, int kt) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int code = 16 * kt + 4 * q + r;
      if (code < K) {
        const float d = (xr + wsq[code]) - 2.0f * acc[r];   // (||x||^2 + ||W||^2) - 2 x.W  (:1234-1238)
        if (argmin_better(d, bd)) {
          bd = d;
          bk = code;
        }
      }
    }
  };
  int kt = wave;
  // two whole code tiles per pass (kt and kt + 4, taken in that order: the lane's codes stay ascending): two independent
  // accumulator chains instead of one 100-long dependent MFMA chain per tile at E = 400 (the reference's own shape, 128 rows:
  // 57 -> 50 us; what remains is eight workgroups each streaming the whole 0.8 MB codebook -- a split over the codes with a
  // second-stage argmin is the next step for small N)
  for (; 16 * (kt + 4) + 16 <= K; kt += 8) {
    f32x4 acc[2] = {(f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}};
    wave_gemm<2>(acc, W, (int64_t)E, wvec, 16 * kt, 64, 16, E, Xs, ldx, lane);
    take(acc[0], kt);
    take(acc[1], kt + 4);
  }
  for (; kt < ntile; kt += 4) {
    f32x4 acc[1] = {(f32x4){0.f, 0.f, 0.f, 0.f}};
    const int nvalid = min(16, K - 16 * kt);
    wave_gemm<1>(acc, W, (int64_t)E, wvec, 16 * kt, 16, nvalid, E, Xs, ldx, lane);
    take(acc[0], kt);
  }
  // merge the 4 lanes (q = 0..3) that hold the same row
  {
    float d2 = __shfl_xor(bd, 16);
    int k2 = __shfl_xor(bk, 16);
    argmin_merge(bd, bk, d2, k2);
    d2 = __shfl_xor(bd, 32);
    k2 = __shfl_xor(bk, 32);
    argmin_merge(bd, bk, d2, k2);
  }
  if (lane < 16) {
    wbest_d[wave * 16 + lane] = bd;
    wbest_k[wave * 16 + lane] = bk;
  }
  __syncthreads();
  if (tid < 16) {
    float d = wbest_d[tid];
    int k = wbest_k[tid];
#pragma unroll
    for (int w = 1; w < 4; ++w) argmin_merge(d, k, wbest_d[w * 16 + tid], wbest_k[w * 16 + tid]);
    best_k[tid] = k;
    if (tid < nrows) {
      idx_out[r0 + tid] = (int64_t)k;
      if (dist_min) dist_min[r0 + tid] = d;
    }
  }
  __syncthreads();
  if (quant) {
    const int row = tid >> 4, part = tid & 15;
    float sse = 0.f;
    if (row < nrows) {
      const float* wq = W + (int64_t)best_k[row] * E;
      const float* zr = z + (int64_t)(r0 + row) * E;
      float* qo = quant + (int64_t)(r0 + row) * E;
      for (int k = part; k < E; k += 16) {
        const float zv = zr[k];
        const float diff = wq[k] - zv;
        qo[k] = zv + diff;                 // inputs + (quantized - inputs).detach()  (:1292)
        sse += diff * diff;
      }
    }
    sse = wave_sum(sse);
    if (lane == 0) red[wave] = sse;
    __syncthreads();
    if (tid == 0 && sse_partial) sse_partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
  }
}

// ---- few rows (the reference's own batch: 128 latent rows at E = 400): the codes are split over gridDim.y workgroups per row tile
// and the row minima meet in a 64-bit atomicMin on (order-preserving image of the distance, code index) -- torch.argmin's rule:
// lowest distance, lowest index among equals, a NaN distance beats every number and the first NaN wins (image 0).  The keys live
// in the idx array itself (set to all ones in front of the launch); vq_assign_finish_kernel turns them into indices and does the
// gather / straight-through / SSE part.  Eight workgroups sweeping the whole codebook each took 50-57 us.
__device__ __forceinline__ unsigned long long argmin_key(float d, int code) {
  unsigned hi = 0u;                                  // NaN
  if (d == d) {
    unsigned u = (d == 0.f) ? 0u : __float_as_uint(d);            // -0 and +0 are one value
    hi = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
  }
  return ((unsigned long long)hi << 32) | (unsigned)code;
}
__device__ __forceinline__ float argmin_key_value(unsigned long long key) {
  const unsigned hi = (unsigned)(key >> 32);
  if (hi == 0u) return __uint_as_float(0x7fc00000u);
  return __uint_as_float((hi & 0x80000000u) ? (hi & 0x7fffffffu) : ~hi);
}

__global__ __launch_bounds__(256) void vq_assign_split_kernel(const float* __restrict__ flat, const float* __restrict__ W,
                                                              const float* __restrict__ wsq, unsigned long long* __restrict__ keys,
                                                              int N, int E, int K, int tiles_per_split) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int Ep = (E + 15) & ~15, ldx = Ep + 4;
  float* Xs = smem;                       // [16][ldx]
  float* xx = Xs + VQ_ROWS * ldx;         // [16]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r0 = blockIdx.x * VQ_ROWS;
  const int nrows = min(VQ_ROWS, N - r0);
  stage_rows(Xs, ldx, Ep, flat + (int64_t)r0 * E, E, nrows, E, tid, 256);
  __syncthreads();
  {  // ||x||^2 per row: 16 threads per row (the arithmetic of vq_assign_kernel)
    const int row = tid >> 4, part = tid & 15;
    float s2 = 0.f;
    for (int k = part; k < E; k += 16) s2 += Xs[row * ldx + k] * Xs[row * ldx + k];
    s2 = reduce16(s2);
    if (part == 0) xx[row] = s2;
  }
  __syncthreads();
  const int i = lane & 15, q = lane >> 4;
  const bool wvec = ptr_vec_ok(W, E);
  const float xr = xx[i];
  float bd = INFINITY;
  int bk = 0;
  const int ntile = (K + 15) >> 4;
  const int t0 = blockIdx.y * tiles_per_split, t1 = min(ntile, t0 + tiles_per_split);
  for (int kt = t0 + wave; kt < t1; kt += 4) {
    f32x4 acc[1] = {(f32x4){0.f, 0.f, 0.f, 0.f}};
    const int nvalid = min(16, K - 16 * kt);
    wave_gemm<1>(acc, W, (int64_t)E, wvec, 16 * kt, 16, nvalid, E, Xs, ldx, lane);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int code = 16 * kt + 4 * q + r;
      if (code < K) {
        const float d = (xr + wsq[code]) - 2.0f * acc[0][r];
        if (argmin_better(d, bd)) {
          bd = d;
          bk = code;
        }
      }
    }
  }
  {
    float d2 = __shfl_xor(bd, 16);
    int k2 = __shfl_xor(bk, 16);
    argmin_merge(bd, bk, d2, k2);
    d2 = __shfl_xor(bd, 32);
    k2 = __shfl_xor(bk, 32);
    argmin_merge(bd, bk, d2, k2);
  }
  // (a wave without a tile in this split, or an all-+inf slice, offers (+inf, its first code or 0): +inf never beats a real
  // candidate, and among all-+inf rows code 0 -- offered by split 0 -- is the lowest index, as torch resolves it)
  const bool has_tile = t0 + wave < t1;
  if (lane < nrows && lane < 16 && (has_tile || (blockIdx.y == 0 && wave == 0))) atomicMin(&keys[r0 + lane], argmin_key(bd, bk));
}

__global__ __launch_bounds__(256) void vq_assign_finish_kernel(unsigned long long* __restrict__ keys_idx, const float* __restrict__ z,
                                                               const float* __restrict__ W, float* __restrict__ quant,
                                                               float* __restrict__ dist_min, float* __restrict__ sse_partial,
                                                               int N, int E) {
  __shared__ int best_k[VQ_ROWS];
  __shared__ float red[4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r0 = blockIdx.x * VQ_ROWS;
  const int nrows = min(VQ_ROWS, N - r0);
  if (tid < nrows) {
    const unsigned long long key = keys_idx[r0 + tid];
    const int k = (int)(unsigned)(key & 0xffffffffull);
    best_k[tid] = k;
    keys_idx[r0 + tid] = (unsigned long long)k;            // the int64 index, in place
    if (dist_min) dist_min[r0 + tid] = argmin_key_value(key);
  }
  __syncthreads();
  if (quant) {
    const int row = tid >> 4, part = tid & 15;
    float sse = 0.f;
    if (row < nrows) {
      const float* wq = W + (int64_t)best_k[row] * E;
      const float* zr = z + (int64_t)(r0 + row) * E;
      float* qo = quant + (int64_t)(r0 + row) * E;
      for (int k = part; k < E; k += 16) {
        const float zv = zr[k];
        const float diff = wq[k] - zv;
        qo[k] = zv + diff;                 // inputs + (quantized - inputs).detach()  (:1292)
        sse += diff * diff;
      }
    }
    sse = wave_sum(sse);
    if (lane == 0) red[wave] = sse;
    __syncthreads();
    if (tid == 0 && sse_partial) sse_partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
  }
}

// ---- any E % 16 == 0, K % 16 == 0 (round 5: the reference's OWN shapes, E = H L = 400 with K = 512 / 400 -- every shipped YAML;
// until now they took vq_assign_kernel above: 48.9 us at N = 4096, 0.22 of the fp32 matrix peak) --------------------------------
// vq_assign_kernel's arithmetic bit for bit (the same row norms, one accumulator chain per code tile in the same k order, the
// same merges: idx, dist_min, quantized and the SSE partials are bitwise equal), restructured around the weight stream:
//  * the codebook comes as the fragment-major image of g2v_vq_pack_codebook: one coalesced 1 KiB block per (code tile, k-step)
//    instead of 16 rows x 64 B -- and a PD-deep ring of them per wave (wave_gemm_p_rows' whole-group / tail form: the wait-count
//    pass waits for the oldest slot only),
//  * EIGHT waves (two per SIMD: twice the bytes in flight, one wave's MFMAs under the other's waits), NT = 4 code tiles per pass:
//    K = 512 is ONE pass per wave, one ring prologue per wave,
//  * NR row tiles per workgroup share every fragment (bulk assignment: NR = 2 / 4 where row tiles outnumber the CUs).
// Per-SIMD MFMA floor at N = 4096, E = 400, K = 512: 800 x 32 cycles = 11.6 us.
template <int NT, int NR, int PD = 8>
__device__ __forceinline__ void wave_gemm_pt_rows(f32x4 (&acc)[NT][NR], const float* const (&pt)[NT], int KS, const float* Xs, int ldx,
                                                  int lane) {
  const int i = lane & 15, q = lane >> 4;
  const float* xrow = Xs + i * ldx + 4 * q;
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
    for (int r = 0; r < NR; ++r) {
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
      const int sn = min(s + PD, KS - 1);
#pragma unroll
      for (int t = 0; t < NT; ++t) ring[j][t] = *reinterpret_cast<const float4*>(pt[t] + (int64_t)sn * 256);
    }
    __builtin_amdgcn_sched_barrier(0);
  };
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

template <int NR, int NT>
__global__ __launch_bounds__(512) void vq_assign_p_kernel(const float* __restrict__ flat, const float* __restrict__ z,
                                                          const float* __restrict__ W, const float* __restrict__ Wf,
                                                          const float* __restrict__ wsq, int64_t* __restrict__ idx_out,
                                                          float* __restrict__ quant, float* __restrict__ dist_min,
                                                          float* __restrict__ sse_partial, int N, int E, int K) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int ROWS = 16 * NR, NW = 8;
  const int ldx = E + 4, KS = E >> 4, ntile = K >> 4;
  float* Xs = smem;                             // [ROWS][ldx]
  float* xx = Xs + ROWS * ldx;                  // [ROWS]
  float* wbest_d = xx + ROWS;                   // [NW][ROWS]
  int* wbest_k = (int*)(wbest_d + NW * ROWS);   // [NW][ROWS]
  int* best_k = wbest_k + NW * ROWS;            // [ROWS]
  float* red = (float*)(best_k + ROWS);         // [NW]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r0 = blockIdx.x * ROWS;
  const int nrows = min(ROWS, N - r0);
  const int E4 = E >> 2;
  for (int e = tid; e < ROWS * E4; e += 512) {
    const int r = e / E4, c = 4 * (e - r * E4);
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (r < nrows) v = *reinterpret_cast<const float4*>(flat + (int64_t)(r0 + r) * E + c);
    *reinterpret_cast<float4*>(Xs + r * ldx + c) = v;
  }
  __syncthreads();
  for (int rb = 0; rb < ROWS; rb += 32) {  // ||x||^2 per row: 16 threads per row, vq_assign_kernel's summation order
    const int row = rb + (tid >> 4), part = tid & 15;
    float sacc = 0.f;
    if (row < ROWS)
      for (int k = part; k < E; k += 16) sacc += Xs[row * ldx + k] * Xs[row * ldx + k];
    sacc = reduce16(sacc);
    if (part == 0 && row < ROWS) xx[row] = sacc;
  }
  __syncthreads();
  const int i = lane & 15, q = lane >> 4;
  float xr[NR], bd[NR];
  int bk[NR];
#pragma unroll
  for (int r = 0; r < NR; ++r) {
    xr[r] = xx[16 * r + i];
    bd[r] = INFINITY;
    bk[r] = 0;
  }
  for (int base = wave; base < ntile; base += NW * NT) {
    const float* pt[NT];
    int tile[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      tile[t] = base + NW * t;
      const int tc = tile[t] < ntile ? tile[t] : base;      // past the last tile: a valid tile again, its result unused
      pt[t] = Wf + ((int64_t)tc * KS * 64 + lane) * 4;
    }
    float4 wq[NT];                                           // ||W||^2 of this lane's four codes per tile, ahead of the product
#pragma unroll
    for (int t = 0; t < NT; ++t) wq[t] = *reinterpret_cast<const float4*>(wsq + 16 * (tile[t] < ntile ? tile[t] : base) + 4 * q);
    f32x4 acc[NT][NR];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int r = 0; r < NR; ++r) acc[t][r] = (f32x4){0.f, 0.f, 0.f, 0.f};
    wave_gemm_pt_rows<NT, NR>(acc, pt, KS, Xs, ldx, lane);
#pragma unroll
    for (int t = 0; t < NT; ++t) {                           // tiles ascending: the lane's codes stay in increasing order
      if (tile[t] >= ntile) continue;
      const float sq[4] = {wq[t].x, wq[t].y, wq[t].z, wq[t].w};
#pragma unroll
      for (int r = 0; r < NR; ++r)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float d = (xr[r] + sq[e]) - 2.0f * acc[t][r][e];   // (||x||^2 + ||W||^2) - 2 x.W  (:1234-1238)
          if (argmin_better(d, bd[r])) {
            bd[r] = d;
            bk[r] = 16 * tile[t] + 4 * q + e;
          }
        }
    }
  }
#pragma unroll
  for (int r = 0; r < NR; ++r) {     // the 4 lanes (q = 0..3) that hold the same row, then the 8 waves
    float d2 = __shfl_xor(bd[r], 16);
    int k2 = __shfl_xor(bk[r], 16);
    argmin_merge(bd[r], bk[r], d2, k2);
    d2 = __shfl_xor(bd[r], 32);
    k2 = __shfl_xor(bk[r], 32);
    argmin_merge(bd[r], bk[r], d2, k2);
    if (lane < 16) {
      wbest_d[wave * ROWS + 16 * r + lane] = bd[r];
      wbest_k[wave * ROWS + 16 * r + lane] = bk[r];
    }
  }
  __syncthreads();
  if (tid < ROWS) {
    float d = wbest_d[tid];
    int k = wbest_k[tid];
#pragma unroll
    for (int w = 1; w < NW; ++w) argmin_merge(d, k, wbest_d[w * ROWS + tid], wbest_k[w * ROWS + tid]);
    best_k[tid] = k;
    if (tid < nrows) {
      idx_out[r0 + tid] = (int64_t)k;
      if (dist_min) dist_min[r0 + tid] = d;
    }
  }
  __syncthreads();
  if (quant) {
    // gather / straight-through / SSE: per 16-row block exactly vq_assign_kernel's (16 threads per row, 256 threads per block,
    // its wave_sum and four-way add): two blocks of 16 rows at a time with 512 threads
    for (int rb = 0; rb < ROWS; rb += 32) {
      const int half = tid >> 8, t2 = tid & 255, w4 = t2 >> 6;
      const int row = rb + 16 * half + (t2 >> 4), part = t2 & 15;
      float sse = 0.f;
      if (row < nrows && row < ROWS) {
        const float* wq2 = W + (int64_t)best_k[row] * E;
        const float* zr = z + (int64_t)(r0 + row) * E;
        float* qo = quant + (int64_t)(r0 + row) * E;
        for (int k = part; k < E; k += 16) {
          const float zv = zr[k];
          const float diff = wq2[k] - zv;
          qo[k] = zv + diff;                 // inputs + (quantized - inputs).detach()  (:1292)
          sse += diff * diff;
        }
      }
      sse = wave_sum(sse);
      __syncthreads();
      if (lane == 0) red[wave] = sse;
      __syncthreads();
      const int blk = (r0 + rb) / 16 + half;
      if (t2 == 0 && sse_partial && rb + 16 * half < ROWS && r0 + rb + 16 * half < N)
        sse_partial[blk] = (red[4 * half] + red[4 * half + 1]) + (red[4 * half + 2] + red[4 * half + 3]);
      (void)w4;
    }
  }
}

// Fast path for E == 128 and K % 128 == 0 (the BASELINE shape E = 128, K = 512).  Same arithmetic, but:
//  * the 16 x E row tile's MFMA B-fragments (8 x float4) live in registers for the whole kernel,
//  * each wave walks its code tiles in PAIRS (two independent accumulator chains: the 16x16x4 fp32 MFMA has a
//    40-cycle dependent latency against a 32-cycle issue slot), and the NEXT pair's 16 fragment loads are issued
//    before the current pair's 64 MFMAs, so the L2 latency of the codebook stream hides behind ~2k cycles of math,
//  * codebook rows are read straight from the row-major (K,E) array: per k-step a wave touches 16 rows x 64 B and
//    the neighbouring k-step consumes the other half of each 128-byte line (L1 hit).
template <int E>
__global__ __launch_bounds__(256) void vq_assign_fast_kernel(const float* __restrict__ flat, const float* __restrict__ z,
                                                             const float* __restrict__ W, const float* __restrict__ wsq,
                                                             int64_t* __restrict__ idx_out, float* __restrict__ quant,
                                                             float* __restrict__ dist_min, float* __restrict__ sse_partial,
                                                             int N, int K) {
  constexpr int KS = E / 16, ldx = E + 4;
  __shared__ __attribute__((aligned(16))) float Xs[VQ_ROWS * ldx];
  __shared__ float xx[16];
  __shared__ float wbest_d[64];
  __shared__ int wbest_k[64];
  __shared__ int best_k[16];
  __shared__ float red[4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r0 = blockIdx.x * VQ_ROWS;
  const int nrows = min(VQ_ROWS, N - r0);
  const int i = lane & 15, q = lane >> 4;
  {  // stage the row tile (coalesced float4) and ||x||^2
    const int row = tid >> 4, part = tid & 15;   // 16 threads per row, 2 float4 each (E = 128)
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < E / 64; ++j) {
      const int c = 4 * (part + 16 * j);
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (row < nrows) v = *reinterpret_cast<const float4*>(flat + (int64_t)(r0 + row) * E + c);
      *reinterpret_cast<float4*>(Xs + row * ldx + c) = v;
      s += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
    }
    s = reduce16(s);
    if (part == 0) xx[row] = s;
  }
  __syncthreads();
  float4 xb[KS];
#pragma unroll
  for (int s = 0; s < KS; ++s) xb[s] = *reinterpret_cast<const float4*>(Xs + i * ldx + 16 * s + 4 * q);
  const float xr = xx[i];
  float bd = INFINITY;
  int bk = 0;
  const int npair = K >> 7;   // tiles of this wave: kt = wave + 4 j ; pairs (j, j+1)  -> K / 16 / 4 / 2
  auto load_pair = [&](int p, float4 (&w0)[KS], float4 (&w1)[KS], float4& q0, float4& q1) {
    const int kt0 = wave + 8 * p, kt1 = kt0 + 4;
    const float* r0p = W + (int64_t)(16 * kt0 + i) * E + 4 * q;
    const float* r1p = W + (int64_t)(16 * kt1 + i) * E + 4 * q;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      w0[s] = *reinterpret_cast<const float4*>(r0p + 16 * s);
      w1[s] = *reinterpret_cast<const float4*>(r1p + 16 * s);
    }
    q0 = *reinterpret_cast<const float4*>(wsq + 16 * kt0 + 4 * q);
    q1 = *reinterpret_cast<const float4*>(wsq + 16 * kt1 + 4 * q);
  };
  float4 wa0[KS], wa1[KS], wb0[KS], wb1[KS], qa0, qa1, qb0, qb1;
  load_pair(0, wa0, wa1, qa0, qa1);
  // consume pair p from (w0, w1) while the fragments of pair pn stream into (n0, n1) BETWEEN the MFMAs: a burst of 18
  // fragment loads in front of the 64 MFMAs would cost the wave ~1.5k cycles of vector-memory issue with an idle matrix
  // pipe; two loads per k-step (8 MFMAs = 256 cycles) are absorbed at the rate the CU's memory front end accepts them.
  auto consume = [&](int p, const float4 (&w0)[KS], const float4 (&w1)[KS], const float4& q0, const float4& q1, bool ld,
                     int pn, float4 (&n0)[KS], float4 (&n1)[KS], float4& nq0, float4& nq1) {
    f32x4 a0 = (f32x4){0.f, 0.f, 0.f, 0.f}, a1 = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int kt0 = wave + 8 * pn, kt1 = kt0 + 4;
    const float* r0p = W + (int64_t)(16 * kt0 + i) * E + 4 * q;
    const float* r1p = W + (int64_t)(16 * kt1 + i) * E + 4 * q;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      a0 = mfma16(w0[s].x, xb[s].x, a0); a1 = mfma16(w1[s].x, xb[s].x, a1);
      a0 = mfma16(w0[s].y, xb[s].y, a0); a1 = mfma16(w1[s].y, xb[s].y, a1);
      if (ld) n0[s] = *reinterpret_cast<const float4*>(r0p + 16 * s);
      __builtin_amdgcn_sched_barrier(0);
      a0 = mfma16(w0[s].z, xb[s].z, a0); a1 = mfma16(w1[s].z, xb[s].z, a1);
      a0 = mfma16(w0[s].w, xb[s].w, a0); a1 = mfma16(w1[s].w, xb[s].w, a1);
      if (ld) n1[s] = *reinterpret_cast<const float4*>(r1p + 16 * s);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (ld) {
      nq0 = *reinterpret_cast<const float4*>(wsq + 16 * kt0 + 4 * q);
      nq1 = *reinterpret_cast<const float4*>(wsq + 16 * kt1 + 4 * q);
    }
    const int c0 = 16 * (wave + 8 * p) + 4 * q, c1 = c0 + 64;
    const float s0[4] = {q0.x, q0.y, q0.z, q0.w}, s1[4] = {q1.x, q1.y, q1.z, q1.w};
#pragma unroll
    for (int r = 0; r < 4; ++r) {      // codes in increasing order within the lane: strict '<' keeps the lowest index
      const float d = (xr + s0[r]) - 2.0f * a0[r];
      if (argmin_better(d, bd)) { bd = d; bk = c0 + r; }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float d = (xr + s1[r]) - 2.0f * a1[r];
      if (argmin_better(d, bd)) { bd = d; bk = c1 + r; }
    }
  };
  for (int p = 0; p < npair; p += 2) {
    consume(p, wa0, wa1, qa0, qa1, p + 1 < npair, p + 1, wb0, wb1, qb0, qb1);
    if (p + 1 < npair) consume(p + 1, wb0, wb1, qb0, qb1, p + 2 < npair, p + 2, wa0, wa1, qa0, qa1);
  }
  {
    float d2 = __shfl_xor(bd, 16);
    int k2 = __shfl_xor(bk, 16);
    argmin_merge(bd, bk, d2, k2);
    d2 = __shfl_xor(bd, 32);
    k2 = __shfl_xor(bk, 32);
    argmin_merge(bd, bk, d2, k2);
  }
  if (lane < 16) {
    wbest_d[wave * 16 + lane] = bd;
    wbest_k[wave * 16 + lane] = bk;
  }
  __syncthreads();
  if (tid < 16) {
    float d = wbest_d[tid];
    int k = wbest_k[tid];
#pragma unroll
    for (int w = 1; w < 4; ++w) argmin_merge(d, k, wbest_d[w * 16 + tid], wbest_k[w * 16 + tid]);
    best_k[tid] = k;
    if (tid < nrows) {
      idx_out[r0 + tid] = (int64_t)k;
      if (dist_min) dist_min[r0 + tid] = d;
    }
  }
  __syncthreads();
  if (quant) {
    const int row = tid >> 4, part = tid & 15;
    float sse = 0.f;
    if (row < nrows) {
      const float* wq = W + (int64_t)best_k[row] * E;
      const float* zr = z + (int64_t)(r0 + row) * E;
      float* qo = quant + (int64_t)(r0 + row) * E;
#pragma unroll
      for (int j = 0; j < E / 64; ++j) {
        const int c = 4 * (part + 16 * j);
        const float4 zv = *reinterpret_cast<const float4*>(zr + c), wv = *reinterpret_cast<const float4*>(wq + c);
        const float4 df = make_float4(wv.x - zv.x, wv.y - zv.y, wv.z - zv.z, wv.w - zv.w);
        *reinterpret_cast<float4*>(qo + c) = make_float4(zv.x + df.x, zv.y + df.y, zv.z + df.z, zv.w + df.w);   // :1292
        sse += df.x * df.x + df.y * df.y + df.z * df.z + df.w * df.w;
      }
    }
    sse = wave_sum(sse);
    if (lane == 0) red[wave] = sse;
    __syncthreads();
    if (tid == 0 && sse_partial) sse_partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
  }
}

// Row-tiled variant: RT row tiles (16*RT rows) per workgroup.  Every codebook fragment a wave pulls from L2 now feeds
// RT MFMAs (one per row tile) instead of one, so the L2 traffic of the codebook stream drops RT-fold.  At N = 4096
// the 16-rows-per-workgroup kernel above is L2-bandwidth-bound (256 workgroups x 256 KB = 64 MB per launch); RT = 2
// halves that at the price of filling only half the CUs; for bulk assignment (N >= 16384) RT = 4 is MFMA-bound.
// LIST: the rows to assign are row_list[0 .. *row_count) (the exact re-check of vq_bx3_sweep_kernel's undecided rows); the
// launch is sized for the worst case and workgroups beyond the count leave at once.  idx only (no quant / dist / SSE).
template <int E, int RT, bool LIST = false>
__global__ __launch_bounds__(256) void vq_assign_rt_kernel(const float* __restrict__ flat, const float* __restrict__ z,
                                                           const float* __restrict__ W, const float* __restrict__ wsq,
                                                           int64_t* __restrict__ idx_out, float* __restrict__ quant,
                                                           float* __restrict__ dist_min, float* __restrict__ sse_partial,
                                                           int N, int K, const int* __restrict__ row_list = nullptr,
                                                           const int* __restrict__ row_count = nullptr) {
  constexpr int KS = E / 16, ldx = E + 4, ROWS = 16 * RT;
  __shared__ __attribute__((aligned(16))) float Xs[ROWS * ldx];
  __shared__ float xx[ROWS];
  __shared__ float wbest_d[4 * ROWS];
  __shared__ int wbest_k[4 * ROWS];
  __shared__ int best_k[ROWS];
  __shared__ float red[4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (LIST) N = min(N, *row_count);
  // LIST (round 6): a bounded grid walks the list in strides -- the launch used to be sized for N rows, 16 k workgroups of which
  // ~150 had work, 64 rows each: one round of 30 us; with 16 rows per workgroup the same rows spread over the chip
  for (int r0 = blockIdx.x * ROWS; r0 < N; r0 += (int)gridDim.x * ROWS) {
  const int nrows = min(ROWS, N - r0);
  const int i = lane & 15, q = lane >> 4;
  for (int e = tid; e < ROWS * (E / 4); e += 256) {     // coalesced float4 staging
    const int row = e / (E / 4), c = 4 * (e - row * (E / 4));
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (row < nrows) v = *reinterpret_cast<const float4*>(flat + (int64_t)(LIST ? row_list[r0 + row] : r0 + row) * E + c);
    *reinterpret_cast<float4*>(Xs + row * ldx + c) = v;
  }
  __syncthreads();
  for (int row = tid >> 4; row < ROWS; row += 16) {     // ||x||^2: 16 threads per row
    const int part = tid & 15;
    float s = 0.f;
    for (int k = part; k < E; k += 16) s += Xs[row * ldx + k] * Xs[row * ldx + k];
    s = reduce16(s);
    if (part == 0) xx[row] = s;
  }
  __syncthreads();
  float xr[RT], bd[RT];
  int bk[RT];
#pragma unroll
  for (int t = 0; t < RT; ++t) { xr[t] = xx[16 * t + i]; bd[t] = INFINITY; bk[t] = 0; }
  const int ntw = K >> 6;     // code tiles of this wave: kt = wave + 4 j
  auto load_tile = [&](int j, float4 (&w)[KS], float4& sq) {
    const int kt = wave + 4 * j;
    const float* rp = W + (int64_t)(16 * kt + i) * E + 4 * q;
#pragma unroll
    for (int s = 0; s < KS; ++s) w[s] = *reinterpret_cast<const float4*>(rp + 16 * s);
    sq = *reinterpret_cast<const float4*>(wsq + 16 * kt + 4 * q);
  };
  // consume tile j from w while the fragments of tile jn stream into wn, one load per k-step BETWEEN the MFMAs
  auto consume = [&](int j, const float4 (&w)[KS], const float4& sq, bool ld, int jn, float4 (&wn)[KS], float4& sqn) {
    f32x4 acc[RT];
#pragma unroll
    for (int t = 0; t < RT; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int ktn = wave + 4 * jn;
    const float* rpn = W + (int64_t)(16 * ktn + i) * E + 4 * q;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      float4 xb[RT];
#pragma unroll
      for (int t = 0; t < RT; ++t) xb[t] = *reinterpret_cast<const float4*>(Xs + (16 * t + i) * ldx + 16 * s + 4 * q);
#pragma unroll
      for (int t = 0; t < RT; ++t) acc[t] = mfma16(w[s].x, xb[t].x, acc[t]);
#pragma unroll
      for (int t = 0; t < RT; ++t) acc[t] = mfma16(w[s].y, xb[t].y, acc[t]);
      if (ld) wn[s] = *reinterpret_cast<const float4*>(rpn + 16 * s);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int t = 0; t < RT; ++t) acc[t] = mfma16(w[s].z, xb[t].z, acc[t]);
#pragma unroll
      for (int t = 0; t < RT; ++t) acc[t] = mfma16(w[s].w, xb[t].w, acc[t]);
    }
    if (ld) sqn = *reinterpret_cast<const float4*>(wsq + 16 * ktn + 4 * q);
    const int c0 = 16 * (wave + 4 * j) + 4 * q;
    const float sv[4] = {sq.x, sq.y, sq.z, sq.w};
#pragma unroll
    for (int t = 0; t < RT; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float d = (xr[t] + sv[r]) - 2.0f * acc[t][r];
        if (argmin_better(d, bd[t])) { bd[t] = d; bk[t] = c0 + r; }
      }
  };
  float4 wa[KS], wb[KS], qa, qb;
  load_tile(0, wa, qa);
  for (int j = 0; j < ntw; j += 2) {
    consume(j, wa, qa, j + 1 < ntw, j + 1, wb, qb);
    if (j + 1 < ntw) consume(j + 1, wb, qb, j + 2 < ntw, j + 2, wa, qa);
  }
#pragma unroll
  for (int t = 0; t < RT; ++t) {
    float d2 = __shfl_xor(bd[t], 16);
    int k2 = __shfl_xor(bk[t], 16);
    argmin_merge(bd[t], bk[t], d2, k2);
    d2 = __shfl_xor(bd[t], 32);
    k2 = __shfl_xor(bk[t], 32);
    argmin_merge(bd[t], bk[t], d2, k2);
    if (lane < 16) {
      wbest_d[wave * ROWS + 16 * t + lane] = bd[t];
      wbest_k[wave * ROWS + 16 * t + lane] = bk[t];
    }
  }
  __syncthreads();
  if (tid < ROWS) {
    float d = wbest_d[tid];
    int k = wbest_k[tid];
#pragma unroll
    for (int w = 1; w < 4; ++w) argmin_merge(d, k, wbest_d[w * ROWS + tid], wbest_k[w * ROWS + tid]);
    best_k[tid] = k;
    if (tid < nrows) {
      idx_out[LIST ? row_list[r0 + tid] : r0 + tid] = (int64_t)k;
      if (dist_min) dist_min[r0 + tid] = d;
    }
  }
  if (LIST) {
    __syncthreads();          // the staging arrays are reused by the next stride
    continue;
  }
  __syncthreads();
  if (quant) {
    float sse = 0.f;
    for (int e = tid; e < ROWS * (E / 4); e += 256) {
      const int row = e / (E / 4), c = 4 * (e - row * (E / 4));
      if (row < nrows) {
        const float4 zv = *reinterpret_cast<const float4*>(z + (int64_t)(r0 + row) * E + c);
        const float4 wv = *reinterpret_cast<const float4*>(W + (int64_t)best_k[row] * E + c);
        const float4 df = make_float4(wv.x - zv.x, wv.y - zv.y, wv.z - zv.z, wv.w - zv.w);
        *reinterpret_cast<float4*>(quant + (int64_t)(r0 + row) * E + c) =
            make_float4(zv.x + df.x, zv.y + df.y, zv.z + df.z, zv.w + df.w);   // :1292
        sse += df.x * df.x + df.y * df.y + df.z * df.z + df.w * df.w;
      }
    }
    sse = wave_sum(sse);
    if (lane == 0) red[wave] = sse;
    __syncthreads();
    if (tid == 0 && sse_partial) sse_partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
  }
  break;                     // (one 16 RT-row block per workgroup unless LIST)
  }
}


// ||x||^2 of a projected row as the fused kernels form it (shared so that vq_fused_assign_kernel and vq_fused_bx_kernel agree
// bitwise): a lane holding 4 consecutive features adds their squares as an fma chain, the four lanes (q) that hold one row of a
// 16-feature tile add up through xor-16 / xor-32 shuffles, and the 8 tile partials of a row are summed pairwise in a fixed order.
__device__ __forceinline__ float sq4_chain(float x, float y, float z, float w) { return fmaf(w, w, fmaf(z, z, fmaf(y, y, x * x))); }
__device__ __forceinline__ float tile_partial_sumsq(float x, float y, float z, float w) {
  float p = sq4_chain(x, y, z, w);
  p += __shfl_xor(p, 16);
  p += __shfl_xor(p, 32);
  return p;
}
__device__ __forceinline__ float sum8_partials(const float* xxp, int row) {        // xxp[tile][16]
  return ((xxp[row] + xxp[16 + row]) + (xxp[32 + row] + xxp[48 + row])) + ((xxp[64 + row] + xxp[80 + row]) + (xxp[96 + row] + xxp[112 + row]));
}

// ---- fused pre_linear + assign for E == 128 (the BASELINE shape) -------------------------------------------------
// One workgroup = 16 rows of z.  Phase 1: flat = z W_pre^T + b (:1230) on MFMA (wave w owns output tiles w and w + 4; the
// 16 weight fragments of a wave are requested up front, the first codebook pair right behind them).  The projected tile
// goes to LDS (it IS the B operand of the distance contraction) and to global (`flat_out`: the code statistics need it).
// Phase 2 = vq_assign_fast_kernel's loop on that tile: distances, running argmin, gather + straight-through + SSE on raw z.
// Replaces the launch sequence  gemm_nt_stream (pre_linear, 18 us) -> assign (12 us)  and one 2 MB round trip of `flat`.
// PACKED: the codebook's MFMA A fragments are read from a fragment-major image Wf (g2v_vq_pack_codebook: tile, k-step, lane ->
// 16 bytes), one contiguous 1 KB run per wave-level load = 8 full cache lines; straight from the row-major matrix the same
// load touches 16 rows x 64 bytes = half of each of 16 lines.  Same values, same arithmetic; the gather of the chosen code
// still reads the row-major matrix.
template <int E, bool PACKED>
__global__ __launch_bounds__(256) void vq_fused_assign_kernel(const float* __restrict__ z, const float* __restrict__ Wp,
                                                              const float* __restrict__ bp, const float* __restrict__ W,
                                                              const float* __restrict__ Wf,
                                                              const float* __restrict__ wsq, float* __restrict__ flat_out,
                                                              int64_t* __restrict__ idx_out, float* __restrict__ quant,
                                                              float* __restrict__ sse_partial, int N, int K) {
  constexpr int KS = E / 16, ldx = E + 4;
  __shared__ __attribute__((aligned(16))) float Xz[VQ_ROWS * ldx];
  __shared__ __attribute__((aligned(16))) float Xf[VQ_ROWS * ldx];
  __shared__ float xx[16];
  __shared__ float xxp[8 * 16];
  __shared__ float wbest_d[64];
  __shared__ int wbest_k[64];
  __shared__ int best_k[16];
  __shared__ float red[4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r0 = blockIdx.x * VQ_ROWS;
  const int nrows = min(VQ_ROWS, N - r0);
  const int i = lane & 15, q = lane >> 4;
  VSTAMP(0);
  // request order = consumption order (vmcnt retires in order): the raw row tile first, then the pre_linear fragments, then
  // the first pair of codebook tiles
  float4 zv[E / 64];
  {
    const int row = tid >> 4, part = tid & 15;
#pragma unroll
    for (int j = 0; j < E / 64; ++j) {
      const int rr = row < nrows ? row : 0;
      zv[j] = *reinterpret_cast<const float4*>(z + (int64_t)(r0 + rr) * E + 4 * (part + 16 * j));
    }
  }
  // pre_linear weight fragments of this wave (tiles wave, wave + 4), straight from the row-major (E,E) matrix
  float4 wp0[KS], wp1[KS];
  {
    const float* p0 = Wp + (int64_t)(16 * wave + i) * E + 4 * q;
    const float* p1 = Wp + (int64_t)(16 * (wave + 4) + i) * E + 4 * q;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      wp0[s] = *reinterpret_cast<const float4*>(p0 + 16 * s);
      wp1[s] = *reinterpret_cast<const float4*>(p1 + 16 * s);
    }
  }
  // ... and the first pair of codebook tiles right behind them: they travel during the staging and the projection
  auto load_pair = [&](int p, float4 (&w0)[KS], float4 (&w1)[KS], float4& q0, float4& q1) {
    const int kt0 = wave + 8 * p, kt1 = kt0 + 4;
    constexpr int FS = PACKED ? 256 : 16;      // floats between the fragments of consecutive k-steps
    const float* r0p = PACKED ? Wf + (int64_t)(kt0 * KS) * 256 + lane * 4 : W + (int64_t)(16 * kt0 + i) * E + 4 * q;
    const float* r1p = PACKED ? Wf + (int64_t)(kt1 * KS) * 256 + lane * 4 : W + (int64_t)(16 * kt1 + i) * E + 4 * q;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      w0[s] = *reinterpret_cast<const float4*>(r0p + FS * s);
      w1[s] = *reinterpret_cast<const float4*>(r1p + FS * s);
    }
    q0 = *reinterpret_cast<const float4*>(wsq + 16 * kt0 + 4 * q);
    q1 = *reinterpret_cast<const float4*>(wsq + 16 * kt1 + 4 * q);
  };
  float4 wa0[KS], wa1[KS], wb0[KS], wb1[KS], qa0, qa1, qb0, qb1;
  load_pair(0, wa0, wa1, qa0, qa1);
  __builtin_amdgcn_sched_barrier(0);
  {  // stage the raw row tile
    const int row = tid >> 4, part = tid & 15;
#pragma unroll
    for (int j = 0; j < E / 64; ++j)
      *reinterpret_cast<float4*>(Xz + row * ldx + 4 * (part + 16 * j)) = row < nrows ? zv[j] : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  lds_barrier();
  VSTAMP(1);
  {  // phase 1: the projection
    f32x4 a0 = (f32x4){0.f, 0.f, 0.f, 0.f}, a1 = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const float4 xb = *reinterpret_cast<const float4*>(Xz + i * ldx + 16 * s + 4 * q);
      a0 = mfma16(wp0[s].x, xb.x, a0); a1 = mfma16(wp1[s].x, xb.x, a1);
      a0 = mfma16(wp0[s].y, xb.y, a0); a1 = mfma16(wp1[s].y, xb.y, a1);
      a0 = mfma16(wp0[s].z, xb.z, a0); a1 = mfma16(wp1[s].z, xb.z, a1);
      a0 = mfma16(wp0[s].w, xb.w, a0); a1 = mfma16(wp1[s].w, xb.w, a1);
    }
    const int f0 = 16 * wave + 4 * q, f1 = f0 + 64;
    const float4 b0 = *reinterpret_cast<const float4*>(bp + f0), b1 = *reinterpret_cast<const float4*>(bp + f1);
    const float4 v0 = make_float4(a0[0] + b0.x, a0[1] + b0.y, a0[2] + b0.z, a0[3] + b0.w);
    const float4 v1 = make_float4(a1[0] + b1.x, a1[1] + b1.y, a1[2] + b1.z, a1[3] + b1.w);
    *reinterpret_cast<float4*>(Xf + i * ldx + f0) = v0;
    *reinterpret_cast<float4*>(Xf + i * ldx + f1) = v1;
    const float p0 = tile_partial_sumsq(v0.x, v0.y, v0.z, v0.w), p1 = tile_partial_sumsq(v1.x, v1.y, v1.z, v1.w);
    if (lane < 16) {
      xxp[wave * 16 + lane] = p0;
      xxp[(wave + 4) * 16 + lane] = p1;
    }
    if (i < nrows) {
      *reinterpret_cast<float4*>(flat_out + (int64_t)(r0 + i) * E + f0) = v0;
      *reinterpret_cast<float4*>(flat_out + (int64_t)(r0 + i) * E + f1) = v1;
    }
  }
  lds_barrier();
  VSTAMP(2);
  if (tid < 16) xx[tid] = sum8_partials(xxp, tid);       // ||x||^2 of the projected rows
  lds_barrier();
  VSTAMP(3);
  float4 xb[KS];
#pragma unroll
  for (int s = 0; s < KS; ++s) xb[s] = *reinterpret_cast<const float4*>(Xf + i * ldx + 16 * s + 4 * q);
  const float xr = xx[i];
  float bd = INFINITY;
  int bk = 0;
  const int npair = K >> 7;
  auto consume = [&](int p, const float4 (&w0)[KS], const float4 (&w1)[KS], const float4& q0, const float4& q1, bool ld,
                     int pn, float4 (&n0)[KS], float4 (&n1)[KS], float4& nq0, float4& nq1) {
    f32x4 a0 = (f32x4){0.f, 0.f, 0.f, 0.f}, a1 = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int kt0 = wave + 8 * pn, kt1 = kt0 + 4;
    constexpr int FS = PACKED ? 256 : 16;
    const float* r0p = PACKED ? Wf + (int64_t)(kt0 * KS) * 256 + lane * 4 : W + (int64_t)(16 * kt0 + i) * E + 4 * q;
    const float* r1p = PACKED ? Wf + (int64_t)(kt1 * KS) * 256 + lane * 4 : W + (int64_t)(16 * kt1 + i) * E + 4 * q;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      a0 = mfma16(w0[s].x, xb[s].x, a0); a1 = mfma16(w1[s].x, xb[s].x, a1);
      a0 = mfma16(w0[s].y, xb[s].y, a0); a1 = mfma16(w1[s].y, xb[s].y, a1);
      if (ld) n0[s] = *reinterpret_cast<const float4*>(r0p + FS * s);
      __builtin_amdgcn_sched_barrier(0);
      a0 = mfma16(w0[s].z, xb[s].z, a0); a1 = mfma16(w1[s].z, xb[s].z, a1);
      a0 = mfma16(w0[s].w, xb[s].w, a0); a1 = mfma16(w1[s].w, xb[s].w, a1);
      if (ld) n1[s] = *reinterpret_cast<const float4*>(r1p + FS * s);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (ld) {
      nq0 = *reinterpret_cast<const float4*>(wsq + 16 * kt0 + 4 * q);
      nq1 = *reinterpret_cast<const float4*>(wsq + 16 * kt1 + 4 * q);
    }
    const int c0 = 16 * (wave + 8 * p) + 4 * q, c1 = c0 + 64;
    const float s0[4] = {q0.x, q0.y, q0.z, q0.w}, s1[4] = {q1.x, q1.y, q1.z, q1.w};
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float d = (xr + s0[r]) - 2.0f * a0[r];
      if (argmin_better(d, bd)) { bd = d; bk = c0 + r; }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float d = (xr + s1[r]) - 2.0f * a1[r];
      if (argmin_better(d, bd)) { bd = d; bk = c1 + r; }
    }
  };
  for (int p = 0; p < npair; p += 2) {
    consume(p, wa0, wa1, qa0, qa1, p + 1 < npair, p + 1, wb0, wb1, qb0, qb1);
    if (p + 1 < npair) consume(p + 1, wb0, wb1, qb0, qb1, p + 2 < npair, p + 2, wa0, wa1, qa0, qa1);
  }
  VSTAMP(4);
  {
    float d2 = __shfl_xor(bd, 16);
    int k2 = __shfl_xor(bk, 16);
    argmin_merge(bd, bk, d2, k2);
    d2 = __shfl_xor(bd, 32);
    k2 = __shfl_xor(bk, 32);
    argmin_merge(bd, bk, d2, k2);
  }
  if (lane < 16) {
    wbest_d[wave * 16 + lane] = bd;
    wbest_k[wave * 16 + lane] = bk;
  }
  __syncthreads();
  if (tid < 16) {
    float d = wbest_d[tid];
    int k = wbest_k[tid];
#pragma unroll
    for (int w = 1; w < 4; ++w) argmin_merge(d, k, wbest_d[w * 16 + tid], wbest_k[w * 16 + tid]);
    best_k[tid] = k;
    if (tid < nrows) idx_out[r0 + tid] = (int64_t)k;
  }
  __syncthreads();
  VSTAMP(5);
  {
    const int row = tid >> 4, part = tid & 15;
    float sse = 0.f;
    if (row < nrows) {
      const float* wq = W + (int64_t)best_k[row] * E;
      float* qo = quant + (int64_t)(r0 + row) * E;
#pragma unroll
      for (int j = 0; j < E / 64; ++j) {
        const int c = 4 * (part + 16 * j);
        const float4 zv = *reinterpret_cast<const float4*>(Xz + row * ldx + c), wv = *reinterpret_cast<const float4*>(wq + c);
        const float4 df = make_float4(wv.x - zv.x, wv.y - zv.y, wv.z - zv.z, wv.w - zv.w);
        *reinterpret_cast<float4*>(qo + c) = make_float4(zv.x + df.x, zv.y + df.y, zv.z + df.z, zv.w + df.w);   // :1292
        sse += df.x * df.x + df.y * df.y + df.z * df.z + df.w * df.w;
      }
    }
    sse = wave_sum(sse);
    if (lane == 0) red[wave] = sse;
    __syncthreads();
    if (tid == 0 && sse_partial) sse_partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
  }
  VSTAMP(6);
}

// ---- K3, tile-owner form: one workgroup OWNS a 16-code x 16-column tile of dw (and, for column tile 0, the 16 counts) and
// contracts over ALL rows: onehot(idx)^T (16 codes x rows) * flat (rows x 16 columns) on MFMA, the four waves taking a
// quarter of the rows each, one in-workgroup sum at the end.  No slabs, no second launch, deterministic; cost independent
// of how the codes are distributed (a collapsed codebook sends every row to one code).  N % 16 == 0, E % 16 == 0, K % 16 == 0.
__global__ __launch_bounds__(256) void vq_stats_owner_kernel(const int64_t* __restrict__ idx, const float* __restrict__ flat,
                                                             float* __restrict__ cnt, float* __restrict__ dw, int N, int E,
                                                             int K) {
  __shared__ float part[4][4][64];     // [wave][r][lane]
  __shared__ float partn[4][4][64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i = lane & 15, q = lane >> 4;
  // (staging the code ids in LDS -- they are half of the loop's requests, 16 useful bytes each -- was measured: 21 -> 32 us)
  // blockIdx.x = COLUMN tile: workgroups are dealt round-robin over the 8 XCDs in linear order, so with E / 16 = 8 column tiles
  // every XCD's L2 fetches one 64-byte column slice of `flat` instead of all of it (round 2: code tile on x -> each of the eight
  // L2s pulled the whole array, 17.3 MB of HBM traffic for 2.4 MB algorithmic).  Speed only, never correctness.
  const int c0 = blockIdx.y * 16, e0 = blockIdx.x * 16;
  const int mycode = c0 + i;
  const int per = ((N / 4) + 3) & ~3;                  // rows per wave (multiple of 4)
  const int mb = wave * per, me = min(N, mb + per);
  const int* idx32 = reinterpret_cast<const int*>(idx);          // low dwords of the int64 indices
  f32x4 acc = {0.f, 0.f, 0.f, 0.f}, accn = {0.f, 0.f, 0.f, 0.f};
  const bool want_cnt = (blockIdx.x == 0);
  constexpr int U = 8;                                  // groups of 4 rows per batch of requests; TWO batches in flight: the loop
  int iv[2][U];                                         // was one L2 round trip per batch (32 batches per wave = the kernel's 25 us)
  float fv[2][U];
  auto request = [&](int m0, int (&ivb)[U], float (&fvb)[U]) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int row = m0 + 4 * u + q;
      const bool ok = row < me;
      const int rc = ok ? row : mb;
      const int v = idx32[2 * (int64_t)rc];
      ivb[u] = ok ? v : -1;
      fvb[u] = flat[(int64_t)rc * E + e0 + i];
    }
  };
  auto consume = [&](const int (&ivb)[U], const float (&fvb)[U]) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const float a = (ivb[u] == mycode) ? 1.0f : 0.0f;
      acc = mfma16(a, fvb[u], acc);
      if (want_cnt) accn = mfma16(a, 1.0f, accn);
    }
  };
  request(mb, iv[0], fv[0]);
  for (int m0 = mb; m0 < me; m0 += 8 * U) {
    request(m0 + 4 * U, iv[1], fv[1]);                  // rows past `me` read a valid row and contribute nothing
    __builtin_amdgcn_sched_barrier(0);
    consume(iv[0], fv[0]);
    request(m0 + 8 * U, iv[0], fv[0]);
    __builtin_amdgcn_sched_barrier(0);
    consume(iv[1], fv[1]);
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    part[wave][r][lane] = acc[r];
    if (want_cnt) partn[wave][r][lane] = accn[r];
  }
  __syncthreads();
  if (wave == 0) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float v = (part[0][r][lane] + part[1][r][lane]) + (part[2][r][lane] + part[3][r][lane]);
      dw[(int64_t)(c0 + 4 * q + r) * E + e0 + i] = v;           // D[4q + r][j = i]: code c0 + 4q + r, column e0 + i
      if (want_cnt && i == 0)
        cnt[c0 + 4 * q + r] = (partn[0][r][lane] + partn[1][r][lane]) + (partn[2][r][lane] + partn[3][r][lane]);
    }
  }
}

// ---- K3: slab[split] = onehot(idx)^T (K x rows) * flat (rows x E); cnt via a ones column --------------
constexpr int SM = 32, SLD = 64 + 16;
__global__ __launch_bounds__(256) void vq_stats_kernel(const int64_t* __restrict__ idx, const float* __restrict__ flat,
                                                       float* __restrict__ slab_dw, float* __restrict__ slab_cnt,
                                                       int N, int E, int K, int rows_per_split) {
  __shared__ __attribute__((aligned(16))) float Xs[SM * SLD];
  __shared__ int Is[SM];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c0 = blockIdx.x * 64, e0 = blockIdx.y * 64, split = blockIdx.z;
  const int mb = split * rows_per_split, me = min(N, mb + rows_per_split);
  const int i = lane & 15, q = lane >> 4;
  f32x4 acc[4], accn = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int t = 0; t < 4; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const int mycode = c0 + 16 * wave + i;
  for (int mc = mb; mc < me; mc += SM) {
    const int c = tid & 63;
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const int r = (tid >> 6) + 4 * it, m = mc + r;
      Xs[r * SLD + c] = (m < me && e0 + c < E) ? flat[(int64_t)m * E + e0 + c] : 0.f;
    }
    if (tid < SM) Is[tid] = (mc + tid < me) ? (int)idx[mc + tid] : -1;
    __syncthreads();
#pragma unroll
    for (int s4 = 0; s4 < SM; s4 += 4) {
      const float a = (Is[s4 + q] == mycode) ? 1.0f : 0.0f;
#pragma unroll
      for (int t = 0; t < 4; ++t) acc[t] = mfma16(a, Xs[(s4 + q) * SLD + 16 * t + i], acc[t]);
      if (blockIdx.y == 0) accn = mfma16(a, 1.0f, accn);
    }
    __syncthreads();
  }
  float* sl = slab_dw + (int64_t)split * K * E;
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int e = e0 + 16 * t + i;
    if (e >= E) continue;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int code = c0 + 16 * wave + 4 * q + r;
      if (code < K) sl[(int64_t)code * E + e] = acc[t][r];
    }
  }
  if (blockIdx.y == 0 && i == 0) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int code = c0 + 16 * wave + 4 * q + r;
      if (code < K) slab_cnt[(int64_t)split * K + code] = accn[r];
    }
  }
}

static int stats_splits(int N, int E, int K) {
  const int tiles = cdiv(K, 64) * cdiv(E, 64);
  int splits = cdiv(512, tiles);
  const int max_splits = cdiv(N, 2 * SM);
  if (splits > max_splits) splits = max_splits;
  if (splits < 1) splits = 1;
  return splits;
}

// ---- K4 scalars: cluster-size EMA + Laplace smoothing, loss, perplexity (one workgroup) --------------
// 256 threads = one wave per SIMD at 32 registers: the kernel runs in the branch beside the persistent forward rollout, which
// leaves ~140 registers per lane of every SIMD free.  Round 3's 1024-thread form (4 waves per SIMD) did not fit beside it: a 5 us
// kernel that reported 294 us because it could not be placed until the rollout had drained, with the rest of its hardware
// queue waiting behind it.
// `fault`: the persistent rollouts' fault latch (dec_persist.hpp).  A latched fault means this step's statistics are garbage:
// the EMA state and the codebook stay as they were (the host raises at its next sync point and repeats the step).
__global__ __launch_bounds__(256) void vq_ema_scalars_kernel(const float* __restrict__ stats,
                                                             const float* __restrict__ sse_partial, int n_sse,
                                                             float* __restrict__ cs, float* __restrict__ scalars,
                                                             int N_loss, int N_cnt, int E, int K, float beta,
                                                             float decay, float eps, int update,
                                                             const unsigned* __restrict__ fault) {
  __shared__ float red[3][4];
  __shared__ float bc[1];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (fault && *fault != 0u) update = 0;
  // perplexity = exp(-sum p log(p + 1e-10)), p = cnt / N   (:1293-1294)
  float ent = 0.f, nsum = 0.f, sse = 0.f;
  for (int k = tid; k < K; k += 256) {
    const float cnt = stats[k];
    const float p = cnt / (float)N_cnt;
    ent += p * logf(p + 1e-10f);
    if (update) {
      const float c = cs[k] * decay + (1.0f - decay) * cnt;   // :1263-1265
      cs[k] = c;
      nsum += c;
    }
  }
  for (int j = tid; j < n_sse; j += 256) sse += sse_partial[j];
  ent = wave_sum(ent);
  nsum = wave_sum(nsum);
  sse = wave_sum(sse);
  if (lane == 0) {
    red[0][wave] = ent;
    red[1][wave] = sse;
    red[2][wave] = nsum;
  }
  __syncthreads();
  if (tid == 0) {
    scalars[1] = expf(-((red[0][0] + red[0][1]) + (red[0][2] + red[0][3])));
    scalars[0] = beta * (((red[1][0] + red[1][1]) + (red[1][2] + red[1][3])) / ((float)N_loss * (float)E));     // beta * mse(q, z)  (:1285-1289)
    bc[0] = (red[2][0] + red[2][1]) + (red[2][2] + red[2][3]);
  }
  __syncthreads();
  if (update) {
    const float n = bc[0];
    for (int k = tid; k < K; k += 256) cs[k] = (cs[k] + eps) / (n + (float)K * eps) * n;   // :1268-1273
  }
}

// ---- K4 rows: ema_w, codebook, ||W||^2 (one wave per code) -------------------------------------------
__global__ void vq_ema_rows_kernel(const float* __restrict__ stats, const float* __restrict__ cs,
                                   float* __restrict__ ema_w, float* __restrict__ W, float* __restrict__ wsq, int E,
                                   int K, float decay, const unsigned* __restrict__ fault) {
  const int code = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
  if (code >= K || (fault && *fault != 0u)) return;
  const float* dw = stats + K + (int64_t)code * E;
  float* ew = ema_w + (int64_t)code * E;
  float* w = W + (int64_t)code * E;
  const float c = cs[code];
  float s = 0.f;
  for (int e = lane; e < E; e += 64) {
    const float v = ew[e] * decay + (1.0f - decay) * dw[e];   // :1276-1278
    ew[e] = v;
    const float wv = v / c;                                    // :1280-1282
    w[e] = wv;
    s += wv * wv;
  }
  s = wave_sum(s);
  if (lane == 0) wsq[code] = s;
}

__global__ void vq_bwd_kernel(const float* __restrict__ gq, const float* __restrict__ gloss,
                              const float* __restrict__ z, const float* __restrict__ W,
                              const int64_t* __restrict__ idx, float* __restrict__ gz, int64_t total, int E,
                              float coef) {
  const float c = gloss ? gloss[0] * coef : 0.f;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t row = e / E;
    const int col = (int)(e - row * E);
    const float q = idx ? W[idx[row] * E + col] : W[e];   // idx == NULL: W is the dense (N,E) quantised tensor
    gz[e] = gq ? fmaf(c, z[e] - q, gq[e]) : c * (z[e] - q);      // (the form gru_bwd_fast_kernel folds into its d_hn load)
  }
}

// codebook gradient of the non-EMA quantiser (VQ_Payam, :1158-1162): d q_latent / dW[k] = 2/(N E) (cnt[k] W[k] - sum_{idx=k} z)
__global__ void vq_codebook_grad_kernel(const float* __restrict__ stats, const float* __restrict__ W,
                                        const float* __restrict__ gloss, float* __restrict__ gW, int K, int E, float coef) {
  const float c = gloss[0] * coef;
  const int64_t total = (int64_t)K * E;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int k = (int)(e / E);
    gW[e] = c * (stats[k] * W[e] - stats[K + e]);
  }
}


// ---- soft quantiser VQ_Payam_GSSoft (reference :1304-1438) ----------------------------------------------------------
// One wave per row n.  dots[n,k] = flat_n . W_k comes from the dense-layer kernel; here
//   d = |flat_n|^2 + |W_k|^2 - 2 dots,  smooth = 1 / exp(logvar)^2,
//   prob = exp(-(d / 400) * (0.5 * smooth)) / sqrt(smooth),  probs = prob / sum_k prob          (:1349-1372, :1396-1411)
// `dots` is overwritten with d (kept for the backward).
__global__ __launch_bounds__(256) void vq_soft_fwd_kernel(const float* __restrict__ flat, float* __restrict__ dots,
                                                          const float* __restrict__ logvar,
                                                          const float* __restrict__ wsq, float* __restrict__ probs,
                                                          int N, int E, int K) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n = blockIdx.x * 4 + wave;
  if (n >= N) return;
  float fs = 0.f;
  for (int e = lane; e < E; e += 64) {
    const float v = flat[(int64_t)n * E + e];
    fs += v * v;
  }
  fs = wave_sum(fs);
  float sum = 0.f;
  for (int k = lane; k < K; k += 64) {
    const int64_t o = (int64_t)n * K + k;
    const float d = (fs + wsq[k]) - 2.0f * dots[o];
    const float ex = expf(logvar[o]);
    const float smooth = 1.0f / (ex * ex);
    const float pr = expf(-((d / 400.0f) * (0.5f * smooth))) / sqrtf(smooth);
    dots[o] = d;
    probs[o] = pr;
    sum += pr;
  }
  sum = wave_sum(sum);
  for (int k = lane; k < K; k += 64) probs[(int64_t)n * K + k] /= sum;
}

// Backward of the map (d, logvar) -> probs for one row per wave:
//   g_k = p_k (dp_k - sum_j dp_j p_j)        [d log(prob_k)]
//   log prob_k = -d_k s_k / 800 + logvar_k,  s_k = exp(-2 logvar_k)
//   dd_k = -g_k s_k / 800 ;  dlogvar_k = g_k (1 + d_k s_k / 400) ;  rowsum[n] = sum_k dd_k
__global__ __launch_bounds__(256) void vq_soft_bwd_kernel(const float* __restrict__ probs, const float* __restrict__ dprobs,
                                                          const float* __restrict__ dist, const float* __restrict__ logvar,
                                                          float* __restrict__ dd, float* __restrict__ dlogvar,
                                                          float* __restrict__ rowsum, int N, int K) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n = blockIdx.x * 4 + wave;
  if (n >= N) return;
  float dot = 0.f;
  for (int k = lane; k < K; k += 64) dot += dprobs[(int64_t)n * K + k] * probs[(int64_t)n * K + k];
  dot = wave_sum(dot);
  float rs = 0.f;
  for (int k = lane; k < K; k += 64) {
    const int64_t o = (int64_t)n * K + k;
    const float g = probs[o] * (dprobs[o] - dot);
    const float ex = expf(logvar[o]);
    const float s_ = 1.0f / (ex * ex);
    const float v = -g * s_ / 800.0f;
    dd[o] = v;
    dlogvar[o] = g * (1.0f + dist[o] * s_ / 400.0f);
    rs += v;
  }
  rs = wave_sum(rs);
  if (lane == 0) rowsum[n] = rs;
}

// out[r,c] = 2 * A[r,c] * v[r] - 2 * T[r,c]    (the two gradients of d = |f|^2 + |W|^2 - 2 f.W)
__global__ __launch_bounds__(256) void rowscale_combine_kernel(const float* __restrict__ A, const float* __restrict__ v,
                                                               const float* __restrict__ Tm, float* __restrict__ out,
                                                               int64_t rows, int cols) {
  const int64_t total = rows * cols;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    const int64_t r = e / cols;
    out[e] = 2.0f * A[e] * v[r] - 2.0f * Tm[e];
  }
}

// perplexity of the mean assignment: avg_k = mean_n probs[n,k]; exp(-sum_k avg_k log(avg_k + 1e-10))   (:1432-1433)
// One workgroup, fixed summation order (column k is summed by thread k % 1024 over n = 0..N-1).
__global__ __launch_bounds__(1024) void vq_soft_perplexity_kernel(const float* __restrict__ probs, float* __restrict__ out,
                                                                  int N, int K) {
  __shared__ float red[16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float ent = 0.f;
  for (int k = tid; k < K; k += 1024) {
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int n = 0;
    for (; n + 3 < N; n += 4) {
      s0 += probs[(int64_t)n * K + k];
      s1 += probs[(int64_t)(n + 1) * K + k];
      s2 += probs[(int64_t)(n + 2) * K + k];
      s3 += probs[(int64_t)(n + 3) * K + k];
    }
    for (; n < N; ++n) s0 += probs[(int64_t)n * K + k];
    const float avg = ((s0 + s1) + (s2 + s3)) / (float)N;
    ent += avg * logf(avg + 1e-10f);
  }
  ent = wave_sum(ent);
  if (lane == 0) red[wave] = ent;
  __syncthreads();
  if (tid == 0) {
    float s = 0.f;
    for (int w = 0; w < 16; ++w) s += red[w];
    out[0] = expf(-s);
  }
}

// the same over the whole device: (1) block j sums its row range of every column into part[j][K] (coalesced rows, fixed order),
// (2) one workgroup sums the partials per column in block order and finishes.  Deterministic; 12 us instead of 255.
constexpr int PERP_ROWS = 16;       // rows per block of stage 1
__global__ __launch_bounds__(256) void vq_soft_colsum_kernel(const float* __restrict__ probs, float* __restrict__ part, int N, int K) {
  const int r0 = blockIdx.x * PERP_ROWS, r1 = min(N, r0 + PERP_ROWS);
  for (int k = threadIdx.x; k < K; k += 256) {
    float s = 0.f;
    for (int n = r0; n < r1; ++n) s += probs[(int64_t)n * K + k];
    part[(int64_t)blockIdx.x * K + k] = s;
  }
}
__global__ __launch_bounds__(1024) void vq_soft_perplexity_finish_kernel(const float* __restrict__ part, float* __restrict__ out,
                                                                         int nblk, int N, int K) {
  __shared__ float red[16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float ent = 0.f;
  for (int k = tid; k < K; k += 1024) {
    float s = 0.f;
    for (int j = 0; j < nblk; ++j) s += part[(int64_t)j * K + k];
    const float avg = s / (float)N;
    ent += avg * logf(avg + 1e-10f);
  }
  ent = wave_sum(ent);
  if (lane == 0) red[wave] = ent;
  __syncthreads();
  if (tid == 0) {
    float s = 0.f;
    for (int w = 0; w < 16; ++w) s += red[w];
    out[0] = expf(-s);
  }
}

// straight-through value z + (q - z) exactly as the reference evaluates it (:1431)
__global__ __launch_bounds__(256) void ste_kernel(const float* __restrict__ z, const float* __restrict__ q,
                                                  float* __restrict__ out, int64_t n) {
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (int64_t)gridDim.x * 256)
    out[e] = z[e] + (q[e] - z[e]);
}

}  // namespace g2v

using namespace g2v;

extern "C" int g2v_vq_codebook_grad(const float* stats, const float* codebook, const float* g_loss, float* g_codebook,
                                    int N, int E, int K, g2v_stream_t stream) {
  G2V_REQUIRE(stats && codebook && g_loss && g_codebook, "null pointer");
  G2V_REQUIRE(N > 0 && E > 0 && K > 0, "non-positive size");
  int blocks = cdiv((int64_t)K * E, 256);
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(vq_codebook_grad_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, stats, codebook, g_loss,
                     g_codebook, K, E, 2.0f / ((float)N * (float)E));
  G2V_CHECK_LAUNCH();
  return G2V_OK;
}

// ---- bulk code assignment on the bf16 matrix pipe, EXACT where it matters: 3-term split screening + fp32 re-check --------
// The distance contraction -2 x W^T in fp32 MFMA is the whole cost of bulk assignment (vq_assign_rt_kernel<128,4>: 64-68 % of
// the fp32 matrix peak, i.e. it cannot get faster in fp32).  gfx950's bf16 MFMA moves 16x the FLOPs per cycle, and
//     x . w  ~=  xh.wh + xh.wl + xl.wh        (xh = bf16(x), xl = bf16(x - xh); same for w)
// misses only xl.wl and the second-order residuals: |error| <= 3 * 2^-16 * |x||w| + the fp32 accumulation of 3E terms.
//   vq_bx3_split_kernel   codebook -> Wh, Wl (bf16, row-major), once per call
//   vq_bx3_sweep_kernel   per row: best code, best and SECOND-best approximate distance; a row whose two best distances are
//                         closer than margin = 2^-11 |x| max_k|w_k| (> 2x the bound above) is UNDECIDED: its id goes to a list
//   vq_assign_rt_kernel<128,4,LIST>   the undecided rows again, in exact fp32 (the kernel the training path and the tests use)
// A decided row's approximate winner beats every other code by more than twice the error bound, so it is the exact-arithmetic
// argmin as well; an undecided row gets exactly what the fp32 kernel gives.  Cost: 12 bf16 MFMAs (16 cycles) per 16 x 16 x 128
// tile instead of 32 fp32 ones (32 cycles) = 5.3x fewer matrix cycles, plus the re-check on the undecided fraction.
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;

__global__ __launch_bounds__(256) void vq_bx3_split_kernel(const float* __restrict__ W, const float* __restrict__ wsq,
                                                           __bf16* __restrict__ Wh, __bf16* __restrict__ Wl,
                                                           float* __restrict__ wn, int64_t n, int K) {
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (int64_t)gridDim.x * 256) {
    const float w = W[e];
    const __bf16 h = (__bf16)w;
    Wh[e] = h;
    Wl[e] = (__bf16)(w - (float)h);
  }
  // |w_k| (rounded up), for the PER-CODE error radius of the sweep
  for (int k = blockIdx.x * 256 + threadIdx.x; k < K; k += gridDim.x * 256) wn[k] = sqrtf(wsq[k]) * 1.001f;
}

// 256 rows per workgroup: wave w owns rows 64 w .. 64 w + 63 (4 row tiles, fragments in registers for the whole sweep) and
// walks ALL code tiles; the codebook passes through LDS in chunks of 64 codes (hi and lo images, double buffered, fetched
// from L2 once per workgroup = 1 KiB per row -- with the codebook pulled per wave, as in vq_assign_rt_kernel, this kernel is
// L2-bandwidth-bound at the fp32 kernel's speed: measured 1.28 vs 1.25 ms for 2^20 rows).
constexpr int BX3_CODES = 64, BX3_LDW = 128 + 8;       // bf16 elements per LDS row: 272 B, conflict-free ds_read_b128 per 16 lanes
// (best, second best) per row are kept as floats that carry the code index in their low mantissa bits (kbits = log2 K): one
// v_and_or packs, med3(d1, d2, p) is the new second best, min(d1, p) the new best -- 3.5 VALU ops per candidate, which fit
// into the issue slots the bf16 MFMAs leave; the 2^-(23 - kbits) relative truncation is added to the margin.
// Round 5: the margin is PER CODE.  The sweep ranks LOWER BOUNDS  L_k = e_k - r_k,  r_k = 2^-12 |x| |w_k| (the 3-term split's
// error bound, x 2 for the -2 x.w, with the same 2x allowance as before); U* = L_a + 2 r_a of the best-ranked code a bounds the
// row's minimum from above, so the row is decided iff the second-smallest lower bound clears U*.  With the global margin
// 2^-11 |x| max_k |w_k| a TRAINED codebook -- the EMA update leaves dead codes with norms hundreds of times the live ones' --
// made every row undecided (all 2^20 rows through the fp32 kernel: 1.7 ms instead of 0.49); per code those are simply far away.
__global__ __launch_bounds__(256, 2) void vq_bx3_sweep_kernel(const float* __restrict__ flat, const __bf16* __restrict__ Wh,
                                                              const __bf16* __restrict__ Wl, const float* __restrict__ wsq,
                                                              const float* __restrict__ wn,
                                                              int64_t* __restrict__ idx_out, int* __restrict__ und_list,
                                                              int* __restrict__ und_count, int N, int K, int kbits) {
  constexpr int E = 128, KB = E / 32, RT = 4;
  __shared__ __attribute__((aligned(16))) __bf16 Ls[2][2][BX3_CODES * BX3_LDW];      // [buffer][hi / lo][code][k]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, i = lane & 15, q = lane >> 4;
  const int r0 = blockIdx.x * 256 + 64 * wave;
  // ---- this wave's rows as B-operand fragments (hi / lo), straight from global; |x|^2 on the way ---------------------------
  bf16x8 xh[RT][KB], xl[RT][KB];
  float xr[RT];
#pragma unroll
  for (int t = 0; t < RT; ++t) {
    const int row = r0 + 16 * t + i;
    const float* xp = flat + (int64_t)(row < N ? row : N - 1) * E + 8 * q;
    float ss = 0.f;
#pragma unroll
    for (int s = 0; s < KB; ++s) {
      const float4 a = *reinterpret_cast<const float4*>(xp + 32 * s), b = *reinterpret_cast<const float4*>(xp + 32 * s + 4);
      const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const __bf16 h = (__bf16)v[j];
        xh[t][s][j] = h;
        xl[t][s][j] = (__bf16)(v[j] - (float)h);
        ss += v[j] * v[j];
      }
    }
    ss += __shfl_xor(ss, 16);
    ss += __shfl_xor(ss, 32);
    xr[t] = ss;
  }
  float cx[RT];                                         // 2^-12 |x|
#pragma unroll
  for (int t = 0; t < RT; ++t) cx[t] = 2.44140625e-4f * sqrtf(xr[t]);
  float d1[RT], d2[RT];
#pragma unroll
  for (int t = 0; t < RT; ++t) { d1[t] = INFINITY; d2[t] = INFINITY; }
  const unsigned kmask = (1u << kbits) - 1u;
  // ---- codebook chunks: thread -> (row tid >> 4 (+16 p), 16-byte piece tid & 15) of the hi and of the lo image -----------
  const int frow = tid >> 4, fpc = tid & 15;
  bf16x8 fh[4], fl[4];
  auto fetch = [&](int c) {
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int64_t off = (int64_t)(c * BX3_CODES + frow + 16 * p) * E + 8 * fpc;
      fh[p] = *reinterpret_cast<const bf16x8*>(Wh + off);
      fl[p] = *reinterpret_cast<const bf16x8*>(Wl + off);
    }
  };
  auto stash = [&](int buf) {
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      *reinterpret_cast<bf16x8*>(&Ls[buf][0][(frow + 16 * p) * BX3_LDW + 8 * fpc]) = fh[p];
      *reinterpret_cast<bf16x8*>(&Ls[buf][1][(frow + 16 * p) * BX3_LDW + 8 * fpc]) = fl[p];
    }
  };
  const int nch = K / BX3_CODES;
  fetch(0);
  stash(0);
  __syncthreads();
  for (int c = 0; c < nch; ++c) {
    const int buf = c & 1;
    if (c + 1 < nch) fetch(c + 1);
#pragma unroll
    for (int tl = 0; tl < BX3_CODES / 16; ++tl) {
      bf16x8 wh[KB], wl[KB];
#pragma unroll
      for (int s = 0; s < KB; ++s) {
        wh[s] = *reinterpret_cast<const bf16x8*>(&Ls[buf][0][(16 * tl + i) * BX3_LDW + 32 * s + 8 * q]);
        wl[s] = *reinterpret_cast<const bf16x8*>(&Ls[buf][1][(16 * tl + i) * BX3_LDW + 32 * s + 8 * q]);
      }
      const int c0 = c * BX3_CODES + 16 * tl + 4 * q;
      const float4 sq = *reinterpret_cast<const float4*>(wsq + c0), nq = *reinterpret_cast<const float4*>(wn + c0);
      f32x4 acc[RT];
#pragma unroll
      for (int t = 0; t < RT; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < KB; ++s) {
#pragma unroll
        for (int t = 0; t < RT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl[s], xh[t][s], acc[t], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < RT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[s], xl[t][s], acc[t], 0, 0, 0);
      }
#pragma unroll
      for (int s = 0; s < KB; ++s) {        // the large term last: the small ones are not absorbed by its rounding
#pragma unroll
        for (int t = 0; t < RT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[s], xh[t][s], acc[t], 0, 0, 0);
      }
      const float sv[4] = {sq.x, sq.y, sq.z, sq.w}, nv[4] = {nq.x, nq.y, nq.z, nq.w};
#pragma unroll
      for (int t = 0; t < RT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float d = (xr[t] + fmaf(-cx[t], nv[r], sv[r])) - 2.0f * acc[t][r];      // the LOWER bound e_k - r_k
          const float pk = __uint_as_float((__float_as_uint(d) & ~kmask) | (unsigned)(c0 + r));
          d2[t] = __builtin_amdgcn_fmed3f(d1[t], d2[t], pk);      // d1 <= d2 always: the median is the second smallest
          d1[t] = fminf(d1[t], pk);                               // a NaN candidate leaves both untouched (minNum)
        }
    }
    if (c + 1 < nch) stash(buf ^ 1);          // buffer buf ^ 1 was last read in iteration c - 1, before the barrier below
    __syncthreads();
  }
  const float trunc = ldexpf(1.0f, kbits - 22);          // 2 x the relative truncation of a packed distance
#pragma unroll
  for (int t = 0; t < RT; ++t) {
#pragma unroll
    for (int o = 16; o <= 32; o <<= 1) {                 // (best, second best) of two disjoint code sets
      const float e1 = __shfl_xor(d1[t], o), e2 = __shfl_xor(d2[t], o);
      d2[t] = fminf(fmaxf(d1[t], e1), fminf(d2[t], e2));
      d1[t] = fminf(d1[t], e1);
    }
    const int row = r0 + 16 * t + i;
    if (q == 0 && row < N) {
      const int ca = (int)(__float_as_uint(d1[t]) & kmask);
      idx_out[row] = (int64_t)ca;
      const float margin = 2.0f * cx[t] * wn[ca < K ? ca : 0] + trunc * fabsf(d2[t]);       // 2 r_a + the packing's truncation
      // NaN / inf - inf (rows with non-finite distances) compare false: undecided, the exact kernel follows torch.argmin
      if (!(d2[t] - d1[t] >= margin)) und_list[atomicAdd(und_count, 1)] = row;
    }
  }
}

// ---- round 6: the sweep again, built around what limited vq_bx3_sweep_kernel (profiles/r06_d_bulk_*) ---------------------------
// That kernel ran 560 us at 2^20 rows against 164 us of bf16 matrix-pipe time: (a) 14 vector instructions per (row, code) value
// -- with two waves per SIMD the SIMD's issue slots, not its matrix pipe, were the bound --, (b) 72 spilled registers (the
// codebook chunks passed through 32 staging registers on their way to LDS beside 128 registers of row fragments).  Here:
//   * the codebook chunks go global -> LDS by LDS-DMA (global_load_lds_dwordx4: no staging registers, no ds_write, no address
//     arithmetic in the loop); the LDS image is linear, the 16-byte pieces of a code row XOR-swizzled by the row (source address
//     and fragment read both: conflict-free ds_read_b128 without padding, which an LDS-DMA destination cannot have);
//   * everything that is constant per code or per row sits in the ACCUMULATOR'S INITIAL VALUE: acc_0 = -|w_k|^2 / 2 + (c_x / 2) |w_k|
//     (c_x = 2^-12 |x|: the per-code error radius of round 5), so that the finished accumulator is -L_k / 2 for the lower bound
//     L_k = e_k - r_k itself -- one fma per value, off the MFMA chain; the codes are ranked by LARGEST accumulator;
//   * ranking = one v_and_or (code index into the low mantissa bits), one v_med3 (second largest), one v_max: 4 instructions per
//     value in all, 64 per code tile and wave beside its 48 MFMAs.
// Decided iff acc_1 - acc_2 >= c_x |w_a| + the packing's truncation (the same rule as before, halved); everything else -- ties,
// non-finite rows -- goes to the list and gets the fp32 kernel's answer.  A codebook with a non-finite |w_k|^2 sends every row
// there (flag from the split kernel): a NaN accumulator would otherwise just lose the ranking.
__global__ __launch_bounds__(256) void vq_bulk_split_kernel(const float* __restrict__ W, const float* __restrict__ wsq,
                                                            __bf16* __restrict__ Wh, __bf16* __restrict__ Wl,
                                                            float2* __restrict__ cst, int* __restrict__ flags, int64_t n, int K) {
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (int64_t)gridDim.x * 256) {
    const float w = W[e];
    const __bf16 h = (__bf16)w;
    Wh[e] = h;
    Wl[e] = (__bf16)(w - (float)h);
  }
  for (int k = blockIdx.x * 256 + threadIdx.x; k < K; k += gridDim.x * 256) {
    const float q = wsq[k];
    cst[k] = make_float2(-0.5f * q, sqrtf(q) * 1.001f);      // {-|w_k|^2 / 2, |w_k| rounded up (the PER-CODE error radius of the sweep)}
    if (!(fabsf(q) < INFINITY)) atomicOr(&flags[1], 1);
  }
}

constexpr int BK2_CODES = 64;          // codes per LDS chunk: 16 KB per image
__global__ __launch_bounds__(256, 2) void vq_bulk_sweep_kernel(const float* __restrict__ flat, const __bf16* __restrict__ Wh,
                                                               const __bf16* __restrict__ Wl, const float2* __restrict__ cst,
                                                               int64_t* __restrict__ idx_out,
                                                               int* __restrict__ und_list, int* __restrict__ und_count, int N, int K,
                                                               int kbits) {
  constexpr int E = 128, KB = E / 32, RT = 4;
  __shared__ __attribute__((aligned(16))) __bf16 Ls[2][2][BK2_CODES * E];       // [buffer][hi / lo][code][k], pieces swizzled
  __shared__ __attribute__((aligned(16))) float Cs[2][BK2_CODES][2];            // [buffer][code]{-|w|^2 / 2, |w|}
  const int tid = threadIdx.x, lane = tid & 63, i = lane & 15, q = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r0 = blockIdx.x * 256 + 64 * wave;
  // ---- the chunk fill: LDS-DMA, 1 KiB (4 code rows) per wave-instruction; wave w moves rows 16 w .. 16 w + 15 of both images ----
  // lane (g, pos) of instruction j lands at row 4 j + g, piece position pos of the linear image and fetches logical piece pos ^ (row & 15)
  const int frow = lane >> 4, fpos = lane & 15;
  auto fill = [&](int c, int buf) {
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
      const int row = 16 * wave + 4 * jj + frow;
      const int64_t off = (int64_t)(c * BK2_CODES + row) * E + 8 * (fpos ^ (row & 15));
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(Wh + off),
                                       (__attribute__((address_space(3))) void*)&Ls[buf][0][(16 * wave + 4 * jj) * E], 16, 0, 0);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(Wl + off),
                                       (__attribute__((address_space(3))) void*)&Ls[buf][1][(16 * wave + 4 * jj) * E], 16, 0, 0);
    }
    // the chunk's constants: ONE more LDS-DMA (half a wave, 512 B).  Two of them -- one per array, as the first version had it --
    // made hipcc put an s_waitcnt vmcnt(0) between them: every wave waited for the chunk it had just requested, at the TOP of
    // the chunk it was about to compute, and no schedule of the loop below changed the kernel's time (profiles/r06_d_bulk_*)
    if (wave == 0 && lane < 32)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(cst + c * BK2_CODES + 2 * lane),
                                       (__attribute__((address_space(3))) void*)&Cs[buf][0][0], 16, 0, 0);
  };
  VSTAMP(0);
  fill(0, 0);
  // ---- this wave's rows as B-operand fragments (hi / lo), straight from global; |x|^2 on the way ---------------------------
  bf16x8 xh[RT][KB], xl[RT][KB];
  float cxh[RT];                                        // c_x / 2 = 2^-13 |x|
#pragma unroll
  for (int t = 0; t < RT; ++t) {
    const int row = r0 + 16 * t + i;
    const float* xp = flat + (int64_t)(row < N ? row : N - 1) * E + 8 * q;
    float ss = 0.f;
#pragma unroll
    for (int s = 0; s < KB; ++s) {
      const float4 a = *reinterpret_cast<const float4*>(xp + 32 * s), b = *reinterpret_cast<const float4*>(xp + 32 * s + 4);
      const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const __bf16 h = (__bf16)v[j];
        xh[t][s][j] = h;
        xl[t][s][j] = (__bf16)(v[j] - (float)h);
        ss += v[j] * v[j];
      }
    }
    ss += __shfl_xor(ss, 16);
    ss += __shfl_xor(ss, 32);
    cxh[t] = 1.220703125e-4f * sqrtf(ss);
  }
  float d1[RT], d2[RT];                                 // largest and second-largest packed accumulator per row (this lane's codes)
#pragma unroll
  for (int t = 0; t < RT; ++t) { d1[t] = -INFINITY; d2[t] = -INFINITY; }
  const unsigned kmask = (1u << kbits) - 1u;
  const int nch = K / BK2_CODES;
  VSTAMP(1);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  VSTAMP(2);
  // ---- the sweep, software-pipelined by hand (three versions, profiles/r06_d_bulk_*): -------------------------------------------
  // a COLUMN = one 16-code tile against the wave's four row tiles: 48 MFMAs in four k-blocks of 12 (wl.xh, wh.xl, wh.xh per row
  // tile), on accumulator set `cur`.  In the shadow of k-block s the wave (a) fetches the fragments of k-block s + 1 (of the next
  // column's k-block 0 behind the last) into the other fragment register set -- they are needed 12 MFMAs later, so the LDS latency
  // that the first version paid in front of every column (41 % of wave time in s_waitcnt) is hidden --, (b) RANKS row tile s of
  // the column BEFORE this one (accumulator set `prev`: 4 v_and_or + 8 v_med3; max(a, b) = med3(a, b, +inf), the plain fmaxf costs a
  // NaN-quieting v_max per operand) and (c) writes the NEXT column's initial values into that freed accumulator (4 fma): 16
  // vector instructions per 12 MFMAs, one behind each MFMA -- an MFMA owns the SIMD's issue port for 8 of its 16 cycles.
  f32x4 acc[2][RT];
#pragma unroll
  for (int t = 0; t < RT; ++t) acc[1][t] = (f32x4){-INFINITY, -INFINITY, -INFINITY, -INFINITY};  // "column before the first": ranks as a no-op
  unsigned codes[2][4] = {{0u, 0u, 0u, 0u}, {0u, 0u, 0u, 0u}};                                    // (-inf packed with code 0 stays -inf)
  float pinf = INFINITY;
  asm volatile("" : "+v"(pinf));        // opaque to the compiler: med3(a, b, +inf) folded to fmaxf brings the quieting v_max back
  auto rank = [&](int t, const f32x4& a, const unsigned (&code)[4]) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float pk = __uint_as_float((__float_as_uint(a[r]) & ~kmask) | code[r]);
      d2[t] = __builtin_amdgcn_fmed3f(d1[t], d2[t], pk);          // d1 >= d2 always: the median is the second largest
      d1[t] = __builtin_amdgcn_fmed3f(d1[t], pk, pinf);           // the larger of the two (a NaN candidate leaves both untouched)
    }
  };
  auto frag_at = [&](int buf, int img, int tl, int s) -> bf16x8 {
    return *reinterpret_cast<const bf16x8*>(&Ls[buf][img][(16 * tl + i) * E + 8 * ((4 * s + q) ^ i)]);
  };
  auto init4 = [&](f32x4& a, int t, const float4& mv, const float4& nv) {
    a = (f32x4){fmaf(cxh[t], nv.x, mv.x), fmaf(cxh[t], nv.y, mv.y), fmaf(cxh[t], nv.z, mv.z), fmaf(cxh[t], nv.w, mv.w)};
  };
  constexpr int NCOL = BK2_CODES / 16;
  for (int c = 0; c < nch; ++c) {
    const int buf = c & 1;
    if (c + 1 < nch) fill(c + 1, buf ^ 1);       // (buffer buf ^ 1 was last read in iteration c - 1, before the barrier below)
    // the chunk's first column: its fragments of k-block 0 and its initial values cannot be prepared across the barrier
    bf16x8 fh[2], fl[2];                                  // two fragment register sets (k-block parity)
    fh[0] = frag_at(buf, 0, 0, 0);
    fl[0] = frag_at(buf, 1, 0, 0);
    auto consts_of = [&](int tl, float4& mv, float4& nv) {          // codes 16 tl + 4 q .. + 3: {m0, n0, m1, n1}, {m2, n2, m3, n3}
      const float4 a = *reinterpret_cast<const float4*>(&Cs[buf][16 * tl + 4 * q][0]);
      const float4 b = *reinterpret_cast<const float4*>(&Cs[buf][16 * tl + 4 * q + 2][0]);
      mv = make_float4(a.x, a.z, b.x, b.z);
      nv = make_float4(a.y, a.w, b.y, b.w);
    };
    {
      float4 mv, nv;
      consts_of(0, mv, nv);
#pragma unroll
      for (int t = 0; t < RT; ++t) init4(acc[0][t], t, mv, nv);     // (set 1 still holds the previous chunk's last column: ranked below)
    }
#pragma unroll
    for (int tl = 0; tl < NCOL; ++tl) {
      const int cs = tl & 1, ps = cs ^ 1;                 // accumulator sets of this column / of the one before it (NCOL is even)
      const unsigned cb = (unsigned)(c * BK2_CODES + 16 * tl + 4 * q);
#pragma unroll
      for (int r = 0; r < 4; ++r) codes[cs][r] = cb + (unsigned)r;
      float4 mvn = make_float4(0.f, 0.f, 0.f, 0.f), nvn = mvn;
      if (tl + 1 < NCOL) consts_of(tl + 1, mvn, nvn);       // the next column's constants (same chunk)
#pragma unroll
      for (int s = 0; s < KB; ++s) {
        const int w = s & 1;                              // (KB is even: k-block 0 of every column sits in fragment set 0)
        if (s + 1 < KB) {
          fh[w ^ 1] = frag_at(buf, 0, tl, s + 1);
          fl[w ^ 1] = frag_at(buf, 1, tl, s + 1);
        } else if (tl + 1 < NCOL) {
          fh[w ^ 1] = frag_at(buf, 0, tl + 1, 0);
          fl[w ^ 1] = frag_at(buf, 1, tl + 1, 0);
        }
#pragma unroll
        for (int t = 0; t < RT; ++t) acc[cs][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fl[w], xh[t][s], acc[cs][t], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < RT; ++t) acc[cs][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fh[w], xl[t][s], acc[cs][t], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < RT; ++t) acc[cs][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fh[w], xh[t][s], acc[cs][t], 0, 0, 0);
        rank(s, acc[ps][s], codes[ps]);                   // row tile s of the column before this one ...
        if (tl + 1 < NCOL) init4(acc[ps][s], s, mvn, nvn);     // ... whose accumulator then takes the next column's initial values
        // the schedule of this k-block: its LDS reads, then one vector instruction behind each MFMA
        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
        }
#pragma unroll
        for (int g = 0; g < 8; ++g) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x002, 1, 0);
        }
      }
    }
    if (c < 6) VSTAMP(3 + 2 * c);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's share of the next chunk has landed ...
    __builtin_amdgcn_s_barrier();                         // ... and so has everybody's; everybody is done with this chunk's buffer
    if (c < 6) VSTAMP(4 + 2 * c);
  }
#pragma unroll
  for (int t = 0; t < RT; ++t) rank(t, acc[1][t], codes[1]);        // the last column (NCOL is even: set 1)
  VSTAMP(15);
  const float trunc = ldexpf(1.0f, kbits - 22);          // 2 x the relative truncation of a packed accumulator
  const bool all_undecided = und_count[1] != 0;          // the codebook has a non-finite code
#pragma unroll
  for (int t = 0; t < RT; ++t) {
#pragma unroll
    for (int o = 16; o <= 32; o <<= 1) {                 // (largest, second largest) of two disjoint code sets
      const float e1 = __shfl_xor(d1[t], o), e2 = __shfl_xor(d2[t], o);
      d2[t] = fmaxf(fminf(d1[t], e1), fmaxf(d2[t], e2));
      d1[t] = fmaxf(d1[t], e1);
    }
    const int row = r0 + 16 * t + i;
    if (q == 0 && row < N) {
      const int ca = (int)(__float_as_uint(d1[t]) & kmask);
      idx_out[row] = (int64_t)(ca < K ? ca : 0);
      // acc = -L / 2: decided iff L_b - L_a >= 2 r_a, i.e. acc_1 - acc_2 >= c_x |w_a| (= 2 cxh |w_a|) + the packing's truncation
      const float margin = 2.0f * cxh[t] * cst[ca < K ? ca : 0].y + trunc * fmaxf(fabsf(d1[t]), fabsf(d2[t]));
      // NaN / inf - inf (rows with non-finite values) compare false: undecided, the exact kernel follows torch.argmin
      if (all_undecided || !(d1[t] - d2[t] >= margin)) und_list[atomicAdd(und_count, 1)] = row;
    }
  }
}

// ---- fused pre_linear + assign with the distance SCREENING on the bf16 matrix pipe, exact by construction (round 3) ------------
// At N = 4096 a batch has one 16-row tile per CU; the fp32 kernel above spends 2.8 k cycles on the projection and then 14 k on the
// 256 fp32 MFMAs per wave of the -2 x W^T contraction, one behind the other.  Only the ARGMIN of that contraction is an output, so
//  (1) it is SCREENED on the bf16 pipe, in z-space:  flat.w_k = (W_pre z + b).w_k = z.u_k + b.w_k  with  u_k = W_pre^T w_k, so
//          e_k = (|w_k|^2 - 2 b.w_k) - 2 (zh + zl).bf16(u_k)      (zh = bf16(z), zl = bf16(z - zh); 8 MFMAs of 16 cycles per
//          16 x 16 x 128 tile instead of 32 of 32 cycles)
//      approximates d_k - |flat|^2 WITHOUT the projected rows: the sweep does not wait for the projection, both are fed by one
//      ~205 KB request stream per CU (z tile 8 KB, W_pre fp32 fragments 64 KB, bf16 fragments of U 128 KB, s'_k and radius
//      coefficients 4 KB) instead of the fp32 kernel's 328 KB;
//  (2) every code that can still be the fp32 kernel's argmin is re-evaluated with the EXACT fp32 chain of vq_fused_assign_kernel
//      on the projected rows (same operands, same k order => the same bits), the row's code being torch.argmin (:1259) over its
//      exact values (LDS atomicMin on (distance, index) keys).
// Which codes: |e_k - (d_k - |flat|^2)| <= r_k for the fp32 kernel's d_k, with the PER-CODE radius
//     r_k = 2^-7 (1 + 2^-5) |z||u_k|                   bf16(u) (2^-8), the z residual (2^-16) and the fp32 accumulation inside the
//                                                      bf16 MFMAs (budgeted 2^-14: 16x the RNE bound), all x 2 for the -2 x.w
//         + 2^-13 |flat|^ |w_k|                        fp32 rounding of U, of the projection and of the fp32 kernel's chain
//                                                      (<= 3 E 2^-24 x 2);  |flat|^ = |W_pre|_F |z| + |b| >= |flat|
//         + 2^-20 (|flat|^^2 + |w_k|^2) + 5e-31        roundings of (|x|^2 + |w|^2) - 2 x.w and of s'_k; denormal flushes
// so d_k - |flat|^2 lies in [e_k - r_k, e_k + r_k]: a code with e_k - r_k > min_j (e_j + r_j) cannot be the argmin; all others are
// candidates.  (A global margin from max|u|, max|w| fails on a TRAINED codebook: the EMA update leaves dead codes with norms
// hundreds of times the live ones', and their bound made every code a candidate.  Per code those are simply far away.)
// r_k = |z| P_k + Q_k + R_row: P_k, Q_k come packed as two bf16 (rounded up) per code from g2v_vq_bx_pack, R_row is added to the
// row's threshold.  The candidates get the fp32 kernel's own arithmetic: flat, idx and quantized are bitwise
// vq_fused_assign_kernel's.  A tile with a non-finite screening value or more than BXF_MAXP candidates takes the exact fp32
// sweep over all K codes instead (exact_only != 0 forces it: the A/B reference).
constexpr int BXF_MAXP = 128;      // (row, candidate) pairs re-evaluated per tile: 8 waves x one 16-pair MFMA tile
constexpr int BXF_LDH = 128 + 8;   // bf16 elements per LDS row of the hi / lo images: 272 B, conflict-free ds_read_b128 per 16 lanes
constexpr int BXF_LDE = 512 + 4;   // floats per LDS row of the screening values

// monotone map float -> uint32 (a < b  <=>  key(a) < key(b); no NaNs reach it), so that (key(d) << 32 | code) orders by
// distance first and by code index second: an LDS atomicMin over a row's candidates IS torch.argmin's tie rule
__device__ __forceinline__ unsigned long long bxf_key(float d, int code) {
  unsigned u = __float_as_uint(d);
  u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
  return ((unsigned long long)u << 32) | (unsigned)code;
}

struct BxfScalars { float wpf2, bb, pad0, pad1; };      // |W_pre|_F^2, |b|^2 (behind the image, g2v_vq_bx_pack)

// Eight waves, all doing the same (role splits -- projection waves beside sweep waves -- were built and measured: whatever puts
// partial-line or duplicate requests of several waves into the CU's one in-order memory pipeline, e.g. every projection wave
// reading the raw tile as fragments from global, or leaves only four waves to issue most of the requests, drops the request stream
// from ~50 to ~20 B/clk).  Phases, one workgroup barrier between two:
//   requests, all up front, in consumption order: raw tile (16 B per thread), this wave's W_pre fragments (output tile `wave`) and
//   bias, its 4 code tiles of the U image + s';  z staged as fp32 + bf16 hi / lo, |z_row|^2 and the row's margin on the way
//   | projection (the fp32 kernel's chain) -> Xf + ||x||^2 tile partials, then the bf16 sweep -> screening values to LDS (16 x K)
//   | scan: 512 threads x 16 codes against the row threshold -> (row, code) pair list
//   | exact distances: ONE pair per thread as a 128-long fmaf chain in the MFMA's k order (s, component, q ascending: measured
//     bitwise equal to the v_mfma_f32_16x16x4_f32 chain, gpurun_tools/mfma_chain_test.hip), the code's row straight from global,
//     LDS atomicMin on (distance, index) keys -- 1.4 k cycles whatever the number of pairs, where a 16 x 16 x 128 MFMA tile per 16
//     pairs cost a gather round trip plus a 32-deep dependent MFMA chain behind yet another barrier
//   | gather + straight-through + SSE.
template <bool EXACT_ONLY>
__device__ __forceinline__ void vq_fused_bx_body(const float* __restrict__ z, const float* __restrict__ Wpf,
                                                 const float* __restrict__ bp, const float* __restrict__ W,
                                                 const __bf16* __restrict__ Uhf, const float* __restrict__ sprime,
                                                 const unsigned* __restrict__ pqk,
                                                 const BxfScalars* __restrict__ scal, const float* __restrict__ wsq,
                                                 float* __restrict__ flat_out, int64_t* __restrict__ idx_out,
                                                 float* __restrict__ quant, float* __restrict__ sse_partial,
                                                 int* __restrict__ diag, int N, int K) {
  constexpr int E = 128, KS = E / 16, KB = E / 32, ldx = E + 4, NT = 5;
  __shared__ __attribute__((aligned(16))) float Xz[VQ_ROWS * ldx];
  __shared__ __attribute__((aligned(16))) float Xf[VQ_ROWS * ldx];
  __shared__ __attribute__((aligned(16))) __bf16 Zh[VQ_ROWS * BXF_LDH];
  __shared__ __attribute__((aligned(16))) __bf16 Zl[VQ_ROWS * BXF_LDH];
  __shared__ __attribute__((aligned(16))) float Es[VQ_ROWS * BXF_LDE];
  __shared__ float xxp[8 * 16];                // ||flat_row||^2 partials per 16-feature tile
  __shared__ float xx[16];
  __shared__ float marg[16];
  __shared__ float znrow[16];
  __shared__ float wmin[8 * 16];
  __shared__ unsigned long long rowbest[16];
  __shared__ int p_row[BXF_MAXP];
  __shared__ int p_code[BXF_MAXP];
  __shared__ float wbest_d[8 * 16];
  __shared__ int wbest_k[8 * 16];
  __shared__ int best_k[16];
  __shared__ float rowsse[16];
  __shared__ int rowcnt[16];
  __shared__ int rowcode[16];
  __shared__ int s_exact, s_np;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r0 = blockIdx.x * VQ_ROWS;
  const int nrows = min(VQ_ROWS, N - r0);
  const int i = lane & 15, q = lane >> 4;
  const int ntile = K >> 4;
  VSTAMP(0);
  const BxfScalars sc = *scal;
  // ---- every request of the first half, in consumption order ----------------------------------------------------------------------
  const int zrow = tid >> 5, zpart = tid & 31;
  const float4 zv = *reinterpret_cast<const float4*>(z + (int64_t)(r0 + (zrow < nrows ? zrow : 0)) * E + 4 * zpart);
  float4 wp[KS];
#pragma unroll
  for (int s = 0; s < KS; ++s) wp[s] = *reinterpret_cast<const float4*>(Wpf + ((int64_t)(wave * KS + s) * 64 + lane) * 4);
  const float4 b0 = *reinterpret_cast<const float4*>(bp + 16 * wave + 4 * q);
  bf16x8 uf[NT][KB];
  float4 sp[NT];
  uint4 pq[NT];
  // code tiles: waves 0-3 (which request theirs first, below) own 5 each -- w + 4 p --, waves 4-7 own 3 each -- 20 + (w - 4) + 4 p
  const int kt0 = wave < 4 ? wave : 16 + wave, nown = wave < 4 ? 5 : 3;
  auto request_u = [&]() {
#pragma unroll
    for (int p = 0; p < NT; ++p) {
      if (p >= nown) break;
      const int kt = kt0 + 4 * p;
      const int ktc = kt < ntile ? kt : 0;       // beyond K / 16: a valid tile, its values are dropped
#pragma unroll
      for (int kb = 0; kb < KB; ++kb) uf[p][kb] = *reinterpret_cast<const bf16x8*>(Uhf + ((int64_t)(ktc * KB + kb) * 64 + lane) * 8);
      sp[p] = *reinterpret_cast<const float4*>(sprime + 16 * ktc + 4 * q);
      pq[p] = *reinterpret_cast<const uint4*>(pqk + 16 * ktc + 4 * q);
    }
  };
  __builtin_amdgcn_sched_barrier(0);
  if (tid < 16) {
    rowbest[tid] = ~0ull;
    rowcnt[tid] = 0;
    rowsse[tid] = 0.f;
  }
  if (tid == 0) {
    s_exact = EXACT_ONLY ? 1 : 0;
    s_np = 0;
  }
  {  // stage the raw tile: fp32 (projection operand, straight-through) and its bf16 hi / lo images (screening operand); the
     // row's 32 threads (two DPP rows of one wave) add up |z_row|^2 and one of them prices the row's margin
    const float4 zs = zrow < nrows ? zv : make_float4(0.f, 0.f, 0.f, 0.f);
    *reinterpret_cast<float4*>(Xz + zrow * ldx + 4 * zpart) = zs;
    typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
    const float v[4] = {zs.x, zs.y, zs.z, zs.w};
    bf16x4 h4, l4;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const __bf16 h = (__bf16)v[j];
      h4[j] = h;
      l4[j] = (__bf16)(v[j] - (float)h);
    }
    *reinterpret_cast<bf16x4*>(Zh + zrow * BXF_LDH + 4 * zpart) = h4;
    *reinterpret_cast<bf16x4*>(Zl + zrow * BXF_LDH + 4 * zpart) = l4;
    float zz = reduce16(sq4_chain(zs.x, zs.y, zs.z, zs.w));
    zz += __shfl_xor(zz, 16);
    if (zpart == 0) {
      const float zn = __builtin_amdgcn_sqrtf(zz) * 1.001f;
      const float fn = __builtin_amdgcn_sqrtf(sc.wpf2) * zn + __builtin_amdgcn_sqrtf(sc.bb);       // >= |flat_row|
      znrow[zrow] = zn;
      marg[zrow] = 2.0f * (9.5367432e-7f * (fn * fn) + 5e-31f);       // 2 R_row: the part of the radius common to the row's codes
    }
  }
  lds_barrier();
  VSTAMP(1);
  // A wave that is issuing requests into the saturated memory pipeline (~130 cycles each) cannot issue MFMAs meanwhile, and a
  // wave inside the projection's dependent MFMA chain issues nothing else: the two waves of a SIMD (w, w + 4) take turns --
  // waves 0-3 request their code tiles first and project afterwards, waves 4-7 project first and request afterwards.
  if (wave < 4) {
    request_u();
    __builtin_amdgcn_sched_barrier(0);
  }
  {  // ---- pre_linear (:1230), output tile `wave`: one fp32 chain over k, the chain of vq_fused_assign_kernel ---------------------
    f32x4 a = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const float4 xb = *reinterpret_cast<const float4*>(Xz + i * ldx + 16 * s + 4 * q);
      a = mfma16(wp[s].x, xb.x, a);
      a = mfma16(wp[s].y, xb.y, a);
      a = mfma16(wp[s].z, xb.z, a);
      a = mfma16(wp[s].w, xb.w, a);
    }
    const float4 v0 = make_float4(a[0] + b0.x, a[1] + b0.y, a[2] + b0.z, a[3] + b0.w);
    *reinterpret_cast<float4*>(Xf + i * ldx + 16 * wave + 4 * q) = v0;
    const float p0 = tile_partial_sumsq(v0.x, v0.y, v0.z, v0.w);
    if (lane < 16) xxp[wave * 16 + lane] = p0;
  }
  VSTAMP(7);
  VSTAMP_B(7);
  if (wave >= 4) {
    __builtin_amdgcn_sched_barrier(0);
    request_u();
    __builtin_amdgcn_sched_barrier(0);
  }
  {  // ---- screening sweep on the bf16 pipe, in z-space ---------------------------------------------------------------------------
    bf16x8 zh[KB], zl[KB];
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) {
      zh[kb] = *reinterpret_cast<const bf16x8*>(Zh + i * BXF_LDH + 32 * kb + 8 * q);
      zl[kb] = *reinterpret_cast<const bf16x8*>(Zl + i * BXF_LDH + 32 * kb + 8 * q);
    }
    const float zn = znrow[i];
    float emin = INFINITY, esum = 0.f;
#pragma unroll
    for (int p = 0; p < NT; ++p) {
      if (p >= nown) break;
      const int kt = kt0 + 4 * p;
      f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kb = 0; kb < KB; ++kb) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(uf[p][kb], zl[kb], acc, 0, 0, 0);
#pragma unroll
      for (int kb = 0; kb < KB; ++kb) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(uf[p][kb], zh[kb], acc, 0, 0, 0);
      if (kt < ntile) {                                    // wave-uniform; every request above is consumed either way
        const float ev_[4] = {sp[p].x - 2.0f * acc[0], sp[p].y - 2.0f * acc[1], sp[p].z - 2.0f * acc[2], sp[p].w - 2.0f * acc[3]};
        const unsigned pk_[4] = {pq[p].x, pq[p].y, pq[p].z, pq[p].w};
        float lo[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          // radius of code k for this row: |z| P_k + Q_k (+ R_row, added to the threshold): lo / hi bracket d_k - |flat|^2
          const float rk = fmaf(zn, __uint_as_float(pk_[r] & 0xffff0000u), __uint_as_float(pk_[r] << 16));
          lo[r] = ev_[r] - rk;
          emin = fminf(emin, ev_[r] + rk);
          esum += ev_[r];                                  // NaN / inf anywhere poisons the sum
        }
        *reinterpret_cast<float4*>(Es + i * BXF_LDE + 16 * kt + 4 * q) = make_float4(lo[0], lo[1], lo[2], lo[3]);
      }
    }
    emin = fminf(emin, __shfl_xor(emin, 16));
    emin = fminf(emin, __shfl_xor(emin, 32));
    if (lane < 16) wmin[wave * 16 + lane] = emin;
    if (__any(!(fabsf(esum) < INFINITY)) && lane == 0) s_exact = 1;
  }
  VSTAMP(8);
  VSTAMP_B(8);
  lds_barrier();
  VSTAMP(2);
  float pf0 = 0.f, pf1 = 0.f, pf2 = 0.f, pf3 = 0.f;
  {  // candidate scan by all 512 threads: thread -> row tid >> 5, 16 of its codes
    const int row = tid >> 5, c0 = 4 * (tid & 31);          // codes c0 + 128 g + (0..3): a wave-level read is 2 x 512 contiguous bytes
    float4 e[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) e[g] = 128 * g < K ? *reinterpret_cast<const float4*>(Es + row * BXF_LDE + c0 + 128 * g) : make_float4(INFINITY, INFINITY, INFINITY, INFINITY);
    float m = wmin[row];
#pragma unroll
    for (int w = 1; w < 8; ++w) m = fminf(m, wmin[w * 16 + row]);
    const float th = m + marg[row];
    if (__any(!(fabsf(th) < INFINITY)) && lane == 0) s_exact = 1;      // non-finite norms: no usable bound
    if (tid < 16) xx[tid] = sum8_partials(xxp, tid);
    const float ev[16] = {e[0].x, e[0].y, e[0].z, e[0].w, e[1].x, e[1].y, e[1].z, e[1].w,
                          e[2].x, e[2].y, e[2].z, e[2].w, e[3].x, e[3].y, e[3].z, e[3].w};
    unsigned cm = 0;                      // candidate bits of this thread's 16 codes: branch-free, ONE conditional block per wave
#pragma unroll
    for (int c = 0; c < 16; ++c) cm |= (ev[c] <= th) ? (1u << c) : 0u;
    if (cm) {                             // about 1.4 threads per row get here
      const int n = __popc(cm);
      int slot = atomicAdd(&s_np, n);
      atomicAdd(&rowcnt[row], n);
      rowcode[row] = c0 + 128 * ((__ffs(cm) - 1) >> 2) + ((__ffs(cm) - 1) & 3);      // THE code of a row that ends with one candidate
      while (cm) {
        const int c = __ffs(cm) - 1;
        cm &= cm - 1;
        const int code = c0 + 128 * (c >> 2) + (c & 3);
        if (slot < BXF_MAXP) {
          p_row[slot] = row;
          p_code[slot] = code;
        }
        ++slot;
        if (slot > BXF_MAXP) break;       // the list is full: this tile takes the exact sweep, nothing more to list or to touch
        // touch the code's four cache lines now: the exact chain (another thread, behind the barrier) then reads them from L1
        // instead of paying an L2 round trip.  Fire and forget: the values are never used, the registers are pinned below.
        const float* wr = W + (int64_t)code * E;
        asm volatile("global_load_dword %0, %4, off\n\tglobal_load_dword %1, %4, off offset:128\n\t"
                     "global_load_dword %2, %4, off offset:256\n\tglobal_load_dword %3, %4, off offset:384"
                     : "=&v"(pf0), "=&v"(pf1), "=&v"(pf2), "=&v"(pf3) : "v"(wr) : "memory");
      }
    }
  }
  VSTAMP(9);
  VSTAMP_B(9);
  lds_barrier();
  VSTAMP(3);
  const int P = s_np;
  bool exact = (s_exact != 0) || P > BXF_MAXP;
  // gather + straight-through + SSE on raw z (:1285-1292) of one row by its 16 threads; the row's SSE goes to rowsse
  auto finish_row = [&](int row, int part, int code) {
    const float* wq_ = W + (int64_t)code * E;
    float* qo = quant + (int64_t)(r0 + row) * E;
    float sse = 0.f;
#pragma unroll
    for (int j = 0; j < E / 64; ++j) {
      const int c = 4 * (part + 16 * j);
      const float4 zzv = *reinterpret_cast<const float4*>(Xz + row * ldx + c), wv = *reinterpret_cast<const float4*>(wq_ + c);
      const float4 df = make_float4(wv.x - zzv.x, wv.y - zzv.y, wv.z - zzv.z, wv.w - zzv.w);
      *reinterpret_cast<float4*>(qo + c) = make_float4(zzv.x + df.x, zzv.y + df.y, zzv.z + df.z, zzv.w + df.w);   // :1292
      sse += df.x * df.x + df.y * df.y + df.z * df.z + df.w * df.w;
    }
    sse = reduce16(sse);                 // the row's 16 threads (DPP); the 16 rows are added in order at the end
    if (part == 0) {
      rowsse[row] = sse;
      idx_out[r0 + row] = (int64_t)code;
    }
  };
  const int frow = (tid & 255) >> 4, fpart = tid & 15;
  if (!exact) {
    if (tid < P) {     // ---- one pair per thread: the fp32 kernel's dot product of (projected row, code) as its fmaf chain -------------
      const int row = p_row[tid], code = p_code[tid];
      const float* wr = W + (int64_t)code * E;
      const float* xr_ = Xf + row * ldx;
      float4 wv[E / 4];
#pragma unroll
      for (int t = 0; t < E / 4; ++t) wv[t] = *reinterpret_cast<const float4*>(wr + 4 * t);
      const float sq = wsq[code];
      float a = 0.f;
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        float4 xv[4];
#pragma unroll
        for (int qq = 0; qq < 4; ++qq) xv[qq] = *reinterpret_cast<const float4*>(xr_ + 16 * s + 4 * qq);
#pragma unroll
        for (int qq = 0; qq < 4; ++qq) a = fmaf(wv[4 * s + qq].x, xv[qq].x, a);
#pragma unroll
        for (int qq = 0; qq < 4; ++qq) a = fmaf(wv[4 * s + qq].y, xv[qq].y, a);
#pragma unroll
        for (int qq = 0; qq < 4; ++qq) a = fmaf(wv[4 * s + qq].z, xv[qq].z, a);
#pragma unroll
        for (int qq = 0; qq < 4; ++qq) a = fmaf(wv[4 * s + qq].w, xv[qq].w, a);
      }
      const float d = (xx[row] + sq) - 2.0f * a;
      if (!(fabsf(d) < INFINITY)) s_exact = 1;              // a non-finite exact value: the exact sweep decides (NaN rules)
      atomicMin(&rowbest[row], bxf_key(d, code));
    } else if (tid >= 256) {
      // ---- beside the chains (waves 4-7): the rows that ended the scan with ONE candidate are decided, finish them now; and the
      // projected rows go to global (the code statistics read them in a later launch)
      if (frow < nrows) {
        if (rowcnt[frow] == 1) finish_row(frow, fpart, rowcode[frow]);
#pragma unroll
        for (int j = 0; j < E / 64; ++j) {
          const int c = 4 * (fpart + 16 * j);
          *reinterpret_cast<float4*>(flat_out + (int64_t)(r0 + frow) * E + c) = *reinterpret_cast<const float4*>(Xf + frow * ldx + c);
        }
      }
    }
    VSTAMP(10);
    VSTAMP_B(10);
    lds_barrier();
    VSTAMP(11);
    exact = s_exact != 0;
    if (!exact && tid < 256 && frow < nrows && rowcnt[frow] != 1)
      finish_row(frow, fpart, (int)(unsigned)(rowbest[frow] & 0xffffffffull));
  }
  VSTAMP(4);
  if (exact) {
    // ---- exact fp32 sweep over every code (vq_fused_assign_kernel's arithmetic; slow path) ---------------------------------------
    const float xr = xx[i];
    float4 xb[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) xb[s] = *reinterpret_cast<const float4*>(Xf + i * ldx + 16 * s + 4 * q);
    float bd = INFINITY;
    int bk = 0;
    for (int kt = wave; kt < ntile; kt += 8) {
      const float* wrow = W + (int64_t)(16 * kt + i) * E + 4 * q;
      float4 wg[KS];
#pragma unroll
      for (int s = 0; s < KS; ++s) wg[s] = *reinterpret_cast<const float4*>(wrow + 16 * s);
      const float4 sq = *reinterpret_cast<const float4*>(wsq + 16 * kt + 4 * q);
      f32x4 a = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        a = mfma16(wg[s].x, xb[s].x, a);
        a = mfma16(wg[s].y, xb[s].y, a);
        a = mfma16(wg[s].z, xb[s].z, a);
        a = mfma16(wg[s].w, xb[s].w, a);
      }
      const float sv[4] = {sq.x, sq.y, sq.z, sq.w};
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float d = (xr + sv[r]) - 2.0f * a[r];
        if (argmin_better(d, bd)) { bd = d; bk = 16 * kt + 4 * q + r; }
      }
    }
    {
      float d2 = __shfl_xor(bd, 16);
      int k2 = __shfl_xor(bk, 16);
      argmin_merge(bd, bk, d2, k2);
      d2 = __shfl_xor(bd, 32);
      k2 = __shfl_xor(bk, 32);
      argmin_merge(bd, bk, d2, k2);
    }
    if (lane < 16) {
      wbest_d[wave * 16 + lane] = bd;
      wbest_k[wave * 16 + lane] = bk;
    }
    lds_barrier();
    if (tid < 16) {
      float d = wbest_d[tid];
      int k = wbest_k[tid];
#pragma unroll
      for (int w = 1; w < 8; ++w) argmin_merge(d, k, wbest_d[w * 16 + tid], wbest_k[w * 16 + tid]);
      best_k[tid] = k;
    }
    lds_barrier();
    if (frow < nrows) {
      if (tid < 256) {
        finish_row(frow, fpart, best_k[frow]);
      } else {
#pragma unroll
        for (int j = 0; j < E / 64; ++j) {
          const int c = 4 * (fpart + 16 * j);
          *reinterpret_cast<float4*>(flat_out + (int64_t)(r0 + frow) * E + c) = *reinterpret_cast<const float4*>(Xf + frow * ldx + c);
        }
      }
    }
  }
  __syncthreads();
  VSTAMP(5);
  if (tid == 0) {
    if (sse_partial) {
      float t = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) t += rowsse[r];
      sse_partial[blockIdx.x] = t;
    }
    if (diag) {
      if (exact) atomicAdd(&diag[0], 1);
      else atomicAdd(&diag[1], P);
    }
  }
  asm volatile("" ::"v"(pf0), "v"(pf1), "v"(pf2), "v"(pf3));          // the prefetch destinations stay reserved until here
  VSTAMP(6);
}

// Two symbols over one body, so that a kernel trace tells them apart (round-3 verdict: the forced exact sweep, 16 us, ran under
// the product kernel's name and spoilt its average in profiles/*kernel_stats*.csv):
//   vq_fused_bx_kernel        the product: bf16 screening + exact fp32 re-evaluation of the candidates
//   vq_fused_bx_exact_kernel  G2V_VQ_BX_EXACT: every tile takes the exact fp32 sweep over all K codes (the A/B reference)
#define VQ_BX_PARAMS                                                                                                       \
  const float *__restrict__ z, const float *__restrict__ Wpf, const float *__restrict__ bp, const float *__restrict__ W,   \
      const __bf16 *__restrict__ Uhf, const float *__restrict__ sprime, const unsigned *__restrict__ pqk,                  \
      const BxfScalars *__restrict__ scal, const float *__restrict__ wsq, float *__restrict__ flat_out,                    \
      int64_t *__restrict__ idx_out, float *__restrict__ quant, float *__restrict__ sse_partial, int *__restrict__ diag,   \
      int N, int K
#define VQ_BX_ARGS z, Wpf, bp, W, Uhf, sprime, pqk, scal, wsq, flat_out, idx_out, quant, sse_partial, diag, N, K
__global__ __launch_bounds__(512) void vq_fused_bx_kernel(VQ_BX_PARAMS) { vq_fused_bx_body<false>(VQ_BX_ARGS); }
__global__ __launch_bounds__(512) void vq_fused_bx_exact_kernel(VQ_BX_PARAMS) { vq_fused_bx_body<true>(VQ_BX_ARGS); }


// Screening operands of vq_fused_bx_kernel, rebuilt whenever the codebook (or pre_linear) changed: one workgroup per 16 codes.
//   U = W W_pre (u_k = W_pre^T w_k), as the bf16 MFMA-fragment image [K/16 tiles][E/32 k-blocks][64 lanes][8]: lane (q, i) of
//   tile kt, block kb holds bf16(U[16 kt + i][32 kb + 8 q .. + 7]);  s'_k = |w_k|^2 - 2 b.w_k;  the per-code radius
//   coefficients P_k = 2^-7 (1 + 2^-5) |u_k| + 2^-13 |W_pre|_F |w_k|,  Q_k = 2^-13 |b| |w_k| + 2^-20 |w_k|^2, each rounded UP to
//   bf16 and packed (P_k << 16 | Q_k);  scalars |W_pre|_F^2, |b|^2.
__device__ __forceinline__ unsigned bf16_ceil_bits(float v) {      // v >= 0 (or NaN / inf, which stay what they are)
  const unsigned u = __float_as_uint(v);
  return (u & 0x7f800000u) == 0x7f800000u ? (u | ((u & 0xffffu) ? 0x10000u : 0u)) : (u + 0xffffu);      // upper 16 bits are the result
}
__global__ __launch_bounds__(256) void vq_bx_pack_kernel(const float* __restrict__ W, const float* __restrict__ wsq,
                                                         const float* __restrict__ Wp, const float* __restrict__ bp,
                                                         __bf16* __restrict__ Uhf, float* __restrict__ sprime,
                                                         unsigned* __restrict__ pqk, BxfScalars* __restrict__ scal, int K) {
  constexpr int E = 128, LDP = E + 1, LDW = E + 4;
  // U tile = Ws (16 codes x E) W_pre (E x E) on the fp32 MFMA: A = W_pre^T (column on the lane) read from an LDS copy of W_pre,
  // B = the 16 code rows.  (The first version contracted on the VALU with W_pre read from global inside the loop: 143 us, and
  // 71 us from LDS -- in the branch beside the encoder GRU that made the quantiser wait.)
  extern __shared__ __attribute__((aligned(16))) float smem_pack[];
  float* WpS = smem_pack;                         // [E][LDP]
  float* Ws = WpS + E * LDP;                      // [16][LDW]   (16-byte aligned: E * LDP * 4 = 66048)
  float* Us = Ws + 16 * LDW;                      // [16][LDW]
  __shared__ float red[4];
  __shared__ float s_wpf2, s_bb;
  const int tid = threadIdx.x, kt = blockIdx.x, lane = tid & 63, wave = tid >> 6, i = lane & 15, q = lane >> 4;
  for (int e4 = tid; e4 < 16 * E / 4; e4 += 256) {
    const int r = e4 / (E / 4), c = 4 * (e4 % (E / 4));
    *reinterpret_cast<float4*>(Ws + r * LDW + c) = *reinterpret_cast<const float4*>(W + (int64_t)(16 * kt + r) * E + c);
  }
  {  // stage W_pre; |W_pre|_F^2 and |b|^2 on the way (every workgroup for itself: the radius coefficients below need them)
    float f = 0.f;
    for (int e4 = tid; e4 < E * E / 4; e4 += 256) {
      const float4 v = reinterpret_cast<const float4*>(Wp)[e4];
      const int r = e4 / (E / 4), c = 4 * (e4 % (E / 4));
      float* d = WpS + r * LDP + c;
      d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
      f = fmaf(v.x, v.x, fmaf(v.y, v.y, fmaf(v.z, v.z, fmaf(v.w, v.w, f))));
    }
    f = wave_sum(f);
    if (lane == 0) red[wave] = f;
  }
  __syncthreads();
  if (tid == 0) {
    float bb = 0.f;
    for (int e = 0; e < E; ++e) bb = fmaf(bp[e], bp[e], bb);
    s_wpf2 = (red[0] + red[1]) + (red[2] + red[3]);
    s_bb = bb;
    if (kt == 0) {
      scal->wpf2 = s_wpf2;
      scal->bb = bb;
      scal->pad0 = 0.f;
      scal->pad1 = 0.f;
    }
  }
#pragma unroll
  for (int t = 0; t < 2; ++t) {                    // column tiles wave, wave + 4 of U
    const int jt = wave + 4 * t;
    f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < E / 16; ++s) {
      const float4 xb = *reinterpret_cast<const float4*>(Ws + i * LDW + 16 * s + 4 * q);
      const float* ap = WpS + (16 * s + 4 * q) * LDP + 16 * jt + i;
      acc = mfma16(ap[0], xb.x, acc);
      acc = mfma16(ap[LDP], xb.y, acc);
      acc = mfma16(ap[2 * LDP], xb.z, acc);
      acc = mfma16(ap[3 * LDP], xb.w, acc);
    }
    // lane (code i, q) holds U[code][16 jt + 4 q .. + 3]
    *reinterpret_cast<float4*>(Us + i * LDW + 16 * jt + 4 * q) = make_float4(acc[0], acc[1], acc[2], acc[3]);
  }
  __syncthreads();
  for (int o = tid; o < 16 * E / 8; o += 256) {        // 16-byte fragment elements of this tile: [kb][lane]
    const int ln = o & 63, kb = o >> 6, ii = ln & 15, qq = ln >> 4;
    bf16x8 h;
#pragma unroll
    for (int e = 0; e < 8; ++e) h[e] = (__bf16)Us[ii * LDW + 32 * kb + 8 * qq + e];
    *reinterpret_cast<bf16x8*>(Uhf + ((int64_t)(kt * (E / 32) + kb) * 64 + ln) * 8) = h;
  }
  if (tid < 16) {
    float uu = 0.f, bw = 0.f;
    for (int e = 0; e < E; ++e) {
      uu = fmaf(Us[tid * LDW + e], Us[tid * LDW + e], uu);
      bw = fmaf(bp[e], Ws[tid * LDW + e], bw);
    }
    const float s = wsq[16 * kt + tid];
    sprime[16 * kt + tid] = s - 2.0f * bw;
    const float un = sqrtf(uu) * 1.0001f, wn = sqrtf(fabsf(s)) * 1.0001f;      // a NaN norm gives NaN coefficients: never a candidate
    const float P = 0.008056641f * un + 1.2207031e-4f * (sqrtf(s_wpf2) * 1.0001f) * wn;      // by the radius, but s' is NaN too ->
    const float Q = 1.2207031e-4f * (sqrtf(s_bb) * 1.0001f) * wn + 9.5367432e-7f * fabsf(s);  // the tile takes the exact sweep
    pqk[16 * kt + tid] = (bf16_ceil_bits(P) & 0xffff0000u) | (bf16_ceil_bits(Q) >> 16);
  }
}

// rows per workgroup: 16 (fill the chip; the codebook stream is then L2-bound), 64 for bulk assignment (MFMA-bound)
static int vq_rows_per_block(int N) { return N >= 16384 ? 64 : 16; }   // measured: 32 rows at N = 4096 is slower (18.8 vs 13.7 us)
// upper bound on the number of SSE partials any path writes for N rows (callers size sse_partial with it)
extern "C" int g2v_vq_assign_blocks(int N) { return N > 0 ? cdiv(N, VQ_ROWS) : 0; }

extern "C" int g2v_vq_code_sqnorm(const float* codebook, float* code_sqnorm, int K, int E, g2v_stream_t stream) {
  G2V_REQUIRE(codebook && code_sqnorm, "null pointer");
  G2V_REQUIRE(K > 0 && E > 0, "non-positive size");
  hipLaunchKernelGGL(code_sqnorm_kernel, dim3(cdiv((int64_t)K * 64, 256)), dim3(256), 0, (hipStream_t)stream, codebook,
                     code_sqnorm, K, E);
  G2V_CHECK_LAUNCH();
  return G2V_OK;
}

extern "C" int g2v_vq_assign_fwd(const float* flat, const float* z, const float* codebook, const float* code_sqnorm,
                                 int64_t* idx, float* quantized, float* dist_min, float* sse_partial, int N, int E,
                                 int K, g2v_stream_t stream) {
  G2V_REQUIRE(flat && codebook && code_sqnorm && idx, "null pointer");
  G2V_REQUIRE(!quantized || z, "z required with quantized");
  G2V_REQUIRE(N > 0 && E > 0 && K > 0, "non-positive size");
  if (E == 128 && (K & 127) == 0 && ptr_aligned16(flat) && ptr_aligned16(codebook) && ptr_aligned16(code_sqnorm) &&
      (!quantized || (ptr_aligned16(z) && ptr_aligned16(quantized)))) {
    const int rows = vq_rows_per_block(N);
    hipStream_t st = (hipStream_t)stream;
    if (sse_partial && rows > VQ_ROWS) {   // fewer, larger blocks leave the tail of the partial array untouched: zero it
      (void)hipMemsetAsync(sse_partial, 0, sizeof(float) * (size_t)cdiv(N, VQ_ROWS), st);
    }
    if (rows == 64)
      hipLaunchKernelGGL((vq_assign_rt_kernel<128, 4>), dim3(cdiv(N, 64)), dim3(256), 0, st, flat, z, codebook,
                         code_sqnorm, idx, quantized, dist_min, sse_partial, N, K);
    else
      hipLaunchKernelGGL(vq_assign_fast_kernel<128>, dim3(cdiv(N, VQ_ROWS)), dim3(256), 0, st, flat, z, codebook,
                         code_sqnorm, idx, quantized, dist_min, sse_partial, N, K);
    G2V_CHECK_LAUNCH();
    return G2V_OK;
  }
  const int Ep = (E + 15) & ~15;
  const size_t lds = (size_t)(VQ_ROWS * (Ep + 4) + 16 + 64 + 64 + 16 + 4) * sizeof(float);
  G2V_REQUIRE(lds <= 160 * 1024, "embedding dim too large for LDS");
  {
    // few row tiles: split the codes over workgroups (see vq_assign_split_kernel)
    const int ntile = cdiv(K, 16), nrt = cdiv(N, VQ_ROWS);
    if (nrt <= 64 && ntile >= 8) {
      int nsplit = ntile / 4;                          // >= one code tile per wave
      if (nsplit > 8) nsplit = 8;
      const int tps = cdiv(ntile, nsplit);
      nsplit = cdiv(ntile, tps);
      hipStream_t st = (hipStream_t)stream;
      if (lds > 48 * 1024)
        (void)hipFuncSetAttribute((const void*)vq_assign_split_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (hipMemsetAsync(idx, 0xFF, sizeof(int64_t) * (size_t)N, st) != hipSuccess) {
        set_error("g2v_vq_assign_fwd: memset failed");
        return G2V_ERR_LAUNCH;
      }
      hipLaunchKernelGGL(vq_assign_split_kernel, dim3(nrt, nsplit), dim3(256), lds, st, flat, codebook, code_sqnorm,
                         reinterpret_cast<unsigned long long*>(idx), N, E, K, tps);
      G2V_CHECK_LAUNCH();
      hipLaunchKernelGGL(vq_assign_finish_kernel, dim3(nrt), dim3(256), 0, st, reinterpret_cast<unsigned long long*>(idx), z, codebook,
                         quantized, dist_min, sse_partial, N, E);
      G2V_CHECK_LAUNCH();
      return G2V_OK;
    }
  }
  if (lds > 48 * 1024)
    (void)hipFuncSetAttribute((const void*)vq_assign_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL(vq_assign_kernel, dim3(cdiv(N, VQ_ROWS)), dim3(256), lds, (hipStream_t)stream, flat, z, codebook,
                     code_sqnorm, idx, quantized, dist_min, sse_partial, N, E, K);
  G2V_CHECK_LAUNCH();
  return G2V_OK;
}

// g2v_vq_assign_fwd on the fragment-major codebook image of g2v_vq_pack_codebook (K * E floats), any E % 16 == 0, K % 16 == 0:
// the same outputs bit for bit (vq_assign_p_kernel above).
extern "C" int g2v_vq_assign_packed_ok(int N, int E, int K) {
  return (N > 0 && E >= 16 && (E & 15) == 0 && K >= 16 && (K & 15) == 0 &&
          ((size_t)64 * (E + 4) + 64 + 2 * 8 * 64 + 64 + 8) * sizeof(float) <= 160 * 1024) ? 1 : 0;
}
extern "C" int g2v_vq_assign_packed_fwd(const float* flat, const float* z, const float* codebook, const float* codebook_frag,
                                        const float* code_sqnorm, int64_t* idx, float* quantized, float* dist_min,
                                        float* sse_partial, int N, int E, int K, g2v_stream_t stream) {
  G2V_REQUIRE(flat && codebook && codebook_frag && code_sqnorm && idx, "null pointer");
  G2V_REQUIRE(!quantized || z, "z required with quantized");
  if (!g2v_vq_assign_packed_ok(N, E, K)) {
    set_error("g2v_vq_assign_packed_fwd: needs E %% 16 == 0, K %% 16 == 0 (g2v_vq_assign_packed_ok)");
    return G2V_ERR_UNSUPPORTED;
  }
  G2V_REQUIRE(ptr_aligned16(flat) && ptr_aligned16(codebook_frag) && ptr_aligned16(code_sqnorm), "16-byte aligned operands");
  hipStream_t st = (hipStream_t)stream;
  // row tiles per workgroup: one while there is a CU per tile or two, more for bulk assignment (every fragment then feeds NR tiles)
  const int nr = N >= 32768 ? 4 : (N >= 8192 ? 2 : 1);
  const size_t lds = ((size_t)16 * nr * (E + 4) + 16 * nr + 2 * 8 * 16 * nr + 16 * nr + 8) * sizeof(float);
#define G2V_VQP(NR_, NT_)                                                                                                  \
  do {                                                                                                                     \
    if (lds > 48 * 1024)                                                                                                   \
      (void)hipFuncSetAttribute((const void*)vq_assign_p_kernel<NR_, NT_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
    hipLaunchKernelGGL((vq_assign_p_kernel<NR_, NT_>), dim3(cdiv(N, 16 * NR_)), dim3(512), lds, st, flat, z, codebook,     \
                       codebook_frag, code_sqnorm, idx, quantized, dist_min, sse_partial, N, E, K);                        \
  } while (0)
  if (nr == 4) G2V_VQP(4, 2);
  else if (nr == 2) G2V_VQP(2, 4);
  else G2V_VQP(1, 4);
#undef G2V_VQP
  G2V_CHECK_LAUNCH();
  return G2V_OK;
}

extern "C" size_t g2v_vq_assign_bulk_workspace(int N, int E, int K) {
  if (N <= 0 || E <= 0 || K <= 0) return 0;
  // [hi image][lo image][counters][|w_k|][-|w_k|^2 / 2][list of undecided rows]
  return 2 * (((size_t)K * E * 2 + 255) & ~(size_t)255) + 256 + 2 * (((size_t)K * 4 + 255) & ~(size_t)255) + (size_t)N * 4;
}

// idx[n] = argmin_k |flat[n] - W[k]|^2 for MANY rows (bulk latent -> code assignment): bf16 split screening + exact fp32
// re-check of the undecided rows (see vq_bx3_sweep_kernel).  undecided (device int, may be NULL) receives their count.
extern "C" int g2v_vq_assign_bulk(const float* flat, const float* codebook, const float* code_sqnorm, int64_t* idx, int N,
                                  int E, int K, void* workspace, size_t workspace_bytes, int* undecided,
                                  g2v_stream_t stream) {
  G2V_REQUIRE(flat && codebook && code_sqnorm && idx && workspace, "null pointer");
  G2V_REQUIRE(N > 0 && E > 0 && K > 0, "non-positive size");
  if (!(E == 128 && (K & 127) == 0 && ptr_aligned16(flat) && ptr_aligned16(codebook) && ptr_aligned16(code_sqnorm) &&
        ptr_aligned16(workspace))) {
    set_error("g2v_vq_assign_bulk: needs E == 128, K %% 128 == 0 and 16-byte aligned operands");
    return G2V_ERR_UNSUPPORTED;
  }
  if (workspace_bytes < g2v_vq_assign_bulk_workspace(N, E, K)) {
    set_error("g2v_vq_assign_bulk: workspace too small");
    return G2V_ERR_WORKSPACE;
  }
  hipStream_t st = (hipStream_t)stream;
  const size_t half = ((size_t)K * E * 2 + 255) & ~(size_t)255;
  char* w = (char*)workspace;
  __bf16* Wh = (__bf16*)w;
  __bf16* Wl = (__bf16*)(w + half);
  int* count = (int*)(w + 2 * half);
  const size_t kpad = ((size_t)K * 4 + 255) & ~(size_t)255;
  float* wn = (float*)(w + 2 * half + 256);
  float* msv = (float*)(w + 2 * half + 256 + kpad);
  int* list = (int*)(w + 2 * half + 256 + 2 * kpad);
  (void)hipMemsetAsync(count, 0, 2 * sizeof(int), st);        // [0] undecided rows, [1] the codebook has a non-finite code
  int kbits = 1;
  while ((1 << kbits) < K) ++kbits;
  G2V_REQUIRE(kbits <= 13, "codebook larger than 8192 codes");
#ifdef G2V_BULK_SWEEP_R5        // A/B build: the round-5 sweep (gpurun_tools/r06_bulk_ab.sh)
  hipLaunchKernelGGL(vq_bx3_split_kernel, dim3(cdiv((int64_t)K * E, 256)), dim3(256), 0, st, codebook, code_sqnorm, Wh, Wl, wn,
                     (int64_t)K * E, K);
  hipLaunchKernelGGL(vq_bx3_sweep_kernel, dim3(cdiv(N, 256)), dim3(256), 0, st, flat, Wh, Wl, code_sqnorm, wn, idx, list, count, N,
                     K, kbits);
#else
  (void)msv;
  hipLaunchKernelGGL(vq_bulk_split_kernel, dim3(cdiv((int64_t)K * E, 256)), dim3(256), 0, st, codebook, code_sqnorm, Wh, Wl, (float2*)wn,
                     count, (int64_t)K * E, K);          // (wn .. : 2 K floats, {-|w_k|^2 / 2, |w_k|} per code)
  hipLaunchKernelGGL(vq_bulk_sweep_kernel, dim3(cdiv(N, 256)), dim3(256), 0, st, flat, Wh, Wl, (const float2*)wn, idx, list, count, N, K,
                     kbits);
#endif
  hipLaunchKernelGGL((vq_assign_rt_kernel<128, 1, true>), dim3(min(cdiv(N, 16), 2048)), dim3(256), 0, st, flat, (const float*)nullptr,
                     codebook, code_sqnorm, idx, (float*)nullptr, (float*)nullptr, (float*)nullptr, N, K, list, count);
  if (undecided) (void)hipMemcpyAsync(undecided, count, sizeof(int), hipMemcpyDeviceToDevice, st);
  G2V_CHECK_LAUNCH();
  return G2V_OK;
}

extern "C" int g2v_vq_fused_assign_fwd(const float* z, const float* w_pre, const float* b_pre, const float* codebook,
                                       const float* code_sqnorm, float* flat_out, int64_t* idx, float* quantized,
                                       float* sse_partial, int N, int E, int K, g2v_stream_t stream) {
  G2V_REQUIRE(z && w_pre && b_pre && codebook && code_sqnorm && flat_out && idx && quantized, "null pointer");
  G2V_REQUIRE(N > 0 && E > 0 && K > 0, "non-positive size");
  if (!(E == 128 && (K & 127) == 0 && ptr_aligned16(z) && ptr_aligned16(w_pre) && ptr_aligned16(b_pre) &&
        ptr_aligned16(codebook) && ptr_aligned16(code_sqnorm) && ptr_aligned16(flat_out) && ptr_aligned16(quantized))) {
    set_error("g2v_vq_fused_assign_fwd: needs E == 128, K %% 128 == 0 and 16-byte aligned operands");
    return G2V_ERR_UNSUPPORTED;
  }
  hipLaunchKernelGGL((vq_fused_assign_kernel<128, false>), dim3(cdiv(N, VQ_ROWS)), dim3(256), 0, (hipStream_t)stream, z, w_pre,
                     b_pre, codebook, (const float*)nullptr, code_sqnorm, flat_out, idx, quantized, sse_partial, N, K);
  G2V_CHECK_LAUNCH();
  return G2V_OK;
}

// fragment-major image of a (K,E) matrix whose 16-row tiles are MFMA A operands: [K/16][E/16][64 lanes][4]; lane (q, i) of
// tile kt, k-step s holds W[16 kt + i][16 s + 4 q .. + 3]
__global__ __launch_bounds__(256) void vq_pack_codebook_kernel(const float* __restrict__ W, float* __restrict__ Wf, int K, int E) {
  const int64_t o = (int64_t)blockIdx.x * 256 + threadIdx.x;         // one 16-byte fragment element per thread
  const int ks = E / 16;
  if (o >= (int64_t)K * E / 4) return;
  const int lane = (int)(o & 63), i = lane & 15, q = lane >> 4;
  const int64_t ts = o >> 6;
  const int s = (int)(ts % ks), kt = (int)(ts / ks);
  reinterpret_cast<float4*>(Wf)[o] = *reinterpret_cast<const float4*>(W + (int64_t)(16 * kt + i) * E + 16 * s + 4 * q);
}

extern "C" int g2v_vq_pack_codebook(const float* codebook, float* codebook_frag, int K, int E, g2v_stream_t stream) {
  G2V_REQUIRE(codebook && codebook_frag, "null pointer");
  G2V_REQUIRE(K > 0 && E > 0 && (K & 15) == 0 && (E & 15) == 0, "K and E must be multiples of 16");
  G2V_REQUIRE(ptr_aligned16(codebook) && ptr_aligned16(codebook_frag), "16-byte aligned operands");
  hipLaunchKernelGGL(vq_pack_codebook_kernel, dim3(cdiv((int64_t)K * E / 4, 256)), dim3(256), 0, (hipStream_t)stream, codebook,
                     codebook_frag, K, E);
  G2V_CHECK_LAUNCH();
  return G2V_OK;
}

extern "C" int g2v_vq_fused_assign_packed_fwd(const float* z, const float* w_pre, const float* b_pre, const float* codebook,
                                              const float* codebook_frag, const float* code_sqnorm, float* flat_out,
                                              int64_t* idx, float* quantized, float* sse_partial, int N, int E, int K,
                                              g2v_stream_t stream) {
  G2V_REQUIRE(z && w_pre && b_pre && codebook && codebook_frag && code_sqnorm && flat_out && idx && quantized, "null pointer");
  G2V_REQUIRE(N > 0 && E > 0 && K > 0, "non-positive size");
  if (!(E == 128 && (K & 127) == 0 && ptr_aligned16(z) && ptr_aligned16(w_pre) && ptr_aligned16(b_pre) &&
        ptr_aligned16(codebook) && ptr_aligned16(codebook_frag) && ptr_aligned16(code_sqnorm) && ptr_aligned16(flat_out) &&
        ptr_aligned16(quantized))) {
    set_error("g2v_vq_fused_assign_packed_fwd: needs E == 128, K %% 128 == 0 and 16-byte aligned operands");
    return G2V_ERR_UNSUPPORTED;
  }
  hipLaunchKernelGGL((vq_fused_assign_kernel<128, true>), dim3(cdiv(N, VQ_ROWS)), dim3(256), 0, (hipStream_t)stream, z, w_pre,
                     b_pre, codebook, codebook_frag, code_sqnorm, flat_out, idx, quantized, sse_partial, N, K);
  G2V_CHECK_LAUNCH();
  return G2V_OK;
}

static inline size_t bx_pad256(size_t n) { return (n + 255) & ~(size_t)255; }
// image layout: [U fragments, bf16: K * E * 2][s'_k: K floats][P_k / Q_k packed: K words][scalars], each part padded to 256 B
extern "C" size_t g2v_vq_bx_image_bytes(int K, int E) {
  if (K <= 0 || E <= 0) return 0;
  return bx_pad256((size_t)K * E * 2) + 2 * bx_pad256((size_t)K * 4) + 256;
}

extern "C" int g2v_vq_fused_assign_bx_ok(int N, int E, int K) { return (N > 0 && E == 128 && (K & 127) == 0 && K >= 128 && K <= 512) ? 1 : 0; }

extern "C" int g2v_vq_bx_pack(const float* codebook, const float* code_sqnorm, const float* w_pre, const float* b_pre, void* image,
                              int K, int E, g2v_stream_t stream) {
  G2V_REQUIRE(codebook && code_sqnorm && w_pre && b_pre && image, "null pointer");
  if (!g2v_vq_fused_assign_bx_ok(1, E, K)) {
    set_error("g2v_vq_bx_pack: needs E == 128, K in {128, 256, 384, 512}");
    return G2V_ERR_UNSUPPORTED;
  }
  G2V_REQUIRE(ptr_aligned16(codebook) && ptr_aligned16(image), "16-byte aligned operands");
  const size_t o1 = bx_pad256((size_t)K * E * 2), o2 = o1 + bx_pad256((size_t)K * 4), o3 = o2 + bx_pad256((size_t)K * 4);
  const size_t pack_lds = ((size_t)E * (E + 1) + 2 * 16 * (E + 4)) * sizeof(float);
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute((const void*)vq_bx_pack_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)pack_lds) != hipSuccess) {
      set_error("g2v_vq_bx_pack: cannot reserve LDS");
      return G2V_ERR_LAUNCH;
    }
    attr_set = true;
  }
  hipLaunchKernelGGL(vq_bx_pack_kernel, dim3(K / 16), dim3(256), pack_lds, (hipStream_t)stream, codebook, code_sqnorm, w_pre, b_pre,
                     (__bf16*)image, (float*)((char*)image + o1), (unsigned*)((char*)image + o2), (BxfScalars*)((char*)image + o3), K);
  G2V_CHECK_LAUNCH();
  return G2V_OK;
}

extern "C" int g2v_vq_fused_assign_bx_fwd(const float* z, const float* w_pre_frag, const float* b_pre, const float* codebook,
                                          const void* image, const float* code_sqnorm, float* flat_out, int64_t* idx,
                                          float* quantized, float* sse_partial, int* diag, int N, int E, int K, int flags,
                                          g2v_stream_t stream) {
  G2V_REQUIRE(z && w_pre_frag && b_pre && codebook && image && code_sqnorm && flat_out && idx && quantized, "null pointer");
  G2V_REQUIRE(N > 0 && E > 0 && K > 0, "non-positive size");
  if (!(g2v_vq_fused_assign_bx_ok(N, E, K) && ptr_aligned16(z) && ptr_aligned16(w_pre_frag) && ptr_aligned16(b_pre) &&
        ptr_aligned16(codebook) && ptr_aligned16(image) && ptr_aligned16(code_sqnorm) && ptr_aligned16(flat_out) &&
        ptr_aligned16(quantized))) {
    set_error("g2v_vq_fused_assign_bx_fwd: needs E == 128, K in {128, 256, 384, 512} and 16-byte aligned operands");
    return G2V_ERR_UNSUPPORTED;
  }
  const size_t o1 = bx_pad256((size_t)K * E * 2), o2 = o1 + bx_pad256((size_t)K * 4), o3 = o2 + bx_pad256((size_t)K * 4);
  if (flags & G2V_VQ_BX_EXACT)
    hipLaunchKernelGGL(vq_fused_bx_exact_kernel, dim3(cdiv(N, VQ_ROWS)), dim3(512), 0, (hipStream_t)stream, z, w_pre_frag, b_pre,
                       codebook, (const __bf16*)image, (const float*)((const char*)image + o1),
                       (const unsigned*)((const char*)image + o2), (const BxfScalars*)((const char*)image + o3), code_sqnorm,
                       flat_out, idx, quantized, sse_partial, diag, N, K);
  else
    hipLaunchKernelGGL(vq_fused_bx_kernel, dim3(cdiv(N, VQ_ROWS)), dim3(512), 0, (hipStream_t)stream, z, w_pre_frag, b_pre,
                       codebook, (const __bf16*)image, (const float*)((const char*)image + o1),
                       (const unsigned*)((const char*)image + o2), (const BxfScalars*)((const char*)image + o3), code_sqnorm,
                       flat_out, idx, quantized, sse_partial, diag, N, K);
  G2V_CHECK_LAUNCH();
  return G2V_OK;
}

extern "C" size_t g2v_vq_stats_workspace(int N, int E, int K) {
  if (N <= 0 || E <= 0 || K <= 0) return 0;
  return (size_t)stats_splits(N, E, K) * ((size_t)K * E + K) * sizeof(float);
}

extern "C" int g2v_vq_stats(const int64_t* idx, const float* flat, float* stats, int N, int E, int K, void* workspace,
                            size_t workspace_bytes, g2v_stream_t stream) {
  G2V_REQUIRE(idx && flat && stats && workspace, "null pointer");
  G2V_REQUIRE(N > 0 && E > 0 && K > 0, "non-positive size");
  if (workspace_bytes < g2v_vq_stats_workspace(N, E, K)) {
    set_error("g2v_vq_stats: workspace too small");
    return G2V_ERR_WORKSPACE;
  }
  // enough rows for a whole-chip launch of tile owners (K / 16 x E / 16 workgroups, every one walking all N rows):
  // one launch, no slabs.  Small problems keep the split kernel (a tile owner would leave most CUs idle).
  if ((E & 15) == 0 && (K & 15) == 0 && N >= 1024 && (K / 16) * (E / 16) >= 128) {
    hipLaunchKernelGGL(vq_stats_owner_kernel, dim3(E / 16, K / 16), dim3(256), 0, (hipStream_t)stream, idx, flat, stats,
                       stats + K, N, E, K);
    G2V_CHECK_LAUNCH();
    return G2V_OK;
  }
  const int splits = stats_splits(N, E, K);
  const int rows_per_split = round_up(cdiv(N, splits), SM);
  float* slab_dw = (float*)workspace;
  float* slab_cnt = slab_dw + (size_t)splits * K * E;
  hipLaunchKernelGGL(vq_stats_kernel, dim3(cdiv(K, 64), cdiv(E, 64), splits), dim3(256), 0, (hipStream_t)stream, idx,
                     flat, slab_dw, slab_cnt, N, E, K, rows_per_split);
  G2V_CHECK_LAUNCH();
  launch_slab_reduce(slab_cnt, splits, (int64_t)K, stats, 0, (hipStream_t)stream);
  G2V_CHECK_LAUNCH();
  launch_slab_reduce(slab_dw, splits, (int64_t)K * E, stats + K, 0, (hipStream_t)stream);
  G2V_CHECK_LAUNCH();
  return G2V_OK;
}

extern "C" int g2v_vq_ema_update(const float* stats, const float* sse_partial, int n_sse_partial,
                                 float* ema_cluster_size, float* ema_w, float* codebook, float* code_sqnorm,
                                 float* scalars, int N_loss, int N_cnt, int E, int K, float beta, float decay,
                                 float eps, int update, g2v_stream_t stream) {
  G2V_REQUIRE(stats && scalars, "null pointer");
  G2V_REQUIRE(!update || (ema_cluster_size && ema_w && codebook && code_sqnorm), "null EMA state");
  G2V_REQUIRE(N_loss > 0 && N_cnt > 0 && E > 0 && K > 0, "non-positive size");
  const unsigned* fault = g2v_internal_persist_fault_ptr();
  hipLaunchKernelGGL(vq_ema_scalars_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, stats, sse_partial,
                     sse_partial ? n_sse_partial : 0, ema_cluster_size, scalars, N_loss, N_cnt, E, K, beta, decay, eps,
                     update, fault);
  G2V_CHECK_LAUNCH();
  if (update) {
    hipLaunchKernelGGL(vq_ema_rows_kernel, dim3(cdiv((int64_t)K * 64, 256)), dim3(256), 0, (hipStream_t)stream, stats,
                       ema_cluster_size, ema_w, codebook, code_sqnorm, E, K, decay, fault);
    G2V_CHECK_LAUNCH();
  }
  return G2V_OK;
}

extern "C" int g2v_vq_bwd(const float* g_quantized, const float* g_loss, const float* z, const float* codebook,
                          const int64_t* idx, float* gz, int N, int E, float beta, g2v_stream_t stream) {
  G2V_REQUIRE(z && codebook && gz, "null pointer");
  G2V_REQUIRE(N > 0 && E > 0, "non-positive size");
  const int64_t total = (int64_t)N * E;
  const float coef = 2.0f * beta / ((float)N * (float)E);
  int blocks = cdiv(total, 256);
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(vq_bwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, g_quantized, g_loss, z, codebook,
                     idx, gz, total, E, coef);
  G2V_CHECK_LAUNCH();
  return G2V_OK;
}

extern "C" int g2v_vq_soft_fwd(const float* flat, float* dots_to_dist, const float* logvar, const float* code_sqnorm,
                               float* probs, float* perplexity, int N, int E, int K, g2v_stream_t stream) {
  G2V_REQUIRE(flat && dots_to_dist && logvar && code_sqnorm && probs, "null pointer");
  G2V_REQUIRE(N > 0 && E > 0 && K > 0, "non-positive size");
  hipLaunchKernelGGL(vq_soft_fwd_kernel, dim3(cdiv(N, 4)), dim3(256), 0, (hipStream_t)stream, flat, dots_to_dist, logvar,
                     code_sqnorm, probs, N, E, K);
  G2V_CHECK_LAUNCH();
  if (perplexity) {
    hipLaunchKernelGGL(vq_soft_perplexity_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, probs, perplexity, N, K);
    G2V_CHECK_LAUNCH();
  }
  return G2V_OK;
}

extern "C" size_t g2v_vq_soft_perplexity_workspace(int N, int K) {
  return (N > 0 && K > 0) ? (size_t)cdiv(N, PERP_ROWS) * K * sizeof(float) : 0;
}
extern "C" int g2v_vq_soft_perplexity(const float* probs, float* perplexity, int N, int K, void* workspace, size_t workspace_bytes,
                                      g2v_stream_t stream) {
  G2V_REQUIRE(probs && perplexity && workspace, "null pointer");
  G2V_REQUIRE(N > 0 && K > 0, "non-positive size");
  if (workspace_bytes < g2v_vq_soft_perplexity_workspace(N, K)) {
    set_error("g2v_vq_soft_perplexity: workspace too small");
    return G2V_ERR_WORKSPACE;
  }
  const int nblk = cdiv(N, PERP_ROWS);
  hipLaunchKernelGGL(vq_soft_colsum_kernel, dim3(nblk), dim3(256), 0, (hipStream_t)stream, probs, (float*)workspace, N, K);
  G2V_CHECK_LAUNCH();
  hipLaunchKernelGGL(vq_soft_perplexity_finish_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, (const float*)workspace,
                     perplexity, nblk, N, K);
  G2V_CHECK_LAUNCH();
  return G2V_OK;
}

extern "C" int g2v_vq_soft_bwd(const float* probs, const float* dprobs, const float* dist, const float* logvar, float* dd,
                               float* dlogvar, float* rowsum, int N, int K, g2v_stream_t stream) {
  G2V_REQUIRE(probs && dprobs && dist && logvar && dd && dlogvar && rowsum, "null pointer");
  G2V_REQUIRE(N > 0 && K > 0, "non-positive size");
  hipLaunchKernelGGL(vq_soft_bwd_kernel, dim3(cdiv(N, 4)), dim3(256), 0, (hipStream_t)stream, probs, dprobs, dist, logvar,
                     dd, dlogvar, rowsum, N, K);
  G2V_CHECK_LAUNCH();
  return G2V_OK;
}

extern "C" int g2v_rowscale_combine(const float* a, const float* v, const float* t, float* out, int64_t rows, int cols,
                                    g2v_stream_t stream) {
  G2V_REQUIRE(a && v && t && out, "null pointer");
  G2V_REQUIRE(rows > 0 && cols > 0, "non-positive size");
  int blocks = cdiv(rows * cols, 256);
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(rowscale_combine_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a, v, t, out, rows, cols);
  G2V_CHECK_LAUNCH();
  return G2V_OK;
}

extern "C" int g2v_ste_f32(const float* z, const float* q, float* out, int64_t n, g2v_stream_t stream) {
  G2V_REQUIRE(z && q && out, "null pointer");
  G2V_REQUIRE(n > 0, "non-positive size");
  int blocks = cdiv(n, 256);
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(ste_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, z, q, out, n);
  G2V_CHECK_LAUNCH();
  return G2V_OK;
}
