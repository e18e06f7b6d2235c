// linear.hip -- dense-layer forward / data-gradient / weight-gradient on fp32 MFMA (gfx950).
//
// Replaces the nn.Linear call sites named in include/g2v.h (EncoderRNN.in_layer
// Autoencoder_VQVAE_model.py:93, VQ_Payam_EMA.pre_linear :1230, nn.GRU input projections,
// DAE_Network model/DAE_model.py:107-110) and all their autograd matmuls.
//
// Shapes on this path are "tall and thin": M = rows (T*B up to ~1.4e5), K,N <= ~600.  Both
// kernels tile 64x64 outputs per 256-thread workgroup, stage operands through LDS and issue
// v_mfma_f32_16x16x4_f32.  They are HBM-bound (one read of x, one write of y); the weight tile
// is re-read from L2.
#include "common.hpp"

namespace g2v {

struct RowMap {
  int64_t ld;
  int rows_inner;
  int64_t so, si;
};
__device__ __forceinline__ int64_t row_off(const RowMap& m, int r) {
  return m.rows_inner > 0 ? (int64_t)(r / m.rows_inner) * m.so + (int64_t)(r % m.rows_inner) * m.si
                          : (int64_t)r * m.ld;
}

constexpr int BM = 64, BN = 64, BC = 32, LDT = BC + 4;

// C[m][n] (+)= act( sum_c A[m][c] * Bop[n][c] + bias[n] )
//   TRANS_B == false: Bop[n][c] = Bm[n*ldb + c]     (forward: Bm = w [N][K])
//   TRANS_B == true : Bop[n][c] = Bm[c*ldb + n]     (data gradient: Bm = w [N][K], output feature = k)
// (Round 4, measured and dropped: 64-wide chunks with the global loads two chunks ahead.  35 KB of LDS and twice the staging
//  registers per workgroup cost more co-resident workgroups -- which is what hides a chunk's memory round trip here -- than the
//  longer lead bought: native shape 7.97 -> 8.24 ms, Part d at B = 4096 6.79 -> 7.11 ms.)
template <bool TRANS_B, bool VEC = false>
__global__ __launch_bounds__(256) void gemm_nt_kernel(const float* __restrict__ A, RowMap am,
                                                      const uint8_t* __restrict__ keep, float scale,
                                                      const float* __restrict__ Bm, int64_t ldb,
                                                      const float* __restrict__ bias, float* __restrict__ Cout,
                                                      int64_t ldc, int M, int C, int N, int act, int accumulate,
                                                      const float* __restrict__ Bm_z1 = nullptr,
                                                      const float* __restrict__ bias_z1 = nullptr,
                                                      float* __restrict__ Cout_z1 = nullptr) {
  __shared__ __attribute__((aligned(16))) float As[BM * LDT];
  __shared__ __attribute__((aligned(16))) float Bs[BN * LDT];
  if (blockIdx.z) {      // grid.z = 2 (g2v_linear_fwd_pair): the same A against a second (weight, bias, output)
    Bm = Bm_z1; bias = bias_z1; Cout = Cout_z1;
  }
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
  const int i = lane & 15, q = lane >> 4;

  f32x4 acc[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // Software pipeline: the global loads of chunk c+1 are issued (into registers) right after chunk c has been written to LDS,
  // so their L2 / HBM round trip runs under the MFMAs of chunk c instead of in front of them.
  if constexpr (VEC) {
    // 16-byte-aligned operands, C % 4 == 0 (and N % 4 == 0 when transposed), no keep mask: one float4 per (row, 4 columns);
    // thread -> row tid >> 3 (+32), columns 4 (tid & 7) of the 32-wide chunk; transposed B: c = tid >> 4 (+16), n = 4 (tid & 15)
    float4 va[2], vb[2];
    const int vr = tid >> 3, vc = 4 * (tid & 7);
    const int wc = tid >> 4, wn = 4 * (tid & 15);
    int64_t ao[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int m = m0 + vr + 32 * h;
      ao[h] = m < M ? row_off(am, m) : -1;
    }
    auto vfetch = [&](int c0) {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int cc = c0 + vc;
        va[h] = ld4_or_zero(A + (ao[h] >= 0 ? ao[h] : 0) + (cc < C ? cc : 0), ao[h] >= 0 && cc < C);
        if (!TRANS_B) {
          const int n = n0 + vr + 32 * h;
          vb[h] = ld4_or_zero(Bm + (int64_t)(n < N ? n : 0) * ldb + (cc < C ? cc : 0), n < N && cc < C);
        } else {
          const int c = c0 + wc + 16 * h, n = n0 + wn;
          vb[h] = ld4_or_zero(Bm + (int64_t)(c < C ? c : 0) * ldb + (n < N ? n : 0), c < C && n < N);
        }
      }
    };
    vfetch(0);
    for (int c0 = 0; c0 < C; c0 += BC) {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        *reinterpret_cast<float4*>(&As[(vr + 32 * h) * LDT + vc]) = va[h];
        if (!TRANS_B) {
          *reinterpret_cast<float4*>(&Bs[(vr + 32 * h) * LDT + vc]) = vb[h];
        } else {
          const int c = wc + 16 * h;
          Bs[(wn + 0) * LDT + c] = vb[h].x; Bs[(wn + 1) * LDT + c] = vb[h].y;
          Bs[(wn + 2) * LDT + c] = vb[h].z; Bs[(wn + 3) * LDT + c] = vb[h].w;
        }
      }
      __syncthreads();
      if (c0 + BC < C) vfetch(c0 + BC);
#pragma unroll
      for (int k0 = 0; k0 < BC; k0 += 16) {
        const float4 xb = *reinterpret_cast<const float4*>(&As[(16 * wave + i) * LDT + k0 + 4 * q]);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const float4 wa = *reinterpret_cast<const float4*>(&Bs[(16 * t + i) * LDT + k0 + 4 * q]);
          acc[t] = mfma16(wa.x, xb.x, acc[t]);
          acc[t] = mfma16(wa.y, xb.y, acc[t]);
          acc[t] = mfma16(wa.z, xb.z, acc[t]);
          acc[t] = mfma16(wa.w, xb.w, acc[t]);
        }
      }
      __syncthreads();
    }
  } else {
  float ra[8], rb[8];
  const int lc = tid & 31, lr = tid >> 5;             // A (and non-transposed B): 32 consecutive threads along c
  const int tr = tid & 63, tc = tid >> 6;             // transposed B: 64 consecutive threads along n
  int64_t arow[8];
#pragma unroll
  for (int it = 0; it < 8; ++it) {
    const int m = m0 + lr + 8 * it;
    arow[it] = m < M ? row_off(am, m) : -1;
  }
  auto fetch = [&](int c0) {
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const int m = m0 + lr + 8 * it, cc = c0 + lc;
      float v = 0.f;
      if (arow[it] >= 0 && cc < C) {
        v = A[arow[it] + cc];
        if (keep) v = keep[(int64_t)m * C + cc] ? v * scale : 0.f;
      }
      ra[it] = v;
    }
    if (!TRANS_B) {
#pragma unroll
      for (int it = 0; it < 8; ++it) {
        const int n = n0 + lr + 8 * it, cc = c0 + lc;
        rb[it] = (n < N && cc < C) ? Bm[(int64_t)n * ldb + cc] : 0.f;
      }
    } else {
#pragma unroll
      for (int it = 0; it < 8; ++it) {
        const int n = n0 + tr, cc = c0 + tc + 4 * it;
        rb[it] = (n < N && cc < C) ? Bm[(int64_t)cc * ldb + n] : 0.f;
      }
    }
  };
  fetch(0);
  for (int c0 = 0; c0 < C; c0 += BC) {
#pragma unroll
    for (int it = 0; it < 8; ++it) As[(lr + 8 * it) * LDT + lc] = ra[it];
    if (!TRANS_B) {
#pragma unroll
      for (int it = 0; it < 8; ++it) Bs[(lr + 8 * it) * LDT + lc] = rb[it];
    } else {
#pragma unroll
      for (int it = 0; it < 8; ++it) Bs[tr * LDT + tc + 4 * it] = rb[it];
    }
    __syncthreads();
    if (c0 + BC < C) fetch(c0 + BC);
#pragma unroll
    for (int k0 = 0; k0 < BC; k0 += 16) {
      const float4 xb = *reinterpret_cast<const float4*>(&As[(16 * wave + i) * LDT + k0 + 4 * q]);
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const float4 wa = *reinterpret_cast<const float4*>(&Bs[(16 * t + i) * LDT + k0 + 4 * q]);
        acc[t] = mfma16(wa.x, xb.x, acc[t]);
        acc[t] = mfma16(wa.y, xb.y, acc[t]);
        acc[t] = mfma16(wa.z, xb.z, acc[t]);
        acc[t] = mfma16(wa.w, xb.w, acc[t]);
      }
    }
    __syncthreads();
  }
  }

  // epilogue: lane holds row m = m0 + 16*wave + (lane&15), features n0 + 16t + 4q + r
  const int m = m0 + 16 * wave + i;
  if (m >= M) return;
  float* crow = Cout + (int64_t)m * ldc;
  const bool cvec = ptr_vec_ok(Cout, ldc);
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int n = n0 + 16 * t + 4 * q;
    if (n >= N) continue;
    float v[4] = {acc[t][0], acc[t][1], acc[t][2], acc[t][3]};
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      if (n + r < N) {
        if (bias) v[r] += bias[n + r];
        if (act == 1) v[r] = fmaxf(v[r], 0.f);
        else if (act == 2) v[r] = tanhf(v[r]);
        if (accumulate) v[r] += crow[n + r];
      }
    }
    if (cvec && n + 3 < N) {
      *reinterpret_cast<float4*>(crow + n) = make_float4(v[0], v[1], v[2], v[3]);
    } else {
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (n + r < N) crow[n + r] = v[r];
    }
  }
}

// ---- wave-autonomous streaming forward / data-gradient kernel ----------------------------------------------------------
// For 16-byte-aligned A rows without row map / keep mask (the GRU input projections, the dX products, pre_linear):
// the (<= 192 x C) weight block is packed in MFMA fragment order into LDS ONCE per workgroup; after that there is NO
// workgroup barrier: every wave walks its own 16-row groups, pulling its MFMA B-fragments straight from global memory
// into registers (the group after next is prefetched while the current one is multiplied), reading the weight fragments
// from LDS with ds_read_b128, and storing its accumulators.  Waves drift apart, so the matrix pipe of a SIMD is shared
// smoothly between its resident waves instead of all waves hitting the same phase at once.
template <int NTW, int KS, bool TRANS_B>
__global__ __launch_bounds__(256) void gemm_nt_stream_kernel(const float* __restrict__ A, int64_t lda,
                                                             const float* __restrict__ Bm, int64_t ldb,
                                                             const float* __restrict__ bias, float* __restrict__ Cout,
                                                             int64_t ldc, int M, int C, int N, int act, int accumulate) {
  extern __shared__ __attribute__((aligned(16))) float smem[];   // packed W: [NTW][KS][64 lanes][4]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i = lane & 15, q = lane >> 4;
  const bool wvec = ((ldb & 3) == 0) && ((reinterpret_cast<uintptr_t>(Bm) & 15) == 0);
  // ---- pack the weights: P[((t*KS + s)*64 + l)*4 + e] = Wop[16 t + (l & 15)][16 s + 4 (l >> 4) + e] ----
  for (int blk = wave; blk < NTW * KS; blk += 4) {
    const int t = blk / KS, s2 = blk - t * KS;
    const int n = 16 * t + i;
    float v[4];
    if (!TRANS_B && wvec) {
      // forward: the fragment IS 4 consecutive weights of row n: one 16-byte load
      const int c = 16 * s2 + 4 * q;
      const bool ok = (n < N) && (c < C);               // C % 4 == 0: the float4 is all-in or all-out
      const float4 w4 = *reinterpret_cast<const float4*>(Bm + (int64_t)(ok ? n : 0) * ldb + (ok ? c : 0));
      v[0] = ok ? w4.x : 0.f; v[1] = ok ? w4.y : 0.f; v[2] = ok ? w4.z : 0.f; v[3] = ok ? w4.w : 0.f;
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int c = 16 * s2 + 4 * q + e;
        const bool ok = (n < N) && (c < C);
        const int64_t off = TRANS_B ? ((int64_t)(ok ? c : 0) * ldb + (ok ? n : 0)) : ((int64_t)(ok ? n : 0) * ldb + (ok ? c : 0));
        const float w = Bm[off];
        v[e] = ok ? w : 0.f;
      }
    }
    *reinterpret_cast<float4*>(smem + ((int64_t)blk * 64 + lane) * 4) = make_float4(v[0], v[1], v[2], v[3]);
  }
  float4 bia[NTW];
#pragma unroll
  for (int t = 0; t < NTW; ++t) {
    bia[t] = make_float4(0.f, 0.f, 0.f, 0.f);
    const int n = 16 * t + 4 * q;
    if (bias && n + 3 < N) bia[t] = *reinterpret_cast<const float4*>(bias + n);
  }
  __syncthreads();
  const int ngroups = (M + 15) >> 4;
  const int gstride = gridDim.x * 4;
  int g = blockIdx.x * 4 + wave;
  float4 xa[KS], xn[KS];
  auto load_group = [&](int grp, float4 (&x)[KS]) {
    const int m = 16 * grp + i;
    const bool ok = m < M;
    const float* p = A + (int64_t)(ok ? m : 0) * lda + 4 * q;
#pragma unroll
    for (int s2 = 0; s2 < KS; ++s2) {
      float4 v = (16 * s2 + 4 * q < C) ? *reinterpret_cast<const float4*>(p + 16 * s2) : make_float4(0.f, 0.f, 0.f, 0.f);
      x[s2] = ok ? v : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  if (g < ngroups) load_group(g, xa);
  for (; g < ngroups; g += gstride) {
    const int gn = g + gstride;
    if (gn < ngroups) load_group(gn, xn);
    const int m = 16 * g + i;
    float* crow = Cout + (int64_t)(m < M ? m : 0) * ldc;
#pragma unroll
    for (int t = 0; t < NTW; ++t) {
      f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
      const float* wp = smem + ((int64_t)t * KS * 64 + lane) * 4;
#pragma unroll
      for (int s2 = 0; s2 < KS; ++s2) {
        const float4 wa = *reinterpret_cast<const float4*>(wp + s2 * 256);
        acc = mfma16(wa.x, xa[s2].x, acc);
        acc = mfma16(wa.y, xa[s2].y, acc);
        acc = mfma16(wa.z, xa[s2].z, acc);
        acc = mfma16(wa.w, xa[s2].w, acc);
      }
      const int n = 16 * t + 4 * q;
      if (m < M && n + 3 < N) {
        float4 v = make_float4(acc[0] + bia[t].x, acc[1] + bia[t].y, acc[2] + bia[t].z, acc[3] + bia[t].w);
        if (act == 1) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
        if (accumulate) {
          const float4 o = *reinterpret_cast<const float4*>(crow + n);
          v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
        }
        *reinterpret_cast<float4*>(crow + n) = v;
      }
    }
#pragma unroll
    for (int s2 = 0; s2 < KS; ++s2) xa[s2] = xn[s2];
  }
}

// ---- small / medium M (a decode step of Part d: 128..4096 x 200 -> 600): one wave per 16 x 16 TN output tile, no LDS ------
// At M <= 512 rows the 64 x 64 LDS-tiled kernel launches a handful of workgroups that each walk the contraction in 32-wide
// chunks with two barriers per chunk: 11-27 us for 15 MFLOP.  Here every 16 x 16 tile of the output is one wave (304 waves
// for 128 x 600) that pulls its operand fragments straight from L2 in MFMA layout -- lane (i, q) reads 4 consecutive
// contraction elements of A row m0 + i and of weight row n0 + i per 16-wide block -- with the next chunk's loads issued
// before the current chunk's MFMAs.  TRANS_B (data gradient, contraction along the rows of w): dword loads of w.
// Requirements: contraction length % 4 == 0, 16-byte aligned rows; identity / ReLU / tanh epilogue, optional keep mask on A.
// out_keep (uint8, [m * N + n], may be NULL): the OUTPUT is multiplied by keep * out_scale (dropout on the quantity whose
// gradient this is: dx = (dgi W_ih) * keep * scale).
template <bool TRANS_B, bool KEEP, int TN, int NB>      // wave tile: 16 rows x 16 TN columns; NB blocks of 16 contraction elements per chunk
__device__ __forceinline__ void smallm_body(const float* __restrict__ A, int64_t lda, const uint8_t* __restrict__ keep,
                                            float scale, const float* __restrict__ Bm, int64_t ldb,
                                            const float* __restrict__ bias, float* __restrict__ Cout, int64_t ldc, int M, int C,
                                            int N, int act, int accumulate, int tile, const uint8_t* __restrict__ out_keep,
                                            float out_scale) {
  constexpr int CH = 16 * NB;
  const int lane = threadIdx.x & 63, i = lane & 15, q = lane >> 4;
  const int tiles_n = (N + 16 * TN - 1) / (16 * TN);
  // consecutive waves share the A rows (same mt) and walk the weight rows: the 4 waves of a workgroup re-use A through L1/L2
  const int mt = tile / tiles_n, nt = tile - mt * tiles_n;
  if (mt * 16 >= M) return;
  const int m = mt * 16 + i;
  const bool mok = m < M;
  const float* arow = A + (int64_t)(mok ? m : 0) * lda;
  const uint8_t* krow = KEEP ? keep + (int64_t)(mok ? m : 0) * C : nullptr;
  const float* brow[TN];
  bool nok[TN];
#pragma unroll
  for (int t = 0; t < TN; ++t) {
    const int n = (nt * TN + t) * 16 + i;
    nok[t] = n < N;
    brow[t] = TRANS_B ? Bm + (nok[t] ? n : 0) : Bm + (int64_t)(nok[t] ? n : 0) * ldb;
  }
  f32x4 acc[TN];
#pragma unroll
  for (int t = 0; t < TN; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float4 xa[NB], wa[TN][NB], xb[NB], wb[TN][NB];
  uint32_t ka[NB], kb[NB];
  auto fetch = [&](int c0, float4 (&x)[NB], float4 (&w)[TN][NB], uint32_t (&kp)[NB]) {
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      const int c = c0 + 16 * b + 4 * q, cc = c < C ? c : 0;          // past the end: a valid address, zeroed in use()
      x[b] = *reinterpret_cast<const float4*>(arow + cc);
      if (KEEP) kp[b] = *reinterpret_cast<const uint32_t*>(krow + cc);
#pragma unroll
      for (int t = 0; t < TN; ++t) {
        if (!TRANS_B) {
          w[t][b] = *reinterpret_cast<const float4*>(brow[t] + cc);
        } else {
          w[t][b] = make_float4(brow[t][(int64_t)cc * ldb], brow[t][(int64_t)(cc + 1) * ldb],
                                brow[t][(int64_t)(cc + 2) * ldb], brow[t][(int64_t)(cc + 3) * ldb]);
        }
      }
    }
  };
  auto use = [&](int c0, const float4 (&x)[NB], const float4 (&w)[TN][NB], const uint32_t (&kp)[NB]) {
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      const bool ok = c0 + 16 * b + 4 * q < C;
      float4 xv = x[b];
      if (KEEP) {
        xv.x = (kp[b] & 0xffu) ? xv.x * scale : 0.f;
        xv.y = (kp[b] & 0xff00u) ? xv.y * scale : 0.f;
        xv.z = (kp[b] & 0xff0000u) ? xv.z * scale : 0.f;
        xv.w = (kp[b] & 0xff000000u) ? xv.w * scale : 0.f;
      }
      if (!(ok && mok)) xv = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int t = 0; t < TN; ++t) {
        float4 wv = w[t][b];
        if (!(ok && nok[t])) wv = make_float4(0.f, 0.f, 0.f, 0.f);
        acc[t] = mfma16(wv.x, xv.x, acc[t]);
        acc[t] = mfma16(wv.y, xv.y, acc[t]);
        acc[t] = mfma16(wv.z, xv.z, acc[t]);
        acc[t] = mfma16(wv.w, xv.w, acc[t]);
      }
    }
  };
  fetch(0, xa, wa, ka);
  for (int c0 = 0; c0 < C; c0 += 2 * CH) {
    if (c0 + CH < C) fetch(c0 + CH, xb, wb, kb);
    use(c0, xa, wa, ka);
    if (c0 + CH >= C) break;
    if (c0 + 2 * CH < C) fetch(c0 + 2 * CH, xa, wa, ka);
    use(c0 + CH, xb, wb, kb);
  }
  // lane holds C[m = mt*16 + (lane & 15)][n = (nt TN + t)*16 + 4 q + r], r = 0..3
  if (!mok) return;
#pragma unroll
  for (int t = 0; t < TN; ++t) {
    const int no = (nt * TN + t) * 16 + 4 * q;
    float* out = Cout + (int64_t)m * ldc + no;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      if (no + r < N) {
        float v = acc[t][r] + (bias ? bias[no + r] : 0.f);
        if (act == 1) v = v > 0.f ? v : 0.f;
        else if (act == 2) v = tanhf(v);
        if (out_keep) v = out_keep[(int64_t)m * N + no + r] ? v * out_scale : 0.f;
        out[r] = accumulate ? out[r] + v : v;
      }
    }
  }
}

template <bool TRANS_B, bool KEEP, int TN, int NB>
__global__ __launch_bounds__(256) void gemm_smallm_kernel(const float* __restrict__ A, int64_t lda,
                                                          const uint8_t* __restrict__ keep, float scale,
                                                          const float* __restrict__ Bm, int64_t ldb,
                                                          const float* __restrict__ bias, float* __restrict__ Cout,
                                                          int64_t ldc, int M, int C, int N, int act, int accumulate) {
  smallm_body<TRANS_B, KEEP, TN, NB>(A, lda, keep, scale, Bm, ldb, bias, Cout, ldc, M, C, N, act, accumulate,
                                     blockIdx.x * 4 + (threadIdx.x >> 6), nullptr, 1.0f);
}

// Two data-gradient products in one launch (blockIdx.y): out_p (+)= A_p W_p with W_p (C x N_p) row-major, contraction over
// its rows -- the backward of a GRU cell: d_hprev += dgh W_hh and dx = (dgi W_ih) * keep * scale (g2v_gru_cell_bwd).
struct DualTrans {
  const float* A[2];
  const float* W[2];
  float* out[2];
  int N[2];
  int accumulate[2];
  const uint8_t* out_keep[2];
  float out_scale[2];
};
__global__ __launch_bounds__(256) void gemm_smallm_dual_kernel(DualTrans d, int M, int C) {
  const int p = blockIdx.y;
  smallm_body<true, false, 1, 8>(d.A[p], (int64_t)C, nullptr, 1.0f, d.W[p], (int64_t)d.N[p], nullptr, d.out[p], (int64_t)d.N[p], M, C,
                                 d.N[p], 0, d.accumulate[p], blockIdx.x * 4 + (threadIdx.x >> 6), d.out_keep[p], d.out_scale[p]);
}

// Two forward products of the SAME rows in one launch (blockIdx.y): y_p = x W_p^T + b_p -- a decode step of Part d with attention
// reads the new top state twice, for the logits and for the next step's attention query (g2v_linear_fwd_dual).
struct DualNT {
  const float* W[2];
  const float* bias[2];
  float* out[2];
  int64_t ldc[2];
  int N[2];
};
__global__ __launch_bounds__(256) void gemm_smallm_dual_nt_kernel(const float* __restrict__ A, int64_t lda, DualNT d, int M, int C) {
  const int p = blockIdx.y;
  smallm_body<false, false, 1, 8>(A, lda, nullptr, 1.0f, d.W[p], (int64_t)C, d.bias[p], d.out[p], d.ldc[p], M, C, d.N[p], 0, 0,
                                  blockIdx.x * 4 + (threadIdx.x >> 6), nullptr, 1.0f);
}

// measured (gpurun_tools/gemm_bench.py, 200..400 -> 200..600): forward 9.7 vs 10.4 us and data gradient 11 vs 22 us at 128..640
// rows, break-even near 2560 rows, the LDS-tiled kernel ahead beyond (50..86 TF/s at 4096..81920 rows)
extern "C" int g2v_linear_set_smallm_rows(int rows) {            // = g2v_ctx_set_option(NULL, G2V_OPT_SMALLM_ROWS, rows) (measurement only)
  return g2v_ctx_set_option(nullptr, G2V_OPT_SMALLM_ROWS, rows);
}

static bool launch_smallm(bool trans_b, const float* A, int64_t lda, const uint8_t* keep, float scale, const float* Bm,
                          int64_t ldb, const float* bias, float* Cout, int64_t ldc, int M, int C, int N, int act,
                          int accumulate, hipStream_t st) {
  if (M > g2v_internal_options().smallm_max_rows || (C & 3) || (lda & 3) || (reinterpret_cast<uintptr_t>(A) & 15)) return false;
  if (!trans_b && ((ldb & 3) || (reinterpret_cast<uintptr_t>(Bm) & 15))) return false;
  if (keep && (reinterpret_cast<uintptr_t>(keep) & 3)) return false;
  // wider wave tiles (more re-use of the A fragment, fewer waves) once 16 x 16 tiles alone fill the chip a few times over
  const int64_t tiles16 = (int64_t)cdiv(M, 16) * cdiv(N, 16);
  const int tn = tiles16 <= 2048 ? 1 : (tiles16 <= 8192 ? 2 : 4);
  const int tiles = cdiv(M, 16) * cdiv(N, 16 * tn);
  const dim3 grid(cdiv(tiles, 4));
#define G2V_SMALLM(TB, KP, TN, NB)                                                                                        \
  hipLaunchKernelGGL((gemm_smallm_kernel<TB, KP, TN, NB>), grid, dim3(256), 0, st, A, lda, keep, scale, Bm, ldb, bias,  \
                     Cout, ldc, M, C, N, act, accumulate)
#define G2V_SMALLM_TN(TB, KP)                     \
  do {                                            \
    if (tn == 1) G2V_SMALLM(TB, KP, 1, 8);        \
    else if (tn == 2) G2V_SMALLM(TB, KP, 2, 4);   \
    else G2V_SMALLM(TB, KP, 4, 4);                \
  } while (0)
  if (trans_b) { if (keep) G2V_SMALLM_TN(true, true); else G2V_SMALLM_TN(true, false); }
  else { if (keep) G2V_SMALLM_TN(false, true); else G2V_SMALLM_TN(false, false); }
#undef G2V_SMALLM_TN
#undef G2V_SMALLM
  return true;
}

}  // namespace g2v
// used by g2v_gru_cell_bwd (gru.hip): d_hprev (B,H) += dgh W_hh, dx (B,in_dim) = (dgi W_ih) * keep * scale; C = 3H
int g2v_internal_cell_bwd_products(const float* dgh, const float* w_hh, float* d_hprev, int H, const float* dgi,
                                   const float* w_ih, float* dx, int in_dim, const uint8_t* x_keep, float x_scale, int B,
                                   hipStream_t st) {
  using namespace g2v;
  DualTrans d;
  d.A[0] = dgh; d.W[0] = w_hh; d.out[0] = d_hprev; d.N[0] = H; d.accumulate[0] = 1; d.out_keep[0] = nullptr; d.out_scale[0] = 1.0f;
  d.A[1] = dgi; d.W[1] = w_ih; d.out[1] = dx; d.N[1] = in_dim; d.accumulate[1] = 0; d.out_keep[1] = x_keep; d.out_scale[1] = x_scale;
  const int nprob = dx ? 2 : 1;
  const int tmax = cdiv(B, 16) * cdiv(nprob == 2 && in_dim > H ? in_dim : H, 16);
  hipLaunchKernelGGL(gemm_smallm_dual_kernel, dim3(cdiv(tmax, 4), nprob), dim3(256), 0, st, d, B, 3 * H);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}
namespace g2v {

template <bool TRANS_B>
static bool launch_stream(const float* A, int64_t lda, const float* Bm, int64_t ldb, const float* bias, float* Cout,
                          int64_t ldc, int M, int C, int N, int act, int accumulate, hipStream_t st) {
  // eligibility: aligned rows, whole float4s, N a multiple of 16 up to 192, identity / ReLU epilogue, enough rows
  if (M < 1024 || (C & 3) || (N & 15) || N > 192 || act == 2) return false;
  if (!ptr_vec_ok(A, lda) || !ptr_vec_ok(Cout, ldc) || (bias && (reinterpret_cast<uintptr_t>(bias) & 15))) return false;
  const int ks = (C + 15) >> 4, ntw = N >> 4;
  // large M: >= 4 row groups per wave (the per-workgroup weight pack amortises); small M: one group per wave so that
  // the launch still covers the chip
  int gx = (M >= 65536) ? cdiv(M, 16 * 4 * 4) : cdiv(M, 16 * 4);
  if (gx > 256) gx = 256;      // one workgroup per CU (measured: 512 workgroups +8..17 %, 1024 +40 %)
#define G2V_STREAM(NTW, KS)                                                                                              \
  do {                                                                                                                   \
    const size_t lds = (size_t)NTW * KS * 256 * sizeof(float);                                                           \
    if (lds > 48 * 1024)                                                                                                 \
      (void)hipFuncSetAttribute((const void*)gemm_nt_stream_kernel<NTW, KS, TRANS_B>,                                    \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                                   \
    hipLaunchKernelGGL((gemm_nt_stream_kernel<NTW, KS, TRANS_B>), dim3(gx), dim3(256), lds, st, A, lda, Bm, ldb, bias,    \
                       Cout, ldc, M, C, N, act, accumulate);                                                             \
    return true;                                                                                                         \
  } while (0)
  if (ntw == 12 && ks == 4) G2V_STREAM(12, 4);       // gi = xin W_ih^T            (64 -> 192)
  if (ntw == 4 && ks == 12) G2V_STREAM(4, 12);       // dxin = dgi W_ih            (192 -> 64)
  if (ntw == 8 && ks == 8) G2V_STREAM(8, 8);         // VQ pre_linear              (128 -> 128)
  if (ntw == 4 && ks == 4) G2V_STREAM(4, 4);         // 64 -> 64
#undef G2V_STREAM
  return false;
}

// ---- wave-autonomous forward for UNALIGNED / row-mapped inputs (encoder in_layer: x (B,T,135) read as (T,B,135)) ----
// Rows of 135 floats are not 16-byte aligned, so the B-operand fragments are pulled with dword loads exactly in MFMA
// layout: lane (i, q) reads x[row m0 + i][4 s + q] for the KS4 = ceil(C / 4) k-steps (a 128-byte line is consumed by
// 8 consecutive loads, the vector L1 serves the repeats); the next 16 rows are in flight while the current 16 are
// multiplied.  N = 64 outputs: the four 16-row weight tiles are interleaved per k-step in LDS so ONE ds_read_b128
// yields the A operands of all four MFMAs.  No workgroup barrier after the weights are staged.
template <int KS4, bool MAPPED>
__global__ __launch_bounds__(256) void gemm_nt_k4_kernel(const float* __restrict__ A, RowMap am,
                                                         const float* __restrict__ W, const float* __restrict__ bias,
                                                         float* __restrict__ Cout, int64_t ldc, int M, int C, int act) {
  // k-slot q of k-step s is input column KS4 q + s (any assignment of columns to (step, slot) pairs works as long as the weight
  // image uses the same one): a lane's KS4 operands are KS4 ADJACENT floats of its row -- 16-byte loads (4-byte aligned: rows of
  // 135 floats; legal on gfx950, gpurun_tools/unaligned_x4_test.hip) instead of KS4 dword loads
  typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
  static_assert(KS4 % 4 == 2, "KS4 = 4 j + 2: whole 16-byte loads plus two dwords");
  __shared__ __attribute__((aligned(16))) float4 Wp[KS4 * 64];   // [s][lane] = W[{0,16,32,48} + i][KS4 q + s]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i = lane & 15, q = lane >> 4;
  for (int e = tid; e < KS4 * 64; e += 256) {
    const int s2 = e >> 6, l = e & 63, c = KS4 * (l >> 4) + s2, n = l & 15;
    const bool ok = c < C;
    const int cc = ok ? c : 0;
    const float w0 = W[(int64_t)n * C + cc], w1 = W[(int64_t)(n + 16) * C + cc], w2 = W[(int64_t)(n + 32) * C + cc],
                w3 = W[(int64_t)(n + 48) * C + cc];
    Wp[e] = ok ? make_float4(w0, w1, w2, w3) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  float4 bia[4];
#pragma unroll
  for (int t = 0; t < 4; ++t)
    bia[t] = bias ? *reinterpret_cast<const float4*>(bias + 16 * t + 4 * q) : make_float4(0.f, 0.f, 0.f, 0.f);
  __syncthreads();
  const int ngroups = M >> 4;                         // M is a multiple of 16 (checked by the launcher)
  const int gstride = gridDim.x * 4;
  // the lane's run of columns, and the clamped column of a slot past the row (its weight is zero: any finite value will do)
  const int c0 = KS4 * q;
  const int c_a = min(c0 + KS4 - 2, C - 1) - c0, c_b = min(c0 + KS4 - 1, C - 1) - c0;
  constexpr int NV = KS4 / 4;                          // 16-byte loads per row
  float xa[KS4], xn[KS4];
  auto row_ptr = [&](int grp) {
    const int m = 16 * grp + i;
    int64_t off;
    if (MAPPED) {
      const int outer = m / am.rows_inner, inner = m - outer * am.rows_inner;
      off = (int64_t)outer * am.so + (int64_t)inner * am.si;
    } else {
      off = (int64_t)m * am.ld;
    }
    return A + off + c0;
  };
  auto load_piece = [&](const float* p, int k, float (&x)[KS4]) {      // piece k of NV + 2
    if (k < NV) {
      const f4u v = *reinterpret_cast<const f4u*>(p + 4 * k);
      x[4 * k] = v[0]; x[4 * k + 1] = v[1]; x[4 * k + 2] = v[2]; x[4 * k + 3] = v[3];
    } else if (k == NV) {
      x[KS4 - 2] = p[c_a];
    } else {
      x[KS4 - 1] = p[c_b];
    }
  };
  int g = blockIdx.x * 4 + wave;
  if (g < ngroups) {
    const float* p = row_ptr(g);
#pragma unroll
    for (int k = 0; k < NV + 2; ++k) load_piece(p, k, xa);
  }
  // one group: multiply `cur` (loaded while the previous group was multiplied), stream the next group into `nxt` BETWEEN the
  // MFMAs, one load every few k-steps (a burst in front of them would stall the wave on the vector-memory front end with an idle
  // matrix pipe).  Called with the two buffers in alternating roles: no register copies, each k-step waits for its own operand.
  auto group = [&](int gc, const float (&cur)[KS4], float (&nxt)[KS4]) {
    const int gn = min(gc + gstride, ngroups - 1);      // past the end: a valid, unused group
    const float* pn = row_ptr(gn);
    f32x4 acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    constexpr int EVERY = KS4 / (NV + 2);              // k-steps between two loads
#pragma unroll
    for (int s2 = 0; s2 < KS4; ++s2) {
      const float4 wa = Wp[s2 * 64 + lane];
      acc[0] = mfma16(wa.x, cur[s2], acc[0]);
      acc[1] = mfma16(wa.y, cur[s2], acc[1]);
      acc[2] = mfma16(wa.z, cur[s2], acc[2]);
      acc[3] = mfma16(wa.w, cur[s2], acc[3]);
      if (s2 % EVERY == 0 && s2 / EVERY < NV + 2) load_piece(pn, s2 / EVERY, nxt);
      __builtin_amdgcn_sched_barrier(0);
    }
    float* crow = Cout + (int64_t)(16 * gc + i) * ldc + 4 * q;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      float4 v = make_float4(acc[t][0] + bia[t].x, acc[t][1] + bia[t].y, acc[t][2] + bia[t].z, acc[t][3] + bia[t].w);
      if (act == 1) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
      *reinterpret_cast<float4*>(crow + 16 * t) = v;
    }
    __builtin_amdgcn_sched_barrier(0);
  };
  // (three buffers, the loads two groups ahead: measured 39.4 us alone against 37.1 with two, the train step the same)
  while (g < ngroups) {
    group(g, xa, xn);
    g += gstride;
    if (g >= ngroups) break;
    group(g, xn, xa);
    g += gstride;
  }
}

static bool launch_k4(const float* A, const RowMap& am, const float* W, const float* bias, float* Cout, int64_t ldc, int M,
                      int C, int N, int act, hipStream_t st) {
  if (N != 64 || C != 135 || M < 4096 || (M & 15) || act == 2) return false;
  if (!ptr_vec_ok(Cout, ldc) || (bias && (reinterpret_cast<uintptr_t>(bias) & 15))) return false;
  int gx = cdiv(M, 16 * 4 * 4);
  if (gx > 256) gx = 256;      // one workgroup per CU, one wave per SIMD: measured optimum (512: +10 %, 320: +40 %, 128: +60 %)
  if (am.rows_inner > 0)
    hipLaunchKernelGGL((gemm_nt_k4_kernel<34, true>), dim3(gx), dim3(256), 0, st, A, am, W, bias, Cout, ldc, M, C, act);
  else
    hipLaunchKernelGGL((gemm_nt_k4_kernel<34, false>), dim3(gx), dim3(256), 0, st, A, am, W, bias, Cout, ldc, M, C, act);
  return true;
}

// ---- weight gradient: slab[split][n][k] = sum_{m in split} dy[m][n] * xin[m][k] -------------------
// One workgroup owns a (64*NTW) x 64 block of dW for a contiguous range of rows m, so with N <= 192 every
// dy / x element is read from HBM exactly once.  Rows are consumed in chunks of TM = 32: the NEXT chunk's global
// loads are issued into registers before the MFMAs of the current chunk (HBM latency hidden behind 8*4*NTW MFMAs),
// then written to LDS.  Wave w owns n-tiles {w, w+4, w+8}; the contraction index (rows) sits on the MFMA k slots.
constexpr int TM = 32, LDX = 64 + 16;

template <int NTW>
__global__ __launch_bounds__(256) void gemm_tn_kernel(const float* __restrict__ dY, int64_t lddy,
                                                      const float* __restrict__ X, RowMap xm,
                                                      const uint8_t* __restrict__ keep, float scale,
                                                      float* __restrict__ slab, float* __restrict__ slab_db,
                                                      int M, int K, int N, int rows_per_split) {
  constexpr int NB = 64 * NTW, LDD = NB + 16;
  __shared__ __attribute__((aligned(16))) float Ds[TM * LDD];
  __shared__ __attribute__((aligned(16))) float Xs[TM * LDX];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n0 = blockIdx.x * NB, k0 = blockIdx.y * 64, split = blockIdx.z;
  const int mb = split * rows_per_split;
  const int me = min(M, mb + rows_per_split);
  const int i = lane & 15, q = lane >> 4;

  f32x4 acc[NTW][4];
#pragma unroll
  for (int j = 0; j < NTW; ++j)
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[j][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float dbsum = 0.f;

  // register staging: thread -> column c = tid & 63 (coalesced), rows r = (tid >> 6) + 4 * it
  float rd[NTW][8], rx[8];
  const int c = tid & 63, rbase = tid >> 6;
  auto load_chunk = [&](int mc) {
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const int m = mc + rbase + 4 * it;
      const bool mv = m < me;
#pragma unroll
      for (int j = 0; j < NTW; ++j) {
        const int n = n0 + 64 * j + c;
        rd[j][it] = (mv && n < N) ? dY[(int64_t)m * lddy + n] : 0.f;
      }
      float xv = 0.f;
      if (mv && k0 + c < K) {
        xv = X[row_off(xm, m) + k0 + c];
        if (keep) xv = keep[(int64_t)m * K + k0 + c] ? xv * scale : 0.f;
      }
      rx[it] = xv;
    }
  };
  if (mb < me) load_chunk(mb);
  for (int mc = mb; mc < me; mc += TM) {
    __syncthreads();   // the previous chunk's MFMAs are done with the LDS tiles
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const int r = rbase + 4 * it;
#pragma unroll
      for (int j = 0; j < NTW; ++j) Ds[r * LDD + 64 * j + c] = rd[j][it];
      Xs[r * LDX + c] = rx[it];
    }
    __syncthreads();
    if (mc + TM < me) load_chunk(mc + TM);   // in flight while this chunk is multiplied
    if (slab_db && blockIdx.y == 0 && tid < NB) {
      float sacc = 0.f;
#pragma unroll 8
      for (int r = 0; r < TM; ++r) sacc += Ds[r * LDD + tid];
      dbsum += sacc;
    }
#pragma unroll
    for (int s4 = 0; s4 < TM; s4 += 4) {
      float a[NTW], bb[4];
#pragma unroll
      for (int j = 0; j < NTW; ++j) a[j] = Ds[(s4 + q) * LDD + 16 * (wave + 4 * j) + i];
#pragma unroll
      for (int t = 0; t < 4; ++t) bb[t] = Xs[(s4 + q) * LDX + 16 * t + i];
#pragma unroll
      for (int j = 0; j < NTW; ++j)
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[j][t] = mfma16(a[j], bb[t], acc[j][t]);
    }
  }
  // lane holds dw[n = n0 + 16*(wave + 4j) + 4q + r][k = k0 + 16t + (lane&15)]
  float* sl = slab + (int64_t)split * N * K;
#pragma unroll
  for (int j = 0; j < NTW; ++j)
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int k = k0 + 16 * t + i;
      if (k >= K) continue;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int n = n0 + 16 * (wave + 4 * j) + 4 * q + r;
        if (n < N) sl[(int64_t)n * K + k] = acc[j][t][r];
      }
    }
  if (slab_db && blockIdx.y == 0 && tid < NB && n0 + tid < N) slab_db[(int64_t)split * N + n0 + tid] = dbsum;
}

// ---- wave-autonomous weight gradient -------------------------------------------------------------------------------
// A wave keeps a (16 TN_) x (16 TK_) block of dW in its accumulators, so there is no LDS tile and no barrier in the
// streaming loop: lane (i, q) pulls dY[m0 + q][n0 + 16 t + i] and X[m0 + q][k0 + 16 u + i] straight into the MFMA
// operand registers with dword loads (no alignment requirement, so D = 135 rows and the (B,T,D) -> (T,B,D) row map
// cost nothing), the next 16 rows are in flight while the current 16 are multiplied, and TN_*TK_ independent
// accumulators keep the matrix pipe issuing back to back.  A workgroup has 8 waves = 4 row ranges x 2 tile groups
// (SN x SK = 2 halves of the output block, <= 256 registers each, so two waves share every SIMD and one fills the
// matrix pipe while the other waits); the four row ranges are summed through LDS once at the end and the workgroup
// partial goes to a slab for the deterministic slab reduction.
// BF3: the products run on the bf16 matrix pipe as a 3-term split (x = hi + lo, two bf16 each; hi*hi + hi*lo + lo*hi,
// fp32 accumulation; the dropped lo*lo term and the truncation of lo are ~2^-16 relative per product), 2.7x fewer
// matrix-pipe cycles than fp32 MFMA.  Opt-in per call (G2V_WGRAD_BF16X3): a weight GRADIENT tolerates 1e-5 relative noise.
typedef short s16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void split_bf16x4(const float (&v)[4], s16x4& hi, s16x4& lo) {
  uint32_t h[4], l[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const uint32_t bits = __float_as_uint(v[j]);
    h[j] = bits & 0xffff0000u;                                   // hi = truncation to bf16
    l[j] = __float_as_uint(v[j] - __uint_as_float(h[j]));         // exact remainder, then truncated to bf16
  }
  hi = (s16x4){(short)(h[0] >> 16), (short)(h[1] >> 16), (short)(h[2] >> 16), (short)(h[3] >> 16)};
  lo = (s16x4){(short)(l[0] >> 16), (short)(l[1] >> 16), (short)(l[2] >> 16), (short)(l[3] >> 16)};
}

// Up to G2V_TN_BATCH problems of identical shape per launch (blockIdx.y picks one): their workgroups fill each other's
// prologue / epilogue / tail, and one slab reduction serves them all.
constexpr int G2V_TN_BATCH = 4;
struct TnBatch {
  const float* dy[G2V_TN_BATCH];
  const float* x[G2V_TN_BATCH];
  float* slab[G2V_TN_BATCH];
  float* slab_db[G2V_TN_BATCH];
  const float* dy2[G2V_TN_BATCH];     // DUAL instantiations: the product is (dy + dy2)^T x, the sum formed at use time
};

template <int TN_, int TK_, int SN, int SK, bool MAPPED, int VW, bool BF3, int NR = 4, bool GEN = false, bool DUAL = false>
__device__ __forceinline__ void tn_wave_body(const TnBatch& bt, int64_t lddy, const RowMap& xm, int M, int K, int N,
                                             int rows_per_wave) {
  const float* __restrict__ dY = bt.dy[blockIdx.y];
  const float* __restrict__ X = bt.x[blockIdx.y];
  float* __restrict__ slab = bt.slab[blockIdx.y];
  float* __restrict__ slab_db = bt.slab_db[blockIdx.y];
  // DUAL (the encoder input layer: dx arrives as one array per GRU direction): the second addend travels in columns
  // TN_ .. 2 TN_ - 1 of the same operand buffers and is added when the fragment is used -- the same sum-then-multiply
  // arithmetic as a separate add pass, without its 3 x M x N x 4 bytes of traffic
  static_assert(!DUAL || (VW == 1 && !BF3 && !GEN), "two-addend operand: dword loads, fp32 products");
  constexpr int TA = DUAL ? 2 * TN_ : TN_;
  const float* __restrict__ dY2 = DUAL ? bt.dy2[blockIdx.y] : nullptr;
  // NR row ranges x two tile groups.  NR = 2 (256 threads, ONE wave per SIMD, one workgroup per CU) is what the fp32 path
  // launches: with the operand loads interleaved between its MFMAs a single wave keeps the matrix pipe busy on its own, and
  // compared with NR = 4 (two waves per SIMD) every wave walks twice the rows and the final sum is 2-way (measured at the
  // BASELINE shape, four 3H x H problems per launch: 137 us against 153 us).  NR = 4 remains for the bf16x3 path, whose
  // split arithmetic leaves the second wave something to overlap.
  constexpr int NG = SN * SK;
  static_assert(NG == 2, "two tile groups per workgroup");
  static_assert(NR == 2 || NR == 4, "two or four row ranges per workgroup");
  constexpr bool TRIPLE = !BF3 && !GEN && !DUAL; // (the 4 x 7 block of the output-blocked variant leaves room for two buffers;
                                                 //  the two-addend variant would need 222 + 64 registers with three)
  static_assert(VW == 1 || (VW == 2 && TN_ % 2 == 0 && TK_ % 2 == 0), "pairs of tiles per 8-byte load");
  // VW == 2 (8-byte-aligned rows, whole tiles): the 16 MFMA rows of a PAIR of tiles are interleaved over 32 columns,
  // lane i <-> columns 32 g + 2 i + {0, 1}, so one global_load_dwordx2 (a full 128-byte line per matrix row) feeds the
  // operands of two tiles.  The accumulator of tile (2g + t', 2h + u') then holds rows n = 32 g + 2 (4 q + r) + t' and
  // column k = 32 h + 2 j + u' (j = lane & 15).
  extern __shared__ __attribute__((aligned(16))) float smem[];   // 4 regions x [TN_*TK_][64][4] + db [8][TN_][16]
  constexpr int NTILE = TN_ * TK_;
  const int tid = threadIdx.x, lane = tid & 63;
  // (GEN: the wave index as a scalar, so that the row bases below live in scalar registers)
  const int wave = GEN ? __builtin_amdgcn_readfirstlane(tid >> 6) : tid >> 6;
  const int i = lane & 15, q = lane >> 4;
  const int rr = wave % NR, grp = wave / NR;           // row range, tile group
  // GEN: dW is larger than one workgroup's accumulators: blockIdx.z picks a (16 TN_ SN) x (16 TK_ SK) output block; tiles
  // (or lanes of a tile) past N / K read an in-range column instead and their accumulators are never stored
  static_assert(!GEN || (VW == 1 && !BF3), "output-blocked mode: dword loads, fp32 products");
  const int nbn = GEN ? (N + 16 * TN_ * SN - 1) / (16 * TN_ * SN) : 1;
  const int obn = GEN ? (int)blockIdx.z % nbn : 0, obk = GEN ? (int)blockIdx.z / nbn : 0;
  const int n0 = 16 * TN_ * (SN * obn + grp % SN), k0 = 16 * TK_ * (SK * obk + grp / SN);
  int offn[GEN ? TN_ : 1], offk[GEN ? TK_ : 1];
  if constexpr (GEN) {
#pragma unroll
    for (int t = 0; t < TN_; ++t) offn[t] = (n0 + 16 * t + i < N) ? 16 * t + i : (n0 < N ? 0 : -n0);
#pragma unroll
    for (int u = 0; u < TK_; ++u) offk[u] = (k0 + 16 * u + i < K) ? 16 * u + i : (k0 < K ? 0 : -k0);
  }
  // GEN with plain rows: every operand load is  (wave-uniform row base, scalar registers) + (one constant 32-bit byte offset
  // per lane and tile)  -- the saddr form of global_load, no per-load 64-bit vector address arithmetic and 2 x 11 fewer
  // live registers (round 4: the address temporaries were reusing registers of loads still in flight, and the loop head
  // waited vmcnt(0) for them)
  constexpr bool SADDR = GEN && !MAPPED;
  uint32_t goa[SADDR ? TN_ : 1], gob[SADDR ? TK_ : 1];
  if constexpr (SADDR) {
#pragma unroll
    for (int t = 0; t < TN_; ++t) goa[t] = (uint32_t)(((int64_t)q * lddy + n0 + offn[t]) * 4);
#pragma unroll
    for (int u = 0; u < TK_; ++u) gob[u] = (uint32_t)(((int64_t)q * xm.ld + k0 + offk[u]) * 4);
  }
  const int mb = (blockIdx.x * NR + rr) * rows_per_wave;
  const int me = min(M, mb + rows_per_wave);

  f32x4 acc[TN_][TK_];
  float dbs[TN_];
#pragma unroll
  for (int t = 0; t < TN_; ++t) {
    dbs[t] = 0.f;
#pragma unroll
    for (int u = 0; u < TK_; ++u) acc[t][u] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }
  // per-lane column offsets: i for whole tiles (immediate offsets 64 t bytes), clamped for a ragged last tile
  const bool okn = n0 + 16 * (TN_ - 1) + i < N, okk = k0 + 16 * (TK_ - 1) + i < K;
  const int in_last = okn ? i : 0, ik_last = okk ? i : 0;

  // row m of a row-major matrix as a pointer in SCALAR registers; a lane offset passed through in_block() is zero-extended
  // in the block that uses it (hoisted out of the loop, the extension hides the base + 32-bit-offset form from instruction
  // selection and every load gets a 64-bit vector add again)
  auto srow = [&](const float* p, int m, int64_t ld) {
    const int64_t off = (int64_t)m * ld * 4;
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)off), hi = __builtin_amdgcn_readfirstlane((uint32_t)(off >> 32));
    return reinterpret_cast<const char*>(p) + (((uint64_t)hi << 32) | lo);
  };
  auto in_block = [](uint32_t o) {
    asm volatile("" : "+v"(o));
    return o;
  };
  float ca[4][TA], cb[4][TK_], na[4][TA], nb[4][TK_], ea[TRIPLE ? 4 : 1][TRIPLE ? TA : 1], eb[TRIPLE ? 4 : 1][TRIPLE ? TK_ : 1];
  auto load_group = [&](int m0, float (&a)[4][TA], float (&b)[4][TK_]) {
    // M and the wave ranges are multiples of 16: every row of a group is valid.  One division per group for the
    // (B,T,D) -> (T,B,D) row map, then the three following row quads step the (outer, inner) pair.
    // fp32 MFMA (16x16x4): k-slot q of step qd is row m0 + 4 qd + q.  bf16 MFMA (16x16x16): lane group q supplies the four
    // consecutive rows m0 + 4 q + j.  (Any assignment works as long as both operands use the same one.)
    constexpr int RSTEP = BF3 ? 1 : 4;
    int mrow = BF3 ? m0 + 4 * q : m0 + q, outer = 0, inner = 0;
    if (MAPPED) { outer = mrow / xm.rows_inner; inner = mrow - outer * xm.rows_inner; }
#pragma unroll
    for (int qd = 0; qd < 4; ++qd) {
      const float* dr = dY + (int64_t)mrow * lddy + n0;
      const float* xr = X + (MAPPED ? (int64_t)outer * xm.so + (int64_t)inner * xm.si : (int64_t)mrow * xm.ld) + k0;
      const float* dri = dr + i;
      const float* xri = xr + i;
      if constexpr (VW == 2) {
#pragma unroll
        for (int g = 0; g < TN_ / 2; ++g) {
          const float2 v = *reinterpret_cast<const float2*>(dr + 32 * g + 2 * i);
          a[qd][2 * g] = v.x;
          a[qd][2 * g + 1] = v.y;
        }
#pragma unroll
        for (int h = 0; h < TK_ / 2; ++h) {
          const float2 v = *reinterpret_cast<const float2*>(xr + 32 * h + 2 * i);
          b[qd][2 * h] = v.x;
          b[qd][2 * h + 1] = v.y;
        }
      } else if constexpr (SADDR) {
        const char* ba = srow(dY, m0 + 4 * qd, lddy);
        const char* bb = srow(X, m0 + 4 * qd, xm.ld);
#pragma unroll
        for (int t = 0; t < TN_; ++t) a[qd][t] = *reinterpret_cast<const float*>(ba + in_block(goa[t]));
#pragma unroll
        for (int u = 0; u < TK_; ++u) b[qd][u] = *reinterpret_cast<const float*>(bb + in_block(gob[u]));
      } else if constexpr (GEN) {
#pragma unroll
        for (int t = 0; t < TN_; ++t) a[qd][t] = dr[offn[t]];
#pragma unroll
        for (int u = 0; u < TK_; ++u) b[qd][u] = xr[offk[u]];
      } else {
#pragma unroll
      for (int t = 0; t < TN_ - 1; ++t) a[qd][t] = dri[16 * t];
      a[qd][TN_ - 1] = dr[16 * (TN_ - 1) + in_last];
#pragma unroll
      for (int u = 0; u < TK_ - 1; ++u) b[qd][u] = xri[16 * u];
      b[qd][TK_ - 1] = xr[16 * (TK_ - 1) + ik_last];
      if constexpr (DUAL) {
        const float* d2 = dY2 + (int64_t)mrow * lddy + n0;
#pragma unroll
        for (int t = 0; t < TN_ - 1; ++t) a[qd][TN_ + t] = d2[16 * t + i];
        a[qd][2 * TN_ - 1] = d2[16 * (TN_ - 1) + in_last];
      }
      }
      mrow += RSTEP;
      if (MAPPED) {
        inner += RSTEP;
        if (inner >= xm.rows_inner) { inner -= xm.rows_inner; ++outer; }
      }
    }
  };
  auto compute = [&](float (&a)[4][TA], float (&b)[4][TK_]) {
#pragma unroll
    for (int qd = 0; qd < 4; ++qd) {
      if constexpr (VW == 1 && !GEN) {
        a[qd][TN_ - 1] = okn ? a[qd][TN_ - 1] : 0.f;   // ragged last tiles: the clamped column is masked at use time
        b[qd][TK_ - 1] = okk ? b[qd][TK_ - 1] : 0.f;
      }
      if constexpr (DUAL) {
#pragma unroll
        for (int t = 0; t < TN_; ++t) a[qd][t] += (t < TN_ - 1 || okn) ? a[qd][TN_ + t] : 0.f;
      }
    }
    if constexpr (BF3) {
      s16x4 bh[TK_], bl[TK_];
#pragma unroll
      for (int u = 0; u < TK_; ++u) {
        const float v[4] = {b[0][u], b[1][u], b[2][u], b[3][u]};
        split_bf16x4(v, bh[u], bl[u]);
      }
#pragma unroll
      for (int t = 0; t < TN_; ++t) {
        const float v[4] = {a[0][t], a[1][t], a[2][t], a[3][t]};
        dbs[t] += (v[0] + v[1]) + (v[2] + v[3]);
        s16x4 ah, al;
        split_bf16x4(v, ah, al);
#pragma unroll
        for (int u = 0; u < TK_; ++u) {
          acc[t][u] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(ah, bh[u], acc[t][u], 0, 0, 0);
          acc[t][u] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(ah, bl[u], acc[t][u], 0, 0, 0);
          acc[t][u] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(al, bh[u], acc[t][u], 0, 0, 0);
        }
      }
    } else {
#pragma unroll
    for (int qd = 0; qd < 4; ++qd) {
#pragma unroll
      for (int t = 0; t < TN_; ++t) {
        dbs[t] += a[qd][t];
#pragma unroll
        for (int u = 0; u < TK_; ++u) acc[t][u] = mfma16(a[qd][t], b[qd][u], acc[t][u]);
      }
    }
    }
  };
  // fp32 path: multiply group (a, b) while the operand loads of group `mn` are issued BETWEEN the MFMAs into (an, bn): a
  // burst of 20-40 loads in front of the 96 MFMAs stalls the wave on the CU's vector-memory front end with an idle matrix
  // pipe; one load per few MFMAs is absorbed at the rate the front end accepts it.
  auto compute_and_load = [&](float (&a)[4][TA], float (&b)[4][TK_], int mn, float (&an)[4][TA], float (&bn)[4][TK_]) {
    constexpr int NMQ = TN_ * TK_, NLQ = (TA + TK_) / VW;
    constexpr int STRIDE = (NMQ / NLQ) > 0 ? (NMQ / NLQ) : 1;
    int mrow = mn + q, outer = 0, inner = 0;
    if (MAPPED) { outer = mrow / xm.rows_inner; inner = mrow - outer * xm.rows_inner; }
#pragma unroll
    for (int qd = 0; qd < 4; ++qd) {
      if constexpr (VW == 1 && !GEN) {
        a[qd][TN_ - 1] = okn ? a[qd][TN_ - 1] : 0.f;
        b[qd][TK_ - 1] = okk ? b[qd][TK_ - 1] : 0.f;
      }
      if constexpr (DUAL) {
#pragma unroll
        for (int t = 0; t < TN_; ++t) a[qd][t] += (t < TN_ - 1 || okn) ? a[qd][TN_ + t] : 0.f;
      }
      const float* dr = dY + (int64_t)mrow * lddy + n0;
      const float* dr2 = DUAL ? dY2 + (int64_t)mrow * lddy + n0 : nullptr;
      const float* xr = X + (MAPPED ? (int64_t)outer * xm.so + (int64_t)inner * xm.si : (int64_t)mrow * xm.ld) + k0;
      const char* sba = SADDR ? srow(dY, mn + 4 * qd, lddy) : nullptr;
      const char* sbb = SADDR ? srow(X, mn + 4 * qd, xm.ld) : nullptr;
      auto issue = [&](int k) {
        if constexpr (VW == 2) {
          if (k < TN_ / 2) {
            const float2 v = *reinterpret_cast<const float2*>(dr + 32 * k + 2 * i);
            an[qd][2 * k] = v.x;
            an[qd][2 * k + 1] = v.y;
          } else {
            const int h = k - TN_ / 2;
            const float2 v = *reinterpret_cast<const float2*>(xr + 32 * h + 2 * i);
            bn[qd][2 * h] = v.x;
            bn[qd][2 * h + 1] = v.y;
          }
        } else if constexpr (SADDR) {
          if (k < TN_) an[qd][k] = *reinterpret_cast<const float*>(sba + in_block(goa[k < TN_ ? k : 0]));
          else bn[qd][k - TN_] = *reinterpret_cast<const float*>(sbb + in_block(gob[k - TN_ < TK_ ? k - TN_ : 0]));
        } else if constexpr (GEN) {
          if (k < TN_) an[qd][k] = dr[offn[k < TN_ ? k : 0]];
          else bn[qd][k - TN_] = xr[offk[k - TN_ < TK_ ? k - TN_ : 0]];
        } else {
          if (k < TN_ - 1) an[qd][k] = dr[16 * k + i];
          else if (k == TN_ - 1) an[qd][k] = dr[16 * (TN_ - 1) + in_last];
          else if (k < TN_ + TK_ - 1) bn[qd][k - TN_] = xr[16 * (k - TN_) + i];
          else if (k == TN_ + TK_ - 1) bn[qd][TK_ - 1] = xr[16 * (TK_ - 1) + ik_last];
          else if constexpr (DUAL) {
            if (k < 2 * TN_ + TK_ - 1) an[qd][k - TK_] = dr2[16 * (k - TN_ - TK_) + i];
            else an[qd][2 * TN_ - 1] = dr2[16 * (TN_ - 1) + in_last];
          }
        }
      };
#pragma unroll
      for (int t = 0; t < TN_; ++t) {
        dbs[t] += a[qd][t];
#pragma unroll
        for (int u = 0; u < TK_; ++u) {
          acc[t][u] = mfma16(a[qd][t], b[qd][u], acc[t][u]);
          const int m = t * TK_ + u;
          if ((m % STRIDE) == STRIDE - 1 && (m / STRIDE) < NLQ) {
            issue(m / STRIDE);
            __builtin_amdgcn_sched_barrier(0);
          }
        }
      }
#pragma unroll
      for (int k = NMQ / STRIDE; k < NLQ; ++k) issue(k);
      mrow += 4;
      if (MAPPED) {
        inner += 4;
        if (inner >= xm.rows_inner) { inner -= xm.rows_inner; ++outer; }
      }
    }
    // (Round 4, measured and dropped: the bias-gradient column sums behind the group's products instead of in front of each row
    //  quad's -- the compiler gathers them at the top of the group, where their wait counts look too strict -- 1.569 -> 1.588 ms;
    //  two operand buffers instead of three for the 44-loads-per-group shapes (2 x 9, 9 x 2): no difference.)
  };
  // ping-pong over two register buffers (no copies): the loads of the next 16 rows are always in flight while the
  // current 16 are multiplied
  int m0 = mb;
  if (m0 < me) load_group(m0, ca, cb);
  for (; m0 + 32 <= me; m0 += 32) {
    if constexpr (BF3) {
      load_group(m0 + 16, na, nb);
      __builtin_amdgcn_sched_barrier(0);
      compute(ca, cb);
      __builtin_amdgcn_sched_barrier(0);
      load_group(min(m0 + 32, M - 16), ca, cb);   // past the end of the range: a valid, unused group
      __builtin_amdgcn_sched_barrier(0);
      compute(na, nb);
      __builtin_amdgcn_sched_barrier(0);
    } else {
      break;      // fp32 path: the triple-buffered loop below
    }
  }
  if constexpr (TRIPLE) {
    // three operand buffers in rotation: the loads for the group AFTER next are issued between the MFMAs of the current
    // group, i.e. two groups (2 x 96 MFMAs per wave, ~6k cycles with two waves per SIMD) ahead of their use - one group
    // of distance does not cover the HBM round trip under load.
    // (Round 4, measured and dropped: whole rotations without the two mid-loop exits + a tail for the 0-2 groups left over.
    //  The wait counts of the 2 x 9 shape become lenient, those of the 9 x 2 shape collapse to vmcnt(0) at 256 registers:
    //  1.587 -> 1.632 ms at the headline shape.)
    auto clamp_g = [&](int m) { return min(m, M - 16); };      // past the end of the range: a valid, unused group
    if (m0 < me) load_group(clamp_g(m0 + 16), na, nb);
    while (m0 < me) {
      compute_and_load(ca, cb, clamp_g(m0 + 32), ea, eb);
      m0 += 16;
      if (m0 >= me) break;
      compute_and_load(na, nb, clamp_g(m0 + 32), ca, cb);
      m0 += 16;
      if (m0 >= me) break;
      compute_and_load(ea, eb, clamp_g(m0 + 32), na, nb);
      m0 += 16;
    }
  } else if constexpr (!BF3) {
    while (m0 < me) {      // two buffers: the next group streams in between the MFMAs of the current one
      compute_and_load(ca, cb, min(m0 + 16, M - 16), na, nb);
      m0 += 16;
      if (m0 >= me) break;
      compute_and_load(na, nb, min(m0 + 16, M - 16), ca, cb);
      m0 += 16;
    }
  } else {
    if (m0 < me) compute(ca, cb);               // odd number of groups: the last one is already loaded
  }
  // ---- sum the four row ranges of each tile group: (2,3) -> LDS -> (0,1); 1 -> LDS -> 0; range 0 writes the slab ----
  float* dbl = smem + NR * NTILE * 256;
#pragma unroll
  for (int t = 0; t < TN_; ++t) {
    float v = dbs[t];
    v += __shfl_xor(v, 16);
    v += __shfl_xor(v, 32);
    if (q == 0) dbl[(wave * TN_ + t) * 16 + i] = v;
  }
  auto dump = [&](float* reg) {
#pragma unroll
    for (int t = 0; t < TN_; ++t)
#pragma unroll
      for (int u = 0; u < TK_; ++u)
        *reinterpret_cast<float4*>(reg + ((t * TK_ + u) * 64 + lane) * 4) =
            make_float4(acc[t][u][0], acc[t][u][1], acc[t][u][2], acc[t][u][3]);
  };
  auto absorb = [&](const float* reg) {
#pragma unroll
    for (int t = 0; t < TN_; ++t)
#pragma unroll
      for (int u = 0; u < TK_; ++u) {
        const float4 v = *reinterpret_cast<const float4*>(reg + ((t * TK_ + u) * 64 + lane) * 4);
        acc[t][u][0] += v.x; acc[t][u][1] += v.y; acc[t][u][2] += v.z; acc[t][u][3] += v.w;
      }
  };
  if constexpr (NR == 4) {
    if (rr >= 2) dump(smem + (2 * grp + rr - 2) * NTILE * 256);
    __syncthreads();
    if (rr < 2) absorb(smem + (2 * grp + rr) * NTILE * 256);
    __syncthreads();
  }
  if (rr == 1) dump(smem + grp * NTILE * 256);
  __syncthreads();
  if (rr != 0) return;
  absorb(smem + grp * NTILE * 256);
  float* sl = slab + (int64_t)blockIdx.x * N * K;
  if constexpr (VW == 2) {
#pragma unroll
    for (int t = 0; t < TN_; ++t)
#pragma unroll
      for (int h = 0; h < TK_ / 2; ++h)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int n = n0 + 32 * (t >> 1) + 2 * (4 * q + r) + (t & 1);
          *reinterpret_cast<float2*>(sl + (int64_t)n * K + k0 + 32 * h + 2 * i) = make_float2(acc[t][2 * h][r], acc[t][2 * h + 1][r]);
        }
  } else {
#pragma unroll
  for (int t = 0; t < TN_; ++t)
#pragma unroll
    for (int u = 0; u < TK_; ++u) {
      const int k = k0 + 16 * u + i;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int n = n0 + 16 * t + 4 * q + r;
        if (n < N && k < K) sl[(int64_t)n * K + k] = acc[t][u][r];
      }
    }
  }
  if (slab_db && grp / SN == 0 && obk == 0) {
#pragma unroll
    for (int t = 0; t < TN_; ++t) {
      const int n = (VW == 2) ? n0 + 32 * (t >> 1) + 2 * i + (t & 1) : n0 + 16 * t + i;
      const float* d0 = dbl + (NR * grp * TN_ + t) * 16 + i;
      if (q == 0 && n < N)
        slab_db[(int64_t)blockIdx.x * N + n] =
            NR == 4 ? (d0[0] + d0[TN_ * 16]) + (d0[2 * TN_ * 16] + d0[3 * TN_ * 16]) : d0[0] + d0[TN_ * 16];
    }
  }
}

template <int TN_, int TK_, int SN, int SK, bool MAPPED, int VW, bool BF3, int NR>
__global__ __launch_bounds__(128 * NR) __attribute__((amdgpu_num_vgpr(128))) void gemm_tn_wave_kernel(TnBatch bt, int64_t lddy,
                                                                                                      RowMap xm, int M, int K,
                                                                                                      int N, int rows_per_wave) {
  static_assert(SN * SK == 2, "two tile groups per workgroup");
  tn_wave_body<TN_, TK_, SN, SK, MAPPED, VW, BF3, NR>(bt, lddy, xm, M, K, N, rows_per_wave);
}

// (dy + dy2)^T x for the 64 x 135 input-layer shape (see DUAL in tn_wave_body)
template <bool MAPPED>
__global__ __launch_bounds__(256) __attribute__((amdgpu_num_vgpr(160))) void gemm_tn_wave_dual_kernel(TnBatch bt, int64_t lddy,
                                                                                                      RowMap xm, int M, int K,
                                                                                                      int N, int rows_per_wave) {
  tn_wave_body<2, 9, 2, 1, MAPPED, 1, false, 2, false, true>(bt, lddy, xm, M, K, N, rows_per_wave);
}

// Output-blocked variant for weight matrices larger than one workgroup's accumulators (H = 200: 600 x 200, 600 x 300 ...):
// grid.z walks (16 TN_ SN) x (16 TK_ SK) blocks of dW, grid.x the row ranges; everything else as above.
// (Round 4, measured: without the register attribute the accumulators move to AGPRs and the loop gains ~1 v_accvgpr copy per
//  MFMA -- native shape at B = 4096 7.26 -> 7.53 ms; three operand buffers do not fit the 6-bit vmcnt: with 88 loads in
//  flight every wait for the oldest group also waits for most of the youngest.)
template <int TN_, int TK_, int SN, int SK, bool MAPPED>
__global__ __launch_bounds__(256) __attribute__((amdgpu_num_vgpr(128))) void gemm_tn_wave_gen_kernel(TnBatch bt, int64_t lddy,
                                                                                                     RowMap xm, int M, int K,
                                                                                                     int N, int rows_per_wave) {
  tn_wave_body<TN_, TK_, SN, SK, MAPPED, 1, false, 2, true>(bt, lddy, xm, M, K, N, rows_per_wave);
}

// (Round 4, tried: a 4 x 13 block for K = 200 -- 13 tiles, no padding tile, no second K block.  208 accumulators + two operand
//  buffers of 4 x 17: 512 registers AND 276-444 bytes of scratch per lane; not launched.)
// block shape of the output-blocked variant: 0 = 192 x 64 (TN 6, TK 4), 1 = 128 x 112 (TN 4, TK 7); the one that pads N x K
// the least.  Returns the number of output blocks; rows per wave / row splits sized for ~256 workgroups per launch.
static int tn_gen_grid(int M, int K, int N, bool has_keep, int nprob, int* cfg, int* rows_per_wave, int* nsplit) {
  if (has_keep || M < 4096 || (M & 15)) return 0;
  const int bn[2] = {192, 128}, bk[2] = {64, 112};
  int best = 0;
  int64_t best_area = -1;
  for (int c = 0; c < 2; ++c) {
    const int64_t area = (int64_t)cdiv(N, bn[c]) * bn[c] * cdiv(K, bk[c]) * bk[c];
    if (best_area < 0 || area < best_area) { best = c; best_area = area; }
  }
  const int nob = cdiv(N, bn[best]) * cdiv(K, bk[best]);
  int splits = 256 / (nob * nprob);
  if (splits < 1) splits = 1;
  const int rpw = round_up(cdiv(M, 2 * splits), 16);
  if (cfg) *cfg = best;
  if (rows_per_wave) *rows_per_wave = rpw;
  if (nsplit) *nsplit = cdiv(M, 2 * rpw);
  return nob;
}

// grid (= number of slabs) and rows per wave of the wave-autonomous path; 0 when the shape is not covered
static int tn_wave_grid(int M, int K, int N, bool has_keep, int* rows_per_wave, int nprob = 1, int nr = 4) {
  const int tn = cdiv(N, 16), tk = cdiv(K, 16);
  const bool covered = (tn == 12 && tk == 4) || (tn == 4 && tk == 9) || (tn == 9 && tk == 4) || (tn == 4 && tk == 4);
  if (!covered || has_keep || M < 4096 || (M & 15)) return 0;
  // one workgroup (4 row ranges) per CU over ALL problems of the launch: a workgroup occupies a whole CU's registers, so
  // more workgroups than CUs only run as further rounds, each paying the prologue, the LDS sum and the slab write again
  int rpw = round_up(cdiv((int64_t)M * nprob, 256 * nr), 16);
  if (rows_per_wave) *rows_per_wave = rpw;
  return cdiv(M, nr * rpw);
}

// two slab families in one launch: blocks [0, nblk_a) reduce (slab_a -> out_a), the rest (slab_b -> out_b).
// A workgroup owns 32 consecutive outputs: 8 float4 columns x 32 slab groups (thread (grp, c4) sums slabs grp, grp+32,
// ... of its float4 column, every load independent), then the 32 group partials are summed through LDS in a fixed
// order.  n % 4 == 0 for the float4 path; otherwise (the 135-wide bias vectors) a scalar path with the same structure.
struct SlabBatch {
  const float* slab_a[G2V_TN_BATCH];
  float* out_a[G2V_TN_BATCH];
  const float* slab_b[G2V_TN_BATCH];
  float* out_b[G2V_TN_BATCH];
};
// one workgroup's 32 outputs of one family: out[e0 .. e0 + 31] (+)= sum over the nsplit slabs, in the fixed order described above
__device__ __forceinline__ void slab_reduce_block(const float* __restrict__ slab, float* __restrict__ out, int64_t n, int64_t e0,
                                                  int nsplit, int accumulate) {
  __shared__ __attribute__((aligned(16))) float red[32][36];
  const int c4 = threadIdx.x & 7, grp = threadIdx.x >> 3;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  if ((n & 3) == 0) {
    const int64_t e = e0 + 4 * c4;
    if (e < n) {
      float4 t0 = s, t1 = s;
      int p = grp;
      for (; p + 32 < nsplit; p += 64) {
        const float4 a = *reinterpret_cast<const float4*>(slab + (int64_t)p * n + e);
        const float4 b2 = *reinterpret_cast<const float4*>(slab + (int64_t)(p + 32) * n + e);
        t0.x += a.x; t0.y += a.y; t0.z += a.z; t0.w += a.w;
        t1.x += b2.x; t1.y += b2.y; t1.z += b2.z; t1.w += b2.w;
      }
      if (p < nsplit) {
        const float4 a = *reinterpret_cast<const float4*>(slab + (int64_t)p * n + e);
        t0.x += a.x; t0.y += a.y; t0.z += a.z; t0.w += a.w;
      }
      s = make_float4(t0.x + t1.x, t0.y + t1.y, t0.z + t1.z, t0.w + t1.w);
    }
  } else {
    float v[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int64_t e = e0 + 4 * c4 + j;
      if (e < n)
        for (int p = grp; p < nsplit; p += 32) v[j] += slab[(int64_t)p * n + e];
    }
    s = make_float4(v[0], v[1], v[2], v[3]);
  }
  *reinterpret_cast<float4*>(&red[grp][4 * c4]) = s;
  __syncthreads();
  if (threadIdx.x < 32) {
    const int64_t e = e0 + threadIdx.x;
    if (e < n) {
      float t = 0.f;
#pragma unroll
      for (int g = 0; g < 32; ++g) t += red[g][threadIdx.x];
      out[e] = accumulate ? out[e] + t : t;
    }
  }
}
__global__ __launch_bounds__(256) void slab_reduce2_kernel(SlabBatch sb, int64_t na, int64_t nb, int nsplit, int accumulate,
                                                           int nblk_a) {
  const bool first = (int)blockIdx.x < nblk_a;
  const float* __restrict__ slab_b = sb.slab_b[blockIdx.y];
  if (!first && slab_b == nullptr) return;            // this problem has no bias gradient
  slab_reduce_block(first ? sb.slab_a[blockIdx.y] : slab_b, first ? sb.out_a[blockIdx.y] : sb.out_b[blockIdx.y], first ? na : nb,
                    (int64_t)(first ? blockIdx.x : blockIdx.x - nblk_a) * 32, nsplit, accumulate);
}

// Round 6: the slab reductions of SEVERAL weight-gradient calls in ONE launch, off the chain between the products (a product used
// to wait for the reduction of the one in front of it, which shared its workspace; beside the encoder's BPTT kernel -- whose two
// workgroups per CU leave a late-dispatched kernel no registers -- such a reduction took 17-95 us instead of 5:
// profiles/r05_az_step_timeline.txt).  A 1-D grid; family f (a weight or a bias slab set) owns blocks [first[f], first[f + 1]).
// The arithmetic of every output is slab_reduce_block's: bitwise the immediate reduction's.
constexpr int SLAB_MULTI = 24;       // families per launch: up to G2V_WGRAD_PENDING_MAX pending calls x 4 problems x {dw, db}, in chunks
struct SlabMulti {
  const float* slab[SLAB_MULTI];
  float* out[SLAB_MULTI];
  int64_t n[SLAB_MULTI];
  int nsplit[SLAB_MULTI];
  int accumulate[SLAB_MULTI];
  int first[SLAB_MULTI + 1];
  int count;
};
__global__ __launch_bounds__(256) void slab_reduce_multi_kernel(SlabMulti sm) {
  int f = 0;
  while (f + 1 < sm.count && (int)blockIdx.x >= sm.first[f + 1]) ++f;          // (wave-uniform: <= 24 scalar compares)
  slab_reduce_block(sm.slab[f], sm.out[f], sm.n[f], (int64_t)((int)blockIdx.x - sm.first[f]) * 32, sm.nsplit[f], sm.accumulate[f]);
}

static int tn_ntw(int N) { return N <= 64 ? 1 : (N <= 128 ? 2 : 3); }
static int tn_splits(int M, int K, int N) {
  const int tiles = cdiv(N, 64 * tn_ntw(N)) * cdiv(K, 64);
  int splits = cdiv(512, tiles);
  const int max_splits = cdiv(M, 2 * TM);
  if (splits > max_splits) splits = max_splits;
  if (splits < 1) splits = 1;
  return splits;
}

}  // namespace g2v

// used by the persistent rollout backward (dec_persist.hip): out_w[p] (n floats) = sum over nsplit slabs slab_w[p][split][n],
// out_b[p] (nb floats) likewise, for up to four problems in one launch; fixed summation order
int g2v_internal_slab_reduce4(const float* const* slab_w, float* const* out_w, const float* const* slab_b, float* const* out_b,
                              int nprob, int64_t n, int64_t nb, int nsplit, hipStream_t st) {
  using namespace g2v;
  SlabBatch sb;
  for (int p = 0; p < G2V_TN_BATCH; ++p) {
    const int pp = p < nprob ? p : 0;
    sb.slab_a[p] = slab_w[pp]; sb.out_a[p] = out_w[pp]; sb.slab_b[p] = slab_b[pp]; sb.out_b[p] = out_b[pp];
  }
  hipLaunchKernelGGL(slab_reduce2_kernel, dim3(cdiv(n, 32) + cdiv(nb, 32), nprob), dim3(256), 0, st, sb, n, nb, nsplit, 0,
                     cdiv(n, 32));
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

using namespace g2v;

extern "C" int g2v_linear_fwd(const float* x, int64_t ldx, int rows_inner, int64_t stride_outer, int64_t stride_inner,
                              const uint8_t* x_keep, float x_scale, const float* w, const float* bias, float* y,
                              int64_t ldy, int M, int K, int N, int act, g2v_stream_t stream) {
  G2V_REQUIRE(x && w && y, "null pointer");
  G2V_REQUIRE(M > 0 && K > 0 && N > 0, "non-positive size");
  G2V_REQUIRE(act >= 0 && act <= 2, "bad activation");
  RowMap am{ldx, rows_inner, stride_outer, stride_inner};
  if (rows_inner == 0 && launch_smallm(false, x, ldx, x_keep, x_scale, w, (int64_t)K, bias, y, ldy, M, K, N, act, 0,
                                       (hipStream_t)stream)) {
    G2V_CHECK_LAUNCH();
    return G2V_OK;
  }
  if (rows_inner == 0 && !x_keep &&
      launch_stream<false>(x, ldx, w, (int64_t)K, bias, y, ldy, M, K, N, act, 0, (hipStream_t)stream)) {
    G2V_CHECK_LAUNCH();
    return G2V_OK;
  }
  if (!x_keep && launch_k4(x, am, w, bias, y, ldy, M, K, N, act, (hipStream_t)stream)) {
    G2V_CHECK_LAUNCH();
    return G2V_OK;
  }
  dim3 grid(cdiv(M, BM), cdiv(N, BN));
  const bool a_vec = !x_keep && (K & 3) == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0 &&
                     (rows_inner > 0 ? ((stride_outer & 3) == 0 && (stride_inner & 3) == 0) : ((ldx & 3) == 0));
  if (a_vec && (reinterpret_cast<uintptr_t>(w) & 15) == 0)
    hipLaunchKernelGGL((gemm_nt_kernel<false, true>), grid, dim3(256), 0, (hipStream_t)stream, x, am, x_keep, x_scale, w,
                       (int64_t)K, bias, y, ldy, M, K, N, act, 0);
  else
  hipLaunchKernelGGL(gemm_nt_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, x, am, x_keep, x_scale, w,
                     (int64_t)K, bias, y, ldy, M, K, N, act, 0);
  G2V_CHECK_LAUNCH();
  return G2V_OK;
}

// ---- two linear layers in a row as ONE: g_p = (x W_in^T + b_in) W_p^T + b_p = x (W_p W_in)^T + (W_p b_in + b_p) --------------------------
// in_layer (D -> H) straight into the GRU input projections (H -> 3H, two directions; ref EncoderRNN :93-94) at the shipped dims,
// where D (40 / 45) < H (200): the composed weights W_p W_in are (3H x D), and projecting x with them is D / H of the arithmetic of
// projecting the layer's output -- whose only other reader, the weight gradient of W_p, is re-associated through in_layer too
// (g2v_linear_bwd_weight_chain2).  wc_p[g][d] = sum_h w_p[g][h] w_in[h][d], bc_p[g] = sum_h w_p[g][h] b_in[h] + b_p[g]; one thread
// per element, h ascending, w_in (+ b_in as a column) and the workgroup's rows of w_p staged in LDS (a 200-long chain of dependent
// global loads per element was 40 us).  Equal to the two-layer form to rounding.
__global__ __launch_bounds__(256) void linear_compose2_kernel(const float* __restrict__ w0, const float* __restrict__ b0,
                                                              const float* __restrict__ w1, const float* __restrict__ b1,
                                                              const float* __restrict__ w_in, const float* __restrict__ b_in,
                                                              float* __restrict__ wc0, float* __restrict__ bc0,
                                                              float* __restrict__ wc1, float* __restrict__ bc1, int G, int H, int D,
                                                              int staged) {
  extern __shared__ float sm[];      // staged: [H][D + 1] = w_in rows with b_in as column D | [rows][H] = this workgroup's rows of w_p
  const float* W = blockIdx.y ? w1 : w0;
  const float* Bp = blockIdx.y ? b1 : b0;
  float* WC = blockIdx.y ? wc1 : wc0;
  float* BC = blockIdx.y ? bc1 : bc0;
  const int D1 = D + 1, tid = threadIdx.x;
  if (staged) {
    const int rows = 256 / D1, g0 = blockIdx.x * rows;      // (staged implies D + 1 <= 256)
    float* wt = sm;
    float* wr = sm + (size_t)H * D1;
    for (int e = tid; e < H * D1; e += 256) {
      const int h = e / D1, d = e - h * D1;
      wt[e] = d < D ? w_in[(int64_t)h * D + d] : b_in[h];
    }
    for (int e = tid; e < rows * H; e += 256) {
      const int r = e / H, g = g0 + r;
      wr[e] = g < G ? W[(int64_t)g * H + (e - r * H)] : 0.f;
    }
    __syncthreads();
    const int r = tid / D1, d = tid - r * D1, g = g0 + r;
    if (r >= rows || g >= G) return;
    const float* a = wr + (size_t)r * H;
    float acc = 0.f;
#pragma unroll 8
    for (int h = 0; h < H; ++h) acc = fmaf(a[h], wt[h * D1 + d], acc);      // (h ascending: the order of the unstaged loop)
    if (d < D) WC[(int64_t)g * D + d] = acc;
    else BC[g] = acc + Bp[g];
    return;
  }
  const int e = blockIdx.x * 256 + tid;
  if (e >= G * D1) return;
  const int g = e / D1, d = e - g * D1;
  const float* wr = W + (int64_t)g * H;
  float acc = 0.f;
  if (d < D) {
    for (int h = 0; h < H; ++h) acc = fmaf(wr[h], w_in[(int64_t)h * D + d], acc);
    WC[(int64_t)g * D + d] = acc;
  } else {
    for (int h = 0; h < H; ++h) acc = fmaf(wr[h], b_in[h], acc);
    BC[g] = acc + Bp[g];
  }
}
extern "C" int g2v_linear_compose2(const float* w0, const float* b0, const float* w1, const float* b1, const float* w_in,
                                   const float* b_in, float* wc0, float* bc0, float* wc1, float* bc1, int G, int H, int D,
                                   g2v_stream_t stream) {
  G2V_REQUIRE(w0 && b0 && w1 && b1 && w_in && b_in && wc0 && bc0 && wc1 && bc1, "null pointer");
  G2V_REQUIRE(G > 0 && H > 0 && D > 0 && (int64_t)G * (D + 1) < (int64_t)1 << 31, "bad size");
  const int rows = D + 1 <= 256 ? 256 / (D + 1) : 0;
  const size_t lds = sizeof(float) * ((size_t)H * (D + 1) + (size_t)rows * H);
  const int staged = rows > 0 && lds <= 60 * 1024 ? 1 : 0;
  hipLaunchKernelGGL(linear_compose2_kernel, dim3(staged ? cdiv(G, rows) : cdiv(G * (D + 1), 256), 2), dim3(256), staged ? lds : 0,
                     (hipStream_t)stream, w0, b0, w1, b1, w_in, b_in, wc0, bc0, wc1, bc1, G, H, D, staged);
  G2V_CHECK_LAUNCH();
  return G2V_OK;
}

// y_a = act(x w_a^T + b_a), y_b = act(x w_b^T + b_b): the two directions' input projections of a bidirectional GRU layer (ref
// Autoencoder_VQVAE_model.py:94, nn.GRU(bidirectional=True): weight_ih_l0 / weight_ih_l0_reverse) read the same x.  At small row
// counts (2560 rows: 400 workgroups per product, 16 us each, back to back) ONE launch with grid.z = 2 runs both; each workgroup's
// arithmetic is that of g2v_linear_fwd's LDS-tiled kernel (bitwise the same y).  Shapes the other dense kernels serve, and
// unaligned operands, are two g2v_linear_fwd calls.
extern "C" int g2v_linear_fwd_pair(const float* x, int64_t ldx, const float* w_a, const float* bias_a, float* y_a, const float* w_b,
                                   const float* bias_b, float* y_b, int64_t ldy, int M, int K, int N, int act,
                                   g2v_stream_t stream) {
  G2V_REQUIRE(x && w_a && w_b && y_a && y_b, "null pointer");
  G2V_REQUIRE(M > 0 && K > 0 && N > 0, "non-positive size");
  G2V_REQUIRE(act >= 0 && act <= 2, "bad activation");
  const bool tiled = M > g2v_internal_options().smallm_max_rows && N > 192 && !(N == 64 && K == 135);      // (neither launch_smallm, _stream nor _k4)
  const bool vec = (K & 3) == 0 && (ldx & 3) == 0 && ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(w_a) |
                                                       reinterpret_cast<uintptr_t>(w_b)) & 15) == 0;
  if (tiled && vec && M <= 16384) {
    RowMap am{ldx, 0, 0, 0};
    hipLaunchKernelGGL((gemm_nt_kernel<false, true>), dim3(cdiv(M, BM), cdiv(N, BN), 2), dim3(256), 0, (hipStream_t)stream, x, am,
                       (const uint8_t*)nullptr, 1.0f, w_a, (int64_t)K, bias_a, y_a, ldy, M, K, N, act, 0, w_b, bias_b, y_b);
    G2V_CHECK_LAUNCH();
    return G2V_OK;
  }
  const int rc = g2v_linear_fwd(x, ldx, 0, 0, 0, nullptr, 1.0f, w_a, bias_a, y_a, ldy, M, K, N, act, stream);
  if (rc != G2V_OK) return rc;
  return g2v_linear_fwd(x, ldx, 0, 0, 0, nullptr, 1.0f, w_b, bias_b, y_b, ldy, M, K, N, act, stream);
}

// y_a = x w_a^T + b_a (M x N_a), y_b = x w_b^T + b_b (M x N_b): two g2v_linear_fwd calls on the same x; ONE launch (bitwise the
// two calls' results) where both would run the small-row-count kernel with 16 x 16 wave tiles, two launches otherwise.
extern "C" int g2v_linear_fwd_dual(const float* x, int64_t ldx, const float* w_a, const float* bias_a, float* y_a, int64_t ldya, int N_a,
                                   const float* w_b, const float* bias_b, float* y_b, int64_t ldyb, int N_b, int M, int K,
                                   g2v_stream_t stream) {
  G2V_REQUIRE(x && w_a && w_b && y_a && y_b, "null pointer");
  G2V_REQUIRE(M > 0 && K > 0 && N_a > 0 && N_b > 0 && ldx >= K && ldya >= N_a && ldyb >= N_b, "bad size");
  auto a16 = [](const void* q_) { return (reinterpret_cast<uintptr_t>(q_) & 15) == 0; };
  const int64_t tiles_a = (int64_t)cdiv(M, 16) * cdiv(N_a, 16), tiles_b = (int64_t)cdiv(M, 16) * cdiv(N_b, 16);
  if (M <= g2v_internal_options().smallm_max_rows && (K & 3) == 0 && (ldx & 3) == 0 && a16(x) && a16(w_a) && a16(w_b) &&
      tiles_a <= 2048 && tiles_b <= 2048) {
    DualNT d;
    d.W[0] = w_a; d.bias[0] = bias_a; d.out[0] = y_a; d.ldc[0] = ldya; d.N[0] = N_a;
    d.W[1] = w_b; d.bias[1] = bias_b; d.out[1] = y_b; d.ldc[1] = ldyb; d.N[1] = N_b;
    const int64_t tiles = tiles_a > tiles_b ? tiles_a : tiles_b;
    hipLaunchKernelGGL(gemm_smallm_dual_nt_kernel, dim3(cdiv(tiles, 4), 2), dim3(256), 0, (hipStream_t)stream, x, ldx, d, M, K);
    G2V_CHECK_LAUNCH();
    return G2V_OK;
  }
  const int rc = g2v_linear_fwd(x, ldx, 0, 0, 0, nullptr, 1.0f, w_a, bias_a, y_a, ldya, M, K, N_a, 0, stream);
  if (rc != G2V_OK) return rc;
  return g2v_linear_fwd(x, ldx, 0, 0, 0, nullptr, 1.0f, w_b, bias_b, y_b, ldyb, M, K, N_b, 0, stream);
}

extern "C" int g2v_linear_bwd_data(const float* dy, int64_t lddy, const float* w, float* dx, int64_t lddx, int M, int K,
                                   int N, int accumulate, g2v_stream_t stream) {
  G2V_REQUIRE(dy && w && dx, "null pointer");
  G2V_REQUIRE(M > 0 && K > 0 && N > 0, "non-positive size");
  RowMap am{lddy, 0, 0, 0};
  if (launch_smallm(true, dy, lddy, nullptr, 1.0f, w, (int64_t)K, nullptr, dx, lddx, M, N, K, 0, accumulate,
                    (hipStream_t)stream)) {
    G2V_CHECK_LAUNCH();
    return G2V_OK;
  }
  if (launch_stream<true>(dy, lddy, w, (int64_t)K, nullptr, dx, lddx, M, N, K, 0, accumulate, (hipStream_t)stream)) {
    G2V_CHECK_LAUNCH();
    return G2V_OK;
  }
  dim3 grid(cdiv(M, BM), cdiv(K, BN));
  // output feature = k (K of them), contraction over n (N): Bop[k][n] = w[n*K + k]
  if ((N & 3) == 0 && (K & 3) == 0 && (lddy & 3) == 0 && (reinterpret_cast<uintptr_t>(dy) & 15) == 0 &&
      (reinterpret_cast<uintptr_t>(w) & 15) == 0)
    hipLaunchKernelGGL((gemm_nt_kernel<true, true>), grid, dim3(256), 0, (hipStream_t)stream, dy, am, (const uint8_t*)nullptr,
                       1.0f, w, (int64_t)K, (const float*)nullptr, dx, lddx, M, N, K, 0, accumulate);
  else
  hipLaunchKernelGGL(gemm_nt_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, dy, am, (const uint8_t*)nullptr,
                     1.0f, w, (int64_t)K, (const float*)nullptr, dx, lddx, M, N, K, 0, accumulate);
  G2V_CHECK_LAUNCH();
  return G2V_OK;
}

extern "C" size_t g2v_linear_bwd_weight_workspace(int M, int K, int N) {
  if (M <= 0 || K <= 0 || N <= 0) return 0;
  if ((M & 15) && M >= 4096 + 16) {      // ragged rows: the whole 16-row groups go through the kernels below as M & ~15 rows
    const size_t a = g2v_linear_bwd_weight_workspace(M & ~15, K, N);
    const size_t b = (size_t)tn_splits(M, K, N) * ((size_t)N * K + N) * sizeof(float);
    return a > b ? a : b;
  }
  int splits = tn_splits(M, K, N);
  for (int nr = 2; nr <= 4; nr += 2) {       // either row-range variant of the wave-autonomous path
    const int wg = tn_wave_grid(M, K, N, false, nullptr, 1, nr);
    if (wg > splits) splits = wg;
  }
  {
    int gs = 0;
    if (tn_gen_grid(M, K, N, false, 1, nullptr, nullptr, &gs) > 0 && gs > splits) splits = gs;
  }
  return (size_t)splits * ((size_t)N * K + N) * sizeof(float);
}

// ---- small M weight gradient (Part d at B = 128: 640 x 600 -> dW 600 x 200): one workgroup per 16 x 16 tile of dW ----------
// dW[n][k] (+)= sum_m dy[m][n] xin[m][k], db[n] (+)= sum_m dy[m][n].  The four waves split the rows, each pulls both operands
// straight from L2 as MFMA fragments (dword loads: 16 lanes x 4 B contiguous along n / k, 4 rows per MFMA; SMW_NB 16-row
// blocks per request burst, the next burst in flight during the MFMAs), the four partial tiles meet in LDS in a fixed order:
// deterministic, no slabs, no second launch.  The LDS-tiled split kernel + slab reduction it replaces here cost 17 + 5 us.
struct SmallWgradBatch {
  const float* dy[G2V_TN_BATCH];
  const float* x[G2V_TN_BATCH];
  float* dw[G2V_TN_BATCH];
  float* db[G2V_TN_BATCH];
};
// Which (problem, tile group) a workgroup takes.  xpp == 0: grid (groups, problems).  xpp > 0 (1-D grid of 8 slots-per-XCD
// workgroups, nprob = 8 / xpp problems): workgroup ids go round the 8 XCDs, so XCD j takes problem j / xpp only and every xpp-th tile
// group of it -- the workgroups of an XCD then stream the rows of ONE problem's operands through that XCD's 4 MiB L2 at about the
// same pace (four 600 x 200 products at 2560 rows are 33 MB: with every XCD holding tiles of all four, each L2 missed on all of it).
// Placement only: which workgroup computes a tile does not change the tile's arithmetic.
__device__ __forceinline__ bool smw_decode_grid(int xpp, int groups, int& prob, int& grp) {
  if (xpp <= 0) {
    prob = blockIdx.y; grp = blockIdx.x;
    return true;
  }
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  prob = xcd / xpp;
  grp = xcd % xpp + xpp * slot;
  return grp < groups;
}
// xmap.rows_inner > 0: row m of x is row (m_base + m) of a row-mapped tensor (the leftover rows of the ragged in_layer gradient)
// NW waves per workgroup split the rows (4, or 16 with half the burst where a product has too few tiles to fill the chip: a wave's
// rows are a chain of memory round trips -- 640 rows in five bursts -- and that chain, not the arithmetic, is the kernel's time).
template <bool KEEP, int NW = 4, int SMW_NB = 8>
__global__ __launch_bounds__(64 * NW) void gemm_tn_smallm_kernel(SmallWgradBatch sb, int64_t lddy, int64_t ldx,
                                                             const uint8_t* __restrict__ keep, float scale, int M, int K,
                                                             int N, int accumulate, RowMap xmap = RowMap{0, 0, 0, 0},
                                                             int m_base = 0, int xpp = 0) {
  __shared__ float red[NW - 1][64 * 4 + 16];
  const int tiles_k = (K + 15) >> 4;
  int prob, grp;
  if (!smw_decode_grid(xpp, ((N + 15) >> 4) * tiles_k, prob, grp)) return;
  const float* __restrict__ dY = sb.dy[prob];
  const float* __restrict__ X = sb.x[prob];
  float* __restrict__ dW = sb.dw[prob];
  float* __restrict__ dB = sb.db[prob];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, i = lane & 15, q = lane >> 4;
  const int nt = grp / tiles_k, kt = grp - nt * tiles_k;
  const int n = nt * 16 + i, k = kt * 16 + i;
  const bool nok = n < N, kok = k < K;
  const float* dyc = dY + (nok ? n : 0);
  const float* xc = X + (kok ? k : 0);
  const uint8_t* kc = KEEP ? keep + (kok ? k : 0) : nullptr;
  const int per = (((M + NW - 1) / NW) + 15) & ~15;                 // rows per wave, whole 16-row blocks
  const int mb = wave * per, me = mb + per < M ? mb + per : M;
  f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
  float dbs = 0.f;
  float a0[SMW_NB][4], b0[SMW_NB][4], a1[SMW_NB][4], b1[SMW_NB][4];
  uint8_t k0[SMW_NB][4], k1[SMW_NB][4];
  auto fetch = [&](int m0, float (&a)[SMW_NB][4], float (&b)[SMW_NB][4], uint8_t (&kp)[SMW_NB][4]) {
#pragma unroll
    for (int blk = 0; blk < SMW_NB; ++blk)
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const int m = m0 + 16 * blk + 4 * q + s, mc = m < me ? m : (M - 1);       // past the range: a valid row, zeroed in use()
        a[blk][s] = dyc[(int64_t)mc * lddy];
        b[blk][s] = xc[xmap.rows_inner > 0 ? row_off(xmap, m_base + mc) : (int64_t)mc * ldx];
        if (KEEP) kp[blk][s] = kc[(int64_t)mc * K];
      }
  };
  auto use = [&](int m0, const float (&a)[SMW_NB][4], const float (&b)[SMW_NB][4], const uint8_t (&kp)[SMW_NB][4]) {
#pragma unroll
    for (int blk = 0; blk < SMW_NB; ++blk)
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const bool ok = m0 + 16 * blk + 4 * q + s < me;
        const float av = (ok && nok) ? a[blk][s] : 0.f;
        float bv = (ok && kok) ? b[blk][s] : 0.f;
        if (KEEP) bv = kp[blk][s] ? bv * scale : 0.f;
        dbs += av;
        acc = mfma16(av, bv, acc);
      }
  };
  constexpr int CH = 16 * SMW_NB;
  if (mb < me) {
    fetch(mb, a0, b0, k0);
    for (int m0 = mb; m0 < me; m0 += 2 * CH) {
      if (m0 + CH < me) fetch(m0 + CH, a1, b1, k1);
      use(m0, a0, b0, k0);
      if (m0 + CH >= me) break;
      if (m0 + 2 * CH < me) fetch(m0 + 2 * CH, a0, b0, k0);
      use(m0 + CH, a1, b1, k1);
    }
  }
  dbs += __shfl_xor(dbs, 16);
  dbs += __shfl_xor(dbs, 32);
  if (wave > 0) {
#pragma unroll
    for (int r = 0; r < 4; ++r) red[wave - 1][lane * 4 + r] = acc[r];
    if (q == 0) red[wave - 1][256 + i] = dbs;
  }
  __syncthreads();
  if (wave != 0) return;
#pragma unroll
  for (int w = 0; w < NW - 1; ++w) {
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[r] += red[w][lane * 4 + r];
    dbs += red[w][256 + i];
  }
  // lane holds dW[n = nt*16 + 4 q + r][k = kt*16 + (lane & 15)]
  if (kok) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int no = nt * 16 + 4 * q + r;
      if (no < N) {
        float* o = dW + (int64_t)no * K + k;
        *o = accumulate ? *o + acc[r] : acc[r];
      }
    }
  }
  if (dB && kt == 0 && q == 0 && nok) dB[n] = accumulate ? dB[n] + dbs : dbs;
}

// The same with TN x TK tiles of dW per workgroup (register tiling): every operand value a wave loads feeds TK (TN) MFMAs
// instead of one, so the L2 traffic per MFMA drops by 2 TN TK / (TN + TK) -- at the reference's own shape (2432 rows, four
// 600 x 200 products per launch) the 1 x 1 kernel moves 150 MB per product through L2 and takes ~100 us per launch.  Same
// row split over the waves, same accumulation order per output element, same cross-wave sum: bitwise the 1 x 1 kernel's dW.
// Plain rows only (no keep mask, no row map).
template <int TN, int TK, int SMW_NB2>
__global__ __launch_bounds__(256) void gemm_tn_smallm_rt_kernel(SmallWgradBatch sb, int64_t lddy, int64_t ldx, int M, int K,
                                                                int N, int accumulate, int xpp) {
  __shared__ float red[3][TN * TK * 256 + TN * 16];
  const int groups_k = (((K + 15) >> 4) + TK - 1) / TK;
  int prob, grp;
  if (!smw_decode_grid(xpp, ((((N + 15) >> 4) + TN - 1) / TN) * groups_k, prob, grp)) return;
  const float* __restrict__ dY = sb.dy[prob];
  const float* __restrict__ X = sb.x[prob];
  float* __restrict__ dW = sb.dw[prob];
  float* __restrict__ dB = sb.db[prob];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, i = lane & 15, q = lane >> 4;
  const int ng = grp / groups_k, kg = grp - ng * groups_k;
  bool nok[TN], kok[TK];
  const float* dyc[TN];
  const float* xc[TK];
#pragma unroll
  for (int t = 0; t < TN; ++t) {
    const int n = (ng * TN + t) * 16 + i;
    nok[t] = n < N;
    dyc[t] = dY + (nok[t] ? n : 0);
  }
#pragma unroll
  for (int u = 0; u < TK; ++u) {
    const int k = (kg * TK + u) * 16 + i;
    kok[u] = k < K;
    xc[u] = X + (kok[u] ? k : 0);
  }
  const int per = (((M + 3) / 4) + 15) & ~15;                       // rows per wave, whole 16-row blocks (as the 1 x 1 kernel)
  const int mb = wave * per, me = mb + per < M ? mb + per : M;
  f32x4 acc[TN][TK];
  float dbs[TN];
#pragma unroll
  for (int t = 0; t < TN; ++t) {
    dbs[t] = 0.f;
#pragma unroll
    for (int u = 0; u < TK; ++u) acc[t][u] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }
  float a0[SMW_NB2][4][TN], b0[SMW_NB2][4][TK], a1[SMW_NB2][4][TN], b1[SMW_NB2][4][TK];
  auto fetch = [&](int m0, float (&a)[SMW_NB2][4][TN], float (&b)[SMW_NB2][4][TK]) {
#pragma unroll
    for (int blk = 0; blk < SMW_NB2; ++blk)
#pragma unroll
      for (int s2 = 0; s2 < 4; ++s2) {
        const int m = m0 + 16 * blk + 4 * q + s2, mc = m < me ? m : (M - 1);       // past the range: a valid row, zeroed in use()
#pragma unroll
        for (int t = 0; t < TN; ++t) a[blk][s2][t] = dyc[t][(int64_t)mc * lddy];
#pragma unroll
        for (int u = 0; u < TK; ++u) b[blk][s2][u] = xc[u][(int64_t)mc * ldx];
      }
  };
  auto use = [&](int m0, const float (&a)[SMW_NB2][4][TN], const float (&b)[SMW_NB2][4][TK]) {
#pragma unroll
    for (int blk = 0; blk < SMW_NB2; ++blk)
#pragma unroll
      for (int s2 = 0; s2 < 4; ++s2) {
        const bool ok = m0 + 16 * blk + 4 * q + s2 < me;
        float av[TN], bv[TK];
#pragma unroll
        for (int t = 0; t < TN; ++t) {
          av[t] = (ok && nok[t]) ? a[blk][s2][t] : 0.f;
          dbs[t] += av[t];
        }
#pragma unroll
        for (int u = 0; u < TK; ++u) bv[u] = (ok && kok[u]) ? b[blk][s2][u] : 0.f;
#pragma unroll
        for (int t = 0; t < TN; ++t)
#pragma unroll
          for (int u = 0; u < TK; ++u) acc[t][u] = mfma16(av[t], bv[u], acc[t][u]);
      }
  };
  constexpr int CH = 16 * SMW_NB2;
  if (mb < me) {
    fetch(mb, a0, b0);
    for (int m0 = mb; m0 < me; m0 += 2 * CH) {
      if (m0 + CH < me) fetch(m0 + CH, a1, b1);
      use(m0, a0, b0);
      if (m0 + CH >= me) break;
      if (m0 + 2 * CH < me) fetch(m0 + 2 * CH, a0, b0);
      use(m0 + CH, a1, b1);
    }
  }
#pragma unroll
  for (int t = 0; t < TN; ++t) {
    dbs[t] += __shfl_xor(dbs[t], 16);
    dbs[t] += __shfl_xor(dbs[t], 32);
  }
  if (wave > 0) {
#pragma unroll
    for (int t = 0; t < TN; ++t) {
#pragma unroll
      for (int u = 0; u < TK; ++u)
#pragma unroll
        for (int r = 0; r < 4; ++r) red[wave - 1][(t * TK + u) * 256 + lane * 4 + r] = acc[t][u][r];
      if (q == 0) red[wave - 1][TN * TK * 256 + t * 16 + i] = dbs[t];
    }
  }
  __syncthreads();
  if (wave != 0) return;
#pragma unroll
  for (int w = 0; w < 3; ++w)
#pragma unroll
    for (int t = 0; t < TN; ++t) {
#pragma unroll
      for (int u = 0; u < TK; ++u)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[t][u][r] += red[w][(t * TK + u) * 256 + lane * 4 + r];
      dbs[t] += red[w][TN * TK * 256 + t * 16 + i];
    }
  // lane holds dW[n = tile_n*16 + 4 q + r][k = tile_k*16 + (lane & 15)]
#pragma unroll
  for (int t = 0; t < TN; ++t) {
#pragma unroll
    for (int u = 0; u < TK; ++u) {
      const int k = (kg * TK + u) * 16 + i;
      if (k < K) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int no = (ng * TN + t) * 16 + 4 * q + r;
          if (no < N) {
            float* o = dW + (int64_t)no * K + k;
            *o = accumulate ? *o + acc[t][u][r] : acc[t][u][r];
          }
        }
      }
    }
    const int n = (ng * TN + t) * 16 + i;
    if (dB && kg == 0 && q == 0 && n < N) dB[n] = accumulate ? dB[n] + dbs[t] : dbs[t];
  }
}


// The same product with the operands STAGED THROUGH LDS: the register-tiled kernel above pulls every MFMA operand as a dword from
// L2 (16 lanes x 4 B per row: one 64-byte segment per row and instruction) and runs at a fifth of the MFMA rate at the reference's
// own shape.  Here a wave copies 32 rows x (TN + TK) x 16 columns of its row range with float4 loads (whole 128 / 256-byte row
// segments) into its own LDS region -- the next 32 rows are in flight in registers meanwhile -- and reads the fragments back as
// dwords (row stride % 8 == 4 floats: the four row groups of a fragment read land on disjoint banks).  Same row split over the
// waves, same rows per MFMA (16 blk + 4 q + s), same order of the accumulations and of the cross-wave sum as the two kernels above:
// bitwise their dW and db.  Needs N, K, lddy, ldx multiples of 4 and 16-byte aligned operands (the launcher checks).
template <int TN, int TK, int NW>
__global__ __launch_bounds__(64 * NW) void gemm_tn_smallm_lds_kernel(SmallWgradBatch sb, int64_t lddy, int64_t ldx, int M, int K,
                                                                     int N, int accumulate, int xpp) {
  constexpr int R = 32, WN = TN * 16, WK = TK * 16, LD = WN + WK + 4;
  static_assert(LD % 8 == 4, "row stride: the four row groups of a fragment read on disjoint banks");
  constexpr int STG = R * LD, RED = TN * TK * 256 + TN * 16;
  constexpr int NA = R * TN * 4 / 64, NB = R * TK * 4 / 64;      // float4 loads per lane and stage
  extern __shared__ __attribute__((aligned(16))) float smem[];   // max(NW STG, (NW - 1) RED) floats
  const int groups_k = (((K + 15) >> 4) + TK - 1) / TK;
  int prob, grp;
  if (!smw_decode_grid(xpp, ((((N + 15) >> 4) + TN - 1) / TN) * groups_k, prob, grp)) return;
  const float* __restrict__ dY = sb.dy[prob];
  const float* __restrict__ X = sb.x[prob];
  float* __restrict__ dW = sb.dw[prob];
  float* __restrict__ dB = sb.db[prob];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, i = lane & 15, q = lane >> 4;
  const int ng = grp / groups_k, kg = grp - ng * groups_k;
  const int n0 = ng * WN, k0 = kg * WK;
  const int per = (((M + NW - 1) / NW) + 15) & ~15;                 // rows per wave, whole 16-row blocks (NW = 4: as the 1 x 1 kernel)
  const int mb = wave * per, me = mb + per < M ? mb + per : M;
  float* stg = smem + wave * STG;
  f32x4 acc[TN][TK];
  float dbs[TN];
#pragma unroll
  for (int t = 0; t < TN; ++t) {
    dbs[t] = 0.f;
#pragma unroll
    for (int u = 0; u < TK; ++u) acc[t][u] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }
  float4 pa[NA], pb[NB];
  auto fetch = [&](int m0, float4* xa, float4* xb) {      // rows past the range and columns past N / K: zeros (they add nothing)
#pragma unroll
    for (int j = 0; j < NA; ++j) {
      const int e = lane + 64 * j, r = e / (TN * 4), c = (e % (TN * 4)) * 4;
      const bool ok = m0 + r < me && n0 + c < N;
      xa[j] = ld4_or_zero(dY + (ok ? (int64_t)(m0 + r) * lddy + n0 + c : (int64_t)0), ok);
    }
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      const int e = lane + 64 * j, r = e / (TK * 4), c = (e % (TK * 4)) * 4;
      const bool ok = m0 + r < me && k0 + c < K;
      xb[j] = ld4_or_zero(X + (ok ? (int64_t)(m0 + r) * ldx + k0 + c : (int64_t)0), ok);
    }
  };
  auto stash = [&](const float4* xa, const float4* xb) {
    __builtin_amdgcn_wave_barrier();      // (the region is this wave's own: its LDS operations execute in order)
#pragma unroll
    for (int j = 0; j < NA; ++j) {
      const int e = lane + 64 * j, r = e / (TN * 4), c = (e % (TN * 4)) * 4;
      *reinterpret_cast<float4*>(stg + r * LD + c) = xa[j];
    }
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      const int e = lane + 64 * j, r = e / (TK * 4), c = (e % (TK * 4)) * 4;
      *reinterpret_cast<float4*>(stg + r * LD + WN + c) = xb[j];
    }
  };
  auto use = [&]() {
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int blk = 0; blk < R / 16; ++blk)
#pragma unroll
      for (int s2 = 0; s2 < 4; ++s2) {
        const float* row = stg + (16 * blk + 4 * q + s2) * LD + i;
        float av[TN], bv[TK];
#pragma unroll
        for (int t = 0; t < TN; ++t) {
          av[t] = row[16 * t];
          dbs[t] += av[t];
        }
#pragma unroll
        for (int u = 0; u < TK; ++u) bv[u] = row[WN + 16 * u];
#pragma unroll
        for (int t = 0; t < TN; ++t)
#pragma unroll
          for (int u = 0; u < TK; ++u) acc[t][u] = mfma16(av[t], bv[u], acc[t][u]);
      }
  };
  if (mb < me) {
    fetch(mb, pa, pb);
    for (int m0 = mb; m0 < me; m0 += R) {      // (two stages in flight, 2 x 4 / 3 x 3 / 1 x 2 tiles: 50-67 us against 53, r05_aa / r05_ab)
      stash(pa, pb);
      if (m0 + R < me) fetch(m0 + R, pa, pb);
      use();
    }
  }
#pragma unroll
  for (int t = 0; t < TN; ++t) {
    dbs[t] += __shfl_xor(dbs[t], 16);
    dbs[t] += __shfl_xor(dbs[t], 32);
  }
  __syncthreads();      // the staging regions are dead: the partial tiles meet in the same memory
  float (*red)[RED] = reinterpret_cast<float (*)[RED]>(smem);
  if (wave > 0) {
#pragma unroll
    for (int t = 0; t < TN; ++t) {
#pragma unroll
      for (int u = 0; u < TK; ++u)
#pragma unroll
        for (int r = 0; r < 4; ++r) red[wave - 1][(t * TK + u) * 256 + lane * 4 + r] = acc[t][u][r];
      if (q == 0) red[wave - 1][TN * TK * 256 + t * 16 + i] = dbs[t];
    }
  }
  __syncthreads();
  if (wave != 0) return;
#pragma unroll
  for (int w = 0; w < NW - 1; ++w)
#pragma unroll
    for (int t = 0; t < TN; ++t) {
#pragma unroll
      for (int u = 0; u < TK; ++u)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[t][u][r] += red[w][(t * TK + u) * 256 + lane * 4 + r];
      dbs[t] += red[w][TN * TK * 256 + t * 16 + i];
    }
#pragma unroll
  for (int t = 0; t < TN; ++t) {
#pragma unroll
    for (int u = 0; u < TK; ++u) {
      const int k = (kg * TK + u) * 16 + i;
      if (k < K) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int no = (ng * TN + t) * 16 + 4 * q + r;
          if (no < N) {
            float* o = dW + (int64_t)no * K + k;
            *o = accumulate ? *o + acc[t][u][r] : acc[t][u][r];
          }
        }
      }
    }
    const int n = (ng * TN + t) * 16 + i;
    if (dB && kg == 0 && q == 0 && n < N) dB[n] = accumulate ? dB[n] + dbs[t] : dbs[t];
  }
}

// ---- the weight gradient of a layer in FRONT of two parallel layers, from their weight-gradient-shaped products --------------------
// y = x W_in^T + b_in feeds two layers g_p = y W_p^T (p = 0, 1: the two directions' input projections of the bidirectional
// encoder GRU).  Instead of dy = dg_0 W_0 + dg_1 W_1 ((M x G)(G x H) twice, M = T B rows) and dW_in = dy^T x,
//   dW_in = W_0^T (dg_0^T x) + W_1^T (dg_1^T x),    db_in = W_0^T (dg_0^T 1) + W_1^T (dg_1^T 1):
// the inner products P_p = dg_p^T x (G x D) and c_p = column sums of dg_p are ordinary weight-gradient products with K = D (D = 40
// against H = 200 at the reference's dims: a fifth of the arithmetic of dy, and dy -- which nothing else reads -- is never formed);
// this kernel is the outer one, contraction over the G rows of (W_p, P_p): one workgroup per 16 x 16 tile of [dW_in | db_in]
// (column D is the bias), 16 waves: waves 0-7 an eighth of pair 0's rows each, waves 8-15 pair 1's (a wave's rows are ONE request
// burst at G = 600: the kernel is a memory round trip and a fixed-order sum of the 16 partial tiles).
__device__ __forceinline__ void wgrad_fold2_body(int bid, float (*red)[64 * 4], const float* __restrict__ w0,
                                                 const float* __restrict__ w1, const float* __restrict__ p0,
                                                 const float* __restrict__ p1, const float* __restrict__ c0,
                                                 const float* __restrict__ c1, float* __restrict__ dw, float* __restrict__ db, int G,
                                                 int H, int D, int accumulate) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, i = lane & 15, q = lane >> 4;
  const int tiles_d = (D + 1 + 15) >> 4;
  const int ht = bid / tiles_d, dt = bid - ht * tiles_d;
  const int h = ht * 16 + i, d = dt * 16 + i;
  const bool hok = h < H, dok = d <= D;
  const float* W = (wave >> 3) ? w1 : w0;
  const float* P = (wave >> 3) ? p1 : p0;
  const float* Cv = (wave >> 3) ? c1 : c0;
  const int per = (((G + 7) / 8) + 15) & ~15;
  const int mb = (wave & 7) * per, me = mb + per < G ? mb + per : G;
  f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
  constexpr int NBK = 5;      // 16-row blocks per request burst
  for (int m0 = mb; m0 < me; m0 += 16 * NBK) {
    float av[NBK][4], bv[NBK][4];
#pragma unroll
    for (int blk = 0; blk < NBK; ++blk)
#pragma unroll
      for (int s2 = 0; s2 < 4; ++s2) {
        const int m = m0 + 16 * blk + 4 * q + s2;
        const bool ok = m < me;
        const int mc = ok ? m : 0;
        av[blk][s2] = (ok && hok) ? W[(int64_t)mc * H + h] : 0.f;
        bv[blk][s2] = (ok && dok) ? (d < D ? P[(int64_t)mc * D + d] : Cv[mc]) : 0.f;
      }
#pragma unroll
    for (int blk = 0; blk < NBK; ++blk)
#pragma unroll
      for (int s2 = 0; s2 < 4; ++s2) acc = mfma16(av[blk][s2], bv[blk][s2], acc);
  }
  if (wave > 0) {
#pragma unroll
    for (int r = 0; r < 4; ++r) red[wave - 1][lane * 4 + r] = acc[r];
  }
  __syncthreads();
  if (wave != 0) return;
#pragma unroll
  for (int w = 0; w < 15; ++w)
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[r] += red[w][lane * 4 + r];
  // lane holds out[h = ht*16 + 4 q + r][d = dt*16 + (lane & 15)]
  if (dok) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int ho = ht * 16 + 4 * q + r;
      if (ho < H) {
        float* o = d < D ? dw + (int64_t)ho * D + d : db + ho;
        *o = accumulate ? *o + acc[r] : acc[r];
      }
    }
  }
}
__global__ __launch_bounds__(1024) void wgrad_fold2_kernel(const float* __restrict__ w0, const float* __restrict__ w1,
                                                          const float* __restrict__ p0, const float* __restrict__ p1,
                                                          const float* __restrict__ c0, const float* __restrict__ c1,
                                                          float* __restrict__ dw, float* __restrict__ db, int G, int H, int D,
                                                          int accumulate) {
  __shared__ float red[15][64 * 4];
  wgrad_fold2_body(blockIdx.x, red, w0, w1, p0, p1, c0, c1, dw, db, G, H, D, accumulate);
}
// ---- the weight gradient of a layer BEHIND a linear layer, from the same products ---------------------------------------------------
// g = y W^T with y = x W_in^T + b_in (the GRU input projection behind in_layer): dW = dg^T y = (dg^T x) W_in^T + (dg^T 1) b_in^T
// = P W_in^T + c b_in^T with the P, c of g2v_linear_bwd_weight_fold2 -- a (G x D)(D x H) product instead of (G x M)(M x H).
// A workgroup takes CHAIN2_ROWS rows g and every h, W_in^T staged in LDS once (where (D + 1) H floats fit); per output element d
// ascending, then the bias term; blockIdx.y = which of the two (P, c, dW).
constexpr int CHAIN2_ROWS = 8, CHAIN2_MAXD = 255;      // rows g per workgroup; largest D of the staged form
// staged form: W_in^T (row D = b_in) and the workgroup's rows of P (column D = c) in LDS; NSUB x 256 threads, sub-group s takes rows
// s, s + NSUB, ... of the workgroup's CHAIN2_ROWS; per output element d ascending, the bias term last
template <int NSUB>
__device__ __forceinline__ void wgrad_chain2_body(int bx, int dir, float* wt, const float* __restrict__ p0, const float* __restrict__ p1,
                                                  const float* __restrict__ c0, const float* __restrict__ c1,
                                                  const float* __restrict__ w_in, const float* __restrict__ b_in,
                                                  float* __restrict__ dw0, float* __restrict__ dw1, int G, int H, int D,
                                                  int accumulate) {
  constexpr int NT = 256 * NSUB, RPS = CHAIN2_ROWS / NSUB;
  float* ps = wt + (size_t)(D + 1) * H;      // [CHAIN2_ROWS][D + 1]
  const float* P = dir ? p1 : p0;
  const float* Cv = dir ? c1 : c0;
  float* dW = dir ? dw1 : dw0;
  const int tid = threadIdx.x, sub = tid >> 8, t = tid & 255, g0 = bx * CHAIN2_ROWS;
  for (int e = tid; e < H * D; e += NT) {
    const int h = e / D, d = e - h * D;
    wt[d * H + h] = w_in[e];
  }
  for (int h = tid; h < H; h += NT) wt[D * H + h] = b_in[h];
  for (int e = tid; e < CHAIN2_ROWS * (D + 1); e += NT) {
    const int r = e / (D + 1), d = e - r * (D + 1), g = g0 + r;
    ps[r * (D + 1) + d] = g < G ? (d < D ? P[(int64_t)g * D + d] : Cv[g]) : 0.f;
  }
  __syncthreads();
  for (int h = t; h < H; h += 256) {
    float acc[RPS];
#pragma unroll
    for (int r = 0; r < RPS; ++r) acc[r] = 0.f;
    for (int d = 0; d <= D; ++d) {
      const float w = wt[d * H + h];
#pragma unroll
      for (int r = 0; r < RPS; ++r) acc[r] = fmaf(ps[(sub + NSUB * r) * (D + 1) + d], w, acc[r]);
    }
#pragma unroll
    for (int r = 0; r < RPS; ++r) {
      const int g = g0 + sub + NSUB * r;
      if (g < G) {
        float* o = dW + (int64_t)g * H + h;
        *o = accumulate ? *o + acc[r] : acc[r];
      }
    }
  }
}
static size_t chain2_lds_bytes(int H, int D) { return sizeof(float) * ((size_t)(D + 1) * H + (size_t)CHAIN2_ROWS * (D + 1)); }
__global__ __launch_bounds__(256) void wgrad_chain2_kernel(const float* __restrict__ p0, const float* __restrict__ p1,
                                                           const float* __restrict__ c0, const float* __restrict__ c1,
                                                           const float* __restrict__ w_in, const float* __restrict__ b_in,
                                                           float* __restrict__ dw0, float* __restrict__ dw1, int G, int H, int D,
                                                           int accumulate, int staged) {
  extern __shared__ float wt[];
  if (staged) {
    wgrad_chain2_body<1>(blockIdx.x, blockIdx.y, wt, p0, p1, c0, c1, w_in, b_in, dw0, dw1, G, H, D, accumulate);
    return;
  }
  const float* P = blockIdx.y ? p1 : p0;
  const float* Cv = blockIdx.y ? c1 : c0;
  float* dW = blockIdx.y ? dw1 : dw0;
  const int tid = threadIdx.x, g0 = blockIdx.x * CHAIN2_ROWS;
  const int g1 = g0 + CHAIN2_ROWS < G ? g0 + CHAIN2_ROWS : G;
  for (int g = g0; g < g1; ++g) {
    const float* pr = P + (int64_t)g * D;
    const float cg = Cv[g];
    for (int h = tid; h < H; h += 256) {
      const float* wr = w_in + (int64_t)h * D;
      float acc = 0.f;
      for (int d = 0; d < D; ++d) acc = fmaf(pr[d], wr[d], acc);
      acc = fmaf(cg, b_in[h], acc);
      float* o = dW + (int64_t)g * H + h;
      *o = accumulate ? *o + acc : acc;
    }
  }
}
// both in ONE launch (they read the same P, c and nothing of each other): workgroups [0, nfold) the fold's tiles, the rest the
// chain's (row block, direction) pairs with four 256-thread sub-groups; each workgroup's arithmetic is that of the two kernels above
__global__ __launch_bounds__(1024) void wgrad_fold_chain2_kernel(const float* __restrict__ w0, const float* __restrict__ w1,
                                                                 const float* __restrict__ p0, const float* __restrict__ p1,
                                                                 const float* __restrict__ c0, const float* __restrict__ c1,
                                                                 const float* __restrict__ w_in, const float* __restrict__ b_in,
                                                                 float* __restrict__ dw_in, float* __restrict__ db_in,
                                                                 float* __restrict__ dw0, float* __restrict__ dw1, int G, int H, int D,
                                                                 int nfold, int nchain_x) {
  extern __shared__ float sm[];      // max(15 x 256 floats of partial tiles, the chain's staging)
  if ((int)blockIdx.x < nfold) {
    wgrad_fold2_body(blockIdx.x, reinterpret_cast<float (*)[64 * 4]>(sm), w0, w1, p0, p1, c0, c1, dw_in, db_in, G, H, D, 0);
  } else {
    const int bb = blockIdx.x - nfold;
    wgrad_chain2_body<4>(bb % nchain_x, bb / nchain_x, sm, p0, p1, c0, c1, w_in, b_in, dw0, dw1, G, H, D, 0);
  }
}
extern "C" int g2v_linear_bwd_weight_chain2(const float* p0, const float* p1, const float* c0, const float* c1, const float* w_in,
                                            const float* b_in, float* dw0, float* dw1, int G, int H, int D, int accumulate,
                                            g2v_stream_t stream) {
  G2V_REQUIRE(p0 && p1 && c0 && c1 && w_in && b_in && dw0 && dw1, "null pointer");
  G2V_REQUIRE(G > 0 && H > 0 && D > 0 && (int64_t)G * H < (int64_t)1 << 31, "bad size");
  const size_t lds = chain2_lds_bytes(H, D);
  const int staged = lds <= 48 * 1024 && D <= CHAIN2_MAXD ? 1 : 0;
  hipLaunchKernelGGL(wgrad_chain2_kernel, dim3(cdiv(G, CHAIN2_ROWS), 2), dim3(256), staged ? lds : 0, (hipStream_t)stream, p0, p1, c0,
                     c1, w_in, b_in, dw0, dw1, G, H, D, accumulate ? 1 : 0, staged);
  G2V_CHECK_LAUNCH();
  return G2V_OK;
}
extern "C" int g2v_linear_bwd_weight_fold2(const float* w0, const float* w1, const float* p0, const float* p1, const float* c0,
                                           const float* c1, float* dw, float* db, int G, int H, int D, int accumulate,
                                           g2v_stream_t stream) {
  G2V_REQUIRE(w0 && w1 && p0 && p1 && c0 && c1 && dw && db, "null pointer");
  G2V_REQUIRE(G > 0 && H > 0 && D > 0, "non-positive size");
  hipLaunchKernelGGL(wgrad_fold2_kernel, dim3(cdiv(H, 16) * cdiv(D + 1, 16)), dim3(1024), 0, (hipStream_t)stream, w0, w1, p0, p1, c0,
                     c1, dw, db, G, H, D, accumulate ? 1 : 0);
  G2V_CHECK_LAUNCH();
  return G2V_OK;
}
// g2v_linear_bwd_weight_fold2 and g2v_linear_bwd_weight_chain2 (overwrite form) in one launch: bitwise their results.
extern "C" int g2v_linear_bwd_weight_fold_chain2(const float* w0, const float* w1, const float* p0, const float* p1, const float* c0,
                                                 const float* c1, const float* w_in, const float* b_in, float* dw_in, float* db_in,
                                                 float* dw0, float* dw1, int G, int H, int D, g2v_stream_t stream) {
  G2V_REQUIRE(w0 && w1 && p0 && p1 && c0 && c1 && w_in && b_in && dw_in && db_in && dw0 && dw1, "null pointer");
  G2V_REQUIRE(G > 0 && H > 0 && D > 0, "non-positive size");
  const size_t lc = chain2_lds_bytes(H, D), lf = sizeof(float) * 15 * 256;
  if (lc > 48 * 1024 || D > CHAIN2_MAXD) {      // the chain's staging does not fit: the two launches
    const int rc = g2v_linear_bwd_weight_fold2(w0, w1, p0, p1, c0, c1, dw_in, db_in, G, H, D, 0, stream);
    if (rc != G2V_OK) return rc;
    return g2v_linear_bwd_weight_chain2(p0, p1, c0, c1, w_in, b_in, dw0, dw1, G, H, D, 0, stream);
  }
  const int nfold = cdiv(H, 16) * cdiv(D + 1, 16), ncx = cdiv(G, CHAIN2_ROWS);
  hipLaunchKernelGGL(wgrad_fold_chain2_kernel, dim3(nfold + 2 * ncx), dim3(1024), lc > lf ? lc : lf, (hipStream_t)stream, w0, w1, p0, p1,
                     c0, c1, w_in, b_in, dw_in, db_in, dw0, dw1, G, H, D, nfold, ncx);
  G2V_CHECK_LAUNCH();
  return G2V_OK;
}

// measured at the reference's own VQ-VAE.yml shape (B = 128, T = 20: 2432 / 2560 rows, 600 x 200): the LDS-tiled kernel + slab
// pass 20.6 + 7 us per product, so the one-launch form keeps the rows BELOW 4096; from 4096 rows the output-blocked wave
// kernel takes over (the soft quantiser's products at N = 4096: 512 x 128 22 us against 32, 128 x 128 16 against 31)
static constexpr int g_smallm_wgrad_rows = 4095;
template <int TN, int TK, int NW>
static void smw_lds_launch(const SmallWgradBatch& sb, int64_t lddy, int64_t ldx, int M, int K, int N, int accumulate, int nprob,
                           int xpp, hipStream_t st) {
  const int groups = cdiv(cdiv(N, 16), TN) * cdiv(cdiv(K, 16), TK);
  constexpr int STG = 32 * (TN * 16 + TK * 16 + 4), RED = TN * TK * 256 + TN * 16;
  constexpr size_t lds = sizeof(float) * (size_t)(NW * STG > (NW - 1) * RED ? NW * STG : (NW - 1) * RED);
  static bool attr = false;
  if (!attr) attr = hipFuncSetAttribute((const void*)gemm_tn_smallm_lds_kernel<TN, TK, NW>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                        (int)lds) == hipSuccess;
  hipLaunchKernelGGL((gemm_tn_smallm_lds_kernel<TN, TK, NW>), xpp ? dim3(8 * cdiv(groups, xpp), 1) : dim3(groups, nprob), dim3(64 * NW),
                     lds, st, sb, lddy, ldx, M, K, N, accumulate, xpp);
}


// nprob problems of one shape: {dy, x, dw, db}[p].  The wave-autonomous path launches them together (grid.y = problem);
// the LDS-tiled fallback runs them one after the other.  `slab_stride` floats of workspace per problem.
struct WgradItem {
  const float* dy; const float* x; float* dw; float* db;
};
// (dy_a + dy_b)^T x is served by the wave-autonomous kernel's two-addend instantiation: 64 x 135-shaped dW, whole 16-row
// groups, more rows than the small-M kernel takes
static bool wgrad_sum2_ok(int M, int K, int N) {
  int rpw = 0;
  return M > g_smallm_wgrad_rows && (M & 15) == 0 && cdiv(N, 16) == 4 && cdiv(K, 16) == 9 &&
         tn_wave_grid(M, K, N, false, &rpw, 1, 2) > 0;
}
static int wgrad_impl(const WgradItem* it, int nprob, int64_t lddy, int64_t ldx, int rows_inner, int64_t stride_outer,
                      int64_t stride_inner, const uint8_t* x_keep, float x_scale, int M, int K, int N, int flags,
                      float* workspace, g2v_stream_t stream, const float* dy2 = nullptr, g2v_wgrad_pending* pend = nullptr) {
  if (pend) pend->nprob = 0;            // (paths without a slab reduction, or that need dw finished at once, leave it empty)
  if (dy2 && !wgrad_sum2_ok(M, K, N)) {
    set_error("g2v_linear_bwd_weight_sum2: shape not served (see g2v_linear_bwd_weight_sum2_ok)");
    return G2V_ERR_UNSUPPORTED;
  }
  const int accumulate = flags & G2V_WGRAD_ACCUMULATE;
  const bool bf3 = (flags & G2V_WGRAD_BF16X3) != 0;
  if ((M & 15) && M >= 4096 + 16 && !x_keep) {
    // ragged row count (T B not a multiple of 16: B = 4100): the wave-autonomous kernels take the whole 16-row groups, the
    // M % 16 leftover rows are added by the small-M kernel (accumulate) -- instead of the whole product falling back to the
    // LDS-tiled kernel (95 vs 35-45 us per product at the BASELINE shape)
    const int Mt = M & 15, Mm = M - Mt;
    const int rc = wgrad_impl(it, nprob, lddy, ldx, rows_inner, stride_outer, stride_inner, nullptr, 1.0f, Mm, K, N, flags,
                              workspace, stream);
    if (rc != G2V_OK) return rc;
    const bool mapped = rows_inner > 0;
    SmallWgradBatch sb;
    for (int p = 0; p < G2V_TN_BATCH; ++p) {
      const int pp = p < nprob ? p : 0;
      sb.dy[p] = it[pp].dy + (int64_t)Mm * lddy; sb.x[p] = mapped ? it[pp].x : it[pp].x + (int64_t)Mm * ldx;
      sb.dw[p] = it[pp].dw; sb.db[p] = it[pp].db;
    }
    hipLaunchKernelGGL(gemm_tn_smallm_kernel<false>, dim3(cdiv(N, 16) * cdiv(K, 16), nprob), dim3(256), 0, (hipStream_t)stream, sb,
                       lddy, ldx, (const uint8_t*)nullptr, 1.0f, Mt, K, N, 1, RowMap{ldx, rows_inner, stride_outer, stride_inner},
                       mapped ? Mm : 0);
    G2V_CHECK_LAUNCH();
    return G2V_OK;
  }
  if (M <= g_smallm_wgrad_rows && (rows_inner == 0 || !x_keep) && !bf3) {
    const bool mapped = rows_inner > 0;      // (a row-mapped x: the one-tile kernels)
    const RowMap xm1 = mapped ? RowMap{ldx, rows_inner, stride_outer, stride_inner} : RowMap{0, 0, 0, 0};
    SmallWgradBatch sb;
    for (int p = 0; p < G2V_TN_BATCH; ++p) {
      const int pp = p < nprob ? p : 0;
      sb.dy[p] = it[pp].dy; sb.x[p] = it[pp].x; sb.dw[p] = it[pp].dw; sb.db[p] = it[pp].db;
    }
    // XCD-aware placement (smw_decode_grid) when the problems divide the 8 XCDs and there is more than one
    const int xpp = nprob > 1 && (8 % nprob) == 0 ? 8 / nprob : 0;
    const int groups1 = cdiv(N, 16) * cdiv(K, 16), groups2 = cdiv(cdiv(N, 16), 2) * cdiv(cdiv(K, 16), 2);
    const dim3 grid = xpp ? dim3(8 * cdiv(groups1, xpp), 1) : dim3(groups1, nprob);
    // enough tiles that 2 x 2 of them per workgroup still cover the chip, enough rows that the streaming dominates (four
    // 600 x 200 products at 2432 rows: 72 us against 94; 4 x 2, 3 x 3 and 4 x 4 tiles measured within +-5 % of 2 x 2)
    bool vec_ok = !x_keep && !mapped && M >= 512 && ((N | K | lddy | ldx) & 3) == 0;
    for (int p2 = 0; p2 < nprob && vec_ok; ++p2)
      vec_ok = ((reinterpret_cast<uintptr_t>(it[p2].dy) | reinterpret_cast<uintptr_t>(it[p2].x)) & 15) == 0;
    if (vec_ok && (int64_t)cdiv(cdiv(N, 16), 2) * cdiv(K, 16) * nprob >= 256) {      // (2 x 1 tiles per workgroup)
      // float4-aligned operands: the LDS-staged form (four 600 x 200 products at 2560 rows: 53 us against 70, bitwise the same dW)
      // (NW = 8 waves per workgroup, 70 KB of LDS: 60 us against 54 alone, and starved beside the GRU backward cluster -- 110 us;
      //  the counters of the 4-wave form: MFMA pipe 32 % busy, waves waiting on memory 52 % of their time, no LDS bank conflicts:
      //  profiles/r05_ai_pmc_smallm_wgrad_4waves.json, r05_aj_lds_wgrad_8waves_ab.log)
      // 2 x 1 tiles: two / four 600 x 200 products at 2560 rows 27.8 / 50.9 us (2 x 2: 38.6 / 56.6, 1 x 1: 34.8 / 67.1; r05_al log)
      smw_lds_launch<2, 1, 4>(sb, lddy, ldx, M, K, N, accumulate, nprob, xpp, (hipStream_t)stream);
    } else if (!x_keep && !mapped && M >= 512 && (int64_t)cdiv(cdiv(N, 16), 2) * cdiv(cdiv(K, 16), 2) * nprob >= 256)
      hipLaunchKernelGGL((gemm_tn_smallm_rt_kernel<2, 2, 4>), xpp ? dim3(8 * cdiv(groups2, xpp), 1) : dim3(groups2, nprob), dim3(256), 0,
                         (hipStream_t)stream, sb, lddy, ldx, M, K, N, accumulate, xpp);
    else if (x_keep)
      hipLaunchKernelGGL(gemm_tn_smallm_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, sb, lddy, ldx, x_keep, x_scale,
                         M, K, N, accumulate, RowMap{0, 0, 0, 0}, 0, xpp);
    else if (M >= 512 && groups1 * nprob <= 256)      // few tiles, many rows: 16 waves split the rows (2560 rows: 30 -> 19 us)
      hipLaunchKernelGGL((gemm_tn_smallm_kernel<false, 16, 4>), grid, dim3(1024), 0, (hipStream_t)stream, sb, lddy, ldx, x_keep,
                         x_scale, M, K, N, accumulate, xm1, 0, xpp);
    else
      hipLaunchKernelGGL(gemm_tn_smallm_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, sb, lddy, ldx, x_keep, x_scale,
                         M, K, N, accumulate, xm1, 0, xpp);
    G2V_CHECK_LAUNCH();
    return G2V_OK;
  }
  int splits = tn_splits(M, K, N);
  int rows_per_split = cdiv(M, splits);
  rows_per_split = round_up(rows_per_split, TM);
  RowMap xm{ldx, rows_inner, stride_outer, stride_inner};
  int rpw = 0;
  const int nr = bf3 ? 4 : 2;                  // row ranges (waves per tile group) per workgroup, see tn_wave_body
  int wg = tn_wave_grid(M, K, N, x_keep != nullptr, &rpw, nprob, nr);
  int gen_cfg = 0, gen_nob = 0;
  if (wg == 0 && !bf3 && (N > 64 || K > 64)) {  // shapes beyond one workgroup's accumulators: output-blocked wave kernel
    int gs = 0;
    gen_nob = tn_gen_grid(M, K, N, x_keep != nullptr, nprob, &gen_cfg, &rpw, &gs);
    if (gen_nob > 0) wg = gs;
  }
  if (wg > 0) splits = wg;
  const size_t slab_stride = (size_t)splits * ((size_t)N * K + N);
  const int64_t n = (int64_t)N * K;
  bool any_db = false;
  for (int p = 0; p < nprob; ++p) any_db = any_db || it[p].db;
  auto slab_of = [&](int p) { return workspace + (size_t)p * slab_stride; };
  auto slab_db_of = [&](int p) { return it[p].db ? slab_of(p) + (size_t)splits * N * K : (float*)nullptr; };
  if (wg > 0) {
    TnBatch bt;
    bool vec2 = (N % 32 == 0) && (K % 32 == 0) && rows_inner == 0 && (lddy % 2 == 0) && (ldx % 2 == 0);
    bool all_db = true;
    for (int p = 0; p < G2V_TN_BATCH; ++p) {
      const int pp = p < nprob ? p : 0;
      bt.dy[p] = it[pp].dy; bt.x[p] = it[pp].x; bt.slab[p] = slab_of(pp); bt.slab_db[p] = slab_db_of(pp);
      bt.dy2[p] = dy2;
      vec2 = vec2 && (reinterpret_cast<uintptr_t>(it[pp].dy) % 8 == 0) && (reinterpret_cast<uintptr_t>(it[pp].x) % 8 == 0);
      all_db = all_db && (it[pp].db != nullptr);
    }
    (void)all_db;
    const int tn = cdiv(N, 16), tk = cdiv(K, 16);
    if (gen_nob > 0) {
#define G2V_TNG(TN_, TK_, MP)                                                                                             \
  do {                                                                                                                   \
    const size_t lds = ((size_t)2 * TN_ * TK_ * 256 + 4 * TN_ * 16) * sizeof(float);                                      \
    (void)hipFuncSetAttribute((const void*)gemm_tn_wave_gen_kernel<TN_, TK_, 2, 1, MP>,                                  \
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                                     \
    hipLaunchKernelGGL((gemm_tn_wave_gen_kernel<TN_, TK_, 2, 1, MP>), dim3(wg, nprob, gen_nob), dim3(256), lds,          \
                       (hipStream_t)stream, bt, lddy, xm, M, K, N, rpw);                                                 \
  } while (0)
      if (gen_cfg == 0) { if (rows_inner > 0) G2V_TNG(6, 4, true); else G2V_TNG(6, 4, false); }
      else { if (rows_inner > 0) G2V_TNG(4, 7, true); else G2V_TNG(4, 7, false); }
#undef G2V_TNG
    } else {
#define G2V_TNW2(TN_, TK_, SN, SK, VW, BF, NR, MP)                                                                       \
  do {                                                                                                                   \
    const size_t lds = ((size_t)NR * TN_ * TK_ * 256 + 2 * NR * TN_ * 16) * sizeof(float);                                \
    (void)hipFuncSetAttribute((const void*)gemm_tn_wave_kernel<TN_, TK_, SN, SK, MP, VW, BF, NR>,                        \
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                                     \
    hipLaunchKernelGGL((gemm_tn_wave_kernel<TN_, TK_, SN, SK, MP, VW, BF, NR>), dim3(wg, nprob), dim3(128 * NR), lds,    \
                       (hipStream_t)stream, bt, lddy, xm, M, K, N, rpw);                                                 \
  } while (0)
#define G2V_TNW(TN_, TK_, SN, SK, VW)                                                                                    \
  do {                                                                                                                   \
    if (bf3) { if (rows_inner > 0) G2V_TNW2(TN_, TK_, SN, SK, VW, true, 4, true); else G2V_TNW2(TN_, TK_, SN, SK, VW, true, 4, false); } \
    else { if (rows_inner > 0) G2V_TNW2(TN_, TK_, SN, SK, VW, false, 2, true); else G2V_TNW2(TN_, TK_, SN, SK, VW, false, 2, false); }  \
  } while (0)
    // 8-byte vector operand loads need 8-byte-aligned rows on both sides and whole tiles (checked above)
    if (dy2) {
      const size_t lds = ((size_t)2 * 2 * 9 * 256 + 2 * 2 * 2 * 16) * sizeof(float);
      if (rows_inner > 0) {
        (void)hipFuncSetAttribute((const void*)gemm_tn_wave_dual_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(gemm_tn_wave_dual_kernel<true>, dim3(wg, 1), dim3(256), lds, (hipStream_t)stream, bt, lddy, xm, M, K, N, rpw);
      } else {
        (void)hipFuncSetAttribute((const void*)gemm_tn_wave_dual_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(gemm_tn_wave_dual_kernel<false>, dim3(wg, 1), dim3(256), lds, (hipStream_t)stream, bt, lddy, xm, M, K, N, rpw);
      }
    }
    else if (tn == 12 && tk == 4) { if (vec2) G2V_TNW(6, 4, 2, 1, 2); else G2V_TNW(6, 4, 2, 1, 1); }
    else if (tn == 4 && tk == 9) G2V_TNW(2, 9, 2, 1, 1);
    else if (tn == 9 && tk == 4) G2V_TNW(9, 2, 1, 2, 1);
    else { if (vec2) G2V_TNW(2, 4, 2, 1, 2); else G2V_TNW(2, 4, 2, 1, 1); }
#undef G2V_TNW
#undef G2V_TNW2
    }
    G2V_CHECK_LAUNCH();
  } else {
    const int ntw = tn_ntw(N);
    dim3 grid(cdiv(N, 64 * ntw), cdiv(K, 64), splits);
    for (int p = 0; p < nprob; ++p) {
      if (ntw == 1)
        hipLaunchKernelGGL(gemm_tn_kernel<1>, grid, dim3(256), 0, (hipStream_t)stream, it[p].dy, lddy, it[p].x, xm, x_keep,
                           x_scale, slab_of(p), slab_db_of(p), M, K, N, rows_per_split);
      else if (ntw == 2)
        hipLaunchKernelGGL(gemm_tn_kernel<2>, grid, dim3(256), 0, (hipStream_t)stream, it[p].dy, lddy, it[p].x, xm, x_keep,
                           x_scale, slab_of(p), slab_db_of(p), M, K, N, rows_per_split);
      else
        hipLaunchKernelGGL(gemm_tn_kernel<3>, grid, dim3(256), 0, (hipStream_t)stream, it[p].dy, lddy, it[p].x, xm, x_keep,
                           x_scale, slab_of(p), slab_db_of(p), M, K, N, rows_per_split);
    }
    G2V_CHECK_LAUNCH();
  }
  // one launch reduces the weight slabs and (where requested) the bias slabs of every problem
  SlabBatch sb;
  for (int p = 0; p < G2V_TN_BATCH; ++p) {
    const int pp = p < nprob ? p : 0;
    sb.slab_a[p] = slab_of(pp); sb.out_a[p] = it[pp].dw; sb.slab_b[p] = slab_db_of(pp); sb.out_b[p] = it[pp].db;
  }
  if (pend) {       // G2V_WGRAD_DEFER_REDUCE: the caller reduces later (g2v_linear_bwd_weight_reduce); the slabs stay in `workspace`
    for (int p = 0; p < nprob; ++p) {
      pend->slab_w[p] = slab_of(p); pend->out_w[p] = it[p].dw; pend->slab_b[p] = slab_db_of(p); pend->out_b[p] = it[p].db;
    }
    pend->n = n; pend->nb = N; pend->nsplit = splits; pend->nprob = nprob; pend->accumulate = accumulate;
    return G2V_OK;
  }
  // problems without a bias gradient: their bias blocks find slab_b == nullptr and return
  hipLaunchKernelGGL(slab_reduce2_kernel, dim3(cdiv(n, 32) + (any_db ? cdiv(N, 32) : 0), nprob), dim3(256), 0, (hipStream_t)stream,
                     sb, n, (int64_t)N, splits, accumulate, cdiv(n, 32));
  G2V_CHECK_LAUNCH();
  return G2V_OK;
}

extern "C" int g2v_linear_bwd_weight(const float* dy, int64_t lddy, const float* x, int64_t ldx, int rows_inner,
                                     int64_t stride_outer, int64_t stride_inner, const uint8_t* x_keep, float x_scale,
                                     float* dw, float* db, int M, int K, int N, int accumulate, void* workspace,
                                     size_t workspace_bytes, g2v_stream_t stream) {
  G2V_REQUIRE(dy && x && dw && workspace, "null pointer");
  G2V_REQUIRE(M > 0 && K > 0 && N > 0, "non-positive size");
  if (workspace_bytes < g2v_linear_bwd_weight_workspace(M, K, N)) {
    set_error("g2v_linear_bwd_weight: workspace too small");
    return G2V_ERR_WORKSPACE;
  }
  const WgradItem item{dy, x, dw, db};
  return wgrad_impl(&item, 1, lddy, ldx, rows_inner, stride_outer, stride_inner, x_keep, x_scale, M, K, N, accumulate,
                    (float*)workspace, stream);
}

extern "C" int g2v_linear_bwd_weight_sum2_ok(int M, int K, int N) { return wgrad_sum2_ok(M, K, N) ? 1 : 0; }

extern "C" int g2v_linear_bwd_weight_sum2(const float* dy_a, const float* dy_b, int64_t lddy, const float* x, int64_t ldx,
                                          int rows_inner, int64_t stride_outer, int64_t stride_inner, float* dw, float* db,
                                          int M, int K, int N, int accumulate, void* workspace, size_t workspace_bytes,
                                          g2v_stream_t stream) {
  G2V_REQUIRE(dy_a && dy_b && x && dw && workspace, "null pointer");
  G2V_REQUIRE(M > 0 && K > 0 && N > 0, "non-positive size");
  if (workspace_bytes < g2v_linear_bwd_weight_workspace(M, K, N)) {
    set_error("g2v_linear_bwd_weight_sum2: workspace too small");
    return G2V_ERR_WORKSPACE;
  }
  const WgradItem item{dy_a, x, dw, db};
  return wgrad_impl(&item, 1, lddy, ldx, rows_inner, stride_outer, stride_inner, nullptr, 1.0f, M, K, N,
                    accumulate ? G2V_WGRAD_ACCUMULATE : 0, (float*)workspace, stream, dy_b);
}

extern "C" int g2v_linear_bwd_weight_batch(const g2v_wgrad_item* items, int nprob, int64_t lddy, int64_t ldx, int M, int K,
                                           int N, int flags, void* workspace, size_t workspace_bytes, g2v_stream_t stream) {
  G2V_REQUIRE(items && workspace, "null pointer");
  G2V_REQUIRE(nprob >= 1 && nprob <= G2V_TN_BATCH, "1..4 problems per call");
  G2V_REQUIRE(M > 0 && K > 0 && N > 0, "non-positive size");
  WgradItem it[G2V_TN_BATCH];
  for (int p = 0; p < nprob; ++p) {
    G2V_REQUIRE(items[p].dy && items[p].x && items[p].dw, "null pointer");
    it[p] = WgradItem{items[p].dy, items[p].x, items[p].dw, items[p].db};
  }
  if (workspace_bytes < (size_t)nprob * g2v_linear_bwd_weight_workspace(M, K, N)) {
    set_error("g2v_linear_bwd_weight_batch: workspace too small");
    return G2V_ERR_WORKSPACE;
  }
  return wgrad_impl(it, nprob, lddy, ldx, 0, 0, 0, nullptr, 1.0f, M, K, N, flags, (float*)workspace, stream);
}
extern "C" int g2v_linear_bwd_weight_batch_mapped(const g2v_wgrad_item* items, int nprob, int64_t lddy, int64_t ldx, int rows_inner,
                                                  int64_t stride_outer, int64_t stride_inner, int M, int K, int N, int flags,
                                                  void* workspace, size_t workspace_bytes, g2v_stream_t stream) {
  G2V_REQUIRE(items && workspace, "null pointer");
  G2V_REQUIRE(nprob >= 1 && nprob <= G2V_TN_BATCH, "1..4 problems per call");
  G2V_REQUIRE(M > 0 && K > 0 && N > 0 && rows_inner >= 0, "bad size");
  WgradItem it[G2V_TN_BATCH];
  for (int p = 0; p < nprob; ++p) {
    G2V_REQUIRE(items[p].dy && items[p].x && items[p].dw, "null pointer");
    it[p] = WgradItem{items[p].dy, items[p].x, items[p].dw, items[p].db};
  }
  if (workspace_bytes < (size_t)nprob * g2v_linear_bwd_weight_workspace(M, K, N)) {
    set_error("g2v_linear_bwd_weight_batch_mapped: workspace too small");
    return G2V_ERR_WORKSPACE;
  }
  return wgrad_impl(it, nprob, lddy, ldx, rows_inner, stride_outer, stride_inner, nullptr, 1.0f, M, K, N, flags, (float*)workspace,
                    stream);
}

// One call for every weight-gradient form (a single product: nprob = 1; the batch; the row-mapped batch; (dy_a + dy_b)^T x: dy_b)
// whose slab reduction is left to the caller: `pending` describes it (empty when the shape's path has none -- small row counts --
// or needs dw at once -- ragged row counts, reduced here), g2v_linear_bwd_weight_reduce runs up to G2V_WGRAD_PENDING_MAX of them
// in one launch.  The slabs live in `workspace` until then: one workspace per pending call.
extern "C" int g2v_linear_bwd_weight_deferred(const g2v_wgrad_item* items, int nprob, int64_t lddy, int64_t ldx, int rows_inner,
                                              int64_t stride_outer, int64_t stride_inner, const float* dy_b, int M, int K, int N,
                                              int flags, void* workspace, size_t workspace_bytes, g2v_wgrad_pending* pending,
                                              g2v_stream_t stream) {
  G2V_REQUIRE(items && workspace && pending, "null pointer");
  G2V_REQUIRE(nprob >= 1 && nprob <= G2V_TN_BATCH && (!dy_b || nprob == 1), "1..4 problems per call (1 with dy_b)");
  G2V_REQUIRE(M > 0 && K > 0 && N > 0 && rows_inner >= 0, "bad size");
  WgradItem it[G2V_TN_BATCH];
  for (int p = 0; p < nprob; ++p) {
    G2V_REQUIRE(items[p].dy && items[p].x && items[p].dw, "null pointer");
    it[p] = WgradItem{items[p].dy, items[p].x, items[p].dw, items[p].db};
  }
  if (workspace_bytes < (size_t)nprob * g2v_linear_bwd_weight_workspace(M, K, N)) {
    set_error("g2v_linear_bwd_weight_deferred: workspace too small");
    return G2V_ERR_WORKSPACE;
  }
  return wgrad_impl(it, nprob, lddy, ldx, rows_inner, stride_outer, stride_inner, nullptr, 1.0f, M, K, N, flags, (float*)workspace,
                    stream, dy_b, pending);
}

extern "C" int g2v_linear_bwd_weight_reduce(const g2v_wgrad_pending* pending, int count, g2v_stream_t stream) {
  G2V_REQUIRE(pending && count >= 1 && count <= G2V_WGRAD_PENDING_MAX, "1..G2V_WGRAD_PENDING_MAX pending reductions");
  SlabMulti sm;
  sm.count = 0;
  sm.first[0] = 0;
  auto flush = [&]() {
    if (sm.count == 0) return;
    for (int f = sm.count; f < SLAB_MULTI; ++f) {
      sm.slab[f] = sm.slab[0]; sm.out[f] = sm.out[0]; sm.n[f] = 0; sm.nsplit[f] = 0; sm.accumulate[f] = 0; sm.first[f + 1] = sm.first[sm.count];
    }
    hipLaunchKernelGGL(slab_reduce_multi_kernel, dim3(sm.first[sm.count]), dim3(256), 0, (hipStream_t)stream, sm);
    sm.count = 0;
  };
  auto add = [&](const float* slab, float* out, int64_t n, int nsplit, int accumulate) {
    if (sm.count == SLAB_MULTI) flush();
    const int f = sm.count++;
    sm.slab[f] = slab; sm.out[f] = out; sm.n[f] = n; sm.nsplit[f] = nsplit; sm.accumulate[f] = accumulate;
    sm.first[f + 1] = sm.first[f] + (int)cdiv(n, 32);
  };
  for (int c = 0; c < count; ++c) {
    const g2v_wgrad_pending& pd = pending[c];
    G2V_REQUIRE(pd.nprob >= 0 && pd.nprob <= G2V_TN_BATCH, "corrupt pending record");
    for (int p = 0; p < pd.nprob; ++p) {
      G2V_REQUIRE(pd.slab_w[p] && pd.out_w[p] && pd.n > 0 && pd.nsplit > 0, "corrupt pending record");
      add(pd.slab_w[p], pd.out_w[p], pd.n, pd.nsplit, pd.accumulate);
      if (pd.slab_b[p] && pd.out_b[p]) add(pd.slab_b[p], pd.out_b[p], pd.nb, pd.nsplit, pd.accumulate);
    }
  }
  flush();
  G2V_CHECK_LAUNCH();
  return G2V_OK;
}
