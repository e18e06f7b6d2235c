// vq_soft.hip -- the as-shipped soft quantiser (VQ_Payam_GSSoft, reference model/Autoencoder_VQVAE_model.py:1304-1438) as TWO
// kernels: everything between the encoder state and the decoder's initial state in one launch, and its backward in one.
//
// The quantiser is row-local: row n of the encoder state x (N rows of E = H L floats) goes through mean_layer (E x E), its
// logvar_layer (K x E) and codebook (K x E) products, a K-wide normalisation, q = probs W and the two latent losses without ever
// meeting another row (only the loss mean and the perplexity's column means do, as per-workgroup partial sums).  As separate
// launches (three dense products, the element-wise kernel, q = probs W, mse, scale, ste: 9 on the chain between the encoder and
// the rollout; 7 between the rollout's backward and the encoder's BPTT) each one pays a launch and an HBM round trip of an (N, K)
// array: 115 us forward and 235 us backward at N = 4096, K = 512 (profiles/, round 4) for 4 GFLOP of products.  Here a workgroup
// (four waves) owns 16 rows; the row tile and its K-wide intermediates stay in LDS, the weights stream from L2 as MFMA operands
// -- rows of a row-major matrix as float4 A fragments where the contraction runs along the row, dword "column" loads where it
// runs across rows (the weight-gradient kernels' access) -- and only what the backward / the weight gradients read is written.
//
// Arithmetic: each element is formed by the same expressions as the separate kernels (vq.hip: vq_soft_fwd_kernel /
// vq_soft_bwd_kernel / rowscale_combine_kernel / ste_kernel / vq_bwd_kernel, misc.hip: mse_kernel); the K-long and E-long sums meet
// in a different (fixed) order, so results agree to fp32 rounding, and are bitwise reproducible run to run.
//
// Served: E == 128 (H = 64, two layers), K % 128 == 0, K <= 1024; anything else stays on the separate kernels
// (g2v_vq_soft_fused_ok).
#include "common.hpp"

namespace g2v {
namespace {

constexpr int SOFT_E = 128;

struct SoftFwdArgs {
  const float* x;                                  // (N,E) encoder state rows
  const float* w_mean; const float* b_mean;        // (E,E), (E)
  const float* w_lv; const float* b_lv;            // (K,E), (K)
  const float* cb; const float* wsq;               // (K,E), (K) = |W_k|^2
  float* flat; float* logvar; float* dist; float* probs; float* q; float* dq; float* quant;
  float* mse_partial;                              // [nblk] sum (q - x)^2 of the workgroup's rows
  float* colsum;                                   // [nblk][K] column sums of probs over the workgroup's rows, or NULL
  float gcoef;                                     // 2 g / (N E): dq = gcoef (q - x)
  int N, K;
};

__device__ __forceinline__ float quad_sum(float v) {      // over the four lanes (i, q = 0..3) of a row i
  v += __shfl_xor(v, 16);
  v += __shfl_xor(v, 32);
  return v;
}
__device__ __forceinline__ void mma4(f32x4& acc, const float4& w, const float4& x) {
  acc = mfma16(w.x, x.x, acc);
  acc = mfma16(w.y, x.y, acc);
  acc = mfma16(w.z, x.z, acc);
  acc = mfma16(w.w, x.w, acc);
}

// LDS (floats): Xs[16][E+4] | Fs[16][E+4] | Ps[16][K+4] | red[64] | red2[4]
template <int E>
__global__ __launch_bounds__(256) void vq_soft_fused_fwd_kernel(SoftFwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int LD = E + 4, KS = E / 16, NTW = E / 64;      // k-steps over E; output tiles over E per wave
  const int K = a.K, ldk = K + 4, N = a.N;
  float* Xs = smem;
  float* Fs = Xs + 16 * LD;
  float* Ps = Fs + 16 * LD;
  float* red = Ps + 16 * ldk;
  float* red2 = red + 64;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i = lane & 15, q = lane >> 4;
  const int n0 = blockIdx.x * 16, nrows = min(16, N - n0);
  const bool rowok = i < nrows;
  const int64_t n = n0 + (rowok ? i : nrows - 1);

  // ---- stage 0: the x tile; the mean_layer fragments of this wave's two output tiles travel meanwhile ------------------------
  float4 wm[NTW][KS];
#pragma unroll
  for (int j = 0; j < NTW; ++j)
#pragma unroll
    for (int s = 0; s < KS; ++s)
      wm[j][s] = *reinterpret_cast<const float4*>(a.w_mean + (int64_t)(16 * (wave + 4 * j) + i) * E + 16 * s + 4 * q);
  for (int e4 = tid; e4 < 16 * E / 4; e4 += 256) {
    const int r = e4 / (E / 4), c = (e4 - r * (E / 4)) * 4;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (r < nrows) v = *reinterpret_cast<const float4*>(a.x + (int64_t)(n0 + r) * E + c);
    *reinterpret_cast<float4*>(Xs + r * LD + c) = v;
  }
  // first logvar_layer / codebook tile of stage 2 (rows k = 16 t + i of both matrices: the contraction runs along the row)
  float4 wlA[KS], wcA[KS], wlB[KS], wcB[KS];
  float4 blA, wqA, blB, wqB;
  auto load_tile = [&](int t, float4 (&wl)[KS], float4 (&wc)[KS], float4& bl, float4& wq) {
    const int64_t row = (int64_t)(16 * t + i) * E + 4 * q;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      wl[s] = *reinterpret_cast<const float4*>(a.w_lv + row + 16 * s);
      wc[s] = *reinterpret_cast<const float4*>(a.cb + row + 16 * s);
    }
    bl = *reinterpret_cast<const float4*>(a.b_lv + 16 * t + 4 * q);
    wq = *reinterpret_cast<const float4*>(a.wsq + 16 * t + 4 * q);
  };
  load_tile(wave, wlA, wcA, blA, wqA);
  lds_barrier();
  // ---- stage 1: flat = mean_layer(x) --------------------------------------------------------------------------------------------
  {
    f32x4 acc[NTW];
#pragma unroll
    for (int j = 0; j < NTW; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const float4 xb = *reinterpret_cast<const float4*>(Xs + i * LD + 16 * s + 4 * q);
#pragma unroll
      for (int j = 0; j < NTW; ++j) mma4(acc[j], wm[j][s], xb);
    }
#pragma unroll
    for (int j = 0; j < NTW; ++j) {
      const int f0 = 16 * (wave + 4 * j) + 4 * q;
      const float4 b4 = *reinterpret_cast<const float4*>(a.b_mean + f0);
      const float4 v = make_float4(acc[j][0] + b4.x, acc[j][1] + b4.y, acc[j][2] + b4.z, acc[j][3] + b4.w);
      *reinterpret_cast<float4*>(Fs + i * LD + f0) = v;
      if (rowok) *reinterpret_cast<float4*>(a.flat + n * E + f0) = v;
    }
  }
  lds_barrier();
  // ---- stages 2 + 3: logvar and distances of the row against every code -> unnormalised probabilities in LDS ------------------
  float fs = 0.f;
  {
#pragma unroll
    for (int c = 0; c < E / 16; ++c) {
      const float4 v = *reinterpret_cast<const float4*>(Fs + i * LD + (E / 4) * q + 4 * c);
      fs += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
    }
    fs = quad_sum(fs);
  }
  const int ntile = K / 64;                 // tiles per wave (even: K % 128 == 0)
  float rsum = 0.f;
  auto tile = [&](int t, const float4 (&wl)[KS], const float4 (&wc)[KS], const float4& bl4, const float4& wq4) {
    f32x4 al = {0.f, 0.f, 0.f, 0.f}, ad = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const float4 fb = *reinterpret_cast<const float4*>(Fs + i * LD + 16 * s + 4 * q);
      mma4(al, wl[s], fb);
      mma4(ad, wc[s], fb);
    }
    const float bl[4] = {bl4.x, bl4.y, bl4.z, bl4.w}, wq[4] = {wq4.x, wq4.y, wq4.z, wq4.w};
    float lv[4], d[4], pr[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      lv[r] = al[r] + bl[r];
      d[r] = (fs + wq[r]) - 2.0f * ad[r];
      const float ex = expf(lv[r]);
      const float smooth = 1.0f / (ex * ex);
      pr[r] = expf(-((d[r] / 400.0f) * (0.5f * smooth))) / sqrtf(smooth);
      rsum += pr[r];
    }
    const int k0 = 16 * t + 4 * q;
    *reinterpret_cast<float4*>(Ps + i * ldk + k0) = make_float4(pr[0], pr[1], pr[2], pr[3]);
    if (rowok) {
      *reinterpret_cast<float4*>(a.logvar + n * K + k0) = make_float4(lv[0], lv[1], lv[2], lv[3]);
      *reinterpret_cast<float4*>(a.dist + n * K + k0) = make_float4(d[0], d[1], d[2], d[3]);
    }
  };
  for (int j = 0; j < ntile; j += 2) {
    load_tile(wave + 4 * (j + 1), wlB, wcB, blB, wqB);
    tile(wave + 4 * j, wlA, wcA, blA, wqA);
    if (j + 2 < ntile) load_tile(wave + 4 * (j + 2), wlA, wcA, blA, wqA);
    tile(wave + 4 * (j + 1), wlB, wcB, blB, wqB);
  }
  // ---- stage 4 operands requested now: the first ring of codebook "columns" (contraction across rows k) ----------------------
  // k slot of MFMA step c in lane group q: k = kb + 4 q + c, so that the LDS operand is one float4 per 16 k
  constexpr int RING = 8;                   // blocks of 16 k in flight per wave
  float wr[RING][NTW][4];
  auto ring_load = [&](int b, int kb) {
#pragma unroll
    for (int j = 0; j < NTW; ++j)
#pragma unroll
      for (int c = 0; c < 4; ++c) wr[b][j][c] = a.cb[(int64_t)(kb + 4 * q + c) * E + 16 * (wave + 4 * j) + i];
  };
#pragma unroll
  for (int b = 0; b < RING; ++b) ring_load(b, 16 * b);
  // row sums over the four waves' tiles, in wave order
  rsum = quad_sum(rsum);
  if (q == 0) red[wave * 16 + i] = rsum;
  lds_barrier();
  const float tot = (red[i] + red[16 + i]) + (red[32 + i] + red[48 + i]);
  for (int j = 0; j < ntile; ++j) {
    const int k0 = 16 * (wave + 4 * j) + 4 * q;
    float4 p = *reinterpret_cast<const float4*>(Ps + i * ldk + k0);
    p.x /= tot; p.y /= tot; p.z /= tot; p.w /= tot;
    *reinterpret_cast<float4*>(Ps + i * ldk + k0) = p;
    if (rowok) *reinterpret_cast<float4*>(a.probs + n * K + k0) = p;
    if (a.colsum) {
      const float c0 = reduce16(rowok ? p.x : 0.f), c1 = reduce16(rowok ? p.y : 0.f), c2 = reduce16(rowok ? p.z : 0.f),
                  c3 = reduce16(rowok ? p.w : 0.f);
      if (i == 0) *reinterpret_cast<float4*>(a.colsum + (int64_t)blockIdx.x * K + k0) = make_float4(c0, c1, c2, c3);
    }
  }
  lds_barrier();
  // ---- stage 4: q = probs W; the two latent losses' common term; the straight-through value -----------------------------------
  {
    f32x4 acc[NTW];
#pragma unroll
    for (int j = 0; j < NTW; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int kg = 0; kg < K; kg += 16 * RING) {
#pragma unroll
      for (int b = 0; b < RING; ++b) {
        const int kb = kg + 16 * b;
        const float4 pb = *reinterpret_cast<const float4*>(Ps + i * ldk + kb + 4 * q);
#pragma unroll
        for (int j = 0; j < NTW; ++j) {
          acc[j] = mfma16(wr[b][j][0], pb.x, acc[j]);
          acc[j] = mfma16(wr[b][j][1], pb.y, acc[j]);
          acc[j] = mfma16(wr[b][j][2], pb.z, acc[j]);
          acc[j] = mfma16(wr[b][j][3], pb.w, acc[j]);
        }
        if (kb + 16 * RING < K) ring_load(b, kb + 16 * RING);
      }
    }
    float msep = 0.f;
#pragma unroll
    for (int j = 0; j < NTW; ++j) {
      const int e0 = 16 * (wave + 4 * j) + 4 * q;
      const float4 x4 = *reinterpret_cast<const float4*>(Xs + i * LD + e0);
      const float xv[4] = {x4.x, x4.y, x4.z, x4.w};
      float dv[4], st[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float d = acc[j][r] - xv[r];
        dv[r] = a.gcoef * d;
        st[r] = xv[r] + (acc[j][r] - xv[r]);
        msep += rowok ? d * d : 0.f;
      }
      if (rowok) {
        *reinterpret_cast<float4*>(a.q + n * E + e0) = make_float4(acc[j][0], acc[j][1], acc[j][2], acc[j][3]);
        *reinterpret_cast<float4*>(a.dq + n * E + e0) = make_float4(dv[0], dv[1], dv[2], dv[3]);
        *reinterpret_cast<float4*>(a.quant + n * E + e0) = make_float4(st[0], st[1], st[2], st[3]);
      }
    }
    msep = wave_sum(msep);
    if (lane == 0) red2[wave] = msep;
    lds_barrier();
    if (tid == 0) a.mse_partial[blockIdx.x] = (red2[0] + red2[1]) + (red2[2] + red2[3]);
  }
}

// mse = sum of the workgroups' partials / (N E), loss_vq = mse (1 + beta); perplexity of the mean assignment from the column sums
// (reference :1424-1427, :1432-1433).  One workgroup, fixed summation order, four independent chains per thread.
__global__ __launch_bounds__(1024) void vq_soft_finish_kernel(const float* __restrict__ mse_partial, const float* __restrict__ colsum,
                                                              int nblk, int N, int K, float inv_ne,
                                                              const float* __restrict__ one_plus_beta, float* __restrict__ mse_out,
                                                              float* __restrict__ loss_out, float* __restrict__ perp_out) {
  __shared__ float red[16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  auto sum4 = [&](const float* p, int64_t stride) {
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int j = 0;
    for (; j + 3 < nblk; j += 4) {
      s0 += p[(int64_t)j * stride];
      s1 += p[(int64_t)(j + 1) * stride];
      s2 += p[(int64_t)(j + 2) * stride];
      s3 += p[(int64_t)(j + 3) * stride];
    }
    for (; j < nblk; ++j) s0 += p[(int64_t)j * stride];
    return (s0 + s1) + (s2 + s3);
  };
  if (colsum && perp_out) {
    float ent = 0.f;
    for (int k = tid; k < K; k += 1024) {
      const float avg = sum4(colsum + k, K) / (float)N;
      ent += avg * logf(avg + 1e-10f);
    }
    ent = wave_sum(ent);
    if (lane == 0) red[wave] = ent;
    __syncthreads();
    if (tid == 0) {
      float s = 0.f;
      for (int w = 0; w < 16; ++w) s += red[w];
      perp_out[0] = expf(-s);
    }
    __syncthreads();
  }
  // the loss: wave 0, lane l sums partials l, l + 64, ...
  if (wave == 0) {
    float s = 0.f;
    for (int j = lane; j < nblk; j += 64) s += mse_partial[j];
    s = wave_sum(s);
    if (lane == 0) {
      const float mse = s * inv_ne;
      if (mse_out) mse_out[0] = mse;
      loss_out[0] = mse * one_plus_beta[0];
    }
  }
}

struct SoftBwdArgs {
  const float* dh;            // (N,E) gradient arriving at the straight-through value
  const float* gvq;           // device scalar: d total / d loss_vq
  const float* x; const float* q; const float* dq; const float* flat;
  const float* probs; const float* dist; const float* logvar;
  const float* w_mean; const float* w_lv; const float* cb;
  float* dd; float* dlv; float* dflat; float* gz;
  float ccoef;                // 2 beta / (N E)
  int N, K;
};

// LDS (floats): DQ[16][E+4] | DF[16][E+4] | DD[16][K+4] | DL[16][K+4] | red[64]
template <int E>
__global__ __launch_bounds__(256) void vq_soft_fused_bwd_kernel(SoftBwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int LD = E + 4, KS = E / 16, NTW = E / 64;
  const int K = a.K, ldk = K + 4, N = a.N;
  float* DQ = smem;
  float* DF = DQ + 16 * LD;
  float* DD = DF + 16 * LD;
  float* DL = DD + 16 * ldk;
  float* red = DL + 16 * ldk;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i = lane & 15, q = lane >> 4;
  const int n0 = blockIdx.x * 16, nrows = min(16, N - n0);
  const bool rowok = i < nrows;
  const int64_t n = n0 + (rowok ? i : nrows - 1);

  for (int e4 = tid; e4 < 16 * E / 4; e4 += 256) {
    const int r = e4 / (E / 4), c = (e4 - r * (E / 4)) * 4;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (r < nrows) v = *reinterpret_cast<const float4*>(a.dq + (int64_t)(n0 + r) * E + c);
    *reinterpret_cast<float4*>(DQ + r * LD + c) = v;
  }
  // ---- dprobs = dq W^T (contraction along the codebook rows), the row's sum_k dprobs_k probs_k ------------------------------------
  float4 wA[KS], wB[KS], pA, pB;
  auto load_tile = [&](int t, float4 (&w)[KS], float4& p) {
    const int64_t row = (int64_t)(16 * t + i) * E + 4 * q;
#pragma unroll
    for (int s = 0; s < KS; ++s) w[s] = *reinterpret_cast<const float4*>(a.cb + row + 16 * s);
    p = *reinterpret_cast<const float4*>(a.probs + n * K + 16 * t + 4 * q);
  };
  load_tile(wave, wA, pA);
  lds_barrier();
  const int ntile = K / 64;
  float dot = 0.f;
  auto tile = [&](int t, const float4 (&w)[KS], const float4& p) {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const float4 b = *reinterpret_cast<const float4*>(DQ + i * LD + 16 * s + 4 * q);
      mma4(acc, w[s], b);
    }
    dot += (acc[0] * p.x + acc[1] * p.y) + (acc[2] * p.z + acc[3] * p.w);
    *reinterpret_cast<float4*>(DD + i * ldk + 16 * t + 4 * q) = make_float4(acc[0], acc[1], acc[2], acc[3]);
  };
  for (int j = 0; j < ntile; j += 2) {
    load_tile(wave + 4 * (j + 1), wB, pB);
    tile(wave + 4 * j, wA, pA);
    if (j + 2 < ntile) load_tile(wave + 4 * (j + 2), wA, pA);
    tile(wave + 4 * (j + 1), wB, pB);
  }
  dot = quad_sum(dot);
  if (q == 0) red[wave * 16 + i] = dot;
  lds_barrier();
  dot = (red[i] + red[16 + i]) + (red[32 + i] + red[48 + i]);
  // ---- probabilities <- distances, logvar (vq_soft_bwd_kernel's expressions); this wave's own tiles ------------------------------
  float rs = 0.f;
  for (int j = 0; j < ntile; ++j) {
    const int k0 = 16 * (wave + 4 * j) + 4 * q;
    const float4 p4 = *reinterpret_cast<const float4*>(a.probs + n * K + k0);
    const float4 l4 = *reinterpret_cast<const float4*>(a.logvar + n * K + k0);
    const float4 d4 = *reinterpret_cast<const float4*>(a.dist + n * K + k0);
    const float4 dp4 = *reinterpret_cast<const float4*>(DD + i * ldk + k0);
    const float pv[4] = {p4.x, p4.y, p4.z, p4.w}, lv[4] = {l4.x, l4.y, l4.z, l4.w}, dv[4] = {d4.x, d4.y, d4.z, d4.w},
                dp[4] = {dp4.x, dp4.y, dp4.z, dp4.w};
    float v[4], dl[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float g = pv[r] * (dp[r] - dot);
      const float ex = expf(lv[r]);
      const float s_ = 1.0f / (ex * ex);
      v[r] = -g * s_ / 800.0f;
      dl[r] = g * (1.0f + dv[r] * s_ / 400.0f);
      rs += v[r];
    }
    *reinterpret_cast<float4*>(DD + i * ldk + k0) = make_float4(v[0], v[1], v[2], v[3]);
    *reinterpret_cast<float4*>(DL + i * ldk + k0) = make_float4(dl[0], dl[1], dl[2], dl[3]);
    if (rowok) {
      *reinterpret_cast<float4*>(a.dd + n * K + k0) = make_float4(v[0], v[1], v[2], v[3]);
      *reinterpret_cast<float4*>(a.dlv + n * K + k0) = make_float4(dl[0], dl[1], dl[2], dl[3]);
    }
  }
  // ---- dflat = (2 flat sum_k dd - 2 dd W) + dlogvar W_lv: two contractions across the rows k ----------------------------------------
  constexpr int RING = 4;
  float wr[RING][NTW][2][4];
  auto ring_load = [&](int b, int kb) {
#pragma unroll
    for (int j = 0; j < NTW; ++j)
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int64_t o = (int64_t)(kb + 4 * q + c) * E + 16 * (wave + 4 * j) + i;
        wr[b][j][0][c] = a.cb[o];
        wr[b][j][1][c] = a.w_lv[o];
      }
  };
#pragma unroll
  for (int b = 0; b < RING; ++b) ring_load(b, 16 * b);
  rs = quad_sum(rs);
  lds_barrier();                          // (every wave has read the row dots)
  if (q == 0) red[wave * 16 + i] = rs;
  lds_barrier();                          // DD / DL complete, row sums visible
  rs = (red[i] + red[16 + i]) + (red[32 + i] + red[48 + i]);
  float4 fl[NTW];
#pragma unroll
  for (int j = 0; j < NTW; ++j) fl[j] = *reinterpret_cast<const float4*>(a.flat + n * E + 16 * (wave + 4 * j) + 4 * q);
  {
    f32x4 at[NTW], au[NTW];
#pragma unroll
    for (int j = 0; j < NTW; ++j) {
      at[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
      au[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    for (int kg = 0; kg < K; kg += 16 * RING) {
#pragma unroll
      for (int b = 0; b < RING; ++b) {
        const int kb = kg + 16 * b;
        const float4 db = *reinterpret_cast<const float4*>(DD + i * ldk + kb + 4 * q);
        const float4 lb = *reinterpret_cast<const float4*>(DL + i * ldk + kb + 4 * q);
#pragma unroll
        for (int j = 0; j < NTW; ++j) {
          at[j] = mfma16(wr[b][j][0][0], db.x, at[j]);
          au[j] = mfma16(wr[b][j][1][0], lb.x, au[j]);
          at[j] = mfma16(wr[b][j][0][1], db.y, at[j]);
          au[j] = mfma16(wr[b][j][1][1], lb.y, au[j]);
          at[j] = mfma16(wr[b][j][0][2], db.z, at[j]);
          au[j] = mfma16(wr[b][j][1][2], lb.z, au[j]);
          at[j] = mfma16(wr[b][j][0][3], db.w, at[j]);
          au[j] = mfma16(wr[b][j][1][3], lb.w, au[j]);
        }
        if (kb + 16 * RING < K) ring_load(b, kb + 16 * RING);
      }
    }
#pragma unroll
    for (int j = 0; j < NTW; ++j) {
      const int e0 = 16 * (wave + 4 * j) + 4 * q;
      const float fv[4] = {fl[j].x, fl[j].y, fl[j].z, fl[j].w};
      float o[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) o[r] = (2.0f * fv[r] * rs - 2.0f * at[j][r]) + au[j][r];
      *reinterpret_cast<float4*>(DF + i * LD + e0) = make_float4(o[0], o[1], o[2], o[3]);
      if (rowok) *reinterpret_cast<float4*>(a.dflat + n * E + e0) = make_float4(o[0], o[1], o[2], o[3]);
    }
  }
  // ---- gz = [dh + c (x - q)] + dflat W_mean (contraction across mean_layer's rows) ------------------------------------------------
  float wmr[KS][NTW][4];
#pragma unroll
  for (int s = 0; s < KS; ++s)
#pragma unroll
    for (int j = 0; j < NTW; ++j)
#pragma unroll
      for (int c = 0; c < 4; ++c) wmr[s][j][c] = a.w_mean[(int64_t)(16 * s + 4 * q + c) * E + 16 * (wave + 4 * j) + i];
  float4 xv[NTW], qv[NTW], hv[NTW];
#pragma unroll
  for (int j = 0; j < NTW; ++j) {
    const int64_t o = n * E + 16 * (wave + 4 * j) + 4 * q;
    xv[j] = *reinterpret_cast<const float4*>(a.x + o);
    qv[j] = *reinterpret_cast<const float4*>(a.q + o);
    hv[j] = a.dh ? *reinterpret_cast<const float4*>(a.dh + o) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  const float cc = a.gvq ? a.gvq[0] * a.ccoef : 0.f;
  lds_barrier();
  {
    f32x4 acc[NTW];
#pragma unroll
    for (int j = 0; j < NTW; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const float4 b = *reinterpret_cast<const float4*>(DF + i * LD + 16 * s + 4 * q);
#pragma unroll
      for (int j = 0; j < NTW; ++j) {
        acc[j] = mfma16(wmr[s][j][0], b.x, acc[j]);
        acc[j] = mfma16(wmr[s][j][1], b.y, acc[j]);
        acc[j] = mfma16(wmr[s][j][2], b.z, acc[j]);
        acc[j] = mfma16(wmr[s][j][3], b.w, acc[j]);
      }
    }
    if (rowok) {
#pragma unroll
      for (int j = 0; j < NTW; ++j) {
        const float4 x4 = xv[j], q4 = qv[j], h4 = hv[j];
        float4 g;
        if (a.dh) {
          g.x = fmaf(cc, x4.x - q4.x, h4.x); g.y = fmaf(cc, x4.y - q4.y, h4.y);
          g.z = fmaf(cc, x4.z - q4.z, h4.z); g.w = fmaf(cc, x4.w - q4.w, h4.w);
        } else {
          g.x = cc * (x4.x - q4.x); g.y = cc * (x4.y - q4.y); g.z = cc * (x4.z - q4.z); g.w = cc * (x4.w - q4.w);
        }
        g.x += acc[j][0]; g.y += acc[j][1]; g.z += acc[j][2]; g.w += acc[j][3];
        *reinterpret_cast<float4*>(a.gz + n * E + 16 * (wave + 4 * j) + 4 * q) = g;
      }
    }
  }
}

}  // namespace
}  // namespace g2v

using namespace g2v;

extern "C" int g2v_vq_soft_fused_ok(int N, int E, int K) {
  return (N > 0 && E == SOFT_E && K > 0 && (K % 128) == 0 && K <= 1024) ? 1 : 0;
}
extern "C" int g2v_vq_soft_fused_blocks(int N) { return N > 0 ? cdiv(N, 16) : 0; }

extern "C" int g2v_vq_soft_fused_fwd(const float* x, const float* w_mean, const float* b_mean, const float* w_logvar,
                                     const float* b_logvar, const float* codebook, const float* code_sqnorm, float* flat,
                                     float* logvar, float* dist, float* probs, float* q, float* dq, float* quant,
                                     float* mse_partial, float* colsum, float g_scale, int N, int E, int K, g2v_stream_t stream) {
  G2V_REQUIRE(x && w_mean && b_mean && w_logvar && b_logvar && codebook && code_sqnorm, "null pointer");
  G2V_REQUIRE(flat && logvar && dist && probs && q && dq && quant && mse_partial, "null output pointer");
  if (!g2v_vq_soft_fused_ok(N, E, K)) {
    set_error("g2v_vq_soft_fused_fwd: shape not served (see g2v_vq_soft_fused_ok)");
    return G2V_ERR_UNSUPPORTED;
  }
  auto a16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
  G2V_REQUIRE(a16(x) && a16(w_mean) && a16(b_mean) && a16(w_logvar) && a16(b_logvar) && a16(codebook) && a16(code_sqnorm) &&
                  a16(flat) && a16(logvar) && a16(dist) && a16(probs) && a16(q) && a16(dq) && a16(quant) && (!colsum || a16(colsum)),
              "16-byte alignment");
  SoftFwdArgs a;
  a.x = x; a.w_mean = w_mean; a.b_mean = b_mean; a.w_lv = w_logvar; a.b_lv = b_logvar; a.cb = codebook; a.wsq = code_sqnorm;
  a.flat = flat; a.logvar = logvar; a.dist = dist; a.probs = probs; a.q = q; a.dq = dq; a.quant = quant;
  a.mse_partial = mse_partial; a.colsum = colsum;
  a.gcoef = 2.0f * g_scale / ((float)N * (float)E);
  a.N = N; a.K = K;
  const size_t lds = (size_t)(2 * 16 * (SOFT_E + 4) + 16 * (K + 4) + 64 + 4) * sizeof(float);
  if (lds > 48 * 1024)
    (void)hipFuncSetAttribute((const void*)vq_soft_fused_fwd_kernel<SOFT_E>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL(vq_soft_fused_fwd_kernel<SOFT_E>, dim3(cdiv(N, 16)), dim3(256), lds, (hipStream_t)stream, a);
  G2V_CHECK_LAUNCH();
  return G2V_OK;
}

extern "C" int g2v_vq_soft_finish(const float* mse_partial, const float* colsum, const float* one_plus_beta, float* mse,
                                  float* loss_vq, float* perplexity, int N, int E, int K, g2v_stream_t stream) {
  G2V_REQUIRE(mse_partial && one_plus_beta && loss_vq, "null pointer");
  G2V_REQUIRE(N > 0 && E > 0 && K > 0, "non-positive size");
  hipLaunchKernelGGL(vq_soft_finish_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, mse_partial, colsum, cdiv(N, 16), N, K,
                     1.0f / ((float)N * (float)E), one_plus_beta, mse, loss_vq, perplexity);
  G2V_CHECK_LAUNCH();
  return G2V_OK;
}

extern "C" int g2v_vq_soft_fused_bwd(const float* dh, const float* g_loss, const float* x, const float* q, const float* dq,
                                     const float* flat, const float* probs, const float* dist, const float* logvar,
                                     const float* w_mean, const float* w_logvar, const float* codebook, float* dd, float* dlogvar,
                                     float* dflat, float* gz, float beta, int N, int E, int K, g2v_stream_t stream) {
  G2V_REQUIRE(x && q && dq && flat && probs && dist && logvar && w_mean && w_logvar && codebook, "null pointer");
  G2V_REQUIRE(dd && dlogvar && dflat && gz, "null output pointer");
  if (!g2v_vq_soft_fused_ok(N, E, K)) {
    set_error("g2v_vq_soft_fused_bwd: shape not served (see g2v_vq_soft_fused_ok)");
    return G2V_ERR_UNSUPPORTED;
  }
  auto a16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
  G2V_REQUIRE((!dh || a16(dh)) && a16(x) && a16(q) && a16(dq) && a16(flat) && a16(probs) && a16(dist) && a16(logvar) &&
                  a16(dd) && a16(dlogvar) && a16(dflat) && a16(gz),
              "16-byte alignment");
  SoftBwdArgs a;
  a.dh = dh; a.gvq = g_loss; a.x = x; a.q = q; a.dq = dq; a.flat = flat; a.probs = probs; a.dist = dist; a.logvar = logvar;
  a.w_mean = w_mean; a.w_lv = w_logvar; a.cb = codebook;
  a.dd = dd; a.dlv = dlogvar; a.dflat = dflat; a.gz = gz;
  a.ccoef = 2.0f * beta / ((float)N * (float)E);
  a.N = N; a.K = K;
  const size_t lds = (size_t)(2 * 16 * (SOFT_E + 4) + 2 * 16 * (K + 4) + 64) * sizeof(float);
  if (lds > 48 * 1024)
    (void)hipFuncSetAttribute((const void*)vq_soft_fused_bwd_kernel<SOFT_E>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL(vq_soft_fused_bwd_kernel<SOFT_E>, dim3(cdiv(N, 16)), dim3(256), lds, (hipStream_t)stream, a);
  G2V_CHECK_LAUNCH();
  return G2V_OK;
}
