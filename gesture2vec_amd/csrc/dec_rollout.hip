// dec_rollout.hip -- autoregressive pose-decoder rollout (K9), forward and BPTT.
//
// Replaces the T-1 step Python loop model/Autoencoder_VQVAE_model.py:1039-1054 over
// Generator.forward (:646-683) -> BahdanauAttnDecoderRNN.forward (:499-592), att off, 2 GRU layers.
//
// Structure: the step is row-local EXCEPT for BatchNorm1d's batch statistics
// (forward) and their gradient sums (backward).  Those grid-wide reductions are the only seams,
// and a kernel boundary (~1.5 us) is cheaper on MI355X than an in-kernel grid barrier (4-5 us),
// so the rollout is ONE LAUNCH PER TIME STEP, each launch fusing everything between two seams:
//
//   fwd kernel t :  [finish BN(u_t) from per-block partials] -> ReLU -> GRU cell 0 -> inter-layer
//                   dropout -> GRU cell 1 -> out_layer -> y_t -> Dropout(0.95) -> pre_linear ->
//                   u_{t+1} + per-block partial sums of (u_{t+1} - b)
//   bwd kernel t :  [finish BN-backward of step t+1 from per-block partials -> du_{t+1} ->
//                   feedback into dy_t] -> out_layer^T -> GRU cell 1 bwd -> GRU cell 0 bwd ->
//                   ReLU bwd -> per-block partial sums for BN-backward of step t
//
// 16 batch rows per 256-thread workgroup; all contractions are v_mfma_f32_16x16x4_f32 with the
// activations staged in LDS (B operand) and the weights streamed from L2 as A fragments.
#include "common.hpp"

namespace g2v {

#ifdef G2V_STAMPS
__device__ unsigned long long g2v_stamps[64 * 16];
#define STAMP(k)                                                                                   \
  do {                                                                                             \
    if (threadIdx.x == 0 && blockIdx.x < 4 && t == 5)                                              \
      g2v_stamps[blockIdx.x * 16 + (k)] = __builtin_amdgcn_s_memtime();                           \
  } while (0)
#else
#define STAMP(k)
#endif

struct DecDims {
  int T, B, D, H;
  float p_drop;
  int n_pre, conditioned, training, nblk;
};

// packed forward weights (fragment-major, see common.hpp): offsets in floats into the workspace
struct DecPackF {
  const float* pre;    // rows H, K = D
  const float* ih0; const float* hh0; const float* ih1; const float* hh1;   // 3 gate groups x H rows, K = H
  const float* out;    // rows D, K = H
};

// ---- one GRU cell for the feature tiles of this wave ------------------------------------------------
// x-operand Xin [16][ldh] (layer input), Xh [16][ldh] (previous hidden).  Writes h_new (after optional
// inter-layer dropout) to `Hnext_lds`, h_new to global h_out, and the gates.
template <int KSH_T>
__device__ __forceinline__ void gru_cell_fwd(const float* __restrict__ p_ih, const float* __restrict__ p_hh,
                                             const float* __restrict__ b_ih, const float* __restrict__ b_hh,
                                             const float* Xin, const float* Xh, int ldh, int H, int Hp,
                                             float* Hnext_lds,            // [16][ldh]: what the next stage consumes
                                             float* __restrict__ h_out,   // global (B,H) row block base (row b0)
                                             float* __restrict__ gates,   // global (B,4H) row block base or null
                                             const uint8_t* __restrict__ keep, float keep_scale,  // inter-layer dropout
                                             float* __restrict__ xdrop_out,  // global (B,H) dropped output or null
                                             int nrows, int lane, int wave) {
  const int i = lane & 15, q = lane >> 4;
  const int ntile = Hp >> 4, KS = Hp >> 4;
  for (int ft = wave; ft < ntile; ft += 4) {
    f32x4 ai[3], ah[3];
#pragma unroll
    for (int g = 0; g < 3; ++g) {
      ai[g] = (f32x4){0.f, 0.f, 0.f, 0.f};
      ah[g] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    wave_gemm_p<3, KSH_T>(ai, p_ih, KS, ft, ntile, Xin, ldh, lane);
    wave_gemm_p<3, KSH_T>(ah, p_hh, KS, ft, ntile, Xh, ldh, lane);
    const int f0 = 16 * ft + 4 * q;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int f = f0 + r;
      if (f >= H) continue;
      const float hp = Xh[i * ldh + f];
      const float rr = sigmoidf_((ai[0][r] + b_ih[f]) + (ah[0][r] + b_hh[f]));
      const float zz = sigmoidf_((ai[1][r] + b_ih[H + f]) + (ah[1][r] + b_hh[H + f]));
      const float ghn = ah[2][r] + b_hh[2 * H + f];
      const float nn = tanhf((ai[2][r] + b_ih[2 * H + f]) + rr * ghn);
      const float hn = (1.0f - zz) * nn + zz * hp;
      float xd = hn;
      if (keep) xd = (i < nrows && keep[(int64_t)i * H + f]) ? hn * keep_scale : 0.f;
      Hnext_lds[i * ldh + f] = xd;
      if (i < nrows) {
        h_out[(int64_t)i * H + f] = hn;
        if (xdrop_out) xdrop_out[(int64_t)i * H + f] = xd;
        if (gates) {
          float* go = gates + (int64_t)i * 4 * H;
          go[f] = rr; go[H + f] = zz; go[2 * H + f] = nn; go[3 * H + f] = ghn;
        }
      }
    }
  }
}

template <int HS>
__global__ __launch_bounds__(256) void dec_step_fwd_kernel(const float* __restrict__ target,
                                                           const float* __restrict__ h_init, g2v_dec_weights w,
                                                           DecPackF pk, g2v_dec_saved sv,
                                                           const uint8_t* __restrict__ keep95,
                                                           const uint8_t* __restrict__ keep_l0, DecDims dm, int t) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int T = dm.T, B = dm.B, D = dm.D, H = dm.H;
  const int Hp = (H + 15) & ~15, Dp = (D + 15) & ~15, ldh = Hp + 4, ldd = Dp + 4;
  float* Xa = smem;                 // a_t               [16][ldh]
  float* Xh0 = Xa + 16 * ldh;       // h0_{t-1}
  float* Xh1 = Xh0 + 16 * ldh;      // h1_{t-1}
  float* Xx1 = Xh1 + 16 * ldh;      // dropped h0_t (input of layer 1)
  float* Xh1n = Xx1 + 16 * ldh;     // h1_t
  float* Xy = Xh1n + 16 * ldh;      // xin_{t+1}         [16][ldd]
  float* st = Xy + 16 * ldd;        // mean[Hp], invstd[Hp]
  float* red = st + 2 * Hp;         // column sums of the BN partials [2H]
  float* red_scratch = red + 2 * Hp;  // [256]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b0 = blockIdx.x * 16;
  const int nrows = min(16, B - b0);
  const int i = lane & 15, q = lane >> 4;
  const bool has_next = (t < T - 1);

  STAMP(0);
  // zero the operand tiles once (padding columns / rows must be 0 for the MFMA contractions)
  for (int e = tid; e < 5 * 16 * ldh + 16 * ldd; e += 256) smem[e] = 0.f;
  __syncthreads();
  STAMP(1);

  if (t == 0) {
    // seed the state arrays: h0[0], h1[0] = h_init (the quantised latent)
    for (int e = tid; e < 16 * H; e += 256) {
      const int r = e / H, f = e - r * H;
      if (r < nrows) {
        sv.h0[(int64_t)(b0 + r) * H + f] = h_init[(int64_t)(b0 + r) * H + f];
        sv.h1[(int64_t)(b0 + r) * H + f] = h_init[((int64_t)B + b0 + r) * H + f];
      }
    }
  } else {
    // ---- (a) BatchNorm statistics of u_t -----------------------------------------------------------
    const float* ut = sv.u + (int64_t)(t - 1) * B * H;
    if (dm.training) {
      const float* part = sv.bn_partial + (int64_t)((t - 1) & 1) * dm.nblk * 2 * H;
      reduce_partials(part, dm.nblk, 2 * H, red, red_scratch, tid);
      STAMP(2);
      for (int f = tid; f < H; f += 256) {
        const float s1 = red[f], s2 = red[H + f];
        const float mv = s1 / (float)B;
        const float var = fmaxf(s2 / (float)B - mv * mv, 0.f);   // biased batch variance
        const float mean = mv + w.b_pre[f];
        st[f] = mean;
        st[Hp + f] = 1.0f / sqrtf(var + 1e-5f);
        if (blockIdx.x == 0) {
          sv.bn_stats[(int64_t)(t - 1) * 2 * H + f] = mean;
          sv.bn_stats[(int64_t)(t - 1) * 2 * H + H + f] = var;
        }
      }
    } else {
      for (int f = tid; f < H; f += 256) {
        st[f] = w.bn_running_mean[f];
        st[Hp + f] = 1.0f / sqrtf(w.bn_running_var[f] + 1e-5f);
      }
    }
    __syncthreads();
    // ---- (b) a_t = ReLU(BN(u_t)); stage previous hidden states ------------------------------------
    for (int e = tid; e < 16 * H; e += 256) {
      const int r = e / H, f = e - r * H;
      if (r >= nrows) continue;
      const int64_t row = (int64_t)(b0 + r) * H + f;
      const float u = ut[row];
      float a = (u - st[f]) * st[Hp + f] * w.bn_w[f] + w.bn_b[f];
      a = fmaxf(a, 0.f);
      Xa[r * ldh + f] = a;
      if (sv.a) sv.a[(int64_t)(t - 1) * B * H + row] = a;
      Xh0[r * ldh + f] = sv.h0[(int64_t)(t - 1) * B * H + row];
      Xh1[r * ldh + f] = sv.h1[(int64_t)(t - 1) * B * H + row];
    }
    __syncthreads();
    STAMP(3);
    // ---- (c) GRU layer 0 ---------------------------------------------------------------------------
    const bool drop = dm.training && keep_l0 && dm.p_drop > 0.f;
    constexpr int KSH_T = HS / 16;
    gru_cell_fwd<KSH_T>(pk.ih0, pk.hh0, w.b_ih0, w.b_hh0, Xa, Xh0, ldh, H, Hp, Xx1,
                 sv.h0 + ((int64_t)t * B + b0) * H,
                 sv.gates0 ? sv.gates0 + ((int64_t)(t - 1) * B + b0) * 4 * H : nullptr,
                 drop ? keep_l0 + ((int64_t)(t - 1) * B + b0) * H : nullptr, 1.0f / (1.0f - dm.p_drop),
                 (drop && sv.x1) ? sv.x1 + ((int64_t)(t - 1) * B + b0) * H : nullptr, nrows, lane, wave);
    __syncthreads();
    STAMP(4);
    // ---- (d) GRU layer 1 ---------------------------------------------------------------------------
    gru_cell_fwd<KSH_T>(pk.ih1, pk.hh1, w.b_ih1, w.b_hh1, Xx1, Xh1, ldh, H, Hp, Xh1n,
                 sv.h1 + ((int64_t)t * B + b0) * H,
                 sv.gates1 ? sv.gates1 + ((int64_t)(t - 1) * B + b0) * 4 * H : nullptr, nullptr, 1.0f, nullptr, nrows,
                 lane, wave);
    __syncthreads();
    STAMP(5);
  }

  // ---- (e) y_t = out_layer(h1_t)  (t == 0: y_0 = target frame 0), next decoder input ---------------
  {
    const int ntile = Dp >> 4;
    for (int ft = wave; ft < ntile; ft += 4) {
      f32x4 acc[1] = {(f32x4){0.f, 0.f, 0.f, 0.f}};
      if (t > 0) wave_gemm_p<1, HS / 16>(acc, pk.out, Hp >> 4, ft, 0, Xh1n, ldh, lane);
      const int d0 = 16 * ft + 4 * q;
      if (i < nrows) {
        const int b = b0 + i;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int d = d0 + r;
          if (d >= D) continue;
          float y;
          if (t == 0) y = target[((int64_t)b * T + 0) * D + d];
          else y = acc[0][r] + w.b_out[d];
          sv.y[((int64_t)t * B + b) * D + d] = y;
          if (has_next) {
            const float src = (t < dm.n_pre) ? target[((int64_t)b * T + t) * D + d] : y;   // :1049-1052
            float xin = 0.f;
            if (dm.conditioned && keep95[((int64_t)t * B + b) * D + d]) xin = src * 20.0f;   // Dropout(0.95): 1/(1-0.95)
            Xy[i * ldd + d] = xin;
            if (sv.xin) sv.xin[((int64_t)t * B + b) * D + d] = xin;
          }
        }
      }
    }
  }
  if (!has_next) return;
  __syncthreads();
  STAMP(6);
  // ---- (f) u_{t+1} = pre_linear.0(xin_{t+1}) and per-block BN partial sums of (u - b) ----------------
  {
    const int ntile = Hp >> 4;
    float* part = sv.bn_partial + ((int64_t)(t & 1) * dm.nblk + blockIdx.x) * 2 * H;
    for (int ft = wave; ft < ntile; ft += 4) {
      f32x4 acc[1] = {(f32x4){0.f, 0.f, 0.f, 0.f}};
      wave_gemm_p<1, 0>(acc, pk.pre, Dp >> 4, ft, 0, Xy, ldd, lane);
      const int f0 = 16 * ft + 4 * q;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int f = f0 + r;
        const float v = (i < nrows && f < H) ? acc[0][r] : 0.f;
        if (i < nrows && f < H) sv.u[((int64_t)t * B + b0 + i) * H + f] = v + w.b_pre[f];
        const float s1 = reduce16(v), s2 = reduce16(v * v);
        if (i == 0 && f < H) {
          part[f] = s1;
          part[H + f] = s2;
        }
      }
    }
  }
  STAMP(7);
}

// running_mean / running_var (momentum 0.1, unbiased variance), applied T-1 times in step order
__global__ void bn_running_update_kernel(const float* __restrict__ bn_stats, float* __restrict__ rm,
                                         float* __restrict__ rv, int steps, int H, int B) {
  const int f = blockIdx.x * blockDim.x + threadIdx.x;
  if (f >= H) return;
  float m = rm[f], v = rv[f];
  const float unbias = (B > 1) ? (float)B / (float)(B - 1) : 1.0f;
  for (int s = 0; s < steps; ++s) {
    m = 0.9f * m + 0.1f * bn_stats[(int64_t)s * 2 * H + f];
    v = 0.9f * v + 0.1f * (bn_stats[(int64_t)s * 2 * H + H + f] * unbias);
  }
  rm[f] = m;
  rv[f] = v;
}

// =====================================================================================================
// backward
// =====================================================================================================
struct DecTW {   // PACKED transposed weights (fragment-major; rows = output feature of the backward contraction)
  const float* w_pre_t;   // rows D, K = H    (W_pre^T)
  const float* w_out_t;   // rows H, K = D    (W_out^T)
  const float* w_ih0_t; const float* w_hh0_t; const float* w_ih1_t; const float* w_hh1_t;   // rows H, K = 3H each
};

// GRU cell backward for the feature tiles of this wave.
//   dh_in(row,f) = carry (global, may be null on the first step) + [add_lds ? Add[row][f] : 0] + acc (from the caller's GEMM)
// writes dgi / dgh (global), Gi / Gh tiles (LDS, MFMA B operands for the next contractions) and
// direct = dh * z into Dd (LDS).
__device__ __forceinline__ void gru_cell_bwd_tile(const f32x4& acc, const float* __restrict__ carry, float extra_scale,
                                                  const uint8_t* __restrict__ keep,   // applied to acc (inter-layer dropout bwd)
                                                  const float* __restrict__ gates, const float* __restrict__ hprev,
                                                  float* __restrict__ dgi, float* __restrict__ dgh, float* Gi, float* Gh,
                                                  int ldg, float* Dd, int ldh, int H, int ft, int nrows, int lane) {
  const int i = lane & 15, q = lane >> 4;
  const int f0 = 16 * ft + 4 * q;
  const int G = 3 * H;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int f = f0 + r;
    if (f >= H) continue;
    float g_r = 0.f, g_z = 0.f, g_n = 0.f, g_hn = 0.f, direct = 0.f;
    if (i < nrows) {
      float dh = acc[r];
      if (keep) dh = keep[(int64_t)i * H + f] ? dh * extra_scale : 0.f;
      if (carry) dh += carry[(int64_t)i * H + f];
      const float* go = gates + (int64_t)i * 4 * H;
      const float rr = go[f], zz = go[H + f], nn = go[2 * H + f], ghn = go[3 * H + f];
      const float hp = hprev[(int64_t)i * H + f];
      const float dn = dh * (1.0f - zz);
      const float dz = dh * (hp - nn);
      const float dnp = dn * (1.0f - nn * nn);
      g_n = dnp;
      g_hn = dnp * rr;
      g_r = dnp * ghn * rr * (1.0f - rr);
      g_z = dz * zz * (1.0f - zz);
      direct = dh * zz;
      float* o1 = dgi + (int64_t)i * G;
      float* o2 = dgh + (int64_t)i * G;
      o1[f] = g_r; o1[H + f] = g_z; o1[2 * H + f] = g_n;
      o2[f] = g_r; o2[H + f] = g_z; o2[2 * H + f] = g_hn;
    }
    Gi[i * ldg + f] = g_r; Gi[i * ldg + H + f] = g_z; Gi[i * ldg + 2 * H + f] = g_n;
    Gh[i * ldg + f] = g_r; Gh[i * ldg + H + f] = g_z; Gh[i * ldg + 2 * H + f] = g_hn;
    Dd[i * ldh + f] = direct;
  }
}

template <int HS>
__global__ __launch_bounds__(256) void dec_step_bwd_kernel(g2v_dec_weights w, DecTW tw, g2v_dec_saved sv,
                                                           g2v_dec_grads gr, const uint8_t* __restrict__ keep95,
                                                           const uint8_t* __restrict__ keep_l0, DecDims dm, int t) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int T = dm.T, B = dm.B, D = dm.D, H = dm.H, G = 3 * H;
  const int Hp = (H + 15) & ~15, Dp = (D + 15) & ~15, Gp = (G + 15) & ~15;
  const int ldh = Hp + 4, ldd = Dp + 4, ldg = Gp + 4;
  float* Xdu = smem;                // du_{t+1}          [16][ldh]
  float* Xdy = Xdu + 16 * ldh;      // dy_t              [16][ldd]
  float* Gi = Xdy + 16 * ldd;       // gate grads (input side)   [16][ldg]
  float* Gh = Gi + 16 * ldg;        // gate grads (hidden side)  [16][ldg]
  float* Dd = Gh + 16 * ldg;        // dh * z            [16][ldh]
  float* Xdx = Dd + 16 * ldh;       // dh0 incoming      [16][ldh]
  float* st = Xdx + 16 * ldh;       // S1[Hp], S2[Hp]
  float* red = st + 2 * Hp;         // [2H]
  float* red_scratch = red + 2 * Hp;  // [256]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b0 = blockIdx.x * 16;
  const int nrows = min(16, B - b0);
  const int i = lane & 15, q = lane >> 4;
  const bool last = (t == T - 1);   // first kernel of the backward sweep
  const int nth = Hp >> 4, ntd = Dp >> 4;

  for (int e = tid; e < 16 * (3 * ldh + ldd + 2 * ldg); e += 256) smem[e] = 0.f;
  __syncthreads();

  // ================= Part A: finish BatchNorm backward of step t+1 ===================================
  if (!last) {
    const float* part = gr.bn_bwd_partial + (int64_t)((t + 1) & 1) * dm.nblk * 2 * H;
    reduce_partials(part, dm.nblk, 2 * H, red, red_scratch, tid);
    for (int f = tid; f < H; f += 256) {
      const float s1 = red[f], s2 = red[H + f];
      st[f] = s1;
      st[Hp + f] = s2;
      if (blockIdx.x == 0) {   // d gamma / d beta accumulate over the steps (one writer, stream ordered)
        const bool first_acc = (t == T - 2);
        gr.d_bn_w[f] = (first_acc ? 0.f : gr.d_bn_w[f]) + s2;
        gr.d_bn_b[f] = (first_acc ? 0.f : gr.d_bn_b[f]) + s1;
      }
    }
    __syncthreads();
    const float invB = 1.0f / (float)B;
    const float* stats = sv.bn_stats + (int64_t)t * 2 * H;   // step t+1 is stored at index t
    for (int e = tid; e < 16 * H; e += 256) {
      const int r = e / H, f = e - r * H;
      if (r >= nrows) continue;
      const int64_t row = ((int64_t)t * B + b0 + r) * H + f;
      const float invstd = 1.0f / sqrtf(stats[H + f] + 1e-5f);
      const float xhat = (sv.u[row] - stats[f]) * invstd;
      const float du = w.bn_w[f] * invstd * (gr.dbn[row] - st[f] * invB - xhat * st[Hp + f] * invB);
      gr.du[row] = du;
      Xdu[r * ldh + f] = du;
    }
    __syncthreads();
  } else if (blockIdx.x == 0 && T == 2) {
    // degenerate: single decode step, no Part A ever accumulates
  }
  if (t == 0) return;   // only the BN finish of step 1 was left (y_0 is data: no feedback needed)

  // ================= Part B: dy_t (loss + feedback) ===================================================
  {
    const bool feedback = (!last) && dm.conditioned && (t >= dm.n_pre);
    for (int ft = wave; ft < ntd; ft += 4) {
      f32x4 acc[1] = {(f32x4){0.f, 0.f, 0.f, 0.f}};
      if (feedback) wave_gemm_p<1, HS / 16>(acc, tw.w_pre_t, Hp >> 4, ft, 0, Xdu, ldh, lane);
      const int d0 = 16 * ft + 4 * q;
      if (i < nrows) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int d = d0 + r;
          if (d >= D) continue;
          const int64_t idx = ((int64_t)t * B + b0 + i) * D + d;
          float dy = gr.dy[idx];
          if (feedback && keep95[idx]) dy += acc[0][r] * 20.0f;
          gr.dy[idx] = dy;
          Xdy[i * ldd + d] = dy;
        }
      }
    }
  }
  __syncthreads();
  const float* carry0 = last ? nullptr : gr.dh_init + (int64_t)b0 * H;
  const float* carry1 = last ? nullptr : gr.dh_init + ((int64_t)B + b0) * H;
  float* carry0_w = gr.dh_init + (int64_t)b0 * H;
  float* carry1_w = gr.dh_init + ((int64_t)B + b0) * H;
  // ---- dh1 = carry + dy W_out ; GRU cell 1 backward -------------------------------------------------
  {
    for (int ft = wave; ft < nth; ft += 4) {
      f32x4 acc[1] = {(f32x4){0.f, 0.f, 0.f, 0.f}};
      wave_gemm_p<1, 0>(acc, tw.w_out_t, Dp >> 4, ft, 0, Xdy, ldd, lane);
      gru_cell_bwd_tile(acc[0], carry1, 1.0f, nullptr, sv.gates1 + ((int64_t)(t - 1) * B + b0) * 4 * H,
                        sv.h1 + ((int64_t)(t - 1) * B + b0) * H, gr.dgi1 + ((int64_t)(t - 1) * B + b0) * G,
                        gr.dgh1 + ((int64_t)(t - 1) * B + b0) * G, Gi, Gh, ldg, Dd, ldh, H, ft, nrows, lane);
    }
  }
  __syncthreads();
  // ---- carry1' = dh1*z + dgh1 W_hh1 ;  dx1 = dgi1 W_ih1 -> dh0 (inter-layer dropout bwd) ----------------
  {
    for (int ft = wave; ft < nth; ft += 4) {
      f32x4 a1[1] = {(f32x4){0.f, 0.f, 0.f, 0.f}}, a2[1] = {(f32x4){0.f, 0.f, 0.f, 0.f}};
      wave_gemm_p<1, 3 * HS / 16>(a1, tw.w_hh1_t, Gp >> 4, ft, 0, Gh, ldg, lane);
      wave_gemm_p<1, 3 * HS / 16>(a2, tw.w_ih1_t, Gp >> 4, ft, 0, Gi, ldg, lane);
      const int f0 = 16 * ft + 4 * q;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int f = f0 + r;
        if (f >= H) continue;
        if (i < nrows) carry1_w[(int64_t)i * H + f] = Dd[i * ldh + f] + a1[0][r];
        Xdx[i * ldh + f] = a2[0][r];
      }
    }
  }
  __syncthreads();
  // ---- GRU cell 0 backward (Gi/Gh/Dd are reused) ------------------------------------------------------
  {
    const bool drop = keep_l0 && dm.p_drop > 0.f;
    for (int ft = wave; ft < nth; ft += 4) {
      const int f0 = 16 * ft + 4 * q;
      f32x4 acc;
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[r] = (f0 + r < H) ? Xdx[i * ldh + f0 + r] : 0.f;
      gru_cell_bwd_tile(acc, carry0, 1.0f / (1.0f - dm.p_drop),
                        drop ? keep_l0 + ((int64_t)(t - 1) * B + b0) * H : nullptr,
                        sv.gates0 + ((int64_t)(t - 1) * B + b0) * 4 * H, sv.h0 + ((int64_t)(t - 1) * B + b0) * H,
                        gr.dgi0 + ((int64_t)(t - 1) * B + b0) * G, gr.dgh0 + ((int64_t)(t - 1) * B + b0) * G, Gi, Gh,
                        ldg, Dd, ldh, H, ft, nrows, lane);
    }
  }
  __syncthreads();
  // ---- carry0' = dh0*z + dgh0 W_hh0 ;  da = dgi0 W_ih0 -> ReLU bwd -> dbn_t + BN-backward partial sums ----
  {
    const float* stats = sv.bn_stats + (int64_t)(t - 1) * 2 * H;
    float* part = gr.bn_bwd_partial + ((int64_t)(t & 1) * dm.nblk + blockIdx.x) * 2 * H;
    for (int ft = wave; ft < nth; ft += 4) {
      f32x4 a1[1] = {(f32x4){0.f, 0.f, 0.f, 0.f}}, a2[1] = {(f32x4){0.f, 0.f, 0.f, 0.f}};
      wave_gemm_p<1, 3 * HS / 16>(a1, tw.w_hh0_t, Gp >> 4, ft, 0, Gh, ldg, lane);
      wave_gemm_p<1, 3 * HS / 16>(a2, tw.w_ih0_t, Gp >> 4, ft, 0, Gi, ldg, lane);
      const int f0 = 16 * ft + 4 * q;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int f = f0 + r;
        float dbn = 0.f, dbx = 0.f;
        if (f < H && i < nrows) {
          carry0_w[(int64_t)i * H + f] = Dd[i * ldh + f] + a1[0][r];
          const int64_t row = ((int64_t)(t - 1) * B + b0 + i) * H + f;
          dbn = (sv.a[row] > 0.f) ? a2[0][r] : 0.f;
          gr.dbn[row] = dbn;
          const float invstd = 1.0f / sqrtf(stats[H + f] + 1e-5f);
          dbx = dbn * ((sv.u[row] - stats[f]) * invstd);
        }
        const float s1 = reduce16(dbn), s2 = reduce16(dbx);
        if (i == 0 && f < H) {
          part[f] = s1;
          part[H + f] = s2;
        }
      }
    }
  }
}

}  // namespace g2v

using namespace g2v;

#ifdef G2V_STAMPS
extern "C" int g2v_read_stamps(unsigned long long* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g2v_stamps), sizeof(unsigned long long) * 64 * 16);
}
#endif

extern "C" int g2v_dec_rollout_blocks(int B) { return B > 0 ? cdiv(B, 16) : 0; }

static size_t dec_fwd_lds(int D, int H) {
  const int Hp = (H + 15) & ~15, Dp = (D + 15) & ~15;
  return (size_t)(5 * 16 * (Hp + 4) + 16 * (Dp + 4) + 4 * Hp + 256) * sizeof(float);
}
static size_t dec_bwd_lds(int D, int H) {
  const int Hp = (H + 15) & ~15, Dp = (D + 15) & ~15, Gp = (3 * H + 15) & ~15;
  return (size_t)(16 * (3 * (Hp + 4) + (Dp + 4) + 2 * (Gp + 4)) + 4 * Hp + 256) * sizeof(float);
}

static size_t pack_fwd_total(int D, int H) {
  return pack_floats(H, 1, D) + 4 * pack_floats(H, 3, H) + pack_floats(D, 1, H);
}
static size_t pack_bwd_total(int D, int H) {
  return pack_floats(D, 1, H) + pack_floats(H, 1, D) + 4 * pack_floats(H, 1, 3 * H);
}

extern "C" size_t g2v_dec_rollout_fwd_workspace(int D, int H) { return pack_fwd_total(D, H) * sizeof(float); }

extern "C" int g2v_dec_rollout_fwd(const float* target, const float* h_init, const g2v_dec_weights* w,
                                   const g2v_dec_saved* s, const uint8_t* keep95, const uint8_t* keep_l0, float p_drop,
                                   int n_pre_poses, int conditioned, int training, int T, int B, int D, int H,
                                   void* workspace, size_t workspace_bytes, g2v_stream_t stream) {
  G2V_REQUIRE(target && h_init && w && s && keep95 && workspace, "null pointer");
  G2V_REQUIRE(T >= 2 && B > 0 && D > 0 && H > 0, "bad size");
  G2V_REQUIRE(s->y && s->u && s->h0 && s->h1 && s->bn_partial, "missing state buffer");
  G2V_REQUIRE(!training || (s->xin && s->a && s->gates0 && s->gates1 && s->bn_stats), "missing saved buffer");
  G2V_REQUIRE(p_drop >= 0.f && p_drop < 1.f, "bad dropout probability");
  if (workspace_bytes < g2v_dec_rollout_fwd_workspace(D, H)) {
    set_error("g2v_dec_rollout_fwd: workspace too small");
    return G2V_ERR_WORKSPACE;
  }
  const size_t lds = dec_fwd_lds(D, H);
  if (lds > 160 * 1024) {
    set_error("g2v_dec_rollout_fwd: D/H too large for LDS");
    return G2V_ERR_UNSUPPORTED;
  }
  hipStream_t st = (hipStream_t)stream;
  // ---- pack the weights into MFMA fragment order (one launch) ----
  float* p = (float*)workspace;
  DecPackF pk;
  PackBatch pb;
  pb.n = 6;
  pb.d[0] = PackDesc{w->w_pre, p, H, 1, 0, D, D, 0}; pk.pre = p; p += pack_floats(H, 1, D);
  pb.d[1] = PackDesc{w->w_ih0, p, H, 3, H, H, H, 0}; pk.ih0 = p; p += pack_floats(H, 3, H);
  pb.d[2] = PackDesc{w->w_hh0, p, H, 3, H, H, H, 0}; pk.hh0 = p; p += pack_floats(H, 3, H);
  pb.d[3] = PackDesc{w->w_ih1, p, H, 3, H, H, H, 0}; pk.ih1 = p; p += pack_floats(H, 3, H);
  pb.d[4] = PackDesc{w->w_hh1, p, H, 3, H, H, H, 0}; pk.hh1 = p; p += pack_floats(H, 3, H);
  pb.d[5] = PackDesc{w->w_out, p, D, 1, 0, H, H, 0}; pk.out = p; p += pack_floats(D, 1, H);
  launch_pack(pb, st);
  G2V_CHECK_LAUNCH();
  DecDims dm{T, B, D, H, p_drop, n_pre_poses, conditioned, training, cdiv(B, 16)};
  const bool fast = (H == 64);
  if (lds > 48 * 1024) {
    (void)hipFuncSetAttribute((const void*)dec_step_fwd_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    (void)hipFuncSetAttribute((const void*)dec_step_fwd_kernel<64>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  }
  for (int t = 0; t < T; ++t) {
    if (fast)
      hipLaunchKernelGGL(dec_step_fwd_kernel<64>, dim3(dm.nblk), dim3(256), lds, st, target, h_init, *w, pk, *s, keep95,
                         keep_l0, dm, t);
    else
      hipLaunchKernelGGL(dec_step_fwd_kernel<0>, dim3(dm.nblk), dim3(256), lds, st, target, h_init, *w, pk, *s, keep95,
                         keep_l0, dm, t);
  }
  G2V_CHECK_LAUNCH();
  if (training) {
    hipLaunchKernelGGL(bn_running_update_kernel, dim3(cdiv(H, 256)), dim3(256), 0, st, s->bn_stats,
                       w->bn_running_mean, w->bn_running_var, T - 1, H, B);
    G2V_CHECK_LAUNCH();
  }
  return G2V_OK;
}

extern "C" size_t g2v_dec_rollout_bwd_workspace(int D, int H) { return pack_bwd_total(D, H) * sizeof(float); }

extern "C" int g2v_dec_rollout_bwd(const g2v_dec_weights* w, const g2v_dec_saved* s, const g2v_dec_grads* g,
                                   const uint8_t* keep95, const uint8_t* keep_l0, float p_drop, int n_pre_poses,
                                   int conditioned, int T, int B, int D, int H, void* workspace, size_t workspace_bytes,
                                   g2v_stream_t stream) {
  G2V_REQUIRE(w && s && g && keep95 && workspace, "null pointer");
  G2V_REQUIRE(T >= 2 && B > 0 && D > 0 && H > 0, "bad size");
  G2V_REQUIRE(g->dy && g->du && g->dbn && g->dgi0 && g->dgh0 && g->dgi1 && g->dgh1 && g->dh_init && g->d_bn_w &&
                  g->d_bn_b && g->bn_bwd_partial,
              "missing gradient buffer");
  G2V_REQUIRE(s->u && s->a && s->h0 && s->h1 && s->gates0 && s->gates1 && s->bn_stats, "missing saved buffer");
  if (workspace_bytes < g2v_dec_rollout_bwd_workspace(D, H)) {
    set_error("g2v_dec_rollout_bwd: workspace too small");
    return G2V_ERR_WORKSPACE;
  }
  const size_t lds = dec_bwd_lds(D, H);
  if (lds > 160 * 1024) {
    set_error("g2v_dec_rollout_bwd: D/H too large for LDS");
    return G2V_ERR_UNSUPPORTED;
  }
  hipStream_t st = (hipStream_t)stream;
  float* p = (float*)workspace;
  DecTW tw;
  PackBatch pb;
  pb.n = 6;
  const int G = 3 * H;
  pb.d[0] = PackDesc{w->w_pre, p, D, 1, 0, H, D, 1}; tw.w_pre_t = p; p += pack_floats(D, 1, H);   // rows d, k = f: W_pre[f][d]
  pb.d[1] = PackDesc{w->w_out, p, H, 1, 0, D, H, 1}; tw.w_out_t = p; p += pack_floats(H, 1, D);   // rows f, k = d: W_out[d][f]
  pb.d[2] = PackDesc{w->w_ih0, p, H, 1, 0, G, H, 1}; tw.w_ih0_t = p; p += pack_floats(H, 1, G);   // rows k, contraction g: W[g][k]
  pb.d[3] = PackDesc{w->w_hh0, p, H, 1, 0, G, H, 1}; tw.w_hh0_t = p; p += pack_floats(H, 1, G);
  pb.d[4] = PackDesc{w->w_ih1, p, H, 1, 0, G, H, 1}; tw.w_ih1_t = p; p += pack_floats(H, 1, G);
  pb.d[5] = PackDesc{w->w_hh1, p, H, 1, 0, G, H, 1}; tw.w_hh1_t = p; p += pack_floats(H, 1, G);
  launch_pack(pb, st);
  G2V_CHECK_LAUNCH();
  if (lds > 48 * 1024) {
    (void)hipFuncSetAttribute((const void*)dec_step_bwd_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    (void)hipFuncSetAttribute((const void*)dec_step_bwd_kernel<64>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  }
  DecDims dm{T, B, D, H, p_drop, n_pre_poses, conditioned, 1, cdiv(B, 16)};
  const bool fast = (H == 64);
  for (int t = T - 1; t >= 0; --t) {
    if (fast)
      hipLaunchKernelGGL(dec_step_bwd_kernel<64>, dim3(dm.nblk), dim3(256), lds, st, *w, tw, *s, *g, keep95, keep_l0, dm, t);
    else
      hipLaunchKernelGGL(dec_step_bwd_kernel<0>, dim3(dm.nblk), dim3(256), lds, st, *w, tw, *s, *g, keep95, keep_l0, dm, t);
  }
  G2V_CHECK_LAUNCH();
  return G2V_OK;
}
