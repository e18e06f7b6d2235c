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

struct DecDims {
  int T, B, D, H;
  float p_drop;
  int n_pre, conditioned, training, nblk;
};

// ---- one GRU cell for the feature tiles of this wave ------------------------------------------------
// x-operand Xin [16][ldh] (layer input), Xh [16][ldh] (previous hidden).  Writes h_new to `Hout` LDS
// (after optional inter-layer dropout into `Xdrop`), to global h_out, and the gates.
__device__ __forceinline__ void gru_cell_fwd(const float* __restrict__ w_ih, const float* __restrict__ w_hh,
                                             const float* __restrict__ b_ih, const float* __restrict__ b_hh,
                                             const float* Xin, const float* Xh, int ldh, int H, int Hp,
                                             float* Hnext_lds,            // [16][ldh]: what the next stage consumes
                                             float* __restrict__ h_out,   // global (B,H) row block base (row b0)
                                             float* __restrict__ gates,   // global (B,4H) row block base or null
                                             const uint8_t* __restrict__ keep, float keep_scale,  // inter-layer dropout
                                             float* __restrict__ xdrop_out,  // global (B,H) dropped output or null
                                             int nrows, int lane, int wave) {
  const int i = lane & 15, q = lane >> 4;
  const bool wv1 = ptr_vec_ok(w_ih, H), wv2 = ptr_vec_ok(w_hh, H);
  const int ntile = Hp >> 4;
  for (int ft = wave; ft < ntile; ft += 4) {
    f32x4 ai[3], ah[3];
#pragma unroll
    for (int g = 0; g < 3; ++g) {
      ai[g] = (f32x4){0.f, 0.f, 0.f, 0.f};
      ah[g] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    const int nvalid = min(16, H - 16 * ft);
    wave_gemm<3>(ai, w_ih, (int64_t)H, wv1, 16 * ft, H, nvalid, H, Xin, ldh, lane);
    wave_gemm<3>(ah, w_hh, (int64_t)H, wv2, 16 * ft, H, nvalid, H, Xh, ldh, lane);
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

__global__ __launch_bounds__(256) void dec_step_fwd_kernel(const float* __restrict__ target,
                                                           const float* __restrict__ h_init, g2v_dec_weights w,
                                                           g2v_dec_saved sv, const uint8_t* __restrict__ keep95,
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

  // zero the operand tiles once (padding columns / rows must be 0 for the MFMA contractions)
  for (int e = tid; e < 5 * 16 * ldh + 16 * ldd; e += 256) smem[e] = 0.f;
  __syncthreads();

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
    // ---- (c) GRU layer 0 ---------------------------------------------------------------------------
    const bool drop = dm.training && keep_l0 && dm.p_drop > 0.f;
    gru_cell_fwd(w.w_ih0, w.w_hh0, w.b_ih0, w.b_hh0, Xa, Xh0, ldh, H, Hp, Xx1,
                 sv.h0 + ((int64_t)t * B + b0) * H,
                 sv.gates0 ? sv.gates0 + ((int64_t)(t - 1) * B + b0) * 4 * H : nullptr,
                 drop ? keep_l0 + ((int64_t)(t - 1) * B + b0) * H : nullptr, 1.0f / (1.0f - dm.p_drop),
                 (drop && sv.x1) ? sv.x1 + ((int64_t)(t - 1) * B + b0) * H : nullptr, nrows, lane, wave);
    __syncthreads();
    // ---- (d) GRU layer 1 ---------------------------------------------------------------------------
    gru_cell_fwd(w.w_ih1, w.w_hh1, w.b_ih1, w.b_hh1, Xx1, Xh1, ldh, H, Hp, Xh1n,
                 sv.h1 + ((int64_t)t * B + b0) * H,
                 sv.gates1 ? sv.gates1 + ((int64_t)(t - 1) * B + b0) * 4 * H : nullptr, nullptr, 1.0f, nullptr, nrows,
                 lane, wave);
    __syncthreads();
  }

  // ---- (e) y_t = out_layer(h1_t)  (t == 0: y_0 = target frame 0), next decoder input ---------------
  {
    const bool wv = ptr_vec_ok(w.w_out, H);
    const int ntile = Dp >> 4;
    for (int ft = wave; ft < ntile; ft += 4) {
      f32x4 acc[1] = {(f32x4){0.f, 0.f, 0.f, 0.f}};
      const int nvalid = min(16, D - 16 * ft);
      if (t > 0) wave_gemm<1>(acc, w.w_out, (int64_t)H, wv, 16 * ft, 16, nvalid, H, Xh1n, ldh, lane);
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
  // ---- (f) u_{t+1} = pre_linear.0(xin_{t+1}) and per-block BN partial sums of (u - b) ----------------
  {
    const bool wv = ptr_vec_ok(w.w_pre, D);
    const int ntile = Hp >> 4;
    float* part = sv.bn_partial + ((int64_t)(t & 1) * dm.nblk + blockIdx.x) * 2 * H;
    for (int ft = wave; ft < ntile; ft += 4) {
      f32x4 acc[1] = {(f32x4){0.f, 0.f, 0.f, 0.f}};
      const int nvalid = min(16, H - 16 * ft);
      wave_gemm<1>(acc, w.w_pre, (int64_t)D, wv, 16 * ft, 16, nvalid, D, Xy, ldd, lane);
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
struct DecTW {   // transposed weights (contraction index contiguous)
  const float* w_pre_t;   // (D,H)   = W_pre^T
  const float* w_out_t;   // (H,D)   = W_out^T
  const float* w_ih0_t; const float* w_hh0_t; const float* w_ih1_t; const float* w_hh1_t;   // (H,3H) each
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
    const bool wv = ptr_vec_ok(tw.w_pre_t, H);
    for (int ft = wave; ft < ntd; ft += 4) {
      f32x4 acc[1] = {(f32x4){0.f, 0.f, 0.f, 0.f}};
      const int nvalid = min(16, D - 16 * ft);
      if (feedback) wave_gemm<1>(acc, tw.w_pre_t, (int64_t)H, wv, 16 * ft, 16, nvalid, H, Xdu, ldh, lane);
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
    const bool wv = ptr_vec_ok(tw.w_out_t, D);
    for (int ft = wave; ft < nth; ft += 4) {
      f32x4 acc[1] = {(f32x4){0.f, 0.f, 0.f, 0.f}};
      const int nvalid = min(16, H - 16 * ft);
      wave_gemm<1>(acc, tw.w_out_t, (int64_t)D, wv, 16 * ft, 16, nvalid, D, Xdy, ldd, lane);
      gru_cell_bwd_tile(acc[0], carry1, 1.0f, nullptr, sv.gates1 + ((int64_t)(t - 1) * B + b0) * 4 * H,
                        sv.h1 + ((int64_t)(t - 1) * B + b0) * H, gr.dgi1 + ((int64_t)(t - 1) * B + b0) * G,
                        gr.dgh1 + ((int64_t)(t - 1) * B + b0) * G, Gi, Gh, ldg, Dd, ldh, H, ft, nrows, lane);
    }
  }
  __syncthreads();
  // ---- carry1' = dh1*z + dgh1 W_hh1 ;  dx1 = dgi1 W_ih1 -> dh0 (inter-layer dropout bwd) ----------------
  {
    const bool wv1 = ptr_vec_ok(tw.w_hh1_t, G), wv2 = ptr_vec_ok(tw.w_ih1_t, G);
    for (int ft = wave; ft < nth; ft += 4) {
      f32x4 a1[1] = {(f32x4){0.f, 0.f, 0.f, 0.f}}, a2[1] = {(f32x4){0.f, 0.f, 0.f, 0.f}};
      const int nvalid = min(16, H - 16 * ft);
      wave_gemm<1>(a1, tw.w_hh1_t, (int64_t)G, wv1, 16 * ft, 16, nvalid, G, Gh, ldg, lane);
      wave_gemm<1>(a2, tw.w_ih1_t, (int64_t)G, wv2, 16 * ft, 16, nvalid, G, Gi, ldg, lane);
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
    const bool wv1 = ptr_vec_ok(tw.w_hh0_t, G), wv2 = ptr_vec_ok(tw.w_ih0_t, G);
    const float* stats = sv.bn_stats + (int64_t)(t - 1) * 2 * H;
    float* part = gr.bn_bwd_partial + ((int64_t)(t & 1) * dm.nblk + blockIdx.x) * 2 * H;
    for (int ft = wave; ft < nth; ft += 4) {
      f32x4 a1[1] = {(f32x4){0.f, 0.f, 0.f, 0.f}}, a2[1] = {(f32x4){0.f, 0.f, 0.f, 0.f}};
      const int nvalid = min(16, H - 16 * ft);
      wave_gemm<1>(a1, tw.w_hh0_t, (int64_t)G, wv1, 16 * ft, 16, nvalid, G, Gh, ldg, lane);
      wave_gemm<1>(a2, tw.w_ih0_t, (int64_t)G, wv2, 16 * ft, 16, nvalid, G, Gi, ldg, lane);
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

extern "C" int g2v_dec_rollout_blocks(int B) { return B > 0 ? cdiv(B, 16) : 0; }

static size_t dec_fwd_lds(int D, int H) {
  const int Hp = (H + 15) & ~15, Dp = (D + 15) & ~15;
  return (size_t)(5 * 16 * (Hp + 4) + 16 * (Dp + 4) + 4 * Hp + 256) * sizeof(float);
}
static size_t dec_bwd_lds(int D, int H) {
  const int Hp = (H + 15) & ~15, Dp = (D + 15) & ~15, Gp = (3 * H + 15) & ~15;
  return (size_t)(16 * (3 * (Hp + 4) + (Dp + 4) + 2 * (Gp + 4)) + 4 * Hp + 256) * sizeof(float);
}

extern "C" int g2v_dec_rollout_fwd(const float* target, const float* h_init, const g2v_dec_weights* w,
                                   const g2v_dec_saved* s, const uint8_t* keep95, const uint8_t* keep_l0, float p_drop,
                                   int n_pre_poses, int conditioned, int training, int T, int B, int D, int H,
                                   g2v_stream_t stream) {
  G2V_REQUIRE(target && h_init && w && s && keep95, "null pointer");
  G2V_REQUIRE(T >= 2 && B > 0 && D > 0 && H > 0, "bad size");
  G2V_REQUIRE(s->y && s->u && s->h0 && s->h1 && s->bn_partial, "missing state buffer");
  G2V_REQUIRE(!training || (s->xin && s->a && s->gates0 && s->gates1 && s->bn_stats), "missing saved buffer");
  G2V_REQUIRE(p_drop >= 0.f && p_drop < 1.f, "bad dropout probability");
  const size_t lds = dec_fwd_lds(D, H);
  if (lds > 160 * 1024) {
    set_error("g2v_dec_rollout_fwd: D/H too large for LDS");
    return G2V_ERR_UNSUPPORTED;
  }
  if (lds > 48 * 1024)
    (void)hipFuncSetAttribute((const void*)dec_step_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  DecDims dm{T, B, D, H, p_drop, n_pre_poses, conditioned, training, cdiv(B, 16)};
  for (int t = 0; t < T; ++t) {
    hipLaunchKernelGGL(dec_step_fwd_kernel, dim3(dm.nblk), dim3(256), lds, (hipStream_t)stream, target, h_init, *w, *s,
                       keep95, keep_l0, dm, t);
  }
  G2V_CHECK_LAUNCH();
  if (training) {
    hipLaunchKernelGGL(bn_running_update_kernel, dim3(cdiv(H, 256)), dim3(256), 0, (hipStream_t)stream, s->bn_stats,
                       w->bn_running_mean, w->bn_running_var, T - 1, H, B);
    G2V_CHECK_LAUNCH();
  }
  return G2V_OK;
}

extern "C" size_t g2v_dec_rollout_bwd_workspace(int D, int H) {
  return (size_t)(2 * D * H + 4 * 3 * H * H) * sizeof(float);
}

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
  float* ws = (float*)workspace;
  DecTW tw;
  float* p = ws;
  launch_transpose(w->w_pre, p, H, D, st); tw.w_pre_t = p; p += (size_t)D * H;       // (H,D) -> (D,H)
  launch_transpose(w->w_out, p, D, H, st); tw.w_out_t = p; p += (size_t)D * H;       // (D,H) -> (H,D)
  launch_transpose(w->w_ih0, p, 3 * H, H, st); tw.w_ih0_t = p; p += (size_t)3 * H * H;
  launch_transpose(w->w_hh0, p, 3 * H, H, st); tw.w_hh0_t = p; p += (size_t)3 * H * H;
  launch_transpose(w->w_ih1, p, 3 * H, H, st); tw.w_ih1_t = p; p += (size_t)3 * H * H;
  launch_transpose(w->w_hh1, p, 3 * H, H, st); tw.w_hh1_t = p; p += (size_t)3 * H * H;
  G2V_CHECK_LAUNCH();
  if (lds > 48 * 1024)
    (void)hipFuncSetAttribute((const void*)dec_step_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  DecDims dm{T, B, D, H, p_drop, n_pre_poses, conditioned, 1, cdiv(B, 16)};
  for (int t = T - 1; t >= 0; --t) {
    hipLaunchKernelGGL(dec_step_bwd_kernel, dim3(dm.nblk), dim3(256), lds, st, *w, tw, *s, *g, keep95, keep_l0, dm, t);
  }
  G2V_CHECK_LAUNCH();
  return G2V_OK;
}
