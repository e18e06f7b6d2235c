// gru.hip -- full-sequence GRU direction, forward and BPTT, as persistent per-batch-tile kernels.
//
// Replaces one direction of one layer of nn.GRU in EncoderRNN
// (model/Autoencoder_VQVAE_model.py:94; model/text2embedding_model.py:131 with packed lengths).
//
// The recurrence is independent across batch rows, so one 256-thread workgroup owns 16 rows for the
// WHOLE sequence: h lives in LDS (double-buffered), the loop over T is inside the kernel, and the only
// per-step global traffic is gi (read), hs / gates (write) plus W_hh fragments from L2.  The hidden
// features are split over the 4 waves in 16-wide MFMA tiles; each tile computes its r,z,n gate
// pre-activations with three v_mfma_f32_16x16x4_f32 accumulators so that the whole gate math for a
// (row, feature) happens in one lane.
#include "common.hpp"

namespace g2v {

__global__ __launch_bounds__(256) void gru_seq_fwd_kernel(const float* __restrict__ gi, const float* __restrict__ w_hh,
                                                          const float* __restrict__ b_hh, const float* __restrict__ h0,
                                                          const int32_t* __restrict__ lengths, int reverse,
                                                          float* __restrict__ hs, int64_t hs_ld, float* __restrict__ h_n,
                                                          float* __restrict__ gates, int T, int B, int H) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int Hp = (H + 15) & ~15, ldx = Hp + 4;
  float* hbuf0 = smem;
  float* hbuf1 = smem + 16 * ldx;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b0 = blockIdx.x * 16;
  const int nrows = min(16, B - b0);
  const int i = lane & 15, q = lane >> 4;
  const int b = b0 + i;
  const bool rvalid = i < nrows;
  const int len = (lengths && rvalid) ? lengths[b] : T;
  const bool wvec = ptr_vec_ok(w_hh, H);
  const bool gvec = ((H & 3) == 0) && ((reinterpret_cast<uintptr_t>(gi) & 15) == 0);

  for (int e = tid; e < 16 * ldx; e += 256) {
    const int r = e / ldx, k = e - r * ldx;
    float v = 0.f;
    if (h0 && r < nrows && k < H) v = h0[(int64_t)(b0 + r) * H + k];
    hbuf0[e] = v;
    hbuf1[e] = 0.f;
  }
  __syncthreads();
  float* cur = hbuf0;
  float* nxt = hbuf1;
  const int ntile = Hp >> 4;
  for (int s = 0; s < T; ++s) {
    const int t = reverse ? (T - 1 - s) : s;
    const bool valid = rvalid && (t < len);
    for (int ft = wave; ft < ntile; ft += 4) {
      f32x4 acc[3];
#pragma unroll
      for (int g = 0; g < 3; ++g) acc[g] = (f32x4){0.f, 0.f, 0.f, 0.f};
      const int nvalid = min(16, H - 16 * ft);
      wave_gemm<3>(acc, w_hh, (int64_t)H, wvec, 16 * ft, H, nvalid, H, cur, ldx, lane);
      const int f0 = 16 * ft + 4 * q;
      const int64_t row = (int64_t)t * B + b;
      float hp[4], hn[4], gr[4], gz[4], gn[4], gh[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) hp[r] = cur[i * ldx + f0 + r];
      if (valid) {
        const float* gir = gi + row * 3 * H;
        float ir[4], iz[4], in_[4];
        if (gvec && f0 + 3 < H) {
          const float4 a = *reinterpret_cast<const float4*>(gir + f0);
          const float4 c = *reinterpret_cast<const float4*>(gir + H + f0);
          const float4 d = *reinterpret_cast<const float4*>(gir + 2 * H + f0);
          ir[0] = a.x; ir[1] = a.y; ir[2] = a.z; ir[3] = a.w;
          iz[0] = c.x; iz[1] = c.y; iz[2] = c.z; iz[3] = c.w;
          in_[0] = d.x; in_[1] = d.y; in_[2] = d.z; in_[3] = d.w;
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const bool ok = f0 + r < H;
            ir[r] = ok ? gir[f0 + r] : 0.f;
            iz[r] = ok ? gir[H + f0 + r] : 0.f;
            in_[r] = ok ? gir[2 * H + f0 + r] : 0.f;
          }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int f = f0 + r;
          if (f < H) {
            const float rr = sigmoidf_(ir[r] + (acc[0][r] + b_hh[f]));
            const float zz = sigmoidf_(iz[r] + (acc[1][r] + b_hh[H + f]));
            const float ghn = acc[2][r] + b_hh[2 * H + f];
            const float nn = tanhf_(in_[r] + rr * ghn);
            hn[r] = (1.0f - zz) * nn + zz * hp[r];
            gr[r] = rr; gz[r] = zz; gn[r] = nn; gh[r] = ghn;
          } else {
            hn[r] = 0.f; gr[r] = gz[r] = gn[r] = gh[r] = 0.f;
          }
        }
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) { hn[r] = hp[r]; gr[r] = gz[r] = gn[r] = gh[r] = 0.f; }
      }
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (f0 + r < H) nxt[i * ldx + f0 + r] = hn[r];
      if (rvalid) {
        float* ho = hs + row * hs_ld;
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (f0 + r < H) ho[f0 + r] = valid ? hn[r] : 0.f;   // padded positions of the output are zero
        if (gates) {
          float* go = gates + row * 4 * H;
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (f0 + r < H) {
              go[f0 + r] = gr[r];
              go[H + f0 + r] = gz[r];
              go[2 * H + f0 + r] = gn[r];
              go[3 * H + f0 + r] = gh[r];
            }
        }
      }
    }
    __syncthreads();
    float* tmp = cur; cur = nxt; nxt = tmp;
  }
  if (h_n)
    for (int e = tid; e < 16 * H; e += 256) {
      const int r = e / H, k = e - r * H;
      if (r < nrows) h_n[(int64_t)(b0 + r) * H + k] = cur[r * ldx + k];
    }
}

// BPTT.  w_hh_t = W_hh^T, (H, 3H) row-major (so that dh_prev = dgh W_hh is again "weights contiguous along
// the contraction").  LDS: Gs [16][3H padded] (dgh tile = MFMA B operand), dhs [16][H padded] (carry).
__global__ __launch_bounds__(256) void gru_seq_bwd_kernel(const float* __restrict__ d_hs, int64_t d_hs_ld,
                                                          const float* __restrict__ d_hn, const float* __restrict__ hs,
                                                          int64_t hs_ld, const float* __restrict__ h0,
                                                          const float* __restrict__ gates,
                                                          const float* __restrict__ w_hh_t,
                                                          const int32_t* __restrict__ lengths, int reverse,
                                                          float* __restrict__ dgi, float* __restrict__ dgh,
                                                          float* __restrict__ dh0, int T, int B, int H) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int Hp = (H + 15) & ~15, G = 3 * H, Gp = (G + 15) & ~15, ldg = Gp + 4, ldh = Hp + 4;
  float* Gs = smem;              // [16][ldg]
  float* dhs = smem + 16 * ldg;  // [16][ldh]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b0 = blockIdx.x * 16;
  const int nrows = min(16, B - b0);
  const int i = lane & 15, q = lane >> 4;
  const bool wvec = ptr_vec_ok(w_hh_t, G);

  for (int e = tid; e < 16 * ldg; e += 256) Gs[e] = 0.f;
  for (int e = tid; e < 16 * ldh; e += 256) {
    const int r = e / ldh, k = e - r * ldh;
    dhs[e] = (d_hn && r < nrows && k < H) ? d_hn[(int64_t)(b0 + r) * H + k] : 0.f;
  }
  __syncthreads();
  const int ntile = Hp >> 4;
  for (int s = T - 1; s >= 0; --s) {
    const int t = reverse ? (T - 1 - s) : s;           // time index processed at forward iteration s
    const int tprev = reverse ? t + 1 : t - 1;         // where h_prev of this step was written
    // phase 1: gate gradients, element-wise
    for (int e = tid; e < 16 * H; e += 256) {
      const int r = e / H, f = e - r * H;
      if (r >= nrows) continue;
      const int b = b0 + r;
      const int len = lengths ? lengths[b] : T;
      const int64_t row = (int64_t)t * B + b;
      float dh = dhs[r * ldh + f];
      float g_r = 0.f, g_z = 0.f, g_n = 0.f, g_hn = 0.f, direct = dh;
      if (t < len) {
        if (d_hs) dh += d_hs[row * d_hs_ld + f];
        const float* go = gates + row * 4 * H;
        const float rr = go[f], zz = go[H + f], nn = go[2 * H + f], ghn = go[3 * H + f];
        float hp;
        if (s == 0) hp = h0 ? h0[(int64_t)b * H + f] : 0.f;
        else if (tprev >= len) hp = h0 ? h0[(int64_t)b * H + f] : 0.f;   // reverse dir: first valid step
        else hp = hs[((int64_t)tprev * B + b) * hs_ld + f];
        const float dn = dh * (1.0f - zz);
        const float dz = dh * (hp - nn);
        const float dnp = dn * (1.0f - nn * nn);
        g_n = dnp;
        g_hn = dnp * rr;
        g_r = dnp * ghn * rr * (1.0f - rr);
        g_z = dz * zz * (1.0f - zz);
        direct = dh * zz;
      }
      dgi[row * G + f] = g_r; dgi[row * G + H + f] = g_z; dgi[row * G + 2 * H + f] = g_n;
      dgh[row * G + f] = g_r; dgh[row * G + H + f] = g_z; dgh[row * G + 2 * H + f] = g_hn;
      Gs[r * ldg + f] = g_r; Gs[r * ldg + H + f] = g_z; Gs[r * ldg + 2 * H + f] = g_hn;
      dhs[r * ldh + f] = direct;
    }
    __syncthreads();
    // phase 2: dh_prev = direct + dgh W_hh
    for (int ft = wave; ft < ntile; ft += 4) {
      f32x4 acc[1] = {(f32x4){0.f, 0.f, 0.f, 0.f}};
      const int nvalid = min(16, H - 16 * ft);
      wave_gemm<1>(acc, w_hh_t, (int64_t)G, wvec, 16 * ft, 16, nvalid, G, Gs, ldg, lane);
      const int f0 = 16 * ft + 4 * q;
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (f0 + r < H) dhs[i * ldh + f0 + r] += acc[0][r];
    }
    __syncthreads();
  }
  if (dh0)
    for (int e = tid; e < 16 * H; e += 256) {
      const int r = e / H, k = e - r * H;
      if (r < nrows) dh0[(int64_t)(b0 + r) * H + k] = dhs[r * ldh + k];
    }
}

}  // namespace g2v

using namespace g2v;

extern "C" int g2v_gru_seq_fwd(const float* gi, const float* w_hh, const float* b_hh, const float* h0,
                               const int32_t* lengths, int reverse, float* hs, int64_t hs_ld, float* h_n, float* gates,
                               int T, int B, int H, g2v_stream_t stream) {
  G2V_REQUIRE(gi && w_hh && b_hh && hs, "null pointer");
  G2V_REQUIRE(T > 0 && B > 0 && H > 0 && hs_ld >= H, "bad size");
  const int Hp = (H + 15) & ~15;
  const size_t lds = (size_t)2 * 16 * (Hp + 4) * sizeof(float);
  G2V_REQUIRE(lds <= 160 * 1024, "hidden size too large for LDS");
  if (lds > 48 * 1024)
    (void)hipFuncSetAttribute((const void*)gru_seq_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL(gru_seq_fwd_kernel, dim3(cdiv(B, 16)), dim3(256), lds, (hipStream_t)stream, gi, w_hh, b_hh, h0,
                     lengths, reverse, hs, hs_ld, h_n, gates, T, B, H);
  G2V_CHECK_LAUNCH();
  return G2V_OK;
}

extern "C" size_t g2v_gru_seq_bwd_workspace(int H) { return (size_t)3 * H * H * sizeof(float); }

extern "C" int g2v_gru_seq_bwd(const float* d_hs, int64_t d_hs_ld, const float* d_hn, const float* hs, int64_t hs_ld,
                               const float* h0, const float* gates, const float* w_hh, const int32_t* lengths,
                               int reverse, float* dgi, float* dgh, float* dh0, int T, int B, int H,
                               void* workspace, size_t workspace_bytes, g2v_stream_t stream) {
  G2V_REQUIRE(hs && gates && w_hh && dgi && dgh && workspace, "null pointer");
  G2V_REQUIRE(T > 0 && B > 0 && H > 0, "bad size");
  if (workspace_bytes < g2v_gru_seq_bwd_workspace(H)) {
    set_error("g2v_gru_seq_bwd: workspace too small");
    return G2V_ERR_WORKSPACE;
  }
  float* wt = (float*)workspace;
  launch_transpose(w_hh, wt, 3 * H, H, (hipStream_t)stream);  // (3H,H) -> (H,3H)
  G2V_CHECK_LAUNCH();
  const int Hp = (H + 15) & ~15, Gp = (3 * H + 15) & ~15;
  const size_t lds = (size_t)16 * ((Gp + 4) + (Hp + 4)) * sizeof(float);
  G2V_REQUIRE(lds <= 160 * 1024, "hidden size too large for LDS");
  if (lds > 48 * 1024)
    (void)hipFuncSetAttribute((const void*)gru_seq_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL(gru_seq_bwd_kernel, dim3(cdiv(B, 16)), dim3(256), lds, (hipStream_t)stream, d_hs, d_hs_ld, d_hn,
                     hs, hs_ld, h0, gates, wt, lengths, reverse, dgi, dgh, dh0, T, B, H);
  G2V_CHECK_LAUNCH();
  return G2V_OK;
}
