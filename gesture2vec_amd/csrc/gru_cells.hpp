// gru_cells.hpp -- one GRU cell (forward / backward) for the feature tiles of a wave, on 16 batch rows staged in LDS and weights
// streamed from L2 as packed MFMA fragments: shared by the per-step kernels of the pose decoder (dec_rollout.hip) and of the
// text -> gesture-code decoder (t2e_rollout.hip).  Moved here from dec_rollout.hip in round 5 (no change of arithmetic).
#pragma once
#include "common.hpp"

namespace g2v {

// Gate math + stores of one GRU cell for the 4 consecutive features [f0, f0+4) of row i held by this lane.
__device__ __forceinline__ void gru_cell_fwd_epilogue(const f32x4 (&ai)[3], const f32x4 (&ah)[3], const float4 (&bi)[3],
                                                      const float4 (&bh)[3], uint32_t kp, bool keep, float keep_scale,
                                                      const float* Xh, int ldh, int H, float* Hnext_lds,
                                                      float* __restrict__ h_out, float* __restrict__ gates,
                                                      float* __restrict__ xdrop_out, int nrows, int i, int f0,
                                                      bool wt = false) {
  const float4 hp4 = *reinterpret_cast<const float4*>(Xh + i * ldh + f0);
  const float hp[4] = {hp4.x, hp4.y, hp4.z, hp4.w};
  const float bir[4] = {bi[0].x, bi[0].y, bi[0].z, bi[0].w}, biz[4] = {bi[1].x, bi[1].y, bi[1].z, bi[1].w},
              bin[4] = {bi[2].x, bi[2].y, bi[2].z, bi[2].w};
  const float bhr[4] = {bh[0].x, bh[0].y, bh[0].z, bh[0].w}, bhz[4] = {bh[1].x, bh[1].y, bh[1].z, bh[1].w},
              bhn[4] = {bh[2].x, bh[2].y, bh[2].z, bh[2].w};
  float hn[4], xd[4], gr_[4], gz_[4], gn_[4], gh_[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const float rr = sigmoidf_((ai[0][r] + bir[r]) + (ah[0][r] + bhr[r]));
    const float zz = sigmoidf_((ai[1][r] + biz[r]) + (ah[1][r] + bhz[r]));
    const float ghn = ah[2][r] + bhn[r];
    const float nn = tanhf_((ai[2][r] + bin[r]) + rr * ghn);
    hn[r] = (1.0f - zz) * nn + zz * hp[r];
    xd[r] = keep ? (((kp >> (8 * r)) & 0xffu) ? hn[r] * keep_scale : 0.f) : hn[r];
    gr_[r] = rr; gz_[r] = zz; gn_[r] = nn; gh_[r] = ghn;
  }
  *reinterpret_cast<float4*>(Hnext_lds + i * ldh + f0) = make_float4(xd[0], xd[1], xd[2], xd[3]);
  if (i < nrows) {
    *reinterpret_cast<float4*>(h_out + (int64_t)i * H + f0) = make_float4(hn[0], hn[1], hn[2], hn[3]);
    if (xdrop_out) st4(xdrop_out + (int64_t)i * H + f0, make_float4(xd[0], xd[1], xd[2], xd[3]), wt);
    if (gates) {
      float* go = gates + (int64_t)i * 4 * H + f0;
      st4(go, make_float4(gr_[0], gr_[1], gr_[2], gr_[3]), wt);
      st4(go + H, make_float4(gz_[0], gz_[1], gz_[2], gz_[3]), wt);
      st4(go + 2 * H, make_float4(gn_[0], gn_[1], gn_[2], gn_[3]), wt);
      st4(go + 3 * H, make_float4(gh_[0], gh_[1], gh_[2], gh_[3]), wt);
    }
  }
}

// ---- one GRU cell for the feature tiles of this wave ------------------------------------------------
// x-operand Xin [16][ldh] (layer input), Xh [16][ldh] (previous hidden).  Writes h_new (after optional
// inter-layer dropout) to `Hnext_lds`, h_new to global h_out, and the gates.  When H % 4 == 0 every global /
// LDS access of the epilogue is a 16-byte vector (4 consecutive features live in one lane).
template <int KSH_T>
__device__ __forceinline__ void gru_cell_fwd(const float* __restrict__ p_ih, const float* __restrict__ p_hh,
                                             const float* __restrict__ b_ih, const float* __restrict__ b_hh,
                                             const float* Xin, const float* Xh, int ldh, int H, int Hp,
                                             float* Hnext_lds,            // [16][ldh]: what the next stage consumes
                                             float* __restrict__ h_out,   // global (B,H) row block base (row b0)
                                             float* __restrict__ gates,   // global (B,4H) row block base or null
                                             const uint8_t* __restrict__ keep, float keep_scale,  // inter-layer dropout
                                             float* __restrict__ xdrop_out,  // global (B,H) dropped output or null
                                             int nrows, int lane, int wave, int nwaves = 4) {
  const int i = lane & 15, q = lane >> 4;
  const int ntile = Hp >> 4, KS = Hp >> 4;
  const bool hvec = (H & 3) == 0;
  for (int ft = wave; ft < ntile; ft += nwaves) {
    const int f0 = 16 * ft + 4 * q;
    const bool vec = hvec && (f0 + 3 < H);
    // biases / keep flags first: independent of the MFMAs below, their latency hides behind them
    float4 bi[3], bh[3];
    uint32_t kp = 0x01010101u;
    if (vec) {
#pragma unroll
      for (int g = 0; g < 3; ++g) {
        bi[g] = *reinterpret_cast<const float4*>(b_ih + g * H + f0);
        bh[g] = *reinterpret_cast<const float4*>(b_hh + g * H + f0);
      }
      if (keep && i < nrows) kp = *reinterpret_cast<const uint32_t*>(keep + (int64_t)i * H + f0);
    }
    f32x4 ai[3], ah[3];
#pragma unroll
    for (int g = 0; g < 3; ++g) {
      ai[g] = (f32x4){0.f, 0.f, 0.f, 0.f};
      ah[g] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    if constexpr (KSH_T > 0) {
      wave_gemm_p2<3, KSH_T>(ai, p_ih, Xin, ah, p_hh, Xh, ft, ntile, ldh, lane);
    } else {
      wave_gemm_p_dual<3>(ai, p_ih, Xin, ah, p_hh, Xh, KS, ft, ntile, ldh, lane);
    }
    if (vec) {
      gru_cell_fwd_epilogue(ai, ah, bi, bh, kp, keep != nullptr, keep_scale, Xh, ldh, H, Hnext_lds, h_out, gates, xdrop_out,
                            nrows, i, f0);
    } else {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int f = f0 + r;
        if (f >= H) continue;
        const float hp = Xh[i * ldh + f];
        const float rr = sigmoidf_((ai[0][r] + b_ih[f]) + (ah[0][r] + b_hh[f]));
        const float zz = sigmoidf_((ai[1][r] + b_ih[H + f]) + (ah[1][r] + b_hh[H + f]));
        const float ghn = ah[2][r] + b_hh[2 * H + f];
        const float nn = tanhf_((ai[2][r] + b_ih[2 * H + f]) + rr * ghn);
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
}

// GRU cell backward for one 16-feature tile of this wave.
//   dh(row,f) = keep ? acc * extra_scale * keep : acc   (+ carry from the later time step)
// writes dgi / dgh (global), Gi / Gh tiles (LDS, MFMA B operands for the next contractions) and
// direct = dh * z into Dd (LDS).  H % 4 == 0: all accesses are 16-byte vectors.
// what gru_cell_bwd_tile reads from memory for one lane (the 16-byte path), so that a caller can request it ahead of the
// products / the barrier in front of the cell (one memory round trip per tile stood in front of every cell epilogue)
struct CellBwdIn {
  float4 r4, z4, n4, h4, hp4, c4;
  uint32_t kp;
};
__device__ __forceinline__ void cell_bwd_prefetch(CellBwdIn& in, const float* __restrict__ carry, const uint8_t* __restrict__ keep,
                                                  const float* __restrict__ gates, const float* __restrict__ hprev, int H, int ft,
                                                  int nrows, int lane, int carry_ld = 0) {
  const int cld = carry_ld > 0 ? carry_ld : H;      // (the single-launch BPTT of t2e_rollout.hip keeps its carries in LDS)
  const int i = lane & 15, q = lane >> 4;
  const int f0 = 16 * ft + 4 * q;
  in.c4 = make_float4(0.f, 0.f, 0.f, 0.f);
  in.r4 = in.z4 = in.n4 = in.h4 = in.hp4 = in.c4;
  in.kp = 0x01010101u;
  if (((H & 3) == 0) && f0 + 3 < H && i < nrows) {
    const float* go = gates + (int64_t)i * 4 * H + f0;
    in.r4 = *reinterpret_cast<const float4*>(go); in.z4 = *reinterpret_cast<const float4*>(go + H);
    in.n4 = *reinterpret_cast<const float4*>(go + 2 * H); in.h4 = *reinterpret_cast<const float4*>(go + 3 * H);
    in.hp4 = *reinterpret_cast<const float4*>(hprev + (int64_t)i * H + f0);
    if (carry) in.c4 = *reinterpret_cast<const float4*>(carry + (int64_t)i * cld + f0);
    if (keep) in.kp = *reinterpret_cast<const uint32_t*>(keep + (int64_t)i * H + f0);
  }
}
__device__ __forceinline__ void gru_cell_bwd_tile(const f32x4& acc, const float* __restrict__ carry, float extra_scale,
                                                  const uint8_t* __restrict__ keep,   // applied to acc (inter-layer dropout bwd)
                                                  const float* __restrict__ gates, const float* __restrict__ hprev,
                                                  float* __restrict__ dgi, float* __restrict__ dgh, float* Gi, float* Gh,
                                                  int ldg, float* Dd, int ldh, int H, int ft, int nrows, int lane,
                                                  bool wt, bool use_pre, const CellBwdIn& pre, int carry_ld = 0) {
  const int cld = carry_ld > 0 ? carry_ld : H;
  const int i = lane & 15, q = lane >> 4;
  const int f0 = 16 * ft + 4 * q;
  const int G = 3 * H;
  if (((H & 3) == 0) && f0 + 3 < H) {
    float g_r[4] = {0.f, 0.f, 0.f, 0.f}, g_z[4] = {0.f, 0.f, 0.f, 0.f}, g_n[4] = {0.f, 0.f, 0.f, 0.f},
          g_hn[4] = {0.f, 0.f, 0.f, 0.f}, direct[4] = {0.f, 0.f, 0.f, 0.f};
    if (i < nrows) {
      const float* go = gates + (int64_t)i * 4 * H + f0;
      float4 r4, z4, n4, h4, hp4, c4 = make_float4(0.f, 0.f, 0.f, 0.f);
      uint32_t kp = 0x01010101u;
      if (use_pre) {
        r4 = pre.r4; z4 = pre.z4; n4 = pre.n4; h4 = pre.h4; hp4 = pre.hp4; c4 = pre.c4; kp = pre.kp;
      } else {
        r4 = *reinterpret_cast<const float4*>(go); z4 = *reinterpret_cast<const float4*>(go + H);
        n4 = *reinterpret_cast<const float4*>(go + 2 * H); h4 = *reinterpret_cast<const float4*>(go + 3 * H);
        hp4 = *reinterpret_cast<const float4*>(hprev + (int64_t)i * H + f0);
        if (carry) c4 = *reinterpret_cast<const float4*>(carry + (int64_t)i * cld + f0);
        if (keep) kp = *reinterpret_cast<const uint32_t*>(keep + (int64_t)i * H + f0);
      }
      const float rr[4] = {r4.x, r4.y, r4.z, r4.w}, zz[4] = {z4.x, z4.y, z4.z, z4.w}, nn[4] = {n4.x, n4.y, n4.z, n4.w},
                  gh[4] = {h4.x, h4.y, h4.z, h4.w}, hp[4] = {hp4.x, hp4.y, hp4.z, hp4.w}, cc[4] = {c4.x, c4.y, c4.z, c4.w};
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float dh = acc[r];
        if (keep) dh = ((kp >> (8 * r)) & 0xffu) ? dh * extra_scale : 0.f;
        dh += cc[r];
        const float dn = dh * (1.0f - zz[r]);
        const float dz = dh * (hp[r] - nn[r]);
        const float dnp = dn * (1.0f - nn[r] * nn[r]);
        g_n[r] = dnp;
        g_hn[r] = dnp * rr[r];
        g_r[r] = dnp * gh[r] * rr[r] * (1.0f - rr[r]);
        g_z[r] = dz * zz[r] * (1.0f - zz[r]);
        direct[r] = dh * zz[r];
      }
      float* o1 = dgi + (int64_t)i * G + f0;
      float* o2 = dgh + (int64_t)i * G + f0;
      const float4 vr = make_float4(g_r[0], g_r[1], g_r[2], g_r[3]), vz = make_float4(g_z[0], g_z[1], g_z[2], g_z[3]),
                   vn = make_float4(g_n[0], g_n[1], g_n[2], g_n[3]), vh = make_float4(g_hn[0], g_hn[1], g_hn[2], g_hn[3]);
      st4(o1, vr, wt); st4(o1 + H, vz, wt); st4(o1 + 2 * H, vn, wt);
      st4(o2, vr, wt); st4(o2 + H, vz, wt); st4(o2 + 2 * H, vh, wt);
    }
    const float4 vr = make_float4(g_r[0], g_r[1], g_r[2], g_r[3]), vz = make_float4(g_z[0], g_z[1], g_z[2], g_z[3]),
                 vn = make_float4(g_n[0], g_n[1], g_n[2], g_n[3]), vh = make_float4(g_hn[0], g_hn[1], g_hn[2], g_hn[3]);
    *reinterpret_cast<float4*>(Gi + i * ldg + f0) = vr;
    *reinterpret_cast<float4*>(Gi + i * ldg + H + f0) = vz;
    *reinterpret_cast<float4*>(Gi + i * ldg + 2 * H + f0) = vn;
    *reinterpret_cast<float4*>(Gh + i * ldg + f0) = vr;
    *reinterpret_cast<float4*>(Gh + i * ldg + H + f0) = vz;
    *reinterpret_cast<float4*>(Gh + i * ldg + 2 * H + f0) = vh;
    *reinterpret_cast<float4*>(Dd + i * ldh + f0) = make_float4(direct[0], direct[1], direct[2], direct[3]);
    return;
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int f = f0 + r;
    if (f >= H) continue;
    float g_r = 0.f, g_z = 0.f, g_n = 0.f, g_hn = 0.f, direct = 0.f;
    if (i < nrows) {
      float dh = acc[r];
      if (keep) dh = keep[(int64_t)i * H + f] ? dh * extra_scale : 0.f;
      if (carry) dh += carry[(int64_t)i * cld + f];
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

}  // namespace g2v
