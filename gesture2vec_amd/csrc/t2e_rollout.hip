// t2e_rollout.hip -- Part d: the greedy gesture-code decoder of text2embedding_model as fused per-step kernels
// (g2v_attn_code_rollout_fwd / _bwd, SURVEY.md 8(b) K13 + K14).
//
// Replaces the loop model/text2embedding_model.py:701-744 over BahdanauAttnDecoderRNN.forward (:338-395) and its autograd:
// per decode step  Embedding + Dropout(0.5) [-> Bahdanau attention context] -> Linear + BatchNorm1d + ReLU -> GRU (2 layers)
// -> Linear(H -> K) -> argmax feedback.  Rounds 1-4 chained one launch per operator from Python (~30 launches per step with
// torch cat / add / copy kernels between them).
//
// Structure (the pose decoder's, dec_rollout.hip): a step is row-local except for BatchNorm1d's batch statistics, and a kernel
// boundary is the cheapest grid-wide seam on this chip, so the forward is S1 + 1 launches of one kernel, 16 batch rows per
// 512-thread workgroup (two waves per SIMD: at H = 200 a wave streams ~0.3 MB of weight fragments per step out of L2):
//
//   launch j:  [tail of step j-1]  finish BN(u_{j-1}) from the per-workgroup partials -> ReLU -> GRU cell 0 -> inter-layer dropout
//                                  -> GRU cell 1 -> out layer -> logits_{j-1} -> row argmax -> id_j
//              [head of step j]    e_j = Embedding(id_j) * keep * 2 [-> hp = W_h h1_j + b -> energies -> softmax -> context]
//                                  -> u_j = pre_linear(x_j) + per-workgroup partial sums of (u_j - b)
//
// Backward without attention: the argmax feedback carries no gradient and du_t feeds only weight / embedding gradients, so NO
// grid-wide quantity sits on the chain between two steps: the BPTT over all S1 steps is ONE launch (state gradients carried in
// LDS), then BatchNorm's backward for all steps at once, then the products over the S1 x B rows (data gradient of pre_linear,
// embedding gradient, every weight gradient) as batched launches of the library's dense kernels.
// Backward with attention: d(context_t) = du_t W_pre[:, H:] needs BatchNorm's backward sums of step t, and flows into h1_t: one
// launch per step, [finish BN backward of step t+1 -> du_{t+1} -> d(ctx) -> attention backward -> d(h1) carry] + [cells of step t].
#include <stdlib.h>

#include "gru_cells.hpp"
#include "dec_persist.hpp"      // the cluster exchange primitives (cx_*); not the fault latch

namespace g2v {

constexpr int CT_NTHR = 512, CT_NW = CT_NTHR / 64;
constexpr int CT_MAX_TW = 64;

struct CodePackF {      // packed forward weights (fragment-major, common.hpp)
  const float* pre;     // rows H, K = Hin
  const float* ih0; const float* hh0; const float* ih1; const float* hh1;   // 3 gate groups x H rows, K = H
  const float* out;     // rows K (tiles padded to a multiple of 4 * CT_NW), K = H
  const float* attn_h;  // rows H, K = H  (W_attn[:, :H])
};
struct CodePackB {      // packed transposed weights for the backward
  const float* out_t;   // rows H, K = Kdim   (W_out^T)
  const float* ih0_t; const float* hh0_t; const float* ih1_t; const float* hh1_t;   // rows H, K = 3H
  const float* pre_ctx_t;   // rows H (context features), K = H: W_pre[:, H:2H]^T          (attention)
  const float* attn_h_t;    // rows H (state features), K = H: W_attn[:, :H]^T              (attention)
};
struct CodeDims {
  int S1, B, H, K, Hin, Tw;
  float p_drop;
  int n_pre, training, nblk, att, scratch;
};

static inline int ct_ktiles_alloc(int K) { return round_up((K + 15) >> 4, 4 * CT_NW); }

// LDS carve of the forward kernel (floats)
struct CtFwdLds {
  int xa, xh0, xh1, xx1, xh1n, xe, st, red, scratch, amv, amk, ids, sc, total;
};
static __host__ __device__ inline CtFwdLds ct_fwd_lds(int H, int Hin, int Tw, int scratch) {
  const int Hp = (H + 15) & ~15, ldh = Hp + 4, ldx = ((Hin + 15) & ~15) + 4;
  CtFwdLds l;
  int o = 0;
  l.xa = o; o += 16 * ldh;
  l.xh0 = o; o += 16 * ldh;
  l.xh1 = o; o += 16 * ldh;
  l.xx1 = o; o += 16 * ldh;
  l.xh1n = o; o += 16 * ldh;
  l.xe = o; o += 16 * ldx;
  l.st = o; o += 2 * Hp;
  l.red = o; o += 2 * Hp;
  l.scratch = o; o += scratch;
  l.amv = o; o += CT_NW * 16;
  l.amk = o; o += CT_NW * 16;
  l.ids = o; o += 16;
  l.sc = o; o += 16 * (Tw > 0 ? Tw : 1);
  l.total = o;
  return l;
}

// =====================================================================================================================
// forward: launch j of S1 + 1
// =====================================================================================================================
__global__ __launch_bounds__(CT_NTHR) void code_step_fwd_kernel(const int64_t* __restrict__ codes, const float* __restrict__ h_init,
                                                                const float* __restrict__ enc, const float* __restrict__ ep,
                                                                g2v_code_dec_weights w, CodePackF pk, g2v_code_dec_saved sv,
                                                                const uint8_t* __restrict__ keep_emb,
                                                                const uint8_t* __restrict__ keep_l0, CodeDims dm, int j) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int NTHR = CT_NTHR, NW = CT_NW;
  const int S1 = dm.S1, B = dm.B, H = dm.H, K = dm.K, Hin = dm.Hin, Tw = dm.Tw;
  const int Hp = (H + 15) & ~15, ldh = Hp + 4, Hinp = (Hin + 15) & ~15, ldx = Hinp + 4;
  const CtFwdLds L = ct_fwd_lds(H, Hin, Tw, dm.scratch);
  float* Xa = smem + L.xa;          // a_t                [16][ldh]   (head, attention: hp)
  float* Xh0 = smem + L.xh0;        // h0_t
  float* Xh1 = smem + L.xh1;        // h1_t
  float* Xx1 = smem + L.xx1;        // dropped h0_{t+1}
  float* Xh1n = smem + L.xh1n;      // h1_{t+1}
  float* Xe = smem + L.xe;          // x_j = [e_j | ctx_j] [16][ldx]
  float* st = smem + L.st;          // mean[Hp], invstd[Hp]
  float* red = smem + L.red;
  float* red_scratch = smem + L.scratch;
  float* amv = smem + L.amv;        // argmax: per wave and row, best value / index
  int* amk = reinterpret_cast<int*>(smem + L.amk);
  int* ids_l = reinterpret_cast<int*>(smem + L.ids);
  float* sc = smem + L.sc;          // attention scores / weights [16][Tw]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b0 = blockIdx.x * 16;
  const int nrows = min(16, B - b0);
  const int i = lane & 15, q = lane >> 4;
  const int H4 = H >> 2;
  const bool training = dm.training != 0;
  const bool has_tail = j > 0, has_head = j < S1;
  const int t = j - 1;                                   // the step whose tail this launch runs
  const int npre = max(1, min(dm.n_pre, S1));            // steps fed from `codes` (teacher forcing :737-739; step 0 always)

  // ---- prefetch this block's rows of u_t, h0_t, h1_t (written by the previous launch on some other CU) -------------------
  constexpr int NPF = 2;                                  // 16 x H / 4 <= 1024 float4 (H <= 256)
  float4 pu[NPF], ph0[NPF], ph1[NPF];
  int pr[NPF], pc[NPF];
  bool pv[NPF];
#pragma unroll
  for (int k = 0; k < NPF; ++k) {
    const int e = tid + k * NTHR;
    pr[k] = e / H4;
    pc[k] = (e - pr[k] * H4) * 4;
    pv[k] = e < 16 * H4 && pr[k] < nrows;
    pu[k] = ph0[k] = ph1[k] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (pv[k]) {
      if (has_tail) {
        const int64_t row = ((int64_t)t * B + b0 + pr[k]) * H + pc[k];
        pu[k] = *reinterpret_cast<const float4*>(sv.u + row);
        ph0[k] = *reinterpret_cast<const float4*>(sv.h0 + row);
        ph1[k] = *reinterpret_cast<const float4*>(sv.h1 + row);
      } else {
        ph0[k] = *reinterpret_cast<const float4*>(h_init + (int64_t)(b0 + pr[k]) * H + pc[k]);
        ph1[k] = *reinterpret_cast<const float4*>(h_init + ((int64_t)B + b0 + pr[k]) * H + pc[k]);
      }
    }
  }
  // zero what the MFMA contractions must see as zero: padding columns, rows >= nrows
  if (nrows < 16 || Hp != H || Hinp != Hin) {
    for (int e = tid; e < 5 * 16 * ldh + 16 * ldx; e += NTHR) smem[e] = 0.f;
  }
  lds_barrier();

  if (!has_tail) {
    // state in front of step 0 = encoder_hidden[:2] (:667-669)
#pragma unroll
    for (int k = 0; k < NPF; ++k)
      if (pv[k]) {
        *reinterpret_cast<float4*>(sv.h0 + (int64_t)(b0 + pr[k]) * H + pc[k]) = ph0[k];
        *reinterpret_cast<float4*>(sv.h1 + (int64_t)(b0 + pr[k]) * H + pc[k]) = ph1[k];
        *reinterpret_cast<float4*>(Xh1n + pr[k] * ldh + pc[k]) = ph1[k];      // (attention of step 0 scores h1_0)
      }
  } else {
    // ---- (a) BatchNorm statistics of u_t -------------------------------------------------------------------------------
    if (training) {
      const float* part = sv.bn_partial + (int64_t)(t & 1) * dm.nblk * 2 * H;
      reduce_partials<NTHR>(part, dm.nblk, 2 * H, red, red_scratch, tid, dm.scratch);
      for (int f = tid; f < H; f += NTHR) {
        const float s1 = red[f], s2 = red[H + f];
        const float mv = s1 / (float)B;
        const float var = fmaxf(s2 / (float)B - mv * mv, 0.f);   // biased batch variance
        const float mean = mv + w.b_pre[f];
        const float invstd = bn_invstd_(var);
        st[f] = mean;
        st[Hp + f] = invstd;
        if (blockIdx.x == 0) {
          sv.bn_stats[(int64_t)t * 2 * H + f] = mean;
          sv.bn_stats[(int64_t)t * 2 * H + H + f] = invstd;
          // running statistics: momentum 0.1, unbiased variance; one update per decode step, in step order (stream order)
          // (bn_running_mean == NULL while training: the caller commits them behind the backward, g2v_bn_running_update_invstd)
          if (w.bn_running_mean) {
            const float unb = (B > 1) ? var * (float)B / (float)(B - 1) : var;
            w.bn_running_mean[f] = 0.9f * w.bn_running_mean[f] + 0.1f * mean;
            w.bn_running_var[f] = 0.9f * w.bn_running_var[f] + 0.1f * unb;
          }
        }
      }
    } else {
      for (int f = tid; f < H; f += NTHR) {
        st[f] = w.bn_running_mean[f];
        st[Hp + f] = bn_invstd_(w.bn_running_var[f]);
      }
    }
    lds_barrier();
    // ---- (b) a_t = ReLU(BN(u_t)); stage the previous hidden states -----------------------------------------------------
#pragma unroll
    for (int k = 0; k < NPF; ++k)
      if (pv[k]) {
        const int c = pc[k];
        const float4 g4 = *reinterpret_cast<const float4*>(w.bn_w + c), b4 = *reinterpret_cast<const float4*>(w.bn_b + c);
        const float4 m4 = *reinterpret_cast<const float4*>(st + c), i4 = *reinterpret_cast<const float4*>(st + Hp + c);
        float4 a4;
        a4.x = fmaxf((pu[k].x - m4.x) * i4.x * g4.x + b4.x, 0.f);
        a4.y = fmaxf((pu[k].y - m4.y) * i4.y * g4.y + b4.y, 0.f);
        a4.z = fmaxf((pu[k].z - m4.z) * i4.z * g4.z + b4.z, 0.f);
        a4.w = fmaxf((pu[k].w - m4.w) * i4.w * g4.w + b4.w, 0.f);
        *reinterpret_cast<float4*>(Xa + pr[k] * ldh + c) = a4;
        *reinterpret_cast<float4*>(Xh0 + pr[k] * ldh + c) = ph0[k];
        *reinterpret_cast<float4*>(Xh1 + pr[k] * ldh + c) = ph1[k];
        if (sv.a) *reinterpret_cast<float4*>(sv.a + ((int64_t)t * B + b0 + pr[k]) * H + c) = a4;
      }
    lds_barrier();
    // ---- (c) GRU layer 0, (d) GRU layer 1 ------------------------------------------------------------------------------
    const bool drop = training && keep_l0 && dm.p_drop > 0.f;
    gru_cell_fwd<0>(pk.ih0, pk.hh0, w.b_ih0, w.b_hh0, Xa, Xh0, ldh, H, Hp, Xx1, sv.h0 + ((int64_t)(t + 1) * B + b0) * H,
                    sv.gates0 ? sv.gates0 + ((int64_t)t * B + b0) * 4 * H : nullptr,
                    drop ? keep_l0 + ((int64_t)t * B + b0) * H : nullptr, 1.0f / (1.0f - dm.p_drop),
                    (drop && sv.x1) ? sv.x1 + ((int64_t)t * B + b0) * H : nullptr, nrows, lane, wave, NW);
    lds_barrier();
    gru_cell_fwd<0>(pk.ih1, pk.hh1, w.b_ih1, w.b_hh1, Xx1, Xh1, ldh, H, Hp, Xh1n, sv.h1 + ((int64_t)(t + 1) * B + b0) * H,
                    sv.gates1 ? sv.gates1 + ((int64_t)t * B + b0) * 4 * H : nullptr, nullptr, 1.0f, nullptr, nrows, lane, wave,
                    NW);
    lds_barrier();
    // ---- (e) logits_t = out(h1_{t+1}); row argmax (greedy feedback :740-744) -------------------------------------------
    {
      const int ntile = (K + 15) >> 4;
      float bv = -INFINITY;
      int bk = 0x7fffffff;
      for (int base = 0; base < ntile; base += 4 * NW) {
        f32x4 acc[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) acc[u] = (f32x4){0.f, 0.f, 0.f, 0.f};
        wave_gemm_p<4, 0>(acc, pk.out, Hp >> 4, base + wave, NW, Xh1n, ldh, lane);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int k0 = 16 * (base + wave + NW * u) + 4 * q;
          if (k0 + 3 < K) {
            const float4 bo = *reinterpret_cast<const float4*>(w.b_out + k0);
            const float v[4] = {acc[u][0] + bo.x, acc[u][1] + bo.y, acc[u][2] + bo.z, acc[u][3] + bo.w};
            if (i < nrows) *reinterpret_cast<float4*>(sv.logits + ((int64_t)t * B + b0 + i) * K + k0) = make_float4(v[0], v[1], v[2], v[3]);
#pragma unroll
            for (int r = 0; r < 4; ++r)
              if (v[r] > bv) { bv = v[r]; bk = k0 + r; }
          } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int k = k0 + r;
              if (k >= K) continue;
              const float v = acc[u][r] + w.b_out[k];
              if (i < nrows) sv.logits[((int64_t)t * B + b0 + i) * K + k] = v;
              if (v > bv) { bv = v; bk = k; }
            }
          }
        }
      }
      // over the four lanes that hold row i (q = 0..3), then over the waves: the largest value, the lowest index among equals
#pragma unroll
      for (int o = 16; o < 64; o <<= 1) {
        const float v2 = __shfl_xor(bv, o);
        const int k2 = __shfl_xor(bk, o);
        if (v2 > bv || (v2 == bv && k2 < bk)) { bv = v2; bk = k2; }
      }
      if (q == 0) {
        amv[wave * 16 + i] = bv;
        amk[wave * 16 + i] = bk;
      }
      lds_barrier();
      if (tid < 16) {
        float v = amv[tid];
        int k = amk[tid];
        for (int wv = 1; wv < NW; ++wv) {
          const float v2 = amv[wv * 16 + tid];
          const int k2 = amk[wv * 16 + tid];
          if (v2 > v || (v2 == v && k2 < k)) { v = v2; k = k2; }
        }
        ids_l[tid] = (k >= 0 && k < K) ? k : 0;      // (a row of NaNs never compares greater: code 0)
      }
    }
  }
  if (!has_head) return;
  lds_barrier();

  // ---- head of step j: id_j -> e_j = Embedding(id_j) * keep * 2  (:340-343) ------------------------------------------------
  if (tid < 16) {
    int id = 0;
    if (tid < nrows) {
      if (j < npre) {
        const int64_t c = codes[(int64_t)j * B + b0 + tid];
        id = (c >= 0 && c < K) ? (int)c : 0;
      } else {
        id = ids_l[tid];
      }
      sv.ids[(int64_t)j * B + b0 + tid] = id;
    }
    ids_l[tid] = id;
  }
  lds_barrier();
  {
    const bool edrop = training && keep_emb != nullptr;
#pragma unroll
    for (int k = 0; k < NPF; ++k)
      if (pv[k]) {
        const int r = pr[k], c = pc[k];
        float4 e4 = *reinterpret_cast<const float4*>(w.emb + (int64_t)ids_l[r] * H + c);
        if (edrop) {
          const uint32_t kp = *reinterpret_cast<const uint32_t*>(keep_emb + ((int64_t)j * B + b0 + r) * H + c);
          e4.x = (kp & 0xffu) ? e4.x * 2.0f : 0.f;
          e4.y = (kp & 0xff00u) ? e4.y * 2.0f : 0.f;
          e4.z = (kp & 0xff0000u) ? e4.z * 2.0f : 0.f;
          e4.w = (kp & 0xff000000u) ? e4.w * 2.0f : 0.f;
        }
        *reinterpret_cast<float4*>(Xe + r * ldx + c) = e4;
        *reinterpret_cast<float4*>(sv.ec + ((int64_t)j * B + b0 + r) * Hin + c) = e4;
      }
  }
  // ---- Bahdanau attention on h1_j (Attn.forward / score :160-198, context :353-359) ----------------------------------------
  if (dm.att) {
    float* Xhp = Xa;                   // a_t is dead
    const int ntile = Hp >> 4;
    lds_barrier();                     // (Xa readers of cell 0 are long past; ids / Xe writes above)
    for (int ft = wave; ft < ntile; ft += NW) {
      const int f0 = 16 * ft + 4 * q;
      f32x4 acc[1] = {(f32x4){0.f, 0.f, 0.f, 0.f}};
      wave_gemm_p<1, 0>(acc, pk.attn_h, Hp >> 4, ft, 0, Xh1n, ldh, lane);
      if (f0 + 3 < H) {
        const float4 ba = *reinterpret_cast<const float4*>(w.b_attn + f0);
        const float4 h4 = make_float4(acc[0][0] + ba.x, acc[0][1] + ba.y, acc[0][2] + ba.z, acc[0][3] + ba.w);
        *reinterpret_cast<float4*>(Xhp + i * ldh + f0) = h4;
        if (i < nrows && sv.hp) *reinterpret_cast<float4*>(sv.hp + ((int64_t)j * B + b0 + i) * H + f0) = h4;
      }
    }
    lds_barrier();
    // scores: one thread per (row, position); 16 x Tw <= 1024 pairs
    for (int p = tid; p < 16 * Tw; p += NTHR) {
      const int r = p / Tw, tw = p - r * Tw;
      float e = 0.f;
      if (r < nrows) {
        const float* epr = ep + ((int64_t)tw * B + b0 + r) * H;
        const float* hpr = Xhp + r * ldh;
        float e0 = 0.f, e1 = 0.f, e2 = 0.f, e3 = 0.f;
        for (int c = 0; c < H; c += 4) {
          const float4 p4 = *reinterpret_cast<const float4*>(epr + c), h4 = *reinterpret_cast<const float4*>(hpr + c);
          const float4 v4 = *reinterpret_cast<const float4*>(w.v_attn + c);
          e0 += v4.x * tanhf_(h4.x + p4.x);
          e1 += v4.y * tanhf_(h4.y + p4.y);
          e2 += v4.z * tanhf_(h4.z + p4.z);
          e3 += v4.w * tanhf_(h4.w + p4.w);
        }
        e = (e0 + e1) + (e2 + e3);
      }
      sc[r * Tw + tw] = e;
    }
    lds_barrier();
    if (tid < 16 && tid < nrows) {      // softmax over ALL Tw positions (the reference does not mask padded positions)
      float mx = -INFINITY;
      for (int tw = 0; tw < Tw; ++tw) mx = fmaxf(mx, sc[tid * Tw + tw]);
      float sum = 0.f;
      for (int tw = 0; tw < Tw; ++tw) sum += expf(sc[tid * Tw + tw] - mx);
      const float inv = 1.0f / sum;
      for (int tw = 0; tw < Tw; ++tw) {
        const float ww = expf(sc[tid * Tw + tw] - mx) * inv;
        sc[tid * Tw + tw] = ww;
        if (sv.attw) sv.attw[((int64_t)j * B + b0 + tid) * Tw + tw] = ww;
      }
    }
    lds_barrier();
#pragma unroll
    for (int k = 0; k < NPF; ++k)
      if (pv[k]) {
        const int r = pr[k], c = pc[k];
        float4 c4 = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int tw = 0; tw < Tw; ++tw) {
          const float ww = sc[r * Tw + tw];
          const float4 n4 = *reinterpret_cast<const float4*>(enc + ((int64_t)tw * B + b0 + r) * H + c);
          c4.x += ww * n4.x; c4.y += ww * n4.y; c4.z += ww * n4.z; c4.w += ww * n4.w;
        }
        *reinterpret_cast<float4*>(Xe + r * ldx + H + c) = c4;
        *reinterpret_cast<float4*>(sv.ec + ((int64_t)j * B + b0 + r) * Hin + H + c) = c4;
      }
  }
  lds_barrier();
  // ---- u_j = pre_linear.0(x_j) and per-workgroup BatchNorm partial sums of (u - b) --------------------------------------------
  {
    const int ntile = Hp >> 4;
    float* part = sv.bn_partial + ((int64_t)(j & 1) * dm.nblk + blockIdx.x) * 2 * H;
    for (int ft = wave; ft < ntile; ft += NW) {
      const int f0 = 16 * ft + 4 * q;
      const bool vec = f0 + 3 < H;       // (H % 4 == 0: a lane's group of four features is whole or padding)
      float4 bp = make_float4(0.f, 0.f, 0.f, 0.f);
      if (vec) bp = *reinterpret_cast<const float4*>(w.b_pre + f0);
      f32x4 acc[1] = {(f32x4){0.f, 0.f, 0.f, 0.f}};
      wave_gemm_p<1, 0>(acc, pk.pre, Hinp >> 4, ft, 0, Xe, ldx, lane);      // (every lane of the wave: an MFMA is not predicated)
      if (!vec) continue;               // (uniform per DPP row of 16 lanes: the reductions below stay whole)
      float s1[4], s2[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float v = (i < nrows) ? acc[0][r] : 0.f;
        s1[r] = reduce16(v);
        s2[r] = reduce16(v * v);
      }
      if (i < nrows)
        *reinterpret_cast<float4*>(sv.u + ((int64_t)j * B + b0 + i) * H + f0) =
            make_float4(acc[0][0] + bp.x, acc[0][1] + bp.y, acc[0][2] + bp.z, acc[0][3] + bp.w);
      if (i == 0) {
        *reinterpret_cast<float4*>(part + f0) = make_float4(s1[0], s1[1], s1[2], s1[3]);
        *reinterpret_cast<float4*>(part + H + f0) = make_float4(s2[0], s2[1], s2[2], s2[3]);
      }
    }
  }
}

// =====================================================================================================================
// backward
// =====================================================================================================================
struct CodeBwdArgs {
  const float* d_logits;     // (S1,B,K)
  const float* enc; const float* ep;
  g2v_code_dec_weights w;
  CodePackB tw;
  g2v_code_dec_saved sv;
  const uint8_t* keep_l0;
  float* dgi0; float* dgh0; float* dgi1; float* dgh1;     // (S1,B,3H)
  float* dbn;                // (S1,B,H)   d loss / d BN output (behind the ReLU mask)
  float* du;                 // (S1,B,H)
  float* dec;                // (S1,B,H)   d loss / d e_t (the embedding half of d x_t)
  float* dhp;                // (S1,B,H)   attention: d loss / d hp_t
  float* d_ep; float* d_enc; // (Tw,B,H)   attention: accumulated over the steps by the owning workgroup
  float* dv_partial;         // (nblk,H)   attention: per-workgroup sums of d v, accumulated over the steps
  float* bn_part;            // (S1,nblk,2,H) per-workgroup sums of BN backward, one slot per step
  float* bn_sums;            // (S1,2,H)
  float* d_hidden0;          // (2,B,H)
};

// LDS carve of the backward kernels (floats).  H = 200, K = 512: 152 KB of the CU's 160 -- the d logits tile [16][K] doubles as
// the layer-0 incoming gradient tile (written after the out-layer product has consumed it), and the attention phase (Part A of
// code_step_bwd_att_kernel, in front of the cells) borrows the gate tiles: nothing of its own.
struct CtBwdLds {
  int xdl, gi, gh, dd, c0, c1, total;
};
static __host__ __device__ inline CtBwdLds ct_bwd_lds(int H, int K) {
  const int Hp = (H + 15) & ~15, ldh = Hp + 4, Gp = (3 * H + 15) & ~15, ldg = Gp + 4, Kp = (K + 15) & ~15, ldk = Kp + 4;
  CtBwdLds l;
  int o = 0;
  l.xdl = o; o += 16 * (ldk > 2 * ldh ? ldk : 2 * ldh);   // d logits tile; then Xdx [16][ldh]; attention: du | d ctx tiles
  l.gi = o; o += 16 * ldg;                                // attention: d hp tile
  l.gh = o; o += 16 * ldg;                                // attention: BatchNorm sums + reduction scratch, then the d v rows
  l.dd = o; o += 16 * ldh;                                // attention: d_w / ds and the attention weights [2][16][Tw]
  l.c0 = o; o += 16 * ldh;
  l.c1 = o; o += 16 * ldh;
  l.total = o;
  return l;
}

// One step of the BPTT for this workgroup's 16 rows.  dh1 arrives as dlogits_t W_out (+ the carry in C1), the carries C0 / C1
// (LDS, [16][ldh]) hold d loss / d h0_t, d h1_t on exit.  Leaves dbn_t in global memory and this workgroup's BN-backward sums.
__device__ __forceinline__ void code_bwd_cells(const CodeBwdArgs& a, const CodeDims& dm, const CtBwdLds& L, float* smem, int t,
                                               int b0, int nrows, int tid, bool first) {
  constexpr int NTHR = CT_NTHR, NW = CT_NW;
  const int B = dm.B, H = dm.H, K = dm.K, G = 3 * H;
  const int Hp = (H + 15) & ~15, ldh = Hp + 4, Gp = (G + 15) & ~15, ldg = Gp + 4, Kp = (K + 15) & ~15, ldk = Kp + 4;
  float* Xdl = smem + L.xdl;
  float* Gi = smem + L.gi;
  float* Gh = smem + L.gh;
  float* Dd = smem + L.dd;
  float* Xdx = smem + L.xdl;      // (the d logits tile is dead once cell 1's products have read it: barrier in between)
  float* C0 = smem + L.c0;
  float* C1 = smem + L.c1;
  const int lane = tid & 63, wave = tid >> 6;
  const int i = lane & 15, q = lane >> 4;
  const int nth = Hp >> 4;
  constexpr int MAXT = 2;                 // H <= 256: at most two feature tiles per wave
  // cell 1's saved inputs and its carry (LDS: final since the previous step's last barrier), requested before the d logits tile
  // (vector-memory results return in order)
  CellBwdIn cin1[MAXT];
#pragma unroll
  for (int m = 0; m < MAXT; ++m)
    if (wave + NW * m < nth)
      cell_bwd_prefetch(cin1[m], first ? nullptr : C1, nullptr, a.sv.gates1 + ((int64_t)t * B + b0) * 4 * H,
                        a.sv.h1 + ((int64_t)t * B + b0) * H, H, wave + NW * m, nrows, lane, ldh);
  // d logits tile of step t: 16 rows x K, 16-byte vectors
  // (padding columns and rows rewritten every step: the region doubles as the Xdx tile, below)
  {
    const int K4 = Kp >> 2;
    for (int e = tid; e < 16 * K4; e += NTHR) {
      const int r = e / K4, c = (e - r * K4) * 4;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (r < nrows && c < K) v = *reinterpret_cast<const float4*>(a.d_logits + ((int64_t)t * B + b0 + r) * K + c);
      *reinterpret_cast<float4*>(Xdl + r * ldk + c) = v;
    }
  }
  CellBwdIn cin0[MAXT];
  const bool drop = a.keep_l0 && dm.p_drop > 0.f;
#pragma unroll
  for (int m = 0; m < MAXT; ++m)
    if (wave + NW * m < nth)
      cell_bwd_prefetch(cin0[m], first ? nullptr : C0, drop ? a.keep_l0 + ((int64_t)t * B + b0) * H : nullptr,
                        a.sv.gates0 + ((int64_t)t * B + b0) * 4 * H, a.sv.h0 + ((int64_t)t * B + b0) * H, H, wave + NW * m,
                        nrows, lane, ldh);
  lds_barrier();
  // ---- dh1 = carry1 + dlogits W_out ; GRU cell 1 backward ---------------------------------------------------------------
#pragma unroll
  for (int m = 0; m < MAXT; ++m) {
    const int ft = wave + NW * m;
    if (ft < nth) {
      f32x4 acc[1] = {(f32x4){0.f, 0.f, 0.f, 0.f}};
      wave_gemm_p<1, 0>(acc, a.tw.out_t, Kp >> 4, ft, 0, Xdl, ldk, lane);
      gru_cell_bwd_tile(acc[0], first ? nullptr : C1, 1.0f, nullptr, a.sv.gates1 + ((int64_t)t * B + b0) * 4 * H,
                        a.sv.h1 + ((int64_t)t * B + b0) * H, a.dgi1 + ((int64_t)t * B + b0) * G, a.dgh1 + ((int64_t)t * B + b0) * G,
                        Gi, Gh, ldg, Dd, ldh, H, ft, nrows, lane, false, true, cin1[m], ldh);
    }
  }
  lds_barrier();
  // ---- carry1' = dh1 * z + dgh1 W_hh1 ;  dx1 = dgi1 W_ih1 -> dh0 ---------------------------------------------------------
  for (int ft = wave; ft < nth; ft += NW) {
    f32x4 a1[1] = {(f32x4){0.f, 0.f, 0.f, 0.f}}, a2[1] = {(f32x4){0.f, 0.f, 0.f, 0.f}};
    wave_gemm_p_dual<1, 8>(a1, a.tw.hh1_t, Gh, a2, a.tw.ih1_t, Gi, Gp >> 4, ft, 0, ldg, lane);
    const int f0 = 16 * ft + 4 * q;
    if (f0 + 3 < H) {
      const float4 d4 = *reinterpret_cast<const float4*>(Dd + i * ldh + f0);
      *reinterpret_cast<float4*>(C1 + i * ldh + f0) = make_float4(d4.x + a1[0][0], d4.y + a1[0][1], d4.z + a1[0][2], d4.w + a1[0][3]);
      *reinterpret_cast<float4*>(Xdx + i * ldh + f0) = make_float4(a2[0][0], a2[0][1], a2[0][2], a2[0][3]);
    }
  }
  lds_barrier();
  // ---- GRU cell 0 backward (Gi / Gh / Dd are reused) ----------------------------------------------------------------------
#pragma unroll
  for (int m = 0; m < MAXT; ++m) {
    const int ft = wave + NW * m;
    if (ft < nth) {
      const int f0 = 16 * ft + 4 * q;
      f32x4 acc;
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[r] = (f0 + r < H) ? Xdx[i * ldh + f0 + r] : 0.f;
      gru_cell_bwd_tile(acc, first ? nullptr : C0, 1.0f / (1.0f - dm.p_drop), drop ? a.keep_l0 + ((int64_t)t * B + b0) * H : nullptr,
                        a.sv.gates0 + ((int64_t)t * B + b0) * 4 * H, a.sv.h0 + ((int64_t)t * B + b0) * H,
                        a.dgi0 + ((int64_t)t * B + b0) * G, a.dgh0 + ((int64_t)t * B + b0) * G, Gi, Gh, ldg, Dd, ldh, H, ft, nrows,
                        lane, false, true, cin0[m], ldh);
    }
  }
  lds_barrier();
  // ---- carry0' = dh0 * z + dgh0 W_hh0 ;  da = dgi0 W_ih0 -> ReLU backward -> dbn_t + BN-backward partial sums --------------
  {
    const float* stats = a.sv.bn_stats + (int64_t)t * 2 * H;
    float* part = a.bn_part + ((int64_t)t * dm.nblk + blockIdx.x) * 2 * H;
    for (int ft = wave; ft < nth; ft += NW) {
      const int f0 = 16 * ft + 4 * q;
      const bool vec = f0 + 3 < H;
      float av[4] = {0.f, 0.f, 0.f, 0.f}, uv[4] = {0.f, 0.f, 0.f, 0.f}, mv[4] = {0.f, 0.f, 0.f, 0.f}, iv[4] = {0.f, 0.f, 0.f, 0.f};
      if (vec) {
        const float4 m4 = *reinterpret_cast<const float4*>(stats + f0), i4 = *reinterpret_cast<const float4*>(stats + H + f0);
        mv[0] = m4.x; mv[1] = m4.y; mv[2] = m4.z; mv[3] = m4.w;
        iv[0] = i4.x; iv[1] = i4.y; iv[2] = i4.z; iv[3] = i4.w;
        if (i < nrows) {
          const int64_t row = ((int64_t)t * B + b0 + i) * H + f0;
          const float4 a4 = *reinterpret_cast<const float4*>(a.sv.a + row), u4 = *reinterpret_cast<const float4*>(a.sv.u + row);
          av[0] = a4.x; av[1] = a4.y; av[2] = a4.z; av[3] = a4.w;
          uv[0] = u4.x; uv[1] = u4.y; uv[2] = u4.z; uv[3] = u4.w;
        }
      }
      f32x4 a1[1] = {(f32x4){0.f, 0.f, 0.f, 0.f}}, a2[1] = {(f32x4){0.f, 0.f, 0.f, 0.f}};
      wave_gemm_p_dual<1, 8>(a1, a.tw.hh0_t, Gh, a2, a.tw.ih0_t, Gi, Gp >> 4, ft, 0, ldg, lane);
      if (!vec) continue;
      float dbn[4], s1[4], s2[4], cw[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const bool ok = i < nrows;
        cw[r] = Dd[i * ldh + f0 + r] + a1[0][r];
        dbn[r] = (ok && av[r] > 0.f) ? a2[0][r] : 0.f;
        const float dbx = ok ? dbn[r] * ((uv[r] - mv[r]) * iv[r]) : 0.f;
        s1[r] = reduce16(dbn[r]);
        s2[r] = reduce16(dbx);
      }
      *reinterpret_cast<float4*>(C0 + i * ldh + f0) = make_float4(cw[0], cw[1], cw[2], cw[3]);
      if (i < nrows)
        *reinterpret_cast<float4*>(a.dbn + ((int64_t)t * B + b0 + i) * H + f0) = make_float4(dbn[0], dbn[1], dbn[2], dbn[3]);
      if (i == 0) {
        *reinterpret_cast<float4*>(part + f0) = make_float4(s1[0], s1[1], s1[2], s1[3]);
        *reinterpret_cast<float4*>(part + H + f0) = make_float4(s2[0], s2[1], s2[2], s2[3]);
      }
    }
  }
  lds_barrier();
}

__device__ __forceinline__ void code_bwd_write_hidden0(const CodeBwdArgs& a, const CodeDims& dm, const CtBwdLds& L, float* smem,
                                                       int b0, int nrows, int tid) {
  const int B = dm.B, H = dm.H, Hp = (H + 15) & ~15, ldh = Hp + 4, H4 = H >> 2;
  const float* C0 = smem + L.c0;
  const float* C1 = smem + L.c1;
  for (int e = tid; e < 16 * H4; e += CT_NTHR) {
    const int r = e / H4, c = (e - r * H4) * 4;
    if (r >= nrows) continue;
    *reinterpret_cast<float4*>(a.d_hidden0 + (int64_t)(b0 + r) * H + c) = *reinterpret_cast<const float4*>(C0 + r * ldh + c);
    *reinterpret_cast<float4*>(a.d_hidden0 + ((int64_t)B + b0 + r) * H + c) = *reinterpret_cast<const float4*>(C1 + r * ldh + c);
  }
}

// no attention: the whole BPTT in one launch
__global__ __launch_bounds__(CT_NTHR) void code_bptt_kernel(CodeBwdArgs a, CodeDims dm) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const CtBwdLds L = ct_bwd_lds(dm.H, dm.K);
  const int tid = threadIdx.x;
  const int b0 = blockIdx.x * 16, nrows = min(16, dm.B - b0);
  for (int e = tid; e < L.total; e += CT_NTHR) smem[e] = 0.f;      // padding columns / rows >= nrows of every operand tile, carries
  lds_barrier();
  for (int t = dm.S1 - 1; t >= 0; --t) code_bwd_cells(a, dm, L, smem, t, b0, nrows, tid, t == dm.S1 - 1);
  code_bwd_write_hidden0(a, dm, L, smem, b0, nrows, tid);
}

// BatchNorm backward of ALL steps (no attention): grid (nblk, S1); workgroup (b, t) sums step t's per-workgroup partials and
// forms du_t for its 16 rows; workgroup (0, t) leaves the sums for the weight / bias gradient.
__global__ __launch_bounds__(256) void code_bn_bwd_apply_kernel(CodeBwdArgs a, CodeDims dm) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int B = dm.B, H = dm.H, t = blockIdx.y, tid = threadIdx.x;
  float* red = smem;               // [2H]
  float* scratch = smem + 2 * H;   // [1024]
  reduce_partials<256>(a.bn_part + (int64_t)t * dm.nblk * 2 * H, dm.nblk, 2 * H, red, scratch, tid, 1024);
  if (blockIdx.x == 0)
    for (int f = tid; f < 2 * H; f += 256) a.bn_sums[(int64_t)t * 2 * H + f] = red[f];
  const int b0 = blockIdx.x * 16, nrows = min(16, B - b0), H4 = H >> 2;
  const float invB = 1.0f / (float)B;
  const float* stats = a.sv.bn_stats + (int64_t)t * 2 * H;
  for (int e = tid; e < 16 * H4; e += 256) {
    const int r = e / H4, c = (e - r * H4) * 4;
    if (r >= nrows) continue;
    const int64_t row = ((int64_t)t * B + b0 + r) * H + c;
    const float4 u4 = *reinterpret_cast<const float4*>(a.sv.u + row), d4 = *reinterpret_cast<const float4*>(a.dbn + row);
    const float4 m4 = *reinterpret_cast<const float4*>(stats + c), i4 = *reinterpret_cast<const float4*>(stats + H + c);
    const float4 g4 = *reinterpret_cast<const float4*>(a.w.bn_w + c);
    const float uu[4] = {u4.x, u4.y, u4.z, u4.w}, db[4] = {d4.x, d4.y, d4.z, d4.w}, mm[4] = {m4.x, m4.y, m4.z, m4.w},
                ii[4] = {i4.x, i4.y, i4.z, i4.w}, gg[4] = {g4.x, g4.y, g4.z, g4.w};
    float du[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float xhat = (uu[k] - mm[k]) * ii[k];
      du[k] = gg[k] * ii[k] * (db[k] - red[c + k] * invB - xhat * red[H + c + k] * invB);
    }
    *reinterpret_cast<float4*>(a.du + row) = make_float4(du[0], du[1], du[2], du[3]);
  }
}

// d bn_w = sum_t S2_t, d bn_b = sum_t S1_t; [attention] d v = sum over the workgroups' partials
__global__ __launch_bounds__(256) void code_small_sums_kernel(const float* __restrict__ bn_sums, int S1, int H,
                                                              float* __restrict__ d_bn_w, float* __restrict__ d_bn_b,
                                                              const float* __restrict__ dv_partial, int nblk,
                                                              float* __restrict__ d_v) {
  for (int f = threadIdx.x; f < H; f += 256) {
    float sw = 0.f, sb = 0.f;
    for (int t = 0; t < S1; ++t) {
      sb += bn_sums[(int64_t)t * 2 * H + f];
      sw += bn_sums[(int64_t)t * 2 * H + H + f];
    }
    d_bn_w[f] = sw;
    d_bn_b[f] = sb;
    if (d_v) {
      float s = 0.f;
      for (int k = 0; k < nblk; ++k) s += dv_partial[(int64_t)k * H + f];
      d_v[f] = s;
    }
  }
}

// attention: contiguous copies of the two column halves a dense kernel cannot address (its weights are [N][K] with ld = K):
// out0 = W_pre[:, :H], out1 = W_attn[:, H:]   (src row stride 2H)
__global__ void code_split_cols_kernel(const float* __restrict__ w_pre, const float* __restrict__ w_attn, float* __restrict__ out0,
                                       float* __restrict__ out1, int H) {
  for (int e = blockIdx.x * 256 + threadIdx.x; e < H * H; e += gridDim.x * 256) {
    const int r = e / H, c = e - r * H;
    out0[e] = w_pre[(int64_t)r * 2 * H + c];
    out1[e] = w_attn[(int64_t)r * 2 * H + H + c];
  }
}
// d attn.attn.weight (H,2H) = [dW_h | dW_e]
__global__ void code_join_cols_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out, int H) {
  for (int e = blockIdx.x * 256 + threadIdx.x; e < H * H; e += gridDim.x * 256) {
    const int r = e / H, c = e - r * H;
    out[(int64_t)r * 2 * H + c] = a[e];
    out[(int64_t)r * 2 * H + H + c] = b[e];
  }
}

// attention: launch for step t (t = S1-1 .. 0), plus a last launch t = -1 that only finishes step 0's BatchNorm / attention
// backward and writes d_hidden0.  Part A (t < S1-1): finish the BN backward of step t+1 from the per-workgroup partials ->
// du_{t+1} -> d x_{t+1} = du W_pre (context half) -> attention backward (d_ep, d_enc accumulated in place by the owning
// workgroup, d hp_{t+1} saved, d v into this workgroup's partial) -> carry1 += d hp W_attn[:, :H].  Part B: the cells of step t.
// The carries travel between the launches in d_hidden0.
__global__ __launch_bounds__(CT_NTHR) void code_step_bwd_att_kernel(CodeBwdArgs a, CodeDims dm, int t) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int NTHR = CT_NTHR, NW = CT_NW;
  const CtBwdLds L = ct_bwd_lds(dm.H, dm.K);
  const int S1 = dm.S1, B = dm.B, H = dm.H, Tw = dm.Tw;
  const int Hp = (H + 15) & ~15, ldh = Hp + 4, H4 = H >> 2;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i = lane & 15, q = lane >> 4;
  const int b0 = blockIdx.x * 16, nrows = min(16, B - b0);
  float* C0 = smem + L.c0;
  float* C1 = smem + L.c1;
  float* Xdu = smem + L.xdl;               // [16][ldh]   (the d logits tile is staged after Part A)
  float* Xdc = Xdu + 16 * ldh;             // [16][ldh]   d ctx
  float* Xdhp = smem + L.gi;               // [16][ldh]   d hp (Gi is free until the cells)
  float* red = smem + L.gh;                // [2H] BatchNorm backward sums (Gh is free until the cells; dead before the d v rows land)
  float* red_scratch = red + 2 * Hp;       // [dm.scratch]
  float* dsc = smem + L.dd;                // [16][Tw] d_w -> ds          (Dd is free until the cells: 2 x 16 x 64 <= 16 x ldh)
  float* wsc = dsc + 16 * Tw;              // [16][Tw] attention weights of step t+1
  const bool last = (t == S1 - 1);
  for (int e = tid; e < L.total; e += NTHR) smem[e] = 0.f;
  lds_barrier();
  if (!last) {
    // carries of step t+1 (written by the previous launch)
    for (int e = tid; e < 16 * H4; e += NTHR) {
      const int r = e / H4, c = (e - r * H4) * 4;
      if (r >= nrows) continue;
      *reinterpret_cast<float4*>(C0 + r * ldh + c) = *reinterpret_cast<const float4*>(a.d_hidden0 + (int64_t)(b0 + r) * H + c);
      *reinterpret_cast<float4*>(C1 + r * ldh + c) = *reinterpret_cast<const float4*>(a.d_hidden0 + ((int64_t)B + b0 + r) * H + c);
    }
    // ================= Part A: BatchNorm backward of step s = t+1, attention backward of step s ==========================
    const int s = t + 1;
    reduce_partials<NTHR>(a.bn_part + (int64_t)s * dm.nblk * 2 * H, dm.nblk, 2 * H, red, red_scratch, tid, dm.scratch);
    if (blockIdx.x == 0)
      for (int f = tid; f < 2 * H; f += NTHR) a.bn_sums[(int64_t)s * 2 * H + f] = red[f];
    const float invB = 1.0f / (float)B;
    const float* stats = a.sv.bn_stats + (int64_t)s * 2 * H;
    for (int e = tid; e < 16 * H4; e += NTHR) {
      const int r = e / H4, c = (e - r * H4) * 4;
      if (r >= nrows) continue;
      const int64_t row = ((int64_t)s * B + b0 + r) * H + c;
      const float4 u4 = *reinterpret_cast<const float4*>(a.sv.u + row), d4 = *reinterpret_cast<const float4*>(a.dbn + row);
      const float4 m4 = *reinterpret_cast<const float4*>(stats + c), i4 = *reinterpret_cast<const float4*>(stats + H + c);
      const float4 g4 = *reinterpret_cast<const float4*>(a.w.bn_w + c);
      const float uu[4] = {u4.x, u4.y, u4.z, u4.w}, db[4] = {d4.x, d4.y, d4.z, d4.w}, mm[4] = {m4.x, m4.y, m4.z, m4.w},
                  ii[4] = {i4.x, i4.y, i4.z, i4.w}, gg[4] = {g4.x, g4.y, g4.z, g4.w};
      float du[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float xhat = (uu[k] - mm[k]) * ii[k];
        du[k] = gg[k] * ii[k] * (db[k] - red[c + k] * invB - xhat * red[H + c + k] * invB);
      }
      const float4 du4 = make_float4(du[0], du[1], du[2], du[3]);
      *reinterpret_cast<float4*>(a.du + row) = du4;
      *reinterpret_cast<float4*>(Xdu + r * ldh + c) = du4;
    }
    // attention weights of step s
    for (int p = tid; p < 16 * Tw; p += NTHR) {
      const int r = p / Tw, tw = p - r * Tw;
      wsc[p] = (r < nrows) ? a.sv.attw[((int64_t)s * B + b0 + r) * Tw + tw] : 0.f;
    }
    lds_barrier();
    // d ctx_s = du_s W_pre[:, H:2H]   (rows = context feature, contraction over the H outputs of pre_linear)
    const int nth = Hp >> 4;
    for (int ft = wave; ft < nth; ft += NW) {
      const int f0 = 16 * ft + 4 * q;
      f32x4 acc[1] = {(f32x4){0.f, 0.f, 0.f, 0.f}};
      wave_gemm_p<1, 0>(acc, a.tw.pre_ctx_t, Hp >> 4, ft, 0, Xdu, ldh, lane);
      if (f0 + 3 < H) {
        const float4 c4 = make_float4(acc[0][0], acc[0][1], acc[0][2], acc[0][3]);
        *reinterpret_cast<float4*>(Xdc + i * ldh + f0) = c4;
      }
    }
    lds_barrier();
    // d_w[r][tw] = <d ctx[r], enc[tw, r]>
    for (int p = tid; p < 16 * Tw; p += NTHR) {
      const int r = p / Tw, tw = p - r * Tw;
      float e = 0.f;
      if (r < nrows) {
        const float* er = a.enc + ((int64_t)tw * B + b0 + r) * H;
        const float* dc = Xdc + r * ldh;
        float e0 = 0.f, e1 = 0.f, e2 = 0.f, e3 = 0.f;
        for (int c = 0; c < H; c += 4) {
          const float4 n4 = *reinterpret_cast<const float4*>(er + c), d4 = *reinterpret_cast<const float4*>(dc + c);
          e0 += d4.x * n4.x; e1 += d4.y * n4.y; e2 += d4.z * n4.z; e3 += d4.w * n4.w;
        }
        e = (e0 + e1) + (e2 + e3);
      }
      dsc[p] = e;
    }
    lds_barrier();
    if (tid < 16) {     // softmax backward: ds = w (d_w - <w, d_w>)
      float dot = 0.f;
      for (int tw = 0; tw < Tw; ++tw) dot += wsc[tid * Tw + tw] * dsc[tid * Tw + tw];
      for (int tw = 0; tw < Tw; ++tw) dsc[tid * Tw + tw] = wsc[tid * Tw + tw] * (dsc[tid * Tw + tw] - dot);
    }
    lds_barrier();
    // per (row, feature group): d hp, d v partial, d_ep / d_enc rows (accumulated over the steps by this workgroup alone)
    {
      float* dvp = smem + L.gh;        // [16][ldh] d v contributions per row (Gh is free until the cells)
      const bool acc_steps = (s != S1 - 1);
      for (int e = tid; e < 16 * H4; e += NTHR) {
        const int r = e / H4, c = (e - r * H4) * 4;
        float dh[4] = {0.f, 0.f, 0.f, 0.f}, dv[4] = {0.f, 0.f, 0.f, 0.f};
        if (r < nrows) {
          const float4 hp4 = *reinterpret_cast<const float4*>(a.sv.hp + ((int64_t)s * B + b0 + r) * H + c);
          const float4 v4 = *reinterpret_cast<const float4*>(a.w.v_attn + c);
          const float4 dc4 = *reinterpret_cast<const float4*>(Xdc + r * ldh + c);
          const float hp[4] = {hp4.x, hp4.y, hp4.z, hp4.w}, vv[4] = {v4.x, v4.y, v4.z, v4.w}, dc[4] = {dc4.x, dc4.y, dc4.z, dc4.w};
          for (int tw = 0; tw < Tw; ++tw) {
            const int64_t row = ((int64_t)tw * B + b0 + r) * H + c;
            const float4 p4 = *reinterpret_cast<const float4*>(a.ep + row);
            const float pp[4] = {p4.x, p4.y, p4.z, p4.w};
            const float ds = dsc[r * Tw + tw], ww = wsc[r * Tw + tw];
            float de[4], dn[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
              const float en = tanhf_(hp[k] + pp[k]);
              de[k] = ds * vv[k] * (1.0f - en * en);
              dh[k] += de[k];
              dv[k] += ds * en;
              dn[k] = ww * dc[k];
            }
            float4 o1 = make_float4(de[0], de[1], de[2], de[3]), o2 = make_float4(dn[0], dn[1], dn[2], dn[3]);
            if (acc_steps) {
              const float4 q1 = *reinterpret_cast<const float4*>(a.d_ep + row), q2 = *reinterpret_cast<const float4*>(a.d_enc + row);
              o1.x += q1.x; o1.y += q1.y; o1.z += q1.z; o1.w += q1.w;
              o2.x += q2.x; o2.y += q2.y; o2.z += q2.z; o2.w += q2.w;
            }
            *reinterpret_cast<float4*>(a.d_ep + row) = o1;
            *reinterpret_cast<float4*>(a.d_enc + row) = o2;
          }
          *reinterpret_cast<float4*>(a.dhp + ((int64_t)s * B + b0 + r) * H + c) = make_float4(dh[0], dh[1], dh[2], dh[3]);
        }
        *reinterpret_cast<float4*>(Xdhp + r * ldh + c) = make_float4(dh[0], dh[1], dh[2], dh[3]);
        *reinterpret_cast<float4*>(dvp + r * ldh + c) = make_float4(dv[0], dv[1], dv[2], dv[3]);
      }
      lds_barrier();
      for (int f = tid; f < H; f += NTHR) {
        float sdv = 0.f;
        for (int r = 0; r < 16; ++r) sdv += dvp[r * ldh + f];
        float* o = a.dv_partial + (int64_t)blockIdx.x * H + f;
        *o = acc_steps ? *o + sdv : sdv;
      }
    }
    // carry1 += d hp W_attn[:, :H]     (the state the attention of step s scored = h1 in front of step s = output of step t)
    for (int ft = wave; ft < nth; ft += NW) {
      const int f0 = 16 * ft + 4 * q;
      f32x4 acc[1] = {(f32x4){0.f, 0.f, 0.f, 0.f}};
      wave_gemm_p<1, 0>(acc, a.tw.attn_h_t, Hp >> 4, ft, 0, Xdhp, ldh, lane);
      if (f0 + 3 < H) {
        float4 c4 = *reinterpret_cast<const float4*>(C1 + i * ldh + f0);
        c4.x += acc[0][0]; c4.y += acc[0][1]; c4.z += acc[0][2]; c4.w += acc[0][3];
        *reinterpret_cast<float4*>(C1 + i * ldh + f0) = c4;
      }
    }
    lds_barrier();
    // Xdl / Gi / Gh / Dd were borrowed: padding back to zero for the cells
    for (int e = tid; e < L.c0; e += NTHR) smem[e] = 0.f;
    lds_barrier();
  }
  if (t >= 0) code_bwd_cells(a, dm, L, smem, t, b0, nrows, tid, last);
  code_bwd_write_hidden0(a, dm, L, smem, b0, nrows, tid);
}

}  // namespace g2v

// ====================================================================================================================
// Small batch (round 5): the whole greedy FORWARD rollout without attention as ONE persistent launch of (hidden-unit tile x row
// group) workgroups -- the cluster scheme of the pose decoder (dec_rollout.hip: dec_cluster_fwd_kernel; protocol in
// dec_persist.hpp) for the code decoder.  At the reference's B = 128 the per-operator path runs ~9 launches per decode step.
// Per step and workgroup: hidden sides (operands in LDS since the step before) | sweep the u_t row + the BatchNorm sums of every
// row group | a_t = ReLU(BN(u_t)) in LDS | cell 0, input side split by gate over three waves | h0 exchange (+ inter-layer dropout) |
// cell 1 | h1 exchange | logits: the workgroup's K tiles (tile, tile + NT, tile + 2 NT), k-steps split over the four waves | the row
// argmax: per-workgroup best (value, index) pairs exchanged, every workgroup reduces them in the same order (lowest index among
// equals, as torch.argmax) | head of the next step: id -> Embedding * keep * 2 gathered by every workgroup, its pre_linear tile and
// BatchNorm partial sums published.  Writes exactly the arrays g2v_attn_code_rollout_fwd's step kernels write, so either backward
// (the fused BPTT of this file, the per-operator chain of rollout_t2e.py) runs on them.  H <= 208, K <= 48 NT, no attention.
// ====================================================================================================================
namespace g2v {
constexpr int CCL_KS = 13;          // k-steps over H (H <= 208)
constexpr int CCL_KT = 3;           // K tiles of the out layer per workgroup
struct CodeClArgs {
  const int64_t* codes; const float* h_init; const uint8_t* keep_emb; const uint8_t* keep_l0;
  g2v_code_dec_weights w; g2v_code_dec_saved sv;
  unsigned long long* xu;      // [2][nblk] row records: u rows
  unsigned long long* xp;      // [2][nblk][2][Hp]: BatchNorm partial sums
  unsigned long long* xh0;     // [2][nblk] row records
  unsigned long long* xh1;     // [2][nblk] row records
  unsigned long long* xa;      // [2][nblk][NT][16][2]: per-workgroup best (value, index) of every row
  unsigned* xcc;               // [nblk][NT]: the XCC every workgroup runs on (cx_cluster_on_one_xcd)
  unsigned* fault;
  int S1, B, H, K, n_pre, training, nt, nblk;
  float p_drop;
};
__device__ __forceinline__ void ccl_publish4(__amdgpu_buffer_rsrc_t rr, unsigned granule, const float* v, unsigned tag) {
  u32x4 a, b;
  a[0] = __float_as_uint(v[0]); a[1] = tag; a[2] = __float_as_uint(v[1]); a[3] = tag;
  b[0] = __float_as_uint(v[2]); b[1] = tag; b[2] = __float_as_uint(v[3]); b[3] = tag;
  px_st(rr, granule * 8u, a);
  px_st(rr, granule * 8u + 16u, b);
}
// column f of the per-row-group partial sums [nblk][2][Hp] (granules), summed over the row groups in ascending order
__device__ __forceinline__ void ccl_sum_partials(const unsigned long long* rec, int nblk, int Hp, int f, unsigned tag, unsigned* fault,
                                                 float& s1, float& s2) {
  s1 = 0.f; s2 = 0.f;
  for (int k0 = 0; k0 < nblk; k0 += 8) {
    unsigned long long a[8], b[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const bool ok = k0 + j < nblk;
      const unsigned long long* p = rec + (size_t)(ok ? k0 + j : 0) * 2 * Hp + f;
      a[j] = ok ? __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : ((unsigned long long)tag << 32);
      b[j] = ok ? __hip_atomic_load(p + Hp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : ((unsigned long long)tag << 32);
    }
    unsigned spins = 0;
    for (;;) {
      bool ok = true;
#pragma unroll
      for (int j = 0; j < 8; ++j) ok &= (unsigned)(a[j] >> 32) == tag && (unsigned)(b[j] >> 32) == tag;
      if (ok) break;
      __builtin_amdgcn_s_sleep(1);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const unsigned long long* p = rec + (size_t)(k0 + j < nblk ? k0 + j : 0) * 2 * Hp + f;
        if ((unsigned)(a[j] >> 32) != tag) a[j] = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((unsigned)(b[j] >> 32) != tag) b[j] = __hip_atomic_load(p + Hp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      if (cx_give_up(spins, fault)) break;
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      s1 += __uint_as_float((unsigned)a[j]);
      s2 += __uint_as_float((unsigned)b[j]);
    }
  }
}
// GRU cell epilogue of the lane's 4 units from the six accumulators in LDS (the arithmetic of gru_cell_fwd_epilogue)
__device__ __forceinline__ void ccl_cell_epilogue(const float4* xcx, const float4* xch, const float4 (*bs)[4], int lane, int q,
                                                  const float (&hown)[4], float (&hn)[4], float (&gr_)[4], float (&gz_)[4],
                                                  float (&gn_)[4], float (&gh_)[4]) {
  const float4 bi0 = bs[0][q], bi1 = bs[1][q], bi2 = bs[2][q], bh0 = bs[3][q], bh1 = bs[4][q], bh2 = bs[5][q];
  const float bir[4] = {bi0.x, bi0.y, bi0.z, bi0.w}, biz[4] = {bi1.x, bi1.y, bi1.z, bi1.w}, bin[4] = {bi2.x, bi2.y, bi2.z, bi2.w};
  const float bhr[4] = {bh0.x, bh0.y, bh0.z, bh0.w}, bhz[4] = {bh1.x, bh1.y, bh1.z, bh1.w}, bhn[4] = {bh2.x, bh2.y, bh2.z, bh2.w};
  const float4 v0 = xch[lane], v1 = xch[64 + lane], v2 = xch[128 + lane];
  const float ah[3][4] = {{v0.x, v0.y, v0.z, v0.w}, {v1.x, v1.y, v1.z, v1.w}, {v2.x, v2.y, v2.z, v2.w}};
  const float4 c0 = xcx[lane], c1 = xcx[64 + lane], c2 = xcx[128 + lane];
  const float acc[3][4] = {{c0.x, c0.y, c0.z, c0.w}, {c1.x, c1.y, c1.z, c1.w}, {c2.x, c2.y, c2.z, c2.w}};
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const float rr = sigmoidf_((acc[0][r] + bir[r]) + (ah[0][r] + bhr[r]));
    const float zz = sigmoidf_((acc[1][r] + biz[r]) + (ah[1][r] + bhz[r]));
    const float ghn = ah[2][r] + bhn[r];
    const float nn = tanhf_((acc[2][r] + bin[r]) + rr * ghn);
    hn[r] = (1.0f - zz) * nn + zz * hown[r];
    gr_[r] = rr; gz_[r] = zz; gn_[r] = nn; gh_[r] = ghn;
  }
}

__global__ __launch_bounds__(256) void code_cluster_fwd_kernel(CodeClArgs a) {
  constexpr int KS = CCL_KS;
  __shared__ float st[2 * 256];                                                   // mean[H], invstd[H]
  __shared__ float bnw_s[2 * 256];                                                // BatchNorm weight[H], bias[H]
  __shared__ __attribute__((aligned(16))) float4 xch2[2][3 * 64];                // [cell][gate] hidden-side accumulators
  __shared__ __attribute__((aligned(16))) float4 xcx[3 * 64];                    // [gate] input-side accumulators
  __shared__ __attribute__((aligned(16))) float4 xs_u[KS][64];                   // u_t rows as B fragments; a_t; then e_{t+1}
  __shared__ __attribute__((aligned(16))) float4 xs_h0[KS][64];
  __shared__ __attribute__((aligned(16))) float4 xs_h1[KS][64];
  __shared__ __attribute__((aligned(16))) float4 bias_s[2][6][4];
  __shared__ __attribute__((aligned(16))) float4 up_s[4][64];                    // the waves' partial pre_linear products
  __shared__ float bv_s[4][16];                                                   // row argmax: per K tile best value / index
  __shared__ int bk_s[4][16];
  __shared__ int ids_l[16];
  __shared__ int xcd_flag;
  extern __shared__ __attribute__((aligned(16))) float4 dyn_s[];                 // W_out fragments [KT][KS][64] | W_pre fragments [KS][64] |
                                                                                  // partial logits [4][KT][64] | Dropout(h0) rows [KS][64]
  const int S1 = a.S1, B = a.B, H = a.H, K = a.K;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, i = lane & 15, q = lane >> 4;
  const bool xw = wave != 1;
  const int gx = wave == 0 ? 0 : wave - 1;
  // 1-D grid; with a row-group count that is a multiple of 8 the row group is the FAST index, so that under round-robin placement
  // the tile workgroups of a row group share an XCD (verified below) and their records go through that XCD's L2
  const int nt = a.nt, nblk = a.nblk;
  const bool rg_fast = (nblk & 7) == 0;
  const int ft = rg_fast ? (int)blockIdx.x / nblk : (int)blockIdx.x % nt, rg = rg_fast ? (int)blockIdx.x % nblk : (int)blockIdx.x / nt;
  const int b0 = rg * 16;
  const int nrows = min(16, B - b0);
  const int Hp = nt << 4, nkt = (K + 15) >> 4;
  const bool rvalid = i < nrows, wrow_ok = 16 * ft + i < H;
  const int b = b0 + (rvalid ? i : 0);
  const int f0 = 16 * ft + 4 * q;
  const bool fok = f0 < H;
  const int64_t BH = (int64_t)B * H;
  const g2v_code_dec_weights& w = a.w;
  const g2v_code_dec_saved& sv = a.sv;
  const bool training = a.training != 0;
  const int npre = max(1, min(a.n_pre, S1));
  float4* wo_s = dyn_s;                                   // [KT][KS][64]
  float4* wp_s = wo_s + (size_t)CCL_KT * KS * 64;         // [KS][64]
  float4* yp_s = wp_s + (size_t)KS * 64;                  // [4][KT][64]
  float4* xs_x1 = yp_s + (size_t)4 * CCL_KT * 64;         // [KS][64]
  // ---- resident: wave 1 W_hh0 (three gates); the gate waves W_ih0[gx], W_ih1[gx], W_hh1[gx] -----------------------------------------
  float4 wreg[3][KS];
#pragma unroll
  for (int m = 0; m < 3; ++m) {
    const float* W = xw ? (m == 0 ? w.w_ih0 : (m == 1 ? w.w_ih1 : w.w_hh1)) : w.w_hh0;
    const float* wr = W + ((int64_t)(xw ? gx : m) * H + 16 * ft + (wrow_ok ? i : 0)) * H;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const int k = 16 * ks + 4 * q;
      const bool kok = k < H;
      wreg[m][ks] = ld4_or_zero(wr + (kok ? k : 0), kok && wrow_ok);
    }
  }
  // LDS operands: W_out rows of the workgroup's K tiles (wave j < KT fills tile j), W_pre rows of the tile (wave 3), biases
  for (int ks = 0; ks < KS; ++ks) {
    const int k = 16 * ks + 4 * q;
    const bool kok = k < H;
    if (wave < CCL_KT) {
      const int kt = ft + nt * wave, row = 16 * kt + i;
      const bool ok = kt < nkt && row < K && kok;
      wo_s[(wave * KS + ks) * 64 + lane] = ld4_or_zero(w.w_out + (int64_t)(ok ? row : 0) * H + (kok ? k : 0), ok);
    } else {
      wp_s[ks * 64 + lane] = ld4_or_zero(w.w_pre + (int64_t)(16 * ft + (wrow_ok ? i : 0)) * H + (kok ? k : 0), kok && wrow_ok);
    }
  }
  if ((wave == 0 || wave == 2) && i == 0) {
    const int cell = wave >> 1;
    const float* bip = cell == 0 ? w.b_ih0 : w.b_ih1;
    const float* bhp = cell == 0 ? w.b_hh0 : w.b_hh1;
#pragma unroll
    for (int g = 0; g < 3; ++g) {
      bias_s[cell][g][q] = ld4_or_zero(bip + g * H + (fok ? f0 : 0), fok);
      bias_s[cell][3 + g][q] = ld4_or_zero(bhp + g * H + (fok ? f0 : 0), fok);
    }
  }
  for (int f = tid; f < 256; f += 256) {
    bnw_s[f] = f < H ? w.bn_w[f] : 0.f;
    bnw_s[256 + f] = f < H ? w.bn_b[f] : 0.f;
  }
  const float4 bp4 = ld4_or_zero(w.b_pre + (fok ? f0 : 0), fok);      // (wave 0's pre_linear epilogue)
  for (int ks = wave; ks < KS; ks += 4) {
    const int k = 16 * ks + 4 * q;
    const bool ok = k < H && rvalid;
    xs_h0[ks][lane] = ld4_or_zero(a.h_init + (int64_t)b * H + (ok ? k : 0), ok);
    xs_h1[ks][lane] = ld4_or_zero(a.h_init + BH + (int64_t)b * H + (ok ? k : 0), ok);
    xs_u[ks][lane] = make_float4(0.f, 0.f, 0.f, 0.f);
    xs_x1[ks * 64 + lane] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  // slot 0 of the saved states = the initial states (the tile's own 16 x 16 block)
  float hown[4] = {0.f, 0.f, 0.f, 0.f};
  if ((wave == 0 || wave == 2) && rvalid && fok) {
    const float4 v = *reinterpret_cast<const float4*>(a.h_init + (wave == 0 ? 0 : BH) + (int64_t)b * H + f0);
    hown[0] = v.x; hown[1] = v.y; hown[2] = v.z; hown[3] = v.w;
    *reinterpret_cast<float4*>((wave == 0 ? sv.h0 : sv.h1) + (int64_t)b * H + f0) = v;
  }
  const bool drop = training && a.keep_l0 && a.p_drop > 0.f;
  const bool edrop = training && a.keep_emb != nullptr;
  const float scale_l0 = 1.0f / (1.0f - a.p_drop);
  const unsigned rowrec = 256u * (unsigned)nt, prec = 2u * (unsigned)Hp, arec = (unsigned)nt * 32u;
  __amdgpu_buffer_rsrc_t r_u = __builtin_amdgcn_make_buffer_rsrc(a.xu, 0, (int)(2u * (unsigned)nblk * rowrec * 8u), 0x00020000);
  __amdgpu_buffer_rsrc_t r_p = __builtin_amdgcn_make_buffer_rsrc(a.xp, 0, (int)(2u * (unsigned)nblk * prec * 8u), 0x00020000);
  __amdgpu_buffer_rsrc_t r_h0 = __builtin_amdgcn_make_buffer_rsrc(a.xh0, 0, (int)(2u * (unsigned)nblk * rowrec * 8u), 0x00020000);
  __amdgpu_buffer_rsrc_t r_h1 = __builtin_amdgcn_make_buffer_rsrc(a.xh1, 0, (int)(2u * (unsigned)nblk * rowrec * 8u), 0x00020000);
  // the row records (u, h0, h1) and the argmax pairs stay inside the row group: through the XCD's L2 when its workgroups share one;
  // the BatchNorm partial sums cross row groups and keep the write-through stores
  const bool l2x = cx_cluster_on_one_xcd(a.xcc + rg * nt, nt, ft, &xcd_flag, tid, a.fault);
  if (tid < 16) {      // the code fed at step 0 (always from `codes`)
    int id = 0;
    if (tid < nrows) {
      const int64_t c = a.codes[b0 + tid];
      id = (c >= 0 && c < K) ? (int)c : 0;
      if (ft == 0) sv.ids[b0 + tid] = id;
    }
    ids_l[tid] = id;
  }
  lds_barrier();
  for (int t = -1; t < S1; ++t) {
    // iteration t = tail of step t (t >= 0) + head of step t + 1 (t + 1 < S1); t = -1: the head of step 0 only
    const unsigned tag = (unsigned)(t + 1);      // every record of step t carries tag t + 1 (u_t was published by the head of step t)
    if (t >= 0) {
      const unsigned ppar = (unsigned)(t & 1);      // parity of every record of step t: u, BatchNorm sums, h0, h1, argmax
      // ---- hidden sides; sweep of the u_t row; BatchNorm statistics ------------------------------------------------------------------
      uint32_t kp = 0x01010101u;
      if (!xw) {
        f32x4 acc[3];
#pragma unroll
        for (int g = 0; g < 3; ++g) acc[g] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          const float4 x4 = xs_h0[ks][lane];
#pragma unroll
          for (int g = 0; g < 3; ++g) {
            acc[g] = mfma16(wreg[g][ks].x, x4.x, acc[g]);
            acc[g] = mfma16(wreg[g][ks].y, x4.y, acc[g]);
            acc[g] = mfma16(wreg[g][ks].z, x4.z, acc[g]);
            acc[g] = mfma16(wreg[g][ks].w, x4.w, acc[g]);
          }
        }
#pragma unroll
        for (int g = 0; g < 3; ++g) xch2[0][g * 64 + lane] = make_float4(acc[g][0], acc[g][1], acc[g][2], acc[g][3]);
      } else {
        f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          const float4 x4 = xs_h1[ks][lane];
          acc = mfma16(wreg[2][ks].x, x4.x, acc);
          acc = mfma16(wreg[2][ks].y, x4.y, acc);
          acc = mfma16(wreg[2][ks].z, x4.z, acc);
          acc = mfma16(wreg[2][ks].w, x4.w, acc);
        }
        xch2[1][gx * 64 + lane] = make_float4(acc[0], acc[1], acc[2], acc[3]);
        cx_sweep_tiles<(KS + 2) / 3>(r_u, (ppar * (unsigned)nblk + (unsigned)rg) * rowrec, gx, 3, nt, nrows, H, tag, &xs_u[0][0], lane, a.fault);
      }
      for (int f = gx * 64 + lane; f < H && xw; f += 192) {
        float mean, invstd;
        if (training) {
          float s1, s2;
          ccl_sum_partials(a.xp + (size_t)ppar * nblk * prec, nblk, Hp, f, tag, a.fault, s1, s2);
          const float mv = s1 / (float)B;
          const float var = fmaxf(s2 / (float)B - mv * mv, 0.f);   // biased batch variance
          mean = mv + w.b_pre[f];
          invstd = bn_invstd_(var);
          if (ft == 0 && rg == 0) {
            sv.bn_stats[(int64_t)t * 2 * H + f] = mean;
            sv.bn_stats[(int64_t)t * 2 * H + H + f] = invstd;
            // running statistics: momentum 0.1, unbiased variance; one update per decode step, in step order; not behind a latched fault
            // (bn_running_mean == NULL while training: the caller commits them behind the backward, g2v_bn_running_update_invstd)
            if (w.bn_running_mean && (a.fault == nullptr || __hip_atomic_load(a.fault, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u)) {
              const float unb = (B > 1) ? var * (float)B / (float)(B - 1) : var;
              w.bn_running_mean[f] = 0.9f * w.bn_running_mean[f] + 0.1f * mean;
              w.bn_running_var[f] = 0.9f * w.bn_running_var[f] + 0.1f * unb;
            }
          }
        } else {
          mean = w.bn_running_mean[f];
          invstd = bn_invstd_(w.bn_running_var[f]);
        }
        st[f] = mean;
        st[256 + f] = invstd;
      }
      lds_barrier();
      // ---- a_t = ReLU(BN(u_t)) in place, a third of the k-steps per gate wave ----------------------------------------------------------
      if (xw) {
        if (wave == 0 && drop && fok && rvalid) kp = *reinterpret_cast<const uint32_t*>(a.keep_l0 + (int64_t)t * BH + (int64_t)b * H + f0);
#pragma unroll
        for (int j = 0; j < (KS + 2) / 3; ++j) {
          const int ks = gx + 3 * j;
          if (ks < KS) {
            const int k = 16 * ks + 4 * q;
            const bool kok = k < H;
            const int kk = kok ? k : 0;
            const float4 g4 = *reinterpret_cast<const float4*>(bnw_s + kk), b4 = *reinterpret_cast<const float4*>(bnw_s + 256 + kk);
            const float4 m4 = *reinterpret_cast<const float4*>(st + kk), i4 = *reinterpret_cast<const float4*>(st + 256 + kk);
            float4 v = xs_u[ks][lane];
            v.x = fmaxf((v.x - m4.x) * i4.x * g4.x + b4.x, 0.f);
            v.y = fmaxf((v.y - m4.y) * i4.y * g4.y + b4.y, 0.f);
            v.z = fmaxf((v.z - m4.z) * i4.z * g4.z + b4.z, 0.f);
            v.w = fmaxf((v.w - m4.w) * i4.w * g4.w + b4.w, 0.f);
            xs_u[ks][lane] = (kok && rvalid) ? v : make_float4(0.f, 0.f, 0.f, 0.f);
            if (ks == ft && sv.a && rvalid && kok) *reinterpret_cast<float4*>(sv.a + (int64_t)t * BH + (int64_t)b * H + k) = v;
          }
        }
      }
      lds_barrier();
      // ---- cell 0 -------------------------------------------------------------------------------------------------------------------------
      if (xw) {
        f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          const float4 x4 = xs_u[ks][lane];
          acc = mfma16(wreg[0][ks].x, x4.x, acc);
          acc = mfma16(wreg[0][ks].y, x4.y, acc);
          acc = mfma16(wreg[0][ks].z, x4.z, acc);
          acc = mfma16(wreg[0][ks].w, x4.w, acc);
        }
        xcx[gx * 64 + lane] = make_float4(acc[0], acc[1], acc[2], acc[3]);
      }
      lds_barrier();
      if (wave == 0 && rvalid && fok) {
        float hn[4], xd[4], gr_[4], gz_[4], gn_[4], gh_[4];
        ccl_cell_epilogue(xcx, xch2[0], bias_s[0], lane, q, hown, hn, gr_, gz_, gn_, gh_);
#pragma unroll
        for (int r = 0; r < 4; ++r) xd[r] = drop ? (((kp >> (8 * r)) & 0xffu) ? hn[r] * scale_l0 : 0.f) : hn[r];
        cx_publish4(r_h0, (ppar * (unsigned)nblk + (unsigned)rg) * rowrec, ft, i, q, hn, tag, l2x);
        *reinterpret_cast<float4*>(sv.h0 + (int64_t)(t + 1) * BH + (int64_t)b * H + f0) = make_float4(hn[0], hn[1], hn[2], hn[3]);
        if (drop && sv.x1) *reinterpret_cast<float4*>(sv.x1 + (int64_t)t * BH + (int64_t)b * H + f0) = make_float4(xd[0], xd[1], xd[2], xd[3]);
        if (sv.gates0) {
          float* go = sv.gates0 + (int64_t)t * 4 * BH + (int64_t)b * 4 * H + f0;
          *reinterpret_cast<float4*>(go) = make_float4(gr_[0], gr_[1], gr_[2], gr_[3]);
          *reinterpret_cast<float4*>(go + H) = make_float4(gz_[0], gz_[1], gz_[2], gz_[3]);
          *reinterpret_cast<float4*>(go + 2 * H) = make_float4(gn_[0], gn_[1], gn_[2], gn_[3]);
          *reinterpret_cast<float4*>(go + 3 * H) = make_float4(gh_[0], gh_[1], gh_[2], gh_[3]);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) hown[r] = hn[r];
      }
      // ---- h0_{t+1} row: waves 1, 2, 3 sweep (wave 0 has just published), each leaves Dropout of its tiles for cell 1 ------------------
      if (wave != 0) {
        uint32_t km[(KS + 2) / 3];
#pragma unroll
        for (int j = 0; j < (KS + 2) / 3; ++j) {
          const int k = 16 * (wave - 1 + 3 * j) + 4 * q;
          const bool ok = drop && wave - 1 + 3 * j < KS && k < H && rvalid;
          km[j] = ok ? *reinterpret_cast<const uint32_t*>(a.keep_l0 + (int64_t)t * BH + (int64_t)b * H + k) : 0u;
        }
        cx_sweep_tiles<(KS + 2) / 3>(r_h0, (ppar * (unsigned)nblk + (unsigned)rg) * rowrec, wave - 1, 3, nt, nrows, H, tag, &xs_h0[0][0], lane, a.fault);
        if (drop) {
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
          for (int j = 0; j < (KS + 2) / 3; ++j) {
            const int ks = wave - 1 + 3 * j;
            if (ks < KS) {
              float4 v = xs_h0[ks][lane];
              const uint32_t m = km[j];
              v.x = (m & 0xffu) ? v.x * scale_l0 : 0.f;
              v.y = ((m >> 8) & 0xffu) ? v.y * scale_l0 : 0.f;
              v.z = ((m >> 16) & 0xffu) ? v.z * scale_l0 : 0.f;
              v.w = ((m >> 24) & 0xffu) ? v.w * scale_l0 : 0.f;
              xs_x1[ks * 64 + lane] = v;
            }
          }
        }
      }
      lds_barrier();
      // ---- cell 1 -------------------------------------------------------------------------------------------------------------------------
      if (xw) {
        const float4* xin1 = drop ? xs_x1 : &xs_h0[0][0];
        f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          const float4 x4 = xin1[ks * 64 + lane];
          acc = mfma16(wreg[1][ks].x, x4.x, acc);
          acc = mfma16(wreg[1][ks].y, x4.y, acc);
          acc = mfma16(wreg[1][ks].z, x4.z, acc);
          acc = mfma16(wreg[1][ks].w, x4.w, acc);
        }
        xcx[gx * 64 + lane] = make_float4(acc[0], acc[1], acc[2], acc[3]);
      }
      lds_barrier();
      if (wave == 2 && rvalid && fok) {
        float hn[4], gr_[4], gz_[4], gn_[4], gh_[4];
        ccl_cell_epilogue(xcx, xch2[1], bias_s[1], lane, q, hown, hn, gr_, gz_, gn_, gh_);
        cx_publish4(r_h1, (ppar * (unsigned)nblk + (unsigned)rg) * rowrec, ft, i, q, hn, tag, l2x);
        *reinterpret_cast<float4*>(sv.h1 + (int64_t)(t + 1) * BH + (int64_t)b * H + f0) = make_float4(hn[0], hn[1], hn[2], hn[3]);
        if (sv.gates1) {
          float* go = sv.gates1 + (int64_t)t * 4 * BH + (int64_t)b * 4 * H + f0;
          *reinterpret_cast<float4*>(go) = make_float4(gr_[0], gr_[1], gr_[2], gr_[3]);
          *reinterpret_cast<float4*>(go + H) = make_float4(gz_[0], gz_[1], gz_[2], gz_[3]);
          *reinterpret_cast<float4*>(go + 2 * H) = make_float4(gn_[0], gn_[1], gn_[2], gn_[3]);
          *reinterpret_cast<float4*>(go + 3 * H) = make_float4(gh_[0], gh_[1], gh_[2], gh_[3]);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) hown[r] = hn[r];
      }
      // ---- logits_t = out(h1_{t+1}): the row is swept by the waves 0, 1, 3; every wave multiplies its k-steps of the K tiles ------------
      if (wave != 2)
        cx_sweep_tiles<(KS + 2) / 3>(r_h1, (ppar * (unsigned)nblk + (unsigned)rg) * rowrec, wave == 3 ? 2 : wave, 3, nt, nrows, H, tag, &xs_h1[0][0],
                                     lane, a.fault);
      lds_barrier();
      {
        f32x4 yq[CCL_KT];
#pragma unroll
        for (int j = 0; j < CCL_KT; ++j) yq[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int jj = 0; jj < (KS + 3) / 4; ++jj) {
          const int ks = wave + 4 * jj;
          if (ks < KS) {
            const float4 x4 = xs_h1[ks][lane];
#pragma unroll
            for (int j = 0; j < CCL_KT; ++j) {
              const float4 w4 = wo_s[(j * KS + ks) * 64 + lane];
              yq[j] = mfma16(w4.x, x4.x, yq[j]);
              yq[j] = mfma16(w4.y, x4.y, yq[j]);
              yq[j] = mfma16(w4.z, x4.z, yq[j]);
              yq[j] = mfma16(w4.w, x4.w, yq[j]);
            }
          }
        }
#pragma unroll
        for (int j = 0; j < CCL_KT; ++j) yp_s[(wave * CCL_KT + j) * 64 + lane] = make_float4(yq[j][0], yq[j][1], yq[j][2], yq[j][3]);
      }
      lds_barrier();
      if (wave < CCL_KT) {
        // wave j: K tile ft + nt j -- the four partial products added in wave order, + bias, stored; the tile's best (value, index) per row
        const int kt = ft + nt * wave, k0 = 16 * kt + 4 * q;
        float v[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int pw = 0; pw < 4; ++pw) {
          const float4 p4 = yp_s[(pw * CCL_KT + wave) * 64 + lane];
          v[0] += p4.x; v[1] += p4.y; v[2] += p4.z; v[3] += p4.w;
        }
        float bv = -INFINITY;
        int bk = 0x7fffffff;
        if (kt < nkt && k0 < K) {      // (K % 4 == 0: four columns or none)
          const float4 bo = *reinterpret_cast<const float4*>(w.b_out + k0);
          v[0] += bo.x; v[1] += bo.y; v[2] += bo.z; v[3] += bo.w;
          if (rvalid) *reinterpret_cast<float4*>(sv.logits + ((int64_t)t * B + b) * K + k0) = make_float4(v[0], v[1], v[2], v[3]);
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (v[r] > bv) { bv = v[r]; bk = k0 + r; }      // (ascending index: the first of equals stays)
        }
#pragma unroll
        for (int o = 16; o < 64; o <<= 1) {
          const float v2 = __shfl_xor(bv, o);
          const int k2 = __shfl_xor(bk, o);
          if (v2 > bv || (v2 == bv && k2 < bk)) { bv = v2; bk = k2; }
        }
        if (q == 0) { bv_s[wave][i] = bv; bk_s[wave][i] = bk; }
      }
      lds_barrier();
      if (t + 1 < S1) {
        // ---- the row argmax over all K: every workgroup publishes its best pair per row, then reduces all NT pairs in the same order ----
        if (wave == 0 && t + 1 >= npre) {
          unsigned long long* rec = a.xa + ((size_t)ppar * nblk + rg) * arec;
          if (lane < 16) {
            float v = bv_s[0][lane];
            int k = bk_s[0][lane];
#pragma unroll
            for (int j = 1; j < CCL_KT; ++j) {
              const float v2 = bv_s[j][lane];
              const int k2 = bk_s[j][lane];
              if (v2 > v || (v2 == v && k2 < k)) { v = v2; k = k2; }
            }
            const unsigned long long g0 = ((unsigned long long)tag << 32) | (unsigned long long)__float_as_uint(v);
            const unsigned long long g1 = ((unsigned long long)tag << 32) | (unsigned long long)(unsigned)k;
            if (l2x) {      // (a plain store: the readers' agent-scope loads find it in this XCD's L2)
              __hip_atomic_store(rec + ((size_t)ft * 16 + lane) * 2, g0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
              __hip_atomic_store(rec + ((size_t)ft * 16 + lane) * 2 + 1, g1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            } else {
              __hip_atomic_store(rec + ((size_t)ft * 16 + lane) * 2, g0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              __hip_atomic_store(rec + ((size_t)ft * 16 + lane) * 2 + 1, g1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
          }
          // lane (row i, group q): the pairs of the producers q, q + 4, ... of row i, all requested at once; then over the four groups.
          // (value descending, index ascending) is a total order: the result does not depend on the order of the comparisons
          unsigned long long gv[4], gk[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int p = q + 4 * j;
            const unsigned long long* pp = rec + ((size_t)(p < nt ? p : 0) * 16 + i) * 2;
            gv[j] = p < nt ? __hip_atomic_load(pp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : ((unsigned long long)tag << 32) | 0xff800000ull;
            gk[j] = p < nt ? __hip_atomic_load(pp + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : ((unsigned long long)tag << 32) | 0x7fffffffull;
          }
          unsigned spins = 0;
          for (;;) {
            bool ok = true;
#pragma unroll
            for (int j = 0; j < 4; ++j) ok &= (unsigned)(gv[j] >> 32) == tag && (unsigned)(gk[j] >> 32) == tag;
            if (ok) break;
            __builtin_amdgcn_s_sleep(1);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const int p = q + 4 * j;
              const unsigned long long* pp = rec + ((size_t)(p < nt ? p : 0) * 16 + i) * 2;
              if ((unsigned)(gv[j] >> 32) != tag) gv[j] = __hip_atomic_load(pp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              if ((unsigned)(gk[j] >> 32) != tag) gk[j] = __hip_atomic_load(pp + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            if (cx_give_up(spins, a.fault)) break;
          }
          float best = -INFINITY;
          int bidx = 0x7fffffff;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const float v2 = __uint_as_float((unsigned)gv[j]);
            const int k2 = (int)(unsigned)gk[j];
            if (v2 > best || (v2 == best && k2 < bidx)) { best = v2; bidx = k2; }
          }
#pragma unroll
          for (int o = 16; o < 64; o <<= 1) {
            const float v2 = __shfl_xor(best, o);
            const int k2 = __shfl_xor(bidx, o);
            if (v2 > best || (v2 == best && k2 < bidx)) { best = v2; bidx = k2; }
          }
          if (q == 0) ids_l[i] = (bidx >= 0 && bidx < K) ? bidx : 0;      // (a row of NaNs never compares greater: code 0)
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        if (tid < 16) {
          int id = ids_l[tid];
          if (t + 1 < npre) {
            id = 0;
            if (tid < nrows) {
              const int64_t c = a.codes[(int64_t)(t + 1) * B + b0 + tid];
              id = (c >= 0 && c < K) ? (int)c : 0;
            }
          }
          if (tid >= nrows) id = 0;
          ids_l[tid] = id;
          if (ft == 0 && tid < nrows) sv.ids[(int64_t)(t + 1) * B + b0 + tid] = id;
        }
        lds_barrier();
      }
    }
    if (t + 1 >= S1) break;
    // ---- head of step j = t + 1: e_j = Embedding(id_j) * keep * 2 (every workgroup gathers the rows), u_j tile = pre_linear(e_j) + BN sums ----
    {
      const int j = t + 1;
      const int idr = ids_l[i];
      for (int ks = wave; ks < KS; ks += 4) {
        const int k = 16 * ks + 4 * q;
        const bool ok = k < H && rvalid;
        float4 e4 = ld4_or_zero(w.emb + (int64_t)idr * H + (ok ? k : 0), ok);
        if (edrop && ok) {
          const uint32_t kp2 = *reinterpret_cast<const uint32_t*>(a.keep_emb + ((int64_t)j * B + b) * H + k);
          e4.x = (kp2 & 0xffu) ? e4.x * 2.0f : 0.f;
          e4.y = (kp2 & 0xff00u) ? e4.y * 2.0f : 0.f;
          e4.z = (kp2 & 0xff0000u) ? e4.z * 2.0f : 0.f;
          e4.w = (kp2 & 0xff000000u) ? e4.w * 2.0f : 0.f;
        }
        xs_u[ks][lane] = e4;
        if (ft == 0 && ok) *reinterpret_cast<float4*>(sv.ec + ((int64_t)j * B + b) * H + k) = e4;
      }
      lds_barrier();
      {
        f32x4 up = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int jj = 0; jj < (KS + 3) / 4; ++jj) {
          const int ks = wave + 4 * jj;
          if (ks < KS) {
            const float4 x4 = xs_u[ks][lane], w4 = wp_s[ks * 64 + lane];
            up = mfma16(w4.x, x4.x, up);
            up = mfma16(w4.y, x4.y, up);
            up = mfma16(w4.z, x4.z, up);
            up = mfma16(w4.w, x4.w, up);
          }
        }
        up_s[wave][lane] = make_float4(up[0], up[1], up[2], up[3]);
      }
      lds_barrier();
      if (wave == 0) {
        float ua[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int pw = 0; pw < 4; ++pw) {
          const float4 p4 = up_s[pw][lane];
          ua[0] += p4.x; ua[1] += p4.y; ua[2] += p4.z; ua[3] += p4.w;
        }
        float s1[4], s2[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float v = (rvalid && fok) ? ua[r] : 0.f;
          s1[r] = reduce16(v);
          s2[r] = reduce16(v * v);
        }
        const unsigned jpar = (unsigned)(j & 1), jtag = (unsigned)(j + 1);
        if (fok) {
          if (rvalid) {
            const float un[4] = {ua[0] + bp4.x, ua[1] + bp4.y, ua[2] + bp4.z, ua[3] + bp4.w};
            cx_publish4(r_u, (jpar * (unsigned)nblk + (unsigned)rg) * rowrec, ft, i, q, un, jtag, l2x);
            *reinterpret_cast<float4*>(sv.u + (int64_t)j * BH + (int64_t)b * H + f0) = make_float4(un[0], un[1], un[2], un[3]);
          }
          if (i == 0) {
            const unsigned g0 = (jpar * (unsigned)nblk + (unsigned)rg) * prec + (unsigned)f0;
            ccl_publish4(r_p, g0, s1, jtag);
            ccl_publish4(r_p, g0 + (unsigned)Hp, s2, jtag);
          }
        }
      }
      lds_barrier();
    }
  }
}

// The BPTT through the two GRU cells of the same rollout as ONE persistent cluster launch (round 5).  Without attention nothing
// but the cells couples two steps (the greedy feedback carries no gradient; BatchNorm's backward and every weight / embedding
// gradient run afterwards over all steps at once), so a step is: cell 1's gate gradients from dh1 = dLogits W_out (formed for all
// steps by one dense launch beforehand) + carry1 | its two 3H-long contractions in the partial-product form of the pose decoder's
// backward cluster (dec_rollout.hip: dec_cluster_bwd_kernel) | cell 0 | its partial products | da_t = dgi0 W_ih0 stored for
// BatchNorm's backward.  Wave 0 owns the element-wise stages and keeps both carries in registers; two exchanges per step.
struct CodeClBwdArgs {
  const float* dh_top;         // (S1,B,H) dLogits_t W_out
  const uint8_t* keep_l0;      // (S1,B,H) or NULL
  g2v_code_dec_weights w; g2v_code_dec_saved sv;
  float* dgi0; float* dgh0; float* dgi1; float* dgh1;      // (S1,B,3H)
  float* da;                   // (S1,B,H) gradient w.r.t. a_t = ReLU(BN(u_t)) (before the ReLU mask)
  float* d_hidden0;            // (2,B,H)
  unsigned long long* xq;      // [nblk][hh1, ih1, hh0, ih0][producer tile] row records: partial products
  unsigned* xcc;               // [nblk][NT]: the XCC every workgroup runs on (cx_cluster_on_one_xcd)
  unsigned* fault;
  int S1, B, H, nt, nblk;
  float p_drop;
};
__device__ __forceinline__ void ccl_cell_bwd(const float (&dh)[4], const float4 (&gt)[4], const float4& hp4, float (&g_r)[4],
                                             float (&g_z)[4], float (&g_n)[4], float (&g_hn)[4], float (&direct)[4]) {
  const float rr[4] = {gt[0].x, gt[0].y, gt[0].z, gt[0].w}, zz[4] = {gt[1].x, gt[1].y, gt[1].z, gt[1].w},
              nn[4] = {gt[2].x, gt[2].y, gt[2].z, gt[2].w}, gh[4] = {gt[3].x, gt[3].y, gt[3].z, gt[3].w},
              hp[4] = {hp4.x, hp4.y, hp4.z, hp4.w};
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const float dn = dh[r] * (1.0f - zz[r]);
    const float dz = dh[r] * (hp[r] - nn[r]);
    const float dnp = dn * (1.0f - nn[r] * nn[r]);
    g_n[r] = dnp;
    g_hn[r] = dnp * rr[r];
    g_r[r] = dnp * gh[r] * rr[r] * (1.0f - rr[r]);
    g_z[r] = dz * zz[r] * (1.0f - zz[r]);
    direct[r] = dh[r] * zz[r];
  }
}
__global__ __launch_bounds__(256) void code_cluster_bptt_kernel(CodeClBwdArgs a) {
  constexpr int KS = CCL_KS, NU = (2 * KS + 3) / 4, NPW = (KS + 1) / 2;
  __shared__ __attribute__((aligned(16))) float4 xs_g[6][64];      // the stage's gate gradients: dgh r z hn, dgi r z n
  __shared__ __attribute__((aligned(16))) float4 dsum[4][64];      // [wave] its share of the partial products, summed
  __shared__ int xcd_flag;
  const int S1 = a.S1, B = a.B, H = a.H, G = 3 * H;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, i = lane & 15, q = lane >> 4;
  // 1-D grid; with a row-group count that is a multiple of 8 the row group is the FAST index, so that under round-robin placement
  // the tile workgroups of a row group share an XCD (verified below) and their records go through that XCD's L2
  const int nt = a.nt, nblk = a.nblk;
  const bool rg_fast = (nblk & 7) == 0;
  const int ft = rg_fast ? (int)blockIdx.x / nblk : (int)blockIdx.x % nt, rg = rg_fast ? (int)blockIdx.x % nblk : (int)blockIdx.x / nt;
  const int b0 = rg * 16;
  const int nrows = min(16, B - b0);
  const bool rvalid = i < nrows;
  const int b = b0 + (rvalid ? i : 0);
  const int f0 = 16 * ft + 4 * q;
  const bool fok = f0 < H, own = rvalid && fok;
  const int64_t BH = (int64_t)B * H, BG = 3 * BH;
  const g2v_code_dec_weights& w = a.w;
  const g2v_code_dec_saved& sv = a.sv;
  // resident: this wave's (matrix, output tile) units of both pairs: rows g H + f0 + e of the matrix at columns 16 ot + i
  float4 wq[2][NU][3];
#pragma unroll
  for (int c = 0; c < 2; ++c)
#pragma unroll
    for (int j = 0; j < NU; ++j) {
      const int u = wave + 4 * j, m = u >= nt ? 1 : 0, ot = u - m * nt;
      const float* W = c == 1 ? (m == 0 ? w.w_hh1 : w.w_ih1) : (m == 0 ? w.w_hh0 : w.w_ih0);
      const int col = 16 * ot + i;
      const bool ok = u < 2 * nt && fok && col < H;
#pragma unroll
      for (int g = 0; g < 3; ++g) {
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = ok ? W[((int64_t)g * H + f0 + e) * H + col] : 0.f;
        wq[c][j][g] = make_float4(v[0], v[1], v[2], v[3]);
      }
    }
  const bool drop = a.keep_l0 && a.p_drop > 0.f;
  const float scale_l0 = 1.0f / (1.0f - a.p_drop);
  const unsigned rowrec = 256u * (unsigned)nt;
  __amdgpu_buffer_rsrc_t r_q = __builtin_amdgcn_make_buffer_rsrc(a.xq, 0, (int)((unsigned)nblk * 4u * (unsigned)nt * rowrec * 8u), 0x00020000);
  const unsigned q_rg = (unsigned)rg * 4u * (unsigned)nt * rowrec;
  const bool l2x = cx_cluster_on_one_xcd(a.xcc + rg * nt, nt, ft, &xcd_flag, tid, a.fault);
  float carry1[4] = {0.f, 0.f, 0.f, 0.f}, carry0[4] = {0.f, 0.f, 0.f, 0.f};      // (wave 0)
  for (int t = S1 - 1; t >= 0; --t) {
    const unsigned tag = (unsigned)(S1 - t);
    const int64_t o4 = (int64_t)t * 4 * BH + (int64_t)b * 4 * H + (fok ? f0 : 0), o1 = (int64_t)t * BH + (int64_t)b * H + (fok ? f0 : 0);
    float direct[4] = {0.f, 0.f, 0.f, 0.f};
    float4 gt0[4], hp0 = make_float4(0.f, 0.f, 0.f, 0.f);
    uint32_t kp = 0x01010101u;
    if (wave == 0) {
      // ---- cell 1: dh1 = dLogits_t W_out + carry1; its gate gradients -------------------------------------------------------------------
      float4 gt1[4];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        gt1[g] = ld4_or_zero(sv.gates1 + o4 + g * H, own);
        gt0[g] = ld4_or_zero(sv.gates0 + o4 + g * H, own);
      }
      const float4 hp1 = ld4_or_zero(sv.h1 + o1, own);      // (slot t: the state in front of step t)
      hp0 = ld4_or_zero(sv.h0 + o1, own);
      if (drop && own) kp = *reinterpret_cast<const uint32_t*>(a.keep_l0 + o1);
      const float4 d4 = ld4_or_zero(a.dh_top + o1, own);
      float dh[4] = {d4.x + carry1[0], d4.y + carry1[1], d4.z + carry1[2], d4.w + carry1[3]};
      float g_r[4], g_z[4], g_n[4], g_hn[4];
      ccl_cell_bwd(dh, gt1, hp1, g_r, g_z, g_n, g_hn, direct);
      const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
      const float4 vr = own ? make_float4(g_r[0], g_r[1], g_r[2], g_r[3]) : z4, vz = own ? make_float4(g_z[0], g_z[1], g_z[2], g_z[3]) : z4;
      const float4 vn = own ? make_float4(g_n[0], g_n[1], g_n[2], g_n[3]) : z4, vh = own ? make_float4(g_hn[0], g_hn[1], g_hn[2], g_hn[3]) : z4;
      xs_g[0][lane] = vr; xs_g[1][lane] = vz; xs_g[2][lane] = vh;
      xs_g[3][lane] = vr; xs_g[4][lane] = vz; xs_g[5][lane] = vn;
      if (own) {
        float* gi = a.dgi1 + (int64_t)t * BG + (int64_t)b * G + f0;
        float* gh = a.dgh1 + (int64_t)t * BG + (int64_t)b * G + f0;
        *reinterpret_cast<float4*>(gi) = vr; *reinterpret_cast<float4*>(gi + H) = vz; *reinterpret_cast<float4*>(gi + 2 * H) = vn;
        *reinterpret_cast<float4*>(gh) = vr; *reinterpret_cast<float4*>(gh + H) = vz; *reinterpret_cast<float4*>(gh + 2 * H) = vh;
      }
    }
#pragma unroll
    for (int c = 1; c >= 0; --c) {
      lds_barrier();      // xs_g of the stage is complete
      {
        const float4 xh0 = xs_g[0][lane], xh1 = xs_g[1][lane], xh2 = xs_g[2][lane];
        const float4 xi0 = xs_g[3][lane], xi1 = xs_g[4][lane], xi2 = xs_g[5][lane];
#pragma unroll
        for (int j = 0; j < NU; ++j) {
          const int u = wave + 4 * j, m = u >= nt ? 1 : 0, ot = u - m * nt;
          if (u < 2 * nt) {      // (uniform)
            const float4 x0 = m ? xi0 : xh0, x1 = m ? xi1 : xh1, x2 = m ? xi2 : xh2;
            f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
            acc = mfma16(wq[c][j][0].x, x0.x, acc); acc = mfma16(wq[c][j][0].y, x0.y, acc);
            acc = mfma16(wq[c][j][0].z, x0.z, acc); acc = mfma16(wq[c][j][0].w, x0.w, acc);
            acc = mfma16(wq[c][j][1].x, x1.x, acc); acc = mfma16(wq[c][j][1].y, x1.y, acc);
            acc = mfma16(wq[c][j][1].z, x1.z, acc); acc = mfma16(wq[c][j][1].w, x1.w, acc);
            acc = mfma16(wq[c][j][2].x, x2.x, acc); acc = mfma16(wq[c][j][2].y, x2.y, acc);
            acc = mfma16(wq[c][j][2].z, x2.z, acc); acc = mfma16(wq[c][j][2].w, x2.w, acc);
            if (rvalid && 16 * ot + 4 * q < H) {
              const float v[4] = {acc[0], acc[1], acc[2], acc[3]};
              cx_publish4(r_q, q_rg + ((unsigned)((1 - c) * 2 + m) * (unsigned)nt + (unsigned)ft) * rowrec, ot, i, q, v, tag, l2x);
            }
          }
        }
      }
      cx_sweep_tile_sum<NPW>(r_q, q_rg + (unsigned)((1 - c) * 2 + (wave & 1)) * (unsigned)nt * rowrec, rowrec, wave >> 1, 2, nt, ft, nrows, H, tag,
                             &dsum[wave][0], lane, a.fault);
      lds_barrier();
      if (wave == 0) {
        const float4 h_a = dsum[0][lane], h_b = dsum[2][lane], i_a = dsum[1][lane], i_b = dsum[3][lane];
        const float shh[4] = {h_a.x + h_b.x, h_a.y + h_b.y, h_a.z + h_b.z, h_a.w + h_b.w};
        const float sih[4] = {i_a.x + i_b.x, i_a.y + i_b.y, i_a.z + i_b.z, i_a.w + i_b.w};
        if (c == 1) {
          float dh[4], g_r[4], g_z[4], g_n[4], g_hn[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            carry1[r] = direct[r] + shh[r];
            float v = sih[r];
            if (drop) v = ((kp >> (8 * r)) & 0xffu) ? v * scale_l0 : 0.f;
            dh[r] = v + carry0[r];
          }
          ccl_cell_bwd(dh, gt0, hp0, g_r, g_z, g_n, g_hn, direct);
          const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
          const float4 vr = own ? make_float4(g_r[0], g_r[1], g_r[2], g_r[3]) : z4, vz = own ? make_float4(g_z[0], g_z[1], g_z[2], g_z[3]) : z4;
          const float4 vn = own ? make_float4(g_n[0], g_n[1], g_n[2], g_n[3]) : z4, vh = own ? make_float4(g_hn[0], g_hn[1], g_hn[2], g_hn[3]) : z4;
          xs_g[0][lane] = vr; xs_g[1][lane] = vz; xs_g[2][lane] = vh;
          xs_g[3][lane] = vr; xs_g[4][lane] = vz; xs_g[5][lane] = vn;
          if (own) {
            float* gi = a.dgi0 + (int64_t)t * BG + (int64_t)b * G + f0;
            float* gh = a.dgh0 + (int64_t)t * BG + (int64_t)b * G + f0;
            *reinterpret_cast<float4*>(gi) = vr; *reinterpret_cast<float4*>(gi + H) = vz; *reinterpret_cast<float4*>(gi + 2 * H) = vn;
            *reinterpret_cast<float4*>(gh) = vr; *reinterpret_cast<float4*>(gh + H) = vz; *reinterpret_cast<float4*>(gh + 2 * H) = vh;
          }
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r) carry0[r] = direct[r] + shh[r];
          if (own) *reinterpret_cast<float4*>(a.da + o1) = make_float4(sih[0], sih[1], sih[2], sih[3]);
        }
      }
    }
    lds_barrier();      // (xs_g / dsum of this step are done with)
  }
  if (wave == 0 && own) {
    *reinterpret_cast<float4*>(a.d_hidden0 + (int64_t)b * H + f0) = make_float4(carry0[0], carry0[1], carry0[2], carry0[3]);
    *reinterpret_cast<float4*>(a.d_hidden0 + BH + (int64_t)b * H + f0) = make_float4(carry1[0], carry1[1], carry1[2], carry1[3]);
  }
}
}  // namespace g2v

using namespace g2v;

// ---------------------------------------------------------------------------------------------------------------------
// host
// ---------------------------------------------------------------------------------------------------------------------
static size_t al256(size_t b) { return (b + 255) & ~(size_t)255; }
static bool ct_al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

extern "C" int g2v_attn_code_rollout_blocks(int B) { return B > 0 ? cdiv(B, 16) : 0; }

extern "C" int g2v_attn_code_rollout_ok(int S1, int B, int H, int K, int Tw, int attention) {
  if (S1 < 1 || B < 1 || H < 16 || (H & 3) || H > 256 || K < 4 || (K & 3) || K > 1024) return 0;
  if (attention && (Tw < 1 || Tw > CT_MAX_TW)) return 0;
  const int Hin = attention ? 2 * H : H;
  const size_t lf = (size_t)ct_fwd_lds(H, Hin, attention ? Tw : 0, 1024).total * 4;
  const size_t lb = (size_t)ct_bwd_lds(H, K).total * 4;
  const int Hp = (H + 15) & ~15;
  if (attention && (2 * Hp + 1024 > 16 * (((3 * H + 15) & ~15) + 4) || 2 * 16 * Tw > 16 * (Hp + 4))) return 0;   // what Part A borrows
  return lf <= 160 * 1024 && lb <= 160 * 1024;
}

static size_t ct_fwd_pack_floats(int H, int K, int att) {
  const int Hin = att ? 2 * H : H;
  return pack_floats(H, 1, Hin) + 4 * pack_floats(H, 3, H) + (size_t)ct_ktiles_alloc(K) * pack_ks(H) * 256 +
         (att ? pack_floats(H, 1, H) : 0);
}
// exchange records of code_cluster_fwd_kernel: three row records + the BatchNorm sums + the argmax pairs per (parity, row group)
static size_t code_cluster_xch_bytes(int nblk, int H) {
  const size_t Hp = (size_t)((H + 15) & ~15), nt = Hp / 16;
  return (size_t)2 * nblk * ((3 * 16 + 2) * Hp + nt * 32) * 8 + (size_t)nblk * nt * sizeof(unsigned) + 16;
}
static size_t code_cluster_rec_bytes(int nblk, int H) { return code_cluster_xch_bytes(nblk, H) - ((size_t)nblk * ((H + 15) >> 4) * sizeof(unsigned) + 16); }
static size_t code_cluster_dyn_lds() { return ((size_t)CCL_KT * CCL_KS + CCL_KS + (size_t)4 * CCL_KT + CCL_KS) * 64 * sizeof(float4); }
static int ct_device_cus() {
  static int n = -1;
  if (n < 0) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) n = 0;
  }
  return n;
}
int g2v_internal_persist_enabled();      // dec_rollout.hip: g2v_dec_rollout_set_persistent != 0
// 1: g2v_attn_code_rollout_fwd runs this shape as ONE persistent cluster launch (no attention, H <= 208, at most three K tiles per
// hidden-unit tile, the (tile x row group) grid with a CU per workgroup) under the current g2v_dec_rollout_set_persistent setting
extern "C" int g2v_attn_code_rollout_cluster_ok(int S1, int B, int H, int K, int attention) {
  if (attention || S1 < 1 || B < 1 || H < 16 || (H & 3) || H > 16 * CCL_KS || K < 4 || (K & 3)) return 0;
  const int nt = (H + 15) >> 4, nblk = (B + 15) >> 4;
  if (((K + 15) >> 4) > CCL_KT * nt || (int64_t)nt * nblk > ct_device_cus()) return 0;
  return g2v_internal_persist_enabled() ? 1 : 0;
}
// The cells' BPTT of the attention-free rollout at small batch as one persistent cluster launch (code_cluster_bptt_kernel; the
// shapes of g2v_attn_code_rollout_cluster_ok).  dh_top (S1,B,H) = dLogits W_out for every step (g2v_linear_bwd_data); writes dgi0 /
// dgh0 / dgi1 / dgh1 (S1,B,3H), da (S1,B,H) = the gradient w.r.t. a_t before its ReLU mask (BatchNorm's backward is the caller's
// next launch), d_hidden0 (2,B,H).  s: gates0 / gates1 / h0 / h1 of the forward.
extern "C" size_t g2v_code_cluster_bptt_workspace(int B, int H) {
  const size_t Hp = (size_t)((H + 15) & ~15), nt = Hp / 16;
  return (size_t)((B + 15) / 16) * 4 * nt * 16 * Hp * 8 + (size_t)((B + 15) / 16) * nt * sizeof(unsigned) + 16;
}
extern "C" int g2v_code_cluster_bptt(const float* dh_top, const g2v_code_dec_weights* w, const g2v_code_dec_saved* s,
                                     const uint8_t* keep_l0, float p_drop, float* dgi0, float* dgh0, float* dgi1, float* dgh1,
                                     float* da, float* d_hidden0, int S1, int B, int H, void* workspace, size_t workspace_bytes,
                                     g2v_stream_t stream) {
  G2V_REQUIRE(dh_top && w && s && dgi0 && dgh0 && dgi1 && dgh1 && da && d_hidden0 && workspace, "null pointer");
  G2V_REQUIRE(w->w_ih0 && w->w_hh0 && w->w_ih1 && w->w_hh1 && s->gates0 && s->gates1 && s->h0 && s->h1, "missing array");
  G2V_REQUIRE(p_drop >= 0.f && p_drop < 1.f, "bad dropout probability");
  if (!g2v_attn_code_rollout_cluster_ok(S1, B, H, 4, 0)) {
    set_error("g2v_code_cluster_bptt: shape not served (g2v_attn_code_rollout_cluster_ok)");
    return G2V_ERR_UNSUPPORTED;
  }
  if (workspace_bytes < g2v_code_cluster_bptt_workspace(B, H)) {
    set_error("g2v_code_cluster_bptt: workspace too small");
    return G2V_ERR_WORKSPACE;
  }
  const void* al[] = {dh_top, keep_l0, dgi0, dgh0, dgi1, dgh1, da, d_hidden0, s->gates0, s->gates1, s->h0, s->h1, workspace};
  for (const void* p : al) G2V_REQUIRE(ct_al16(p), "16-byte alignment");
  hipStream_t st = (hipStream_t)stream;
  const void* fn = (const void*)code_cluster_bptt_kernel;
  int nocc = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nocc, fn, 256, 0) != hipSuccess || nocc < 1) {
    set_error("g2v_code_cluster_bptt: the kernel does not fit a CU");
    return G2V_ERR_LAUNCH;
  }
  CodeClBwdArgs ca;
  ca.dh_top = dh_top; ca.keep_l0 = keep_l0; ca.w = *w; ca.sv = *s;
  ca.dgi0 = dgi0; ca.dgh0 = dgh0; ca.dgi1 = dgi1; ca.dgh1 = dgh1; ca.da = da; ca.d_hidden0 = d_hidden0;
  ca.xq = reinterpret_cast<unsigned long long*>(workspace);
  ca.nt = (H + 15) >> 4; ca.nblk = cdiv(B, 16);
  ca.xcc = reinterpret_cast<unsigned*>((char*)workspace + (size_t)ca.nblk * 4 * ca.nt * 16 * (ca.nt * 16) * 8);
  ca.fault = const_cast<unsigned*>(g2v_internal_persist_fault_ptr());
  ca.S1 = S1; ca.B = B; ca.H = H; ca.p_drop = p_drop;
  if (hipMemsetAsync(workspace, 0, g2v_code_cluster_bptt_workspace(B, H), st) != hipSuccess) {
    set_error("g2v_code_cluster_bptt: clearing the exchange records failed");
    return G2V_ERR_LAUNCH;
  }
  hipLaunchKernelGGL(code_cluster_bptt_kernel, dim3(((H + 15) >> 4) * cdiv(B, 16)), dim3(256), 0, st, ca);
  G2V_CHECK_LAUNCH();
  return G2V_OK;
}
extern "C" size_t g2v_attn_code_rollout_fwd_workspace(int H, int K, int attention) {
  size_t x = al256(ct_fwd_pack_floats(H, K, attention) * sizeof(float));
  if (!attention && (H & 3) == 0 && H >= 16 && H <= 16 * CCL_KS) {      // the cluster kernel's exchange records at its largest grid
    const int nt = (H + 15) >> 4, cus = ct_device_cus() > 0 ? ct_device_cus() : 256;
    const size_t c = code_cluster_xch_bytes(cus / nt + 1, H);
    if (c > x) x = c;
  }
  return x;
}

static int ct_scratch(size_t base_floats) { return (base_floats + 2048) * sizeof(float) <= 160 * 1024 ? 2048 : 1024; }

extern "C" int g2v_attn_code_rollout_fwd(const int64_t* codes, const float* h_init, const float* enc, const float* enc_proj,
                                         const g2v_code_dec_weights* w, const g2v_code_dec_saved* s, const uint8_t* keep_emb,
                                         const uint8_t* keep_l0, float p_drop, int n_pre, int training, int S1, int B, int H,
                                         int K, int Tw, void* workspace, size_t workspace_bytes, g2v_stream_t stream) {
  G2V_REQUIRE(codes && h_init && w && s && workspace, "null pointer");
  const int att = w->w_attn != nullptr;
  G2V_REQUIRE((training ? (w->bn_running_mean == nullptr) == (w->bn_running_var == nullptr) : (w->bn_running_mean && w->bn_running_var)),
              "BatchNorm running statistics: both or (training only) neither");
  G2V_REQUIRE(w->emb && w->w_pre && w->b_pre && w->bn_w && w->bn_b && w->w_ih0 &&
              w->w_hh0 && w->b_ih0 && w->b_hh0 && w->w_ih1 && w->w_hh1 && w->b_ih1 && w->b_hh1 && w->w_out && w->b_out,
              "missing weight");
  G2V_REQUIRE(!att || (w->b_attn && w->v_attn && enc && enc_proj && s->hp && s->attw), "attention: missing array");
  G2V_REQUIRE(s->ids && s->ec && s->u && s->h0 && s->h1 && s->logits && s->bn_partial, "missing state buffer");
  G2V_REQUIRE(!training || (s->a && s->bn_stats && s->gates0 && s->gates1), "missing saved buffer");
  G2V_REQUIRE(p_drop >= 0.f && p_drop < 1.f, "bad dropout probability");
  if (!g2v_attn_code_rollout_ok(S1, B, H, K, Tw, att)) {
    set_error("g2v_attn_code_rollout_fwd: shape not served (g2v_attn_code_rollout_ok)");
    return G2V_ERR_UNSUPPORTED;
  }
  if (workspace_bytes < g2v_attn_code_rollout_fwd_workspace(H, K, att)) {
    set_error("g2v_attn_code_rollout_fwd: workspace too small");
    return G2V_ERR_WORKSPACE;
  }
  const void* al[] = {h_init, enc, enc_proj, w->emb, w->b_pre, w->bn_w, w->bn_b, w->b_ih0, w->b_hh0, w->b_ih1, w->b_hh1, w->b_out,
                      w->b_attn, w->v_attn, s->ec, s->u, s->a, s->h0, s->h1, s->x1, s->gates0, s->gates1, s->logits, s->bn_partial,
                      s->hp, keep_emb, keep_l0, workspace};
  for (const void* p : al) G2V_REQUIRE(ct_al16(p), "16-byte alignment");
  hipStream_t st = (hipStream_t)stream;
  const int Hin = att ? 2 * H : H;
  if (g2v_attn_code_rollout_cluster_ok(S1, B, H, K, att) && code_cluster_xch_bytes(cdiv(B, 16), H) <= workspace_bytes &&
      (!training || (s->a && s->bn_stats && s->gates0 && s->gates1))) {
    // small batch: ONE persistent launch of (hidden-unit tile x row group) workgroups (code_cluster_fwd_kernel)
    static bool attr_set = false;
    const void* fn = (const void*)code_cluster_fwd_kernel;
    int nocc = 0;
    if (!attr_set) attr_set = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)code_cluster_dyn_lds()) == hipSuccess;
    if (attr_set && hipOccupancyMaxActiveBlocksPerMultiprocessor(&nocc, fn, 256, code_cluster_dyn_lds()) == hipSuccess && nocc >= 1) {
      const int nblk = cdiv(B, 16), nt = (H + 15) >> 4;
      const size_t Hp = (size_t)nt * 16, rowrec = (size_t)2 * nblk * 16 * Hp;
      unsigned long long* x0 = reinterpret_cast<unsigned long long*>(workspace);
      CodeClArgs ca;
      ca.codes = codes; ca.h_init = h_init; ca.keep_emb = keep_emb; ca.keep_l0 = keep_l0; ca.w = *w; ca.sv = *s;
      ca.xu = x0; ca.xh0 = x0 + rowrec; ca.xh1 = x0 + 2 * rowrec; ca.xp = x0 + 3 * rowrec; ca.xa = ca.xp + (size_t)2 * nblk * 2 * Hp;
      ca.xcc = reinterpret_cast<unsigned*>((char*)workspace + code_cluster_rec_bytes(nblk, H));
      ca.fault = const_cast<unsigned*>(g2v_internal_persist_fault_ptr());
      ca.S1 = S1; ca.B = B; ca.H = H; ca.K = K; ca.n_pre = n_pre; ca.training = training; ca.p_drop = p_drop;
      ca.nt = nt; ca.nblk = nblk;
      if (hipMemsetAsync(x0, 0, code_cluster_xch_bytes(nblk, H), st) != hipSuccess) {
        set_error("g2v_attn_code_rollout_fwd: clearing the exchange records failed");
        return G2V_ERR_LAUNCH;
      }
      hipLaunchKernelGGL(code_cluster_fwd_kernel, dim3(nt * nblk), dim3(256), code_cluster_dyn_lds(), st, ca);
      G2V_CHECK_LAUNCH();
      return G2V_OK;
    }
  }
  // ---- pack the weights into MFMA fragment order (one launch) ----
  float* p = (float*)workspace;
  PackBatch pb;
  CodePackF pk{};
  pb.n = 0;
  pb.d[pb.n++] = PackDesc{w->w_pre, p, H, 1, 0, Hin, Hin, 0, 0}; pk.pre = p; p += pack_floats(H, 1, Hin);
  pb.d[pb.n++] = PackDesc{w->w_ih0, p, H, 3, H, H, H, 0, 0}; pk.ih0 = p; p += pack_floats(H, 3, H);
  pb.d[pb.n++] = PackDesc{w->w_hh0, p, H, 3, H, H, H, 0, 0}; pk.hh0 = p; p += pack_floats(H, 3, H);
  pb.d[pb.n++] = PackDesc{w->w_ih1, p, H, 3, H, H, H, 0, 0}; pk.ih1 = p; p += pack_floats(H, 3, H);
  pb.d[pb.n++] = PackDesc{w->w_hh1, p, H, 3, H, H, H, 0, 0}; pk.hh1 = p; p += pack_floats(H, 3, H);
  pb.d[pb.n++] = PackDesc{w->w_out, p, K, 1, 0, H, H, 0, ct_ktiles_alloc(K)}; pk.out = p; p += (size_t)ct_ktiles_alloc(K) * pack_ks(H) * 256;
  if (att) {
    pb.d[pb.n++] = PackDesc{w->w_attn, p, H, 1, 0, H, 2 * H, 0, 0}; pk.attn_h = p; p += pack_floats(H, 1, H);
  }
  launch_pack(pb, st);
  G2V_CHECK_LAUNCH();
  const int scratch = ct_scratch((size_t)ct_fwd_lds(H, Hin, att ? Tw : 0, 0).total);
  CodeDims dm{S1, B, H, K, Hin, att ? Tw : 0, p_drop, n_pre, training, cdiv(B, 16), att, scratch};
  const size_t lds = (size_t)ct_fwd_lds(H, Hin, dm.Tw, scratch).total * sizeof(float);
  if (lds > 48 * 1024)
    (void)hipFuncSetAttribute((const void*)code_step_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  for (int j = 0; j <= S1; ++j)
    hipLaunchKernelGGL(code_step_fwd_kernel, dim3(dm.nblk), dim3(CT_NTHR), lds, st, codes, h_init, enc, enc_proj, *w, pk, *s,
                       keep_emb, keep_l0, dm, j);
  G2V_CHECK_LAUNCH();
  return G2V_OK;
}

// ---- backward workspace layout (bytes, every region 256-byte aligned) ----------------------------------------------------
struct CtBwdWs {
  size_t pack, dgi0, dgh0, dgi1, dgh1, dbn, du, dec, dhp, d_ep, dvp, bn_part, bn_sums, x1m, wg, emb, total;
};
static CtBwdWs ct_bwd_ws(int S1, int B, int H, int K, int Tw, int att) {
  const size_t M = (size_t)S1 * B, G = 3 * (size_t)H, Hin = att ? 2 * (size_t)H : (size_t)H, nblk = cdiv(B, 16);
  CtBwdWs l;
  size_t o = 0;
  auto take = [&](size_t bytes) { const size_t at = o; o += al256(bytes); return at; };
  l.pack = take((pack_floats(H, 1, K) + 4 * pack_floats(H, 1, 3 * H) + (att ? 2 * pack_floats(H, 1, H) : 0)) * 4);
  l.dgi0 = take(M * G * 4); l.dgh0 = take(M * G * 4); l.dgi1 = take(M * G * 4); l.dgh1 = take(M * G * 4);
  l.dbn = take(M * H * 4);
  l.du = take(M * H * 4);
  l.dec = take(M * H * 4);
  l.dhp = take(att ? M * H * 4 : 0);
  l.d_ep = take(att ? (size_t)Tw * B * H * 4 : 0);
  l.dvp = take(att ? nblk * H * 4 : 0);
  l.bn_part = take((size_t)S1 * nblk * 2 * H * 4);
  l.bn_sums = take((size_t)S1 * 2 * H * 4);
  l.x1m = take(att ? (size_t)4 * H * H * 4 : 0);      // attention: W_pre[:, :H], W_attn[:, H:], dW_h, dW_e
  size_t wg = 4 * g2v_linear_bwd_weight_workspace((int)M, H, 3 * H);
  wg = wg > g2v_linear_bwd_weight_workspace((int)M, H, K) ? wg : g2v_linear_bwd_weight_workspace((int)M, H, K);
  wg = wg > g2v_linear_bwd_weight_workspace((int)M, (int)Hin, H) ? wg : g2v_linear_bwd_weight_workspace((int)M, (int)Hin, H);
  if (att) {
    const size_t w2 = g2v_linear_bwd_weight_workspace(Tw * B, H, H);
    wg = wg > w2 ? wg : w2;
  }
  l.wg = take(wg);
  l.emb = take(g2v_embedding_bwd_ws_bytes((int64_t)M, H, K));
  l.total = o;
  return l;
}
extern "C" size_t g2v_attn_code_rollout_bwd_workspace(int S1, int B, int H, int K, int Tw, int attention) {
  if (S1 < 1 || B < 1 || H < 1 || K < 1) return 0;
  return ct_bwd_ws(S1, B, H, K, Tw, attention).total;
}

extern "C" int g2v_attn_code_rollout_bwd(const float* d_logits, const float* enc, const float* enc_proj,
                                         const g2v_code_dec_weights* w, const g2v_code_dec_saved* s, const g2v_code_dec_grads* g,
                                         const uint8_t* keep_emb, const uint8_t* keep_l0, float p_drop, int S1, int B, int H, int K,
                                         int Tw, void* workspace, size_t workspace_bytes, g2v_stream_t stream) {
  G2V_REQUIRE(d_logits && w && s && g && workspace, "null pointer");
  const int att = w->w_attn != nullptr;
  G2V_REQUIRE(s->ids && s->ec && s->u && s->a && s->bn_stats && s->h0 && s->h1 && s->gates0 && s->gates1, "missing saved buffer");
  G2V_REQUIRE(g->d_hidden0 && g->d_emb && g->d_w_pre && g->d_b_pre && g->d_bn_w && g->d_bn_b && g->d_w_ih0 && g->d_w_hh0 &&
              g->d_b_ih0 && g->d_b_hh0 && g->d_w_ih1 && g->d_w_hh1 && g->d_b_ih1 && g->d_b_hh1 && g->d_w_out && g->d_b_out,
              "missing gradient buffer");
  G2V_REQUIRE(!att || (enc && enc_proj && s->hp && s->attw && g->d_w_attn && g->d_b_attn && g->d_v_attn && g->d_enc),
              "attention: missing array");
  const bool drop = keep_l0 && p_drop > 0.f;
  G2V_REQUIRE(!drop || s->x1, "inter-layer dropout: x1 was not saved");
  if (!g2v_attn_code_rollout_ok(S1, B, H, K, Tw, att)) {
    set_error("g2v_attn_code_rollout_bwd: shape not served (g2v_attn_code_rollout_ok)");
    return G2V_ERR_UNSUPPORTED;
  }
  const CtBwdWs L = ct_bwd_ws(S1, B, H, K, Tw, att);
  if (workspace_bytes < L.total) {
    set_error("g2v_attn_code_rollout_bwd: workspace too small");
    return G2V_ERR_WORKSPACE;
  }
  G2V_REQUIRE(ct_al16(workspace) && ct_al16(d_logits) && ct_al16(g->d_hidden0), "16-byte alignment");
  hipStream_t st = (hipStream_t)stream;
  char* base = (char*)workspace;
  const int Hin = att ? 2 * H : H, G = 3 * H, M = S1 * B;
  // ---- transposed packs ----
  float* p = (float*)(base + L.pack);
  PackBatch pb;
  CodeBwdArgs a{};
  pb.n = 0;
  pb.d[pb.n++] = PackDesc{w->w_out, p, H, 1, 0, K, H, 1, 0}; a.tw.out_t = p; p += pack_floats(H, 1, K);      // rows f, k = code: W_out[k][f]
  pb.d[pb.n++] = PackDesc{w->w_ih0, p, H, 1, 0, G, H, 1, 0}; a.tw.ih0_t = p; p += pack_floats(H, 1, G);
  pb.d[pb.n++] = PackDesc{w->w_hh0, p, H, 1, 0, G, H, 1, 0}; a.tw.hh0_t = p; p += pack_floats(H, 1, G);
  pb.d[pb.n++] = PackDesc{w->w_ih1, p, H, 1, 0, G, H, 1, 0}; a.tw.ih1_t = p; p += pack_floats(H, 1, G);
  pb.d[pb.n++] = PackDesc{w->w_hh1, p, H, 1, 0, G, H, 1, 0}; a.tw.hh1_t = p; p += pack_floats(H, 1, G);
  if (att) {
    // rows c (context feature), k = f: W_pre[f][H + c]  -> src offset H, ld = 2H, transposed
    pb.d[pb.n++] = PackDesc{w->w_pre + H, p, H, 1, 0, H, 2 * H, 1, 0}; a.tw.pre_ctx_t = p; p += pack_floats(H, 1, H);
    // rows c (state feature), k = f: W_attn[f][c]
    pb.d[pb.n++] = PackDesc{w->w_attn, p, H, 1, 0, H, 2 * H, 1, 0}; a.tw.attn_h_t = p; p += pack_floats(H, 1, H);
  }
  launch_pack(pb, st);
  G2V_CHECK_LAUNCH();
  a.d_logits = d_logits; a.enc = enc; a.ep = enc_proj; a.w = *w; a.sv = *s; a.keep_l0 = drop ? keep_l0 : nullptr;
  a.dgi0 = (float*)(base + L.dgi0); a.dgh0 = (float*)(base + L.dgh0); a.dgi1 = (float*)(base + L.dgi1); a.dgh1 = (float*)(base + L.dgh1);
  a.dbn = (float*)(base + L.dbn); a.du = (float*)(base + L.du); a.dec = (float*)(base + L.dec);
  a.dhp = att ? (float*)(base + L.dhp) : nullptr; a.d_ep = att ? (float*)(base + L.d_ep) : nullptr; a.d_enc = att ? g->d_enc : nullptr;
  a.dv_partial = att ? (float*)(base + L.dvp) : nullptr;
  a.bn_part = (float*)(base + L.bn_part); a.bn_sums = (float*)(base + L.bn_sums); a.d_hidden0 = g->d_hidden0;
  const int Hp = (H + 15) & ~15;
  const int scratch = (2 * Hp + 2048 <= 16 * (((3 * H + 15) & ~15) + 4)) ? 2048 : 1024;      // (Part A's reduction scratch lives in Gh)
  CodeDims dm{S1, B, H, K, Hin, att ? Tw : 0, drop ? p_drop : 0.f, 0, 1, cdiv(B, 16), att, scratch};
  const size_t lds = (size_t)ct_bwd_lds(H, K).total * sizeof(float);
  void* wgws = base + L.wg;
  const size_t wgn = L.emb - L.wg;
  int rc;
  float* wpe = (float*)(base + L.x1m);      // attention: W_pre[:, :H] | W_attn[:, H:] | dW_h | dW_e, H x H each
  float* wae = wpe + (size_t)H * H;
  float* dwh = wae + (size_t)H * H;
  float* dwe = dwh + (size_t)H * H;
  if (!att) {
    if (lds > 48 * 1024) (void)hipFuncSetAttribute((const void*)code_bptt_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(code_bptt_kernel, dim3(dm.nblk), dim3(CT_NTHR), lds, st, a, dm);
    hipLaunchKernelGGL(code_bn_bwd_apply_kernel, dim3(dm.nblk, S1), dim3(256), (2 * H + 1024) * sizeof(float), st, a, dm);
    G2V_CHECK_LAUNCH();
    // d e = du W_pre over all S1 x B rows
    if ((rc = g2v_linear_bwd_data(a.du, H, w->w_pre, a.dec, H, M, H, H, 0, stream)) != G2V_OK) return rc;
  } else {
    hipLaunchKernelGGL(code_split_cols_kernel, dim3(cdiv(H * H, 256)), dim3(256), 0, st, w->w_pre, w->w_attn, wpe, wae, H);
    if (lds > 48 * 1024)
      (void)hipFuncSetAttribute((const void*)code_step_bwd_att_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    for (int t = S1 - 1; t >= -1; --t)
      hipLaunchKernelGGL(code_step_bwd_att_kernel, dim3(dm.nblk), dim3(CT_NTHR), lds, st, a, dm, t);
    G2V_CHECK_LAUNCH();
    // the embedding half of d x = du W_pre[:, :H] (the context half was formed step by step)
    if ((rc = g2v_linear_bwd_data(a.du, H, wpe, a.dec, H, M, H, H, 0, stream)) != G2V_OK) return rc;
  }
  hipLaunchKernelGGL(code_small_sums_kernel, dim3(1), dim3(256), 0, st, a.bn_sums, S1, H, g->d_bn_w, g->d_bn_b,
                     att ? a.dv_partial : nullptr, dm.nblk, att ? g->d_v_attn : nullptr);
  G2V_CHECK_LAUNCH();
  // ---- embedding gradient, weight gradients: one launch over the S1 x B rows each -------------------------------------------
  if ((rc = g2v_embedding_bwd(a.dec, s->ids, keep_emb, keep_emb ? 2.0f : 1.0f, g->d_emb, M, H, K, 1, base + L.emb,
                              L.total - L.emb, stream)) != G2V_OK)
    return rc;
  if ((rc = g2v_linear_bwd_weight(d_logits, K, s->h1 + (size_t)B * H, H, 0, 0, 0, nullptr, 1.0f, g->d_w_out, g->d_b_out, M, H, K, 0,
                                  wgws, wgn, stream)) != G2V_OK)
    return rc;
  if ((rc = g2v_linear_bwd_weight(a.du, H, s->ec, Hin, 0, 0, 0, nullptr, 1.0f, g->d_w_pre, g->d_b_pre, M, Hin, H, 0, wgws, wgn,
                                  stream)) != G2V_OK)
    return rc;
  {
    g2v_wgrad_item it[4];
    const float* x1 = drop ? s->x1 : s->h0 + (size_t)B * H;     // layer 1's input: dropped h0_{t+1}
    it[0] = g2v_wgrad_item{a.dgi0, s->a, g->d_w_ih0, g->d_b_ih0};
    it[1] = g2v_wgrad_item{a.dgh0, s->h0, g->d_w_hh0, g->d_b_hh0};
    it[2] = g2v_wgrad_item{a.dgi1, x1, g->d_w_ih1, g->d_b_ih1};
    it[3] = g2v_wgrad_item{a.dgh1, s->h1, g->d_w_hh1, g->d_b_hh1};
    if ((rc = g2v_linear_bwd_weight_batch(it, 4, G, H, M, H, G, 0, wgws, wgn, stream)) != G2V_OK) return rc;
  }
  if (att) {
    // attn.attn.weight = [W_h | W_e]: d W_h = dhp^T h1_t (the state in front of each step), d W_e = d_ep^T enc; bias from dhp
    if ((rc = g2v_linear_bwd_weight(a.dhp, H, s->h1, H, 0, 0, 0, nullptr, 1.0f, dwh, g->d_b_attn, M, H, H, 0, wgws, wgn, stream)) !=
        G2V_OK)
      return rc;
    if ((rc = g2v_linear_bwd_weight(a.d_ep, H, enc, H, 0, 0, 0, nullptr, 1.0f, dwe, nullptr, Tw * B, H, H, 0, wgws, wgn, stream)) !=
        G2V_OK)
      return rc;
    hipLaunchKernelGGL(code_join_cols_kernel, dim3(cdiv(H * H, 256)), dim3(256), 0, st, dwh, dwe, g->d_w_attn, H);
    G2V_CHECK_LAUNCH();
    // d enc (the context term, written by the step kernels) += d_ep W_attn[:, H:]
    if ((rc = g2v_linear_bwd_data(a.d_ep, H, wae, g->d_enc, H, Tw * B, H, H, 1, stream)) != G2V_OK) return rc;
  }
  return G2V_OK;
}
