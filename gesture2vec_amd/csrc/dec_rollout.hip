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
#include <stdlib.h>

#include "dec_persist.hpp"
#include "gru_cells.hpp"

// dec_persist.hip
int dec_persist_fwd_launch(const float* target, const float* h_init, const g2v_dec_weights* w, const g2v_dec_saved* s,
                           const uint8_t* keep95, const uint8_t* keep_l0, float p_drop, int n_pre, int conditioned,
                           int training, int T, int B, const float* p_pre, const float* p_ih0, const float* p_hh0,
                           const float* p_ih1, const float* p_hh1, const float* p_out, void* xbase, hipStream_t st,
                           bool clear, int tiles_per_wg);
int dec_persist_bwd_launch(const g2v_dec_weights* w, const g2v_dec_saved* s, const g2v_dec_grads* g, const uint8_t* keep95,
                           const uint8_t* keep_l0, float p_drop, int n_pre, int conditioned, int T, int B,
                           const float* p_pre_t, const float* p_out_t, const float* p_ih0_t, const float* p_hh0_t,
                           const float* p_ih1_t, const float* p_hh1_t, void* xbase, hipStream_t st, bool clear, float* wslab,
                           int tiles_per_wg);
int dec_persist_loss_chase_launch(const float* target, const g2v_dec_saved* s, const uint8_t* keep95, int T, int B, void* xbase,
                                  hipStream_t st);
namespace g2v {
size_t dec_persist_bwd_wgrad_slab_floats();
}

namespace g2v {

// The persistent path needs every workgroup of its launch resident at once: one per CU.
static int device_cu_count() {
  static int n = -1;
  if (n < 0) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) n = 0;
  }
  return n;
}
// (the switch lives in the calling thread's context: g2v_ctx, G2V_OPT_PERSISTENT; the library reads no environment variable)
static bool persist_enabled() { return g2v_internal_options().persist != 0; }
// Row tiles per workgroup of a persistent rollout over `nblk` row tiles: 1 while there is a CU per tile, 2 or 3 beyond that
// (dec_persist.hip, the *_mt kernels); 0: not offered.  G2V_OPT_PERSISTENT 2 / 3 asks for at least that many (parity tests of the
// multi-tile kernels at small batches).
constexpr int PERSIST_MAX_TILES_PER_WG = 3;
static int persist_tiles_per_wg(int nblk) {
  if (!persist_enabled() || nblk <= 0) return 0;
  const int cus = device_cu_count() < PX_MAX_NBLK ? device_cu_count() : PX_MAX_NBLK;
  if (cus <= 0) return 0;
  int r = cdiv(nblk, cus);
  const int want = g2v_internal_options().persist;
  if (want > r) r = want < nblk ? want : (nblk > 1 ? nblk : 1);
  return r <= PERSIST_MAX_TILES_PER_WG ? r : 0;
}

#ifdef G2V_STAMPS
__device__ unsigned long long g2v_stamps[64 * 16];
__device__ unsigned long long g2v_span[1024 * 4];      // [block][0 = bwd start, 1 = bwd end, 2 = fwd start, 3 = fwd end], realtime
#define SPAN(slot)                                                                                 \
  do {                                                                                             \
    if (threadIdx.x == 0 && blockIdx.x < 1024 && t == 5)                                           \
      g2v_span[blockIdx.x * 4 + (slot)] = __builtin_readcyclecounter();                            \
  } while (0)
#define STAMP(k)                                                                                   \
  do {                                                                                             \
    const int sb_ = blockIdx.x < 4 ? (int)blockIdx.x                                               \
                    : (blockIdx.x == 100 ? 4 : (blockIdx.x == 150 ? 5 : (blockIdx.x == 200 ? 6 : (blockIdx.x == 255 ? 7 : -1)))); \
    if (threadIdx.x == 0 && sb_ >= 0 && t == 5)                                                    \
      g2v_stamps[sb_ * 16 + (k)] = __builtin_amdgcn_s_memtime();                                   \
  } while (0)
#define STAMPB(k) STAMP(8 + (k))
#else
#define SPAN(slot)
#define STAMP(k)
#define STAMPB(k)
#endif

struct DecDims {
  int T, B, D, H;
  float p_drop;
  int n_pre, conditioned, training, nblk;
  int wt;   // write-through (sc1) stores for the arrays only later kernels read (see st4 in common.hpp)
  int scratch;      // floats of reduce_partials scratch at the end of the kernel's LDS (1024, or 2048 where it fits)
};
static int dec_wt_stores() { return 1; }

// packed forward weights (fragment-major, see common.hpp): offsets in floats into the workspace
struct DecPackF {
  const float* pre;    // rows H, K = D
  const float* ih0; const float* hh0; const float* ih1; const float* hh1;   // 3 gate groups x H rows, K = H
  const float* out;    // rows D, K = H
};

template <int HS, int DS>
__global__ __launch_bounds__(HS > 0 ? 256 : 512) void dec_step_fwd_kernel(const float* __restrict__ target,
                                                           const float* __restrict__ h_init, g2v_dec_weights w,
                                                           DecPackF pk, g2v_dec_saved sv,
                                                           const uint8_t* __restrict__ keep95,
                                                           const uint8_t* __restrict__ keep_l0, DecDims dm, int t) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  // HS / DS > 0: the model dims are compile-time constants (every stride, tile count and k-loop folds)
  // Generic dims run with EIGHT waves (two per SIMD): at H = 200 a wave streams ~0.5 MB of weight fragments per step out of
  // L2 through an 8-deep register ring, which bounds it at ring bytes / L2 latency; a second wave per SIMD doubles the bytes in
  // flight and takes the matrix pipe while the first one waits (the feature tiles are dealt over 8 waves instead of 4).
  constexpr int NTHR = HS > 0 ? 256 : 512, NW = NTHR / 64;
  constexpr int NPF = HS > 0 ? 1 : 2;      // float4 per thread of the 16 x H prefetch below
  const int T = dm.T, B = dm.B, D = DS > 0 ? DS : dm.D, H = HS > 0 ? HS : dm.H;
  constexpr int KSD_T = (DS + 15) / 16;
  const int Hp = (H + 15) & ~15, Dp = (D + 15) & ~15, ldh = Hp + 4, ldd = Dp + 4;
  float* Xa = smem;                 // a_t               [16][ldh]
  float* Xh0 = Xa + 16 * ldh;       // h0_{t-1}
  float* Xh1 = Xh0 + 16 * ldh;      // h1_{t-1}
  float* Xx1 = Xh1 + 16 * ldh;      // dropped h0_t (input of layer 1)
  float* Xh1n = Xx1 + 16 * ldh;     // h1_t
  float* Xy = Xh1n + 16 * ldh;      // xin_{t+1}         [16][ldd]
  float* st = Xy + 16 * ldd;        // mean[Hp], invstd[Hp]
  float* red = st + 2 * Hp;         // column sums of the BN partials [2H]
  float* red_scratch = red + 2 * Hp;  // [1024]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b0 = blockIdx.x * 16;
  const int nrows = min(16, B - b0);
  const int i = lane & 15, q = lane >> 4;
  const bool has_next = (t < T - 1);
  const bool hvec = (H & 3) == 0;
  const int H4 = H >> 2;

  SPAN(2);
  STAMP(0);
  // Prefetch this block's rows of u_t, h0_{t-1}, h1_{t-1} (written by the previous launch on some other CU: each is
  // an L2 miss).  Issued first so that their latency overlaps the BatchNorm partial reduction below.
  float4 pu = make_float4(0.f, 0.f, 0.f, 0.f), ph0 = pu, ph1 = pu;
  const bool pre_ok = hvec && (16 * H4 <= NPF * NTHR) && t > 0;     // NPF float4 per thread cover the 16 x H tile
  const int pr = pre_ok ? tid / H4 : 0, pc = pre_ok ? (tid - pr * H4) * 4 : 0;
  const bool pvalid = pre_ok && tid < 16 * H4 && pr < nrows;
  if (pvalid) {
    const int64_t row = ((int64_t)(t - 1) * B + b0 + pr) * H + pc;
    pu = *reinterpret_cast<const float4*>(sv.u + row);
    ph0 = *reinterpret_cast<const float4*>(sv.h0 + row);
    ph1 = *reinterpret_cast<const float4*>(sv.h1 + row);
  }
  // second float4 of the generic prefetch (elements NTHR .. 2 NTHR - 1 of the tile)
  float4 pu2 = make_float4(0.f, 0.f, 0.f, 0.f), ph02 = pu2, ph12 = pu2;
  const int e2 = tid + NTHR;
  const int pr2 = (NPF > 1 && pre_ok) ? e2 / H4 : 0, pc2 = (NPF > 1 && pre_ok) ? (e2 - pr2 * H4) * 4 : 0;
  const bool pvalid2 = NPF > 1 && pre_ok && e2 < 16 * H4 && pr2 < nrows;
  if (pvalid2) {
    const int64_t row = ((int64_t)(t - 1) * B + b0 + pr2) * H + pc2;
    pu2 = *reinterpret_cast<const float4*>(sv.u + row);
    ph02 = *reinterpret_cast<const float4*>(sv.h0 + row);
    ph12 = *reinterpret_cast<const float4*>(sv.h1 + row);
  }
  // Fast shape (compile-time dims, H = 64 so wave w owns feature tile w, full 16-row tile): the hidden-side GRU
  // products gh0 = W_hh0 h0_{t-1} and gh1 = W_hh1 h1_{t-1} do not depend on this step's BatchNorm, so their weight
  // fragments are requested now and the products run INSIDE the L2 round trip of the BatchNorm partials (below).
  constexpr bool FASTC = (HS == 64);
  const bool early = FASTC && pre_ok && (nrows == 16);
  constexpr int KSH_F = FASTC ? HS / 16 : 1;
  constexpr int KSD_F = FASTC ? KSD_T : 1;
  WFrag<3, KSH_F> f_hh0, f_hh1, f_ih0, f_ih1, f_out;
  WFrag<1, KSD_F> f_pre;
  f32x4 gh0[3], gh1[3];
  float4 bi0[3], bh0[3], bi1[3], bh1[3];           // GRU biases of this lane's 4 features (early path: prefetched)
#pragma unroll
  for (int g = 0; g < 3; ++g) {
    gh0[g] = (f32x4){0.f, 0.f, 0.f, 0.f};
    gh1[g] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }
  if (early) {
    frag_load(f_hh0, pk.hh0, wave, 4, lane);
    frag_load(f_hh1, pk.hh1, wave, 4, lane);
  }
  // keep flags of the Dropout(0.95) on y_t (dense epilogue below): requested now, consumed after the out_layer
  const bool dense_e = (nrows == 16) && t > 0 && !(has_next && t < dm.n_pre) && ((((int64_t)t * B + b0) * D) & 3) == 0 &&
                       ((16 * D) & 3) == 0 && (16 * D <= 2 * 16 * ldh) && dm.conditioned;
  constexpr int KPRE = FASTC ? (16 * DS / 4 + 255) / 256 : 1;
  uint32_t kpre[KPRE];
  if (FASTC && dense_e && has_next) {
    const int n4 = (16 * D) >> 2;
    const uint32_t* kp4 = reinterpret_cast<const uint32_t*>(keep95 + ((int64_t)t * B + b0) * D);
#pragma unroll
    for (int j = 0; j < KPRE; ++j) {
      const int e4 = tid + 256 * j;
      kpre[j] = kp4[e4 < n4 ? e4 : 0];
    }
  }
  auto early_products = [&](auto issue2) {
    // stage h0_{t-1}, h1_{t-1} and run the two hidden-side products (3 gates x 4 k-steps x 4 MFMAs each)
    if (pvalid) {
      *reinterpret_cast<float4*>(Xh0 + pr * ldh + pc) = ph0;
      *reinterpret_cast<float4*>(Xh1 + pr * ldh + pc) = ph1;
    }
    lds_barrier();
    // ... and while they run, the fragments of the two input-side products stream in BETWEEN the MFMAs
    const int fb = 16 * wave + 4 * q;
    auto frag_or_bias = [&](int k, WFrag<3, KSH_F>& fr, const float* P, float4 (&b_i)[3], float4 (&b_h)[3], const float* bih,
                            const float* bhh) {
      if (k < 3 * KSH_F) {
        const int t2 = k / KSH_F, s2 = k % KSH_F;
        fr.w[t2][s2] = *reinterpret_cast<const float4*>(P + ((int64_t)((wave + 4 * t2) * KSH_F + s2) * 64 + lane) * 4);
      } else if (k < 3 * KSH_F + 3) {
        b_i[k - 3 * KSH_F] = *reinterpret_cast<const float4*>(bih + (k - 3 * KSH_F) * HS + fb);
      } else {
        b_h[k - 3 * KSH_F - 3] = *reinterpret_cast<const float4*>(bhh + (k - 3 * KSH_F - 3) * HS + fb);
      }
    };
    frag_mma_issue<3 * KSH_F + 6>(gh0, f_hh0, Xh0, ldh, lane,
                                  [&](int k) { frag_or_bias(k, f_ih0, pk.ih0, bi0, bh0, w.b_ih0, w.b_hh0); });
    // the second product also carries the second batch of BatchNorm-partial rows (issue2)
    frag_mma_issue<3 * KSH_F + 6 + 16>(gh1, f_hh1, Xh1, ldh, lane, [&](int k) {
      if (k < 3 * KSH_F + 6) frag_or_bias(k, f_ih1, pk.ih1, bi1, bh1, w.b_ih1, w.b_hh1);
      else issue2(k - (3 * KSH_F + 6));
    });
  };
  // Zero what the MFMA contractions must see as zero: padding columns and rows >= nrows of every operand tile.
  // (full tiles of an H % 16 == 0 model have no H padding at all: only the D padding of Xy is touched)
  {
    const bool full = (nrows == 16);
    if (!full || Hp != H) {
      for (int e = tid; e < 5 * 16 * ldh; e += NTHR) smem[e] = 0.f;
    }
    if (!full) {
      for (int e = tid; e < 16 * ldd; e += NTHR) Xy[e] = 0.f;
    } else {
      const int padc = ldd - D;
      for (int e = tid; e < 16 * padc; e += NTHR) Xy[(e / padc) * ldd + D + (e % padc)] = 0.f;
    }
  }
  lds_barrier();
  STAMP(1);

  if (t == 0) {
    // seed the state arrays: h0[0], h1[0] = h_init (the quantised latent)
    for (int e = tid; e < 16 * H; e += NTHR) {
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
      if (early) reduce_partials_128_hook(part, dm.nblk, red, red_scratch, tid, early_products);
      else reduce_partials<NTHR>(part, dm.nblk, 2 * H, red, red_scratch, tid, dm.scratch);
      STAMP(2);
      for (int f = tid; f < H; f += NTHR) {
        const float s1 = red[f], s2 = red[H + f];
        const float mv = s1 / (float)B;
        const float var = fmaxf(s2 / (float)B - mv * mv, 0.f);   // biased batch variance
        const float mean = mv + w.b_pre[f];
        st[f] = mean;
        st[Hp + f] = bn_invstd_(var);
        if (blockIdx.x == 0) {
          sv.bn_stats[(int64_t)(t - 1) * 2 * H + f] = mean;
          sv.bn_stats[(int64_t)(t - 1) * 2 * H + H + f] = var;
        }
      }
    } else {
      if (early) early_products([](int) {});
      for (int f = tid; f < H; f += NTHR) {
        st[f] = w.bn_running_mean[f];
        st[Hp + f] = bn_invstd_(w.bn_running_var[f]);
      }
    }
    lds_barrier();
    // ---- (b) a_t = ReLU(BN(u_t)); stage previous hidden states ------------------------------------
    if (pre_ok) {
      if (pvalid) {
        const float4 g4 = *reinterpret_cast<const float4*>(w.bn_w + pc), b4 = *reinterpret_cast<const float4*>(w.bn_b + pc);
        const float4 m4 = *reinterpret_cast<const float4*>(st + pc), i4 = *reinterpret_cast<const float4*>(st + Hp + pc);
        float4 a4;
        a4.x = fmaxf((pu.x - m4.x) * i4.x * g4.x + b4.x, 0.f);
        a4.y = fmaxf((pu.y - m4.y) * i4.y * g4.y + b4.y, 0.f);
        a4.z = fmaxf((pu.z - m4.z) * i4.z * g4.z + b4.z, 0.f);
        a4.w = fmaxf((pu.w - m4.w) * i4.w * g4.w + b4.w, 0.f);
        *reinterpret_cast<float4*>(Xa + pr * ldh + pc) = a4;
        if (!early) {
          *reinterpret_cast<float4*>(Xh0 + pr * ldh + pc) = ph0;
          *reinterpret_cast<float4*>(Xh1 + pr * ldh + pc) = ph1;
        }
        if (sv.a) st4(sv.a + ((int64_t)(t - 1) * B + b0 + pr) * H + pc, a4, dm.wt != 0);
      }
      if (pvalid2) {
        const float4 g4 = *reinterpret_cast<const float4*>(w.bn_w + pc2), b4 = *reinterpret_cast<const float4*>(w.bn_b + pc2);
        const float4 m4 = *reinterpret_cast<const float4*>(st + pc2), i4 = *reinterpret_cast<const float4*>(st + Hp + pc2);
        float4 a4;
        a4.x = fmaxf((pu2.x - m4.x) * i4.x * g4.x + b4.x, 0.f);
        a4.y = fmaxf((pu2.y - m4.y) * i4.y * g4.y + b4.y, 0.f);
        a4.z = fmaxf((pu2.z - m4.z) * i4.z * g4.z + b4.z, 0.f);
        a4.w = fmaxf((pu2.w - m4.w) * i4.w * g4.w + b4.w, 0.f);
        *reinterpret_cast<float4*>(Xa + pr2 * ldh + pc2) = a4;
        *reinterpret_cast<float4*>(Xh0 + pr2 * ldh + pc2) = ph02;
        *reinterpret_cast<float4*>(Xh1 + pr2 * ldh + pc2) = ph12;
        if (sv.a) st4(sv.a + ((int64_t)(t - 1) * B + b0 + pr2) * H + pc2, a4, dm.wt != 0);
      }
    } else {
      for (int e = tid; e < 16 * H; e += NTHR) {
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
    }
    lds_barrier();
    STAMP(3);
    // ---- (c) GRU layer 0 ---------------------------------------------------------------------------
    const bool drop = dm.training && keep_l0 && dm.p_drop > 0.f;
    constexpr int KSH_T = HS / 16;
    if (early) {
      // input-side products only: the hidden-side halves gh0 / gh1 are already in registers
      const int f0 = 16 * wave + 4 * q;
      const uint8_t* kl0 = drop ? keep_l0 + ((int64_t)(t - 1) * B + b0) * H : nullptr;
      uint32_t kp = 0x01010101u;
      if (kl0) kp = *reinterpret_cast<const uint32_t*>(kl0 + (int64_t)i * H + f0);
      f32x4 ai[3];
#pragma unroll
      for (int g = 0; g < 3; ++g) ai[g] = (f32x4){0.f, 0.f, 0.f, 0.f};
      frag_mma_pf(ai, f_ih0, Xa, ldh, lane, f_out, pk.out, wave, 4, true);          // out_layer fragments ride along
      gru_cell_fwd_epilogue(ai, gh0, bi0, bh0, kp, kl0 != nullptr, 1.0f / (1.0f - dm.p_drop), Xh0, ldh, H, Xx1,
                            sv.h0 + ((int64_t)t * B + b0) * H,
                            sv.gates0 ? sv.gates0 + ((int64_t)(t - 1) * B + b0) * 4 * H : nullptr,
                            (drop && sv.x1) ? sv.x1 + ((int64_t)(t - 1) * B + b0) * H : nullptr, nrows, i, f0, dm.wt != 0);
      lds_barrier();
      STAMP(4);
#pragma unroll
      for (int g = 0; g < 3; ++g) ai[g] = (f32x4){0.f, 0.f, 0.f, 0.f};
      frag_mma_pf(ai, f_ih1, Xx1, ldh, lane, f_pre, pk.pre, wave, 0, has_next);      // pre_linear fragments ride along
      gru_cell_fwd_epilogue(ai, gh1, bi1, bh1, 0x01010101u, false, 1.0f, Xh1, ldh, H, Xh1n,
                            sv.h1 + ((int64_t)t * B + b0) * H,
                            sv.gates1 ? sv.gates1 + ((int64_t)(t - 1) * B + b0) * 4 * H : nullptr, nullptr, nrows, i, f0, dm.wt != 0);
      lds_barrier();
      STAMP(5);
    } else {
    gru_cell_fwd<KSH_T>(pk.ih0, pk.hh0, w.b_ih0, w.b_hh0, Xa, Xh0, ldh, H, Hp, Xx1,
                 sv.h0 + ((int64_t)t * B + b0) * H,
                 sv.gates0 ? sv.gates0 + ((int64_t)(t - 1) * B + b0) * 4 * H : nullptr,
                 drop ? keep_l0 + ((int64_t)(t - 1) * B + b0) * H : nullptr, 1.0f / (1.0f - dm.p_drop),
                 (drop && sv.x1) ? sv.x1 + ((int64_t)(t - 1) * B + b0) * H : nullptr, nrows, lane, wave, NW);
    lds_barrier();
    STAMP(4);
    // ---- (d) GRU layer 1 ---------------------------------------------------------------------------
    gru_cell_fwd<KSH_T>(pk.ih1, pk.hh1, w.b_ih1, w.b_hh1, Xx1, Xh1, ldh, H, Hp, Xh1n,
                 sv.h1 + ((int64_t)t * B + b0) * H,
                 sv.gates1 ? sv.gates1 + ((int64_t)(t - 1) * B + b0) * 4 * H : nullptr, nullptr, 1.0f, nullptr, nrows,
                 lane, wave, NW);
    lds_barrier();
    STAMP(5);
    }
  }

  // ---- (e) y_t = out_layer(h1_t)  (t == 0: y_0 = target frame 0), next decoder input ---------------
  // The 16 rows of this block are CONTIGUOUS in the time-major (T,B,D) arrays (y, xin, keep95): 16*D floats.
  // Fast path: accumulators -> dense LDS tile -> coalesced 16-byte global stores by all 256 threads
  // (the per-lane path touches 64 different cache lines per instruction when D = 135).
  if (dense_e) {
    float* Yt = Xa;   // Xa|Xh0 are dead by now: 2*16*ldh floats >= 16*D
    const int ntile = Dp >> 4;
    const bool three = ntile > 4;
    const int per = three ? 3 : 1, group = 4 * per;
    for (int base = 0; base < ntile; base += group) {
      if (wave >= 4) continue;      // (generic dims: 8 waves; the few D tiles stay dealt over the first four)
      float bo[3][4];
#pragma unroll
      for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int d = 16 * (base + wave + 4 * j) + 4 * q + r;
          bo[j][r] = (j < per && d < D) ? w.b_out[d] : 0.f;
        }
      f32x4 acc[3];
#pragma unroll
      for (int j = 0; j < 3; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (FASTC && KSD_T > 4 && KSD_T <= 12 && early) {
        frag_mma(acc, f_out, Xh1n, ldh, lane);     // one group of 12 tiles; fragments streamed in during GRU layer 0
      } else if (three) {
        wave_gemm_p<3, HS / 16>(acc, pk.out, Hp >> 4, base + wave, 4, Xh1n, ldh, lane);
      } else {
        f32x4 a1[1] = {acc[0]};
        wave_gemm_p<1, HS / 16>(a1, pk.out, Hp >> 4, base + wave, 0, Xh1n, ldh, lane);
        acc[0] = a1[0];
      }
#pragma unroll
      for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int d = 16 * (base + wave + 4 * j) + 4 * q + r;
          if (j < per && d < D) Yt[i * D + d] = acc[j][r] + bo[j][r];
        }
    }
    lds_barrier();
    const int64_t tile = ((int64_t)t * B + b0) * D;     // element offset of this block's 16 x D tile
    const int n4 = (16 * D) >> 2;
    int jj = 0;
    for (int e4 = tid; e4 < n4; e4 += NTHR, ++jj) {
      const float4 y4 = reinterpret_cast<const float4*>(Yt)[e4];
      st4(sv.y + tile + 4 * (int64_t)e4, y4, dm.wt != 0);
      if (has_next) {
        uint32_t k4;
        if constexpr (FASTC) {
          k4 = kpre[0];
#pragma unroll
          for (int j = 1; j < KPRE; ++j) k4 = (jj == j) ? kpre[j] : k4;
        } else {
          k4 = reinterpret_cast<const uint32_t*>(keep95 + tile)[e4];
        }
        float4 x4;
        x4.x = (k4 & 0xffu) ? y4.x * 20.0f : 0.f;            // Dropout(0.95): 1/(1-0.95)
        x4.y = (k4 & 0xff00u) ? y4.y * 20.0f : 0.f;
        x4.z = (k4 & 0xff0000u) ? y4.z * 20.0f : 0.f;
        x4.w = (k4 & 0xff000000u) ? y4.w * 20.0f : 0.f;
        if (sv.xin) st4(sv.xin + tile + 4 * (int64_t)e4, x4, dm.wt != 0);
        const int e = 4 * e4;
        const float xv[4] = {x4.x, x4.y, x4.z, x4.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int r = (e + j) / D, c = (e + j) - r * D;
          Xy[r * ldd + c] = xv[j];
        }
      }
    }
  } else
  // D tiles are dealt to the waves in groups of 12 (3 per wave, one NT=3 contraction that shares the X fragments);
  // the packed matrix is zero-padded to a multiple of 12 tiles (4 when D <= 64, then one tile per wave).
  {
    const int ntile = Dp >> 4;
    const bool three = ntile > 4;
    const int per = three ? 3 : 1, group = 4 * per;
    const bool teacher = (t < dm.n_pre);
    const int b = b0 + i;
    for (int base = 0; base < ntile; base += group) {
      if (wave >= 4) continue;
      float bo[3][4], tg[3][4];
      uint8_t kp[3][4];
#pragma unroll
      for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          bo[j][r] = 0.f; tg[j][r] = 0.f; kp[j][r] = 0;
          const int d = 16 * (base + wave + 4 * j) + 4 * q + r;
          if (j < per && i < nrows && d < D) {
            bo[j][r] = w.b_out[d];
            if (t == 0 || (has_next && teacher)) tg[j][r] = target[((int64_t)b * T + t) * D + d];
            if (has_next && dm.conditioned) kp[j][r] = keep95[((int64_t)t * B + b) * D + d];
          }
        }
      f32x4 acc[3];
#pragma unroll
      for (int j = 0; j < 3; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (t > 0) {
        if (FASTC && KSD_T > 4 && KSD_T <= 12 && early) {
          frag_mma(acc, f_out, Xh1n, ldh, lane);
        } else if (three) {
          wave_gemm_p<3, HS / 16>(acc, pk.out, Hp >> 4, base + wave, 4, Xh1n, ldh, lane);
        } else {
          f32x4 a1[1] = {acc[0]};
          wave_gemm_p<1, HS / 16>(a1, pk.out, Hp >> 4, base + wave, 0, Xh1n, ldh, lane);
          acc[0] = a1[0];
        }
      }
      if (i < nrows) {
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int d = 16 * (base + wave + 4 * j) + 4 * q + r;
            if (j >= per || d >= D) continue;
            const float y = (t == 0) ? tg[j][r] : acc[j][r] + bo[j][r];
            sv.y[((int64_t)t * B + b) * D + d] = y;
            if (has_next) {
              const float src = teacher ? tg[j][r] : y;                                 // :1049-1052
              const float xin = kp[j][r] ? src * 20.0f : 0.f;                            // Dropout(0.95): 1/(1-0.95)
              Xy[i * ldd + d] = xin;
              if (sv.xin) sv.xin[((int64_t)t * B + b) * D + d] = xin;
            }
          }
      }
    }
  }
  if (!has_next) return;
  lds_barrier();
  STAMP(6);
  // ---- (f) u_{t+1} = pre_linear.0(xin_{t+1}) and per-block BN partial sums of (u - b) ----------------
  {
    const int ntile = Hp >> 4;
    float* part = sv.bn_partial + ((int64_t)(t & 1) * dm.nblk + blockIdx.x) * 2 * H;
    for (int ft = wave; ft < ntile; ft += NW) {
      const int f0 = 16 * ft + 4 * q;
      const bool vec = hvec && (f0 + 3 < H);
      float4 bp = make_float4(0.f, 0.f, 0.f, 0.f);
      if (vec) bp = *reinterpret_cast<const float4*>(w.b_pre + f0);
      f32x4 acc[1] = {(f32x4){0.f, 0.f, 0.f, 0.f}};
      if (FASTC && early) frag_mma(acc, f_pre, Xy, ldd, lane);     // fragments streamed in during GRU layer 1
      else wave_gemm_p<1, KSD_T>(acc, pk.pre, Dp >> 4, ft, 0, Xy, ldd, lane);
      float s1[4], s2[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int f = f0 + r;
        const float v = (i < nrows && f < H) ? acc[0][r] : 0.f;
        s1[r] = reduce16(v);
        s2[r] = reduce16(v * v);
      }
      if (vec) {
        if (i < nrows)
          *reinterpret_cast<float4*>(sv.u + ((int64_t)t * B + b0 + i) * H + f0) =
              make_float4(acc[0][0] + bp.x, acc[0][1] + bp.y, acc[0][2] + bp.z, acc[0][3] + bp.w);
        if (i == 0) {
          *reinterpret_cast<float4*>(part + f0) = make_float4(s1[0], s1[1], s1[2], s1[3]);
          *reinterpret_cast<float4*>(part + H + f0) = make_float4(s2[0], s2[1], s2[2], s2[3]);
        }
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int f = f0 + r;
          if (f >= H) continue;
          if (i < nrows) sv.u[((int64_t)t * B + b0 + i) * H + f] = acc[0][r] + w.b_pre[f];
          if (i == 0) {
            part[f] = s1[r];
            part[H + f] = s2[r];
          }
        }
      }
    }
  }
  STAMP(7);
  SPAN(3);
}

// running_mean / running_var (momentum 0.1, unbiased variance), applied T-1 times in step order
__global__ void bn_running_update_kernel(const float* __restrict__ bn_stats, float* __restrict__ rm,
                                         float* __restrict__ rv, int steps, int H, int B, const unsigned* __restrict__ fault) {
  const int f = blockIdx.x * blockDim.x + threadIdx.x;
  if (f >= H) return;
  if (fault && *fault != 0u) return;      // a latched fault of the persistent rollouts: the statistics are garbage, the state stays
  float m = rm[f], v = rv[f];
  const float unbias = (B > 1) ? (float)B / (float)(B - 1) : 1.0f;
  for (int s0 = 0; s0 < steps; s0 += 8) {   // 16 independent loads in flight, then the 8 dependent updates
    float sm[8], svv[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int s = min(s0 + j, steps - 1);
      sm[j] = bn_stats[(int64_t)s * 2 * H + f];
      svv[j] = bn_stats[(int64_t)s * 2 * H + H + f];
    }
#pragma unroll
    for (int j = 0; j < 8; ++j)
      if (s0 + j < steps) {
        m = 0.9f * m + 0.1f * sm[j];
        v = 0.9f * v + 0.1f * (svv[j] * unbias);
      }
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

template <int HS, int DS>
__global__ __launch_bounds__(HS > 0 ? 256 : 512) void dec_step_bwd_kernel(g2v_dec_weights w, DecTW tw, g2v_dec_saved sv,
                                                           g2v_dec_grads gr, const uint8_t* __restrict__ keep95,
                                                           const uint8_t* __restrict__ keep_l0, DecDims dm, int t) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int NTHR = HS > 0 ? 256 : 512, NW = NTHR / 64, NPF = HS > 0 ? 1 : 2;      // generic dims: eight waves (see the forward)
  const int T = dm.T, B = dm.B, D = DS > 0 ? DS : dm.D, H = HS > 0 ? HS : dm.H, G = 3 * H;
  constexpr int KSD_T = (DS + 15) / 16;
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
  float* red_scratch = red + 2 * Hp;  // [1024]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b0 = blockIdx.x * 16;
  const int nrows = min(16, B - b0);
  const int i = lane & 15, q = lane >> 4;
  const bool last = (t == T - 1);   // first kernel of the backward sweep
  const int nth = Hp >> 4, ntd = Dp >> 4;
  const bool hvec = (H & 3) == 0;
  const int H4 = H >> 2;

  SPAN(0);
  // prefetch this block's rows of u_{t+1} and dbn_{t+1} (Part A inputs; L2 misses) before anything else
  float4 pu = make_float4(0.f, 0.f, 0.f, 0.f), pdb = pu;
  const bool pre_ok = hvec && (16 * H4 <= NPF * NTHR) && !last;
  const int pr = pre_ok ? tid / H4 : 0, pc = pre_ok ? (tid - pr * H4) * 4 : 0;
  const bool pvalid = pre_ok && tid < 16 * H4 && pr < nrows;
  if (pvalid) {
    const int64_t row = ((int64_t)t * B + b0 + pr) * H + pc;
    pu = *reinterpret_cast<const float4*>(sv.u + row);
    pdb = *reinterpret_cast<const float4*>(gr.dbn + row);
  }
  float4 pu2 = make_float4(0.f, 0.f, 0.f, 0.f), pdb2 = pu2;      // second float4 of the generic prefetch
  const int e2 = tid + NTHR;
  const int pr2 = (NPF > 1 && pre_ok) ? e2 / H4 : 0, pc2 = (NPF > 1 && pre_ok) ? (e2 - pr2 * H4) * 4 : 0;
  const bool pvalid2 = NPF > 1 && pre_ok && e2 < 16 * H4 && pr2 < nrows;
  if (pvalid2) {
    const int64_t row = ((int64_t)t * B + b0 + pr2) * H + pc2;
    pu2 = *reinterpret_cast<const float4*>(sv.u + row);
    pdb2 = *reinterpret_cast<const float4*>(gr.dbn + row);
  }
  // generic dims: what the two cell epilogues read from HBM (saved gates, previous state, carry, keep flags) is requested well
  // ahead -- cell 1's here, cell 0's in front of cell 1 (vector-memory results return in order: a request placed directly in
  // front of a product's L2 weight stream delays that stream by an HBM round trip, measured) -- instead of one round trip per
  // tile in front of each epilogue (17 + 21 k of the kernel's 184 k ticks, gpurun_tools/stamps_native.py); H <= 256: <= 2 tiles per wave
  constexpr int MAXT = HS > 0 ? 1 : 2;
  const bool pf = HS == 0 && hvec && nth <= MAXT * NW && t > 0;
  CellBwdIn cin[MAXT];
  if (pf) {
    const float* carry1_p = last ? nullptr : gr.dh_init + ((int64_t)B + b0) * H;
#pragma unroll
    for (int j = 0; j < MAXT; ++j)
      if (wave + NW * j < nth)
        cell_bwd_prefetch(cin[j], carry1_p, nullptr, sv.gates1 + ((int64_t)(t - 1) * B + b0) * 4 * H,
                          sv.h1 + ((int64_t)(t - 1) * B + b0) * H, H, wave + NW * j, nrows, lane);
  }
  STAMPB(0);
  // zero padding columns / rows of the MFMA operand tiles
  {
    const bool full = (nrows == 16);
    if (!full || Hp != H || Gp != G) {
      for (int e = tid; e < 16 * (3 * ldh + ldd + 2 * ldg); e += NTHR) smem[e] = 0.f;
    } else {
      const int padc = ldd - D;
      for (int e = tid; e < 16 * padc; e += NTHR) Xdy[(e / padc) * ldd + D + (e % padc)] = 0.f;
      if (last)
        for (int e = tid; e < 16 * ldh; e += NTHR) Xdu[e] = 0.f;
    }
  }
  lds_barrier();

  // ================= Part A: finish BatchNorm backward of step t+1 ===================================
  if (!last) {
    const float* part = gr.bn_bwd_partial + (int64_t)((t + 1) & 1) * dm.nblk * 2 * H;
    reduce_partials<NTHR>(part, dm.nblk, 2 * H, red, red_scratch, tid, dm.scratch);
    STAMPB(1);
    for (int f = tid; f < H; f += NTHR) {
      const float s1 = red[f], s2 = red[H + f];
      st[f] = s1;
      st[Hp + f] = s2;
      if (blockIdx.x == 0) {   // d gamma / d beta accumulate over the steps (one writer, stream ordered)
        const bool first_acc = (t == T - 2);
        gr.d_bn_w[f] = (first_acc ? 0.f : gr.d_bn_w[f]) + s2;
        gr.d_bn_b[f] = (first_acc ? 0.f : gr.d_bn_b[f]) + s1;
      }
    }
    lds_barrier();
    const float invB = 1.0f / (float)B;
    const float* stats = sv.bn_stats + (int64_t)t * 2 * H;   // step t+1 is stored at index t
    if (pre_ok) {
      auto du_of = [&](const float4& pu, const float4& pdb, int pr, int pc) {
        const float4 mean4 = *reinterpret_cast<const float4*>(stats + pc), var4 = *reinterpret_cast<const float4*>(stats + H + pc);
        const float4 g4 = *reinterpret_cast<const float4*>(w.bn_w + pc);
        const float4 s14 = *reinterpret_cast<const float4*>(st + pc), s24 = *reinterpret_cast<const float4*>(st + Hp + pc);
        const float uu[4] = {pu.x, pu.y, pu.z, pu.w}, db[4] = {pdb.x, pdb.y, pdb.z, pdb.w};
        const float mm[4] = {mean4.x, mean4.y, mean4.z, mean4.w}, vv[4] = {var4.x, var4.y, var4.z, var4.w};
        const float gg[4] = {g4.x, g4.y, g4.z, g4.w}, a1[4] = {s14.x, s14.y, s14.z, s14.w}, a2[4] = {s24.x, s24.y, s24.z, s24.w};
        float du[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float invstd = bn_invstd_(vv[r]);
          const float xhat = (uu[r] - mm[r]) * invstd;
          du[r] = gg[r] * invstd * (db[r] - a1[r] * invB - xhat * a2[r] * invB);
        }
        const float4 du4 = make_float4(du[0], du[1], du[2], du[3]);
        st4(gr.du + ((int64_t)t * B + b0 + pr) * H + pc, du4, dm.wt != 0);
        *reinterpret_cast<float4*>(Xdu + pr * ldh + pc) = du4;
      };
      if (pvalid) du_of(pu, pdb, pr, pc);
      if (pvalid2) du_of(pu2, pdb2, pr2, pc2);
    } else {
      for (int e = tid; e < 16 * H; e += NTHR) {
        const int r = e / H, f = e - r * H;
        if (r >= nrows) continue;
        const int64_t row = ((int64_t)t * B + b0 + r) * H + f;
        const float invstd = bn_invstd_(stats[H + f]);
        const float xhat = (sv.u[row] - stats[f]) * invstd;
        const float du = w.bn_w[f] * invstd * (gr.dbn[row] - st[f] * invB - xhat * st[Hp + f] * invB);
        gr.du[row] = du;
        Xdu[r * ldh + f] = du;
      }
    }
    lds_barrier();
  }
  STAMPB(2);
  if (t == 0) return;   // only the BN finish of step 1 was left (y_0 is data: no feedback needed)

  // ================= Part B: dy_t (loss + feedback) ===================================================
  // Same contiguity as in the forward: the block's 16 x D tile of dy / keep95 is one dense run in memory.
  const bool dense_b = (nrows == 16) && ((((int64_t)t * B + b0) * D) & 3) == 0 && ((16 * D) & 3) == 0 && (16 * D <= 16 * ldg);
  if (dense_b) {
    const bool feedback = (!last) && dm.conditioned && (t >= dm.n_pre);
    float* Dt = Gi;                                        // dense dy tile [16*D]
    uint32_t* Kt = reinterpret_cast<uint32_t*>(Gh);        // keep95 bytes of the tile, 4 per word
    const int64_t tile = ((int64_t)t * B + b0) * D;
    const int n4 = (16 * D) >> 2;
    for (int e4 = tid; e4 < n4; e4 += NTHR) {
      reinterpret_cast<float4*>(Dt)[e4] = reinterpret_cast<const float4*>(gr.dy + tile)[e4];
      if (feedback) Kt[e4] = reinterpret_cast<const uint32_t*>(keep95 + tile)[e4];
    }
    lds_barrier();
    const bool three = ntd > 4;
    const int per = three ? 3 : 1, group = 4 * per;
    const uint8_t* Kb = reinterpret_cast<const uint8_t*>(Kt);
    for (int base = 0; base < ntd; base += group) {
      if (wave >= 4) continue;      // (generic dims: 8 waves; the few D tiles stay dealt over the first four)
      f32x4 acc[3];
#pragma unroll
      for (int j = 0; j < 3; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (feedback) {
        if (three) {
          wave_gemm_p<3, HS / 16>(acc, tw.w_pre_t, Hp >> 4, base + wave, 4, Xdu, ldh, lane);
        } else {
          f32x4 a1[1] = {acc[0]};
          wave_gemm_p<1, HS / 16>(a1, tw.w_pre_t, Hp >> 4, base + wave, 0, Xdu, ldh, lane);
          acc[0] = a1[0];
        }
      }
#pragma unroll
      for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int d = 16 * (base + wave + 4 * j) + 4 * q + r;
          if (j >= per || d >= D) continue;
          float dy = Dt[i * D + d];
          if (feedback && Kb[i * D + d]) dy += acc[j][r] * 20.0f;
          Dt[i * D + d] = dy;
          Xdy[i * ldd + d] = dy;
        }
    }
    lds_barrier();
    if (feedback)
      for (int e4 = tid; e4 < n4; e4 += NTHR) st4(gr.dy + tile + 4 * (int64_t)e4, reinterpret_cast<const float4*>(Dt)[e4], dm.wt != 0);
    lds_barrier();
  } else
  {
  {
    const bool feedback = (!last) && dm.conditioned && (t >= dm.n_pre);
    const bool three = ntd > 4;
    const int per = three ? 3 : 1, group = 4 * per;
    for (int base = 0; base < ntd; base += group) {
      if (wave >= 4) continue;
      float dyv[3][4];
      uint8_t kp[3][4];
#pragma unroll
      for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          dyv[j][r] = 0.f; kp[j][r] = 0;
          const int d = 16 * (base + wave + 4 * j) + 4 * q + r;
          if (j < per && i < nrows && d < D) {
            const int64_t idx = ((int64_t)t * B + b0 + i) * D + d;
            dyv[j][r] = gr.dy[idx];
            if (feedback) kp[j][r] = keep95[idx];
          }
        }
      f32x4 acc[3];
#pragma unroll
      for (int j = 0; j < 3; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (feedback) {
        if (three) {
          wave_gemm_p<3, HS / 16>(acc, tw.w_pre_t, Hp >> 4, base + wave, 4, Xdu, ldh, lane);
        } else {
          f32x4 a1[1] = {acc[0]};
          wave_gemm_p<1, HS / 16>(a1, tw.w_pre_t, Hp >> 4, base + wave, 0, Xdu, ldh, lane);
          acc[0] = a1[0];
        }
      }
      if (i < nrows) {
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int d = 16 * (base + wave + 4 * j) + 4 * q + r;
            if (j >= per || d >= D) continue;
            const int64_t idx = ((int64_t)t * B + b0 + i) * D + d;
            float dy = dyv[j][r];
            if (kp[j][r]) dy += acc[j][r] * 20.0f;
            if (feedback) gr.dy[idx] = dy;
            Xdy[i * ldd + d] = dy;
          }
      }
    }
  }
  lds_barrier();
  }
  STAMPB(3);
  const float* carry0 = last ? nullptr : gr.dh_init + (int64_t)b0 * H;
  const float* carry1 = last ? nullptr : gr.dh_init + ((int64_t)B + b0) * H;
  float* carry0_w = gr.dh_init + (int64_t)b0 * H;
  float* carry1_w = gr.dh_init + ((int64_t)B + b0) * H;
  // (cell 0's inputs: requested here, consumed two phases on)
  constexpr int MAXT0 = MAXT;
  const bool pf0 = pf;
  CellBwdIn cin0[MAXT0];
  if (pf0) {
    const bool drop0 = keep_l0 && dm.p_drop > 0.f;
#pragma unroll
    for (int j = 0; j < MAXT0; ++j)
      if (wave + NW * j < nth)
        cell_bwd_prefetch(cin0[j], carry0, drop0 ? keep_l0 + ((int64_t)(t - 1) * B + b0) * H : nullptr,
                          sv.gates0 + ((int64_t)(t - 1) * B + b0) * 4 * H, sv.h0 + ((int64_t)(t - 1) * B + b0) * H, H,
                          wave + NW * j, nrows, lane);
  }
  // ---- dh1 = carry + dy W_out ; GRU cell 1 backward -------------------------------------------------
  {
#pragma unroll
    for (int j = 0; j < MAXT; ++j) {
      for (int ft = wave + NW * j; ft < nth; ft += NW * MAXT) {
        f32x4 acc[1] = {(f32x4){0.f, 0.f, 0.f, 0.f}};
        wave_gemm_p<1, KSD_T>(acc, tw.w_out_t, Dp >> 4, ft, 0, Xdy, ldd, lane);
        gru_cell_bwd_tile(acc[0], carry1, 1.0f, nullptr, sv.gates1 + ((int64_t)(t - 1) * B + b0) * 4 * H,
                          sv.h1 + ((int64_t)(t - 1) * B + b0) * H, gr.dgi1 + ((int64_t)(t - 1) * B + b0) * G,
                          gr.dgh1 + ((int64_t)(t - 1) * B + b0) * G, Gi, Gh, ldg, Dd, ldh, H, ft, nrows, lane, dm.wt != 0,
                          pf && ft == wave + NW * j, cin[j]);
      }
    }
  }
  lds_barrier();
  STAMPB(4);
  // ---- carry1' = dh1*z + dgh1 W_hh1 ;  dx1 = dgi1 W_ih1 -> dh0 (inter-layer dropout bwd) ----------------
  {
    for (int ft = wave; ft < nth; ft += NW) {
      f32x4 a1[1] = {(f32x4){0.f, 0.f, 0.f, 0.f}}, a2[1] = {(f32x4){0.f, 0.f, 0.f, 0.f}};
      if constexpr (HS == 0) {
        wave_gemm_p_dual<1, 8>(a1, tw.w_hh1_t, Gh, a2, tw.w_ih1_t, Gi, Gp >> 4, ft, 0, ldg, lane);
      } else {
        wave_gemm_p<1, 3 * HS / 16>(a1, tw.w_hh1_t, Gp >> 4, ft, 0, Gh, ldg, lane);
        wave_gemm_p<1, 3 * HS / 16>(a2, tw.w_ih1_t, Gp >> 4, ft, 0, Gi, ldg, lane);
      }
      const int f0 = 16 * ft + 4 * q;
      if (hvec && f0 + 3 < H) {
        const float4 d4 = *reinterpret_cast<const float4*>(Dd + i * ldh + f0);
        if (i < nrows)
          *reinterpret_cast<float4*>(carry1_w + (int64_t)i * H + f0) =
              make_float4(d4.x + a1[0][0], d4.y + a1[0][1], d4.z + a1[0][2], d4.w + a1[0][3]);
        *reinterpret_cast<float4*>(Xdx + i * ldh + f0) = make_float4(a2[0][0], a2[0][1], a2[0][2], a2[0][3]);
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int f = f0 + r;
          if (f >= H) continue;
          if (i < nrows) carry1_w[(int64_t)i * H + f] = Dd[i * ldh + f] + a1[0][r];
          Xdx[i * ldh + f] = a2[0][r];
        }
      }
    }
  }
  lds_barrier();
  STAMPB(5);
  // ---- GRU cell 0 backward (Gi/Gh/Dd are reused) ------------------------------------------------------
  {
    const bool drop = keep_l0 && dm.p_drop > 0.f;
#pragma unroll
    for (int j = 0; j < MAXT0; ++j) {
      for (int ft = wave + NW * j; ft < nth; ft += NW * MAXT0) {
        const int f0 = 16 * ft + 4 * q;
        f32x4 acc;
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[r] = (f0 + r < H) ? Xdx[i * ldh + f0 + r] : 0.f;
        gru_cell_bwd_tile(acc, carry0, 1.0f / (1.0f - dm.p_drop),
                          drop ? keep_l0 + ((int64_t)(t - 1) * B + b0) * H : nullptr,
                          sv.gates0 + ((int64_t)(t - 1) * B + b0) * 4 * H, sv.h0 + ((int64_t)(t - 1) * B + b0) * H,
                          gr.dgi0 + ((int64_t)(t - 1) * B + b0) * G, gr.dgh0 + ((int64_t)(t - 1) * B + b0) * G, Gi, Gh,
                          ldg, Dd, ldh, H, ft, nrows, lane, dm.wt != 0, pf0 && ft == wave + NW * j, cin0[j]);
      }
    }
  }
  lds_barrier();
  STAMPB(6);
  // ---- carry0' = dh0*z + dgh0 W_hh0 ;  da = dgi0 W_ih0 -> ReLU bwd -> dbn_t + BN-backward partial sums ----
  {
    const float* stats = sv.bn_stats + (int64_t)(t - 1) * 2 * H;
    float* part = gr.bn_bwd_partial + ((int64_t)(t & 1) * dm.nblk + blockIdx.x) * 2 * H;
    for (int ft = wave; ft < nth; ft += NW) {
      const int f0 = 16 * ft + 4 * q;
      const bool vec = hvec && (f0 + 3 < H);
      // inputs of the epilogue that do not depend on the MFMAs
      float av[4] = {0.f, 0.f, 0.f, 0.f}, uv[4] = {0.f, 0.f, 0.f, 0.f}, mv[4] = {0.f, 0.f, 0.f, 0.f}, vv[4] = {1.f, 1.f, 1.f, 1.f};
      if (vec) {
        const float4 m4 = *reinterpret_cast<const float4*>(stats + f0), v4 = *reinterpret_cast<const float4*>(stats + H + f0);
        mv[0] = m4.x; mv[1] = m4.y; mv[2] = m4.z; mv[3] = m4.w;
        vv[0] = v4.x; vv[1] = v4.y; vv[2] = v4.z; vv[3] = v4.w;
        if (i < nrows) {
          const int64_t row = ((int64_t)(t - 1) * B + b0 + i) * H + f0;
          const float4 a4 = *reinterpret_cast<const float4*>(sv.a + row), u4 = *reinterpret_cast<const float4*>(sv.u + row);
          av[0] = a4.x; av[1] = a4.y; av[2] = a4.z; av[3] = a4.w;
          uv[0] = u4.x; uv[1] = u4.y; uv[2] = u4.z; uv[3] = u4.w;
        }
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int f = f0 + r;
          if (f < H) {
            mv[r] = stats[f];
            vv[r] = stats[H + f];
            if (i < nrows) {
              const int64_t row = ((int64_t)(t - 1) * B + b0 + i) * H + f;
              av[r] = sv.a[row];
              uv[r] = sv.u[row];
            }
          }
        }
      }
      f32x4 a1[1] = {(f32x4){0.f, 0.f, 0.f, 0.f}}, a2[1] = {(f32x4){0.f, 0.f, 0.f, 0.f}};
      if constexpr (HS == 0) {
        wave_gemm_p_dual<1, 8>(a1, tw.w_hh0_t, Gh, a2, tw.w_ih0_t, Gi, Gp >> 4, ft, 0, ldg, lane);
      } else {
        wave_gemm_p<1, 3 * HS / 16>(a1, tw.w_hh0_t, Gp >> 4, ft, 0, Gh, ldg, lane);
        wave_gemm_p<1, 3 * HS / 16>(a2, tw.w_ih0_t, Gp >> 4, ft, 0, Gi, ldg, lane);
      }
      float dbn[4], s1[4], s2[4], cw[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int f = f0 + r;
        const bool ok = (f < H) && (i < nrows);
        cw[r] = Dd[i * ldh + ((f < Hp) ? f : 0)] + a1[0][r];
        dbn[r] = (ok && av[r] > 0.f) ? a2[0][r] : 0.f;
        const float invstd = bn_invstd_(vv[r]);
        const float dbx = ok ? dbn[r] * ((uv[r] - mv[r]) * invstd) : 0.f;
        s1[r] = reduce16(dbn[r]);
        s2[r] = reduce16(dbx);
      }
      if (vec) {
        if (i < nrows) {
          *reinterpret_cast<float4*>(carry0_w + (int64_t)i * H + f0) = make_float4(cw[0], cw[1], cw[2], cw[3]);
          *reinterpret_cast<float4*>(gr.dbn + ((int64_t)(t - 1) * B + b0 + i) * H + f0) = make_float4(dbn[0], dbn[1], dbn[2], dbn[3]);
        }
        if (i == 0) {
          *reinterpret_cast<float4*>(part + f0) = make_float4(s1[0], s1[1], s1[2], s1[3]);
          *reinterpret_cast<float4*>(part + H + f0) = make_float4(s2[0], s2[1], s2[2], s2[3]);
        }
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int f = f0 + r;
          if (f >= H) continue;
          if (i < nrows) {
            carry0_w[(int64_t)i * H + f] = cw[r];
            gr.dbn[((int64_t)(t - 1) * B + b0 + i) * H + f] = dbn[r];
          }
          if (i == 0) {
            part[f] = s1[r];
            part[H + f] = s2[r];
          }
        }
      }
    }
  }
  STAMPB(7);
  SPAN(1);
}

}  // namespace g2v

using namespace g2v;

#ifdef G2V_STAMPS
extern "C" int g2v_read_stamps(unsigned long long* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g2v_stamps), sizeof(unsigned long long) * 64 * 16);
}
extern "C" int g2v_read_spans(unsigned long long* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g2v_span), sizeof(unsigned long long) * 1024 * 4);
}
#endif

int g2v_internal_persist_enabled() { return g2v_internal_options().persist != 0 ? 1 : 0; }      // (t2e_rollout.hip: the code decoder's cluster kernel)
extern "C" int g2v_dec_rollout_blocks(int B) { return B > 0 ? cdiv(B, 16) : 0; }

int g2v_internal_preclear_take(const void* p, size_t need);      // dec_persist.hip
void g2v_internal_preclear_note(const void* p, size_t n);
void g2v_internal_preclear_drop(const void* base, size_t bytes);
extern "C" int g2v_dec_rollout_set_persistent(int enable) {      // = g2v_ctx_set_option(NULL, G2V_OPT_PERSISTENT, enable)
  return g2v_ctx_set_option(nullptr, G2V_OPT_PERSISTENT, enable);
}

// reduce_partials scratch: 2048 floats (every row segment of the 512-thread generic kernels) where the LDS has room, else 1024
static int dec_scratch(size_t base_floats) { return (base_floats + 2048) * sizeof(float) <= 160 * 1024 ? 2048 : 1024; }
static size_t dec_fwd_lds(int D, int H, int* scratch = nullptr) {
  const int Hp = (H + 15) & ~15, Dp = (D + 15) & ~15;
  const size_t base = (size_t)(5 * 16 * (Hp + 4) + 16 * (Dp + 4) + 4 * Hp);
  if (scratch) *scratch = dec_scratch(base);
  return (base + dec_scratch(base)) * sizeof(float);
}
static size_t dec_bwd_lds(int D, int H, int* scratch = nullptr) {
  const int Hp = (H + 15) & ~15, Dp = (D + 15) & ~15, Gp = (3 * H + 15) & ~15;
  const size_t base = (size_t)(16 * (3 * (Hp + 4) + (Dp + 4) + 2 * (Gp + 4)) + 4 * Hp);
  if (scratch) *scratch = dec_scratch(base);
  return (base + dec_scratch(base)) * sizeof(float);
}

static int dtiles_pad(int D) {   // D-row matrices: tiles padded to a multiple of 12 (3 per wave) or 4 (1 per wave)
  const int nt = (D + 15) >> 4;
  return nt > 4 ? (nt + 11) / 12 * 12 : 4;
}
static size_t pack_fwd_total(int D, int H) {
  return pack_floats(H, 1, D) + 4 * pack_floats(H, 3, H) + (size_t)dtiles_pad(D) * pack_ks(H) * 256;
}
static size_t pack_bwd_total(int D, int H) {
  return (size_t)dtiles_pad(D) * pack_ks(H) * 256 + pack_floats(H, 1, D) + 4 * pack_floats(H, 1, 3 * H);
}

// [packed forward weights | (256-byte aligned) exchange state of the persistent kernel]
static size_t fwd_pack_bytes_aligned(int D, int H) { return (pack_fwd_total(D, H) * sizeof(float) + 255) / 256 * 256; }
// exchange records of dec_cluster_fwd_kernel: three (16 x Hp) row records + one (2 x Hp) record of partial sums per (parity, row group)
static size_t dec_cluster_fwd_xch_bytes(int nblk, int H) {      // (+ one XCC word per workgroup)
  const size_t Hp = (size_t)((H + 15) & ~15);
  return (size_t)2 * nblk * (3 * 16 + 2) * Hp * 8 + (size_t)nblk * (Hp / 16) * 4 + 16;
}
// (H <= 208 = 13 k-steps: with 16 the weight fragments + a sweep's granules no longer fit the register file -- 186 spills)
constexpr int DCL_KS = 13;
// backward: dbn row records + partial sums (both parities) and four partial-product row records per (row group, producer tile)
static size_t dec_cluster_bwd_xch_bytes(int nblk, int H) {
  const size_t Hp = (size_t)((H + 15) & ~15), nt = Hp / 16;
  return ((size_t)2 * nblk * (16 + 2) * Hp + (size_t)nblk * 4 * nt * 16 * Hp) * 8 + (size_t)nblk * nt * 4 + 16;
}
static size_t dec_cluster_bwd_dyn_lds() { return (size_t)4 * 13 /* DCL_KS */ * 64 * sizeof(float4); }
static size_t dec_cluster_fwd_dyn_lds() {      // W_out fragments, partial out-layer products, Dropout(h0) rows, input-side accumulators
  return ((size_t)4 * DCL_KS + (size_t)4 * 4 /* DSPLIT_DT */ + (size_t)DCL_KS + 3) * 64 * sizeof(float4);
}
static bool dec_cluster_shape(int D, int H) { return !(H == 64 && D == 135) && (H & 3) == 0 && H <= 16 * DCL_KS && D <= 64 && H >= 4; }
extern "C" size_t g2v_dec_rollout_fwd_workspace(int D, int H) {
  size_t x = PX_BYTES;
  if (dec_cluster_shape(D, H)) {      // the largest grid the cluster kernel is admitted for: one workgroup per CU
    const int nt = (H + 15) >> 4, cus = device_cu_count() > 0 ? device_cu_count() : 256;
    const size_t c = dec_cluster_fwd_xch_bytes(cus / nt + 1, H);
    if (c > x) x = c;
  }
  return fwd_pack_bytes_aligned(D, H) + x;
}

// ====================================================================================================================
// Small batch, generic dims: the forward step t >= 1 as THREE launches over (16 rows x 16 hidden units) workgroups.
// dec_step_fwd_kernel gives a step only B/16 workgroups, each walking all of the step's ~16 k MFMAs (H = 200) on one CU:
// 95 us per step at B = 128 with 8 of 256 CUs busy.  A GRU cell's hidden unit needs only its own rows of W_ih / W_hh, and
// a pre_linear feature only its own row of W_pre, so the step splits over 16-unit tiles; what does need a whole row of the
// previous stage (the cell inputs, the out layer) makes a launch boundary:
//   dec_cell_split_kernel<true>  : [finish BN(u_t)] a_t = ReLU(BN(u_t)), GRU cell 0 for the tile  -> h0_t, gates0, x1
//   dec_cell_split_kernel<false> : GRU cell 1 for the tile                                         -> h1_t, gates1
//   dec_out_pre_split_kernel     : y_t = out_layer(h1_t) (recomputed by every tile workgroup: <= 4 D tiles), Dropout(0.95),
//                                  u_{t+1} tile = pre_linear(xin_{t+1}) + the BN partial sums of the tile
// Operands are requested up front as MFMA fragments in registers (one memory round trip per launch); the y accumulators of
// the out layer ARE the B fragments of the pre_linear product (same lane <-> (row, 4 consecutive columns) mapping).
// Writes exactly the arrays dec_step_fwd_kernel writes, so the backward is unchanged.  H % 4 == 0, H <= 256, D <= 64.
// ====================================================================================================================
constexpr int DSPLIT_KS = 16;       // k-steps over H (H <= 256)
constexpr int DSPLIT_MAX_NBLK = 64; // row groups (B <= 1024): beyond that the fused step kernels fill the chip on their own
constexpr int DSPLIT_DT = 4;        // D tiles (D <= 64)

// column f of the (nblk, 2H) per-row-group sums: eight row groups' loads in flight at a time, fixed summation order
__device__ __forceinline__ void sum_partials(const float* __restrict__ part, int nblk, int H, int f, float& s1, float& s2) {
  s1 = 0.f; s2 = 0.f;
  for (int k0 = 0; k0 < nblk; k0 += 8) {
    float v1[8], v2[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const bool ok = k0 + j < nblk;
      const float* p = part + (int64_t)(ok ? k0 + j : 0) * 2 * H;
      v1[j] = ok ? p[f] : 0.f;
      v2[j] = ok ? p[H + f] : 0.f;
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) { s1 += v1[j]; s2 += v2[j]; }
  }
}

struct DecCellArgs {
  const float* x;         // (B,H) rows of the cell input (u_t for the BN variant)
  const float* h_prev;    // (B,H)
  const float* w_ih; const float* w_hh; const float* b_ih; const float* b_hh;   // (3H,H) row-major, (3H)
  float* h_out;           // (B,H)
  float* gates;           // (B,4H) or NULL
  const uint8_t* keep;    // (B,H) inter-layer dropout keep flags or NULL
  float keep_scale;
  float* xdrop_out;       // (B,H) or NULL
  // BN variant only
  const float* bn_partial;   // (nblk, 2H) sums of (u - b), (u - b)^2 of this step
  const float* b_pre; const float* bn_w; const float* bn_b; const float* run_mean; const float* run_var;
  float* a_out;           // (B,H) or NULL
  float* bn_stats;        // (2H) mean / biased var of this step (training) or NULL
  int nblk, training;
};

template <bool BN>
__global__ __launch_bounds__(128) void dec_cell_split_kernel(DecCellArgs a, int B, int H) {
  __shared__ float st[2 * 16 * DSPLIT_KS];            // mean[H], invstd[H] (BN variant)
  __shared__ __attribute__((aligned(16))) float4 xch[3 * 64];   // h-side accumulators handed to the x-side wave
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i = lane & 15, q = lane >> 4;
  const int b0 = blockIdx.x * 16, ft = blockIdx.y;
  const int nrows = min(16, B - b0);
  const int nks = (H + 15) >> 4;
  const bool rvalid = i < nrows, wrow_ok = 16 * ft + i < H;
  const int b = b0 + (rvalid ? i : 0);
  // ---- this wave's operands: wave 0 = input side (W_ih, x), wave 1 = hidden side (W_hh, h_prev) -------------------------
  const float* W = (wave == 0 ? a.w_ih : a.w_hh) + (int64_t)(16 * ft + (wrow_ok ? i : 0)) * H;
  const float* X = (wave == 0 ? a.x : a.h_prev) + (int64_t)b * H;
  float4 wa[3][DSPLIT_KS], xb[DSPLIT_KS];
#pragma unroll
  for (int ks = 0; ks < DSPLIT_KS; ++ks) {
    const int k = 16 * ks + 4 * q;
    const bool kok = ks < nks && k < H;
#pragma unroll
    for (int g = 0; g < 3; ++g) wa[g][ks] = ld4_or_zero(W + (int64_t)g * H * H + (kok ? k : 0), kok && wrow_ok);
    xb[ks] = ld4_or_zero(X + (kok ? k : 0), kok && rvalid);
  }
  // epilogue inputs of the x-side wave (4 hidden units f0 .. f0 + 3 of batch row i)
  const int f0 = 16 * ft + 4 * q;
  const bool fok = f0 < H;
  float4 bi[3], bh[3], hp4 = make_float4(0.f, 0.f, 0.f, 0.f);
  uint32_t kp = 0x01010101u;
  if (wave == 0) {
#pragma unroll
    for (int g = 0; g < 3; ++g) {
      bi[g] = ld4_or_zero(a.b_ih + g * H + (fok ? f0 : 0), fok);
      bh[g] = ld4_or_zero(a.b_hh + g * H + (fok ? f0 : 0), fok);
    }
    hp4 = ld4_or_zero(a.h_prev + (int64_t)b * H + (fok ? f0 : 0), fok && rvalid);
    if (a.keep && fok && rvalid) kp = *reinterpret_cast<const uint32_t*>(a.keep + (int64_t)b * H + f0);
  }
  if constexpr (BN) {
    // ---- BatchNorm statistics of the step for every feature (each workgroup needs the whole input row) -----------------
    for (int f = tid; f < H; f += 128) {
      float mean, var;
      if (a.training) {
        float s1, s2;
        sum_partials(a.bn_partial, a.nblk, H, f, s1, s2);
        const float mv = s1 / (float)B;
        var = fmaxf(s2 / (float)B - mv * mv, 0.f);     // biased batch variance
        mean = mv + a.b_pre[f];
        if (a.bn_stats && blockIdx.x == 0 && blockIdx.y == 0) {
          a.bn_stats[f] = mean;
          a.bn_stats[H + f] = var;
        }
      } else {
        mean = a.run_mean[f];
        var = a.run_var[f];
      }
      st[f] = mean;
      st[16 * DSPLIT_KS + f] = bn_invstd_(var);
    }
    __syncthreads();
    if (wave == 0) {
      // a_t = ReLU(BN(u_t)) on the fragments; the workgroup's own 16 columns (k-step ft) are the ones it writes out
#pragma unroll
      for (int ks = 0; ks < DSPLIT_KS; ++ks) {
        const int k = 16 * ks + 4 * q;
        if (ks < nks && k < H) {
          const float4 g4 = *reinterpret_cast<const float4*>(a.bn_w + k), b4 = *reinterpret_cast<const float4*>(a.bn_b + k);
          const float4 m4 = *reinterpret_cast<const float4*>(st + k), i4 = *reinterpret_cast<const float4*>(st + 16 * DSPLIT_KS + k);
          float4 v = xb[ks];
          v.x = fmaxf((v.x - m4.x) * i4.x * g4.x + b4.x, 0.f);
          v.y = fmaxf((v.y - m4.y) * i4.y * g4.y + b4.y, 0.f);
          v.z = fmaxf((v.z - m4.z) * i4.z * g4.z + b4.z, 0.f);
          v.w = fmaxf((v.w - m4.w) * i4.w * g4.w + b4.w, 0.f);
          xb[ks] = rvalid ? v : make_float4(0.f, 0.f, 0.f, 0.f);
          if (ks == ft && a.a_out && rvalid) *reinterpret_cast<float4*>(a.a_out + (int64_t)b * H + k) = v;
        }
      }
    }
  }
  // ---- products -----------------------------------------------------------------------------------------------------------
  f32x4 acc[3];
#pragma unroll
  for (int g = 0; g < 3; ++g) acc[g] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int ks = 0; ks < DSPLIT_KS; ++ks) {
    if (ks < nks) {
#pragma unroll
      for (int g = 0; g < 3; ++g) {
        acc[g] = mfma16(wa[g][ks].x, xb[ks].x, acc[g]);
        acc[g] = mfma16(wa[g][ks].y, xb[ks].y, acc[g]);
        acc[g] = mfma16(wa[g][ks].z, xb[ks].z, acc[g]);
        acc[g] = mfma16(wa[g][ks].w, xb[ks].w, acc[g]);
      }
    }
  }
  if (wave == 1) {
#pragma unroll
    for (int g = 0; g < 3; ++g) xch[g * 64 + lane] = make_float4(acc[g][0], acc[g][1], acc[g][2], acc[g][3]);
  }
  __syncthreads();
  if (wave != 0 || !rvalid || !fok) return;
  // ---- GRU cell epilogue (same arithmetic order as gru_cell_fwd_epilogue) -------------------------------------------------
  float ah[3][4];
#pragma unroll
  for (int g = 0; g < 3; ++g) {
    const float4 v = xch[g * 64 + lane];
    ah[g][0] = v.x; ah[g][1] = v.y; ah[g][2] = v.z; ah[g][3] = v.w;
  }
  const float hp[4] = {hp4.x, hp4.y, hp4.z, hp4.w};
  const float bir[4] = {bi[0].x, bi[0].y, bi[0].z, bi[0].w}, biz[4] = {bi[1].x, bi[1].y, bi[1].z, bi[1].w},
              bin[4] = {bi[2].x, bi[2].y, bi[2].z, bi[2].w};
  const float bhr[4] = {bh[0].x, bh[0].y, bh[0].z, bh[0].w}, bhz[4] = {bh[1].x, bh[1].y, bh[1].z, bh[1].w},
              bhn[4] = {bh[2].x, bh[2].y, bh[2].z, bh[2].w};
  float hn[4], xd[4], gr_[4], gz_[4], gn_[4], gh_[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const float rr = sigmoidf_((acc[0][r] + bir[r]) + (ah[0][r] + bhr[r]));
    const float zz = sigmoidf_((acc[1][r] + biz[r]) + (ah[1][r] + bhz[r]));
    const float ghn = ah[2][r] + bhn[r];
    const float nn = tanhf_((acc[2][r] + bin[r]) + rr * ghn);
    hn[r] = (1.0f - zz) * nn + zz * hp[r];
    xd[r] = a.keep ? (((kp >> (8 * r)) & 0xffu) ? hn[r] * a.keep_scale : 0.f) : hn[r];
    gr_[r] = rr; gz_[r] = zz; gn_[r] = nn; gh_[r] = ghn;
  }
  *reinterpret_cast<float4*>(a.h_out + (int64_t)b * H + f0) = make_float4(hn[0], hn[1], hn[2], hn[3]);
  if (a.xdrop_out) *reinterpret_cast<float4*>(a.xdrop_out + (int64_t)b * H + f0) = make_float4(xd[0], xd[1], xd[2], xd[3]);
  if (a.gates) {
    float* go = a.gates + (int64_t)b * 4 * H + f0;
    *reinterpret_cast<float4*>(go) = make_float4(gr_[0], gr_[1], gr_[2], gr_[3]);
    *reinterpret_cast<float4*>(go + H) = make_float4(gz_[0], gz_[1], gz_[2], gz_[3]);
    *reinterpret_cast<float4*>(go + 2 * H) = make_float4(gn_[0], gn_[1], gn_[2], gn_[3]);
    *reinterpret_cast<float4*>(go + 3 * H) = make_float4(gh_[0], gh_[1], gh_[2], gh_[3]);
  }
}

struct DecOutPreArgs {
  const float* h1;        // (B,H) h1_t
  const float* w_out; const float* b_out;   // (D,H), (D)
  const float* w_pre; const float* b_pre;   // (H,D), (H)
  const float* target;    // (B,T,D)
  const uint8_t* keep95;  // (B,D) flags of xin_{t+1} (index t of the (T-1,B,D) array)
  float* y;               // (B,D) y_t
  float* xin;             // (B,D) xin_{t+1} or NULL
  float* u_next;          // (B,H) u_{t+1}
  float* part;            // (nblk, 2H) BN partial sums of u_{t+1}
  int t, T, has_next, teacher, conditioned;
};

__global__ __launch_bounds__(64) void dec_out_pre_split_kernel(DecOutPreArgs a, int B, int D, int H) {
  const int lane = threadIdx.x, i = lane & 15, q = lane >> 4;
  const int b0 = blockIdx.x * 16, ft = blockIdx.y;
  const int nrows = min(16, B - b0);
  const int nks = (H + 15) >> 4, ndt = (D + 15) >> 4;
  const bool rvalid = i < nrows;
  const int b = b0 + (rvalid ? i : 0);
  // ---- every load of the launch -------------------------------------------------------------------------------------------
  float4 wo[DSPLIT_DT][DSPLIT_KS], xb[DSPLIT_KS];
  const float* X = a.h1 + (int64_t)b * H;
#pragma unroll
  for (int ks = 0; ks < DSPLIT_KS; ++ks) {
    const int k = 16 * ks + 4 * q;
    const bool kok = ks < nks && k < H;
    xb[ks] = ld4_or_zero(X + (kok ? k : 0), kok && rvalid);
#pragma unroll
    for (int dt = 0; dt < DSPLIT_DT; ++dt) {
      const int d = 16 * dt + i;
      const bool dok = dt < ndt && d < D;
      wo[dt][ks] = ld4_or_zero(a.w_out + (int64_t)(dok ? d : 0) * H + (kok ? k : 0), kok && dok);
    }
  }
  // lane (i, q) ends up with y[row i][16 dt + 4 q + r]: bias, target and keep flags of exactly those elements
  float bo[DSPLIT_DT][4], tg[DSPLIT_DT][4];
  uint8_t kp[DSPLIT_DT][4];
#pragma unroll
  for (int dt = 0; dt < DSPLIT_DT; ++dt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int d = 16 * dt + 4 * q + r;
      const bool ok = dt < ndt && d < D && rvalid;
      bo[dt][r] = ok ? a.b_out[d] : 0.f;
      tg[dt][r] = (ok && a.has_next && a.teacher) ? a.target[((int64_t)b * a.T + a.t) * D + d] : 0.f;
      kp[dt][r] = (ok && a.has_next && a.conditioned) ? a.keep95[(int64_t)b * D + d] : 0;
    }
  // pre_linear fragments of this workgroup's 16 features: W_pre[16 ft + i][d], d along k (row stride D: scalar loads)
  float4 wp[DSPLIT_DT];
  const bool prow_ok = 16 * ft + i < H;
#pragma unroll
  for (int dt = 0; dt < DSPLIT_DT; ++dt) {
    float v[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int d = 16 * dt + 4 * q + e;
      v[e] = (a.has_next && prow_ok && dt < ndt && d < D) ? a.w_pre[(int64_t)(16 * ft + i) * D + d] : 0.f;
    }
    wp[dt] = make_float4(v[0], v[1], v[2], v[3]);
  }
  const int f0 = 16 * ft + 4 * q;
  const bool fok = f0 < H;
  const float4 bp = ld4_or_zero(a.b_pre + (fok ? f0 : 0), fok && a.has_next);
  // ---- y_t = out_layer(h1_t) ----------------------------------------------------------------------------------------------
  f32x4 ya[DSPLIT_DT];
#pragma unroll
  for (int dt = 0; dt < DSPLIT_DT; ++dt) ya[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int ks = 0; ks < DSPLIT_KS; ++ks) {
    if (ks < nks) {
#pragma unroll
      for (int dt = 0; dt < DSPLIT_DT; ++dt) {
        if (dt < ndt) {
          ya[dt] = mfma16(wo[dt][ks].x, xb[ks].x, ya[dt]);
          ya[dt] = mfma16(wo[dt][ks].y, xb[ks].y, ya[dt]);
          ya[dt] = mfma16(wo[dt][ks].z, xb[ks].z, ya[dt]);
          ya[dt] = mfma16(wo[dt][ks].w, xb[ks].w, ya[dt]);
        }
      }
    }
  }
  // y, the next decoder input xin = Dropout(0.95)(teacher ? target : y) (zeros when !conditioned), kept as MFMA B fragments
  float4 xf[DSPLIT_DT];
#pragma unroll
  for (int dt = 0; dt < DSPLIT_DT; ++dt) {
    float xv[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int d = 16 * dt + 4 * q + r;
      const float y = ya[dt][r] + bo[dt][r];
      const float src = a.teacher ? tg[dt][r] : y;                                  // :1049-1052
      xv[r] = kp[dt][r] ? src * 20.0f : 0.f;                                         // Dropout(0.95): 1/(1-0.95)
      if (ft == 0 && rvalid && dt < ndt && d < D) {
        a.y[(int64_t)b * D + d] = y;
        if (a.has_next && a.xin) a.xin[(int64_t)b * D + d] = xv[r];
      }
    }
    xf[dt] = make_float4(xv[0], xv[1], xv[2], xv[3]);
  }
  if (!a.has_next) return;
  // ---- u_{t+1} tile = pre_linear.0(xin_{t+1}) and the BN partial sums of (u - b) over this workgroup's rows -------------------
  f32x4 ua = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int dt = 0; dt < DSPLIT_DT; ++dt) {
    if (dt < ndt) {
      ua = mfma16(wp[dt].x, xf[dt].x, ua);
      ua = mfma16(wp[dt].y, xf[dt].y, ua);
      ua = mfma16(wp[dt].z, xf[dt].z, ua);
      ua = mfma16(wp[dt].w, xf[dt].w, ua);
    }
  }
  float s1[4], s2[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const float v = (rvalid && fok) ? ua[r] : 0.f;
    s1[r] = reduce16(v);
    s2[r] = reduce16(v * v);
  }
  if (!fok) return;
  if (rvalid)
    *reinterpret_cast<float4*>(a.u_next + (int64_t)b * H + f0) = make_float4(ua[0] + bp.x, ua[1] + bp.y, ua[2] + bp.z, ua[3] + bp.w);
  if (i == 0) {
    float* part = a.part + (int64_t)blockIdx.x * 2 * H;
    *reinterpret_cast<float4*>(part + f0) = make_float4(s1[0], s1[1], s1[2], s1[3]);
    *reinterpret_cast<float4*>(part + H + f0) = make_float4(s2[0], s2[1], s2[2], s2[3]);
  }
}

// ---- the same three stages for ALL steps t >= 1 in ONE launch (round 5): a persistent cluster of the tile workgroups -------------
// At B = 128, H = 200 the three launches of a step take 27 us for ~5 us of arithmetic: each is a kernel boundary plus two or three
// dependent memory round trips from cold registers, and each re-requests its weight rows.  Here the (hidden-unit tile x row group)
// workgroups stay resident for the whole rollout -- four waves, one per (cell, side): W_ih0 / W_hh0 / W_ih1 / W_hh1 rows of the tile
// in registers, plus one D tile of W_out each -- and what the kernel boundaries did is an exchange of 8-byte self-validating
// granules {value, tag = step} through memory (write-through stores, sc1 loads; gru.hip's cluster kernels and dec_persist.hpp have
// the protocol and its measurements):
//   u_t tiles + the BatchNorm partial sums of every row group  (out/pre stage of step t-1  ->  cell 0 of step t)
//   h0_t tiles of the row group                                (cell 0 -> cell 1; kept in LDS for cell 0's hidden side at t+1)
//   h1_t tiles of the row group                                (cell 1 -> out layer; kept in LDS for cell 1's hidden side at t+1)
// A record is swept by ONE wave per workgroup (sc1 loads are served at the fabric, ~3 TB/s chip-wide) and passed on through LDS.
// Arithmetic and summation orders are those of the three kernels above (same saved arrays, same backward).  Records are double-
// buffered by step parity; a workgroup reaches step t+2 only after every workgroup has published step t+1, i.e. consumed step t.
// Residency: every workgroup of the grid must be resident at once (the BN sums couple all row groups): the launcher admits the
// path while the grid has at most one workgroup per CU; every spin is bounded and latches the persistent kernels' fault word.
#ifdef G2V_STAMPS      // diagnostic build only (gpurun_tools/stamps_dcl.py): s_memtime stamps of step t = 5, two workgroups, every wave
#define DCL_STAMP(k)                                                                                                              \
  do {                                                                                                                            \
    const int sw_ = (blockIdx.x == 0 && blockIdx.y == 0) ? 0 : ((blockIdx.x == 5 && blockIdx.y == 3) ? 1 : -1);                  \
    if ((threadIdx.x & 63) == 0 && sw_ >= 0 && t == 5) g2v_stamps[(sw_ * 4 + (threadIdx.x >> 6)) * 16 + (k)] = __builtin_amdgcn_s_memtime(); \
  } while (0)
#define DCB_STAMP(k)                                                                                                              \
  do {                                                                                                                            \
    const int sw_ = (blockIdx.x == 0 && blockIdx.y == 0) ? 0 : ((blockIdx.x == 5 && blockIdx.y == 3) ? 1 : -1);                  \
    if ((threadIdx.x & 63) == 0 && sw_ >= 0 && t == 5) g2v_stamps[256 + (sw_ * 4 + (threadIdx.x >> 6)) * 16 + (k)] = __builtin_amdgcn_s_memtime(); \
  } while (0)
#else
#define DCL_STAMP(k)
#define DCB_STAMP(k)
#endif
struct DecClFwdArgs {
  const float* target; const float* h_init; const uint8_t* keep95; const uint8_t* keep_l0;
  g2v_dec_weights w; g2v_dec_saved sv;
  unsigned long long* xu;      // [2][nblk][16][Hp]  u rows
  unsigned long long* xp;      // [2][nblk][2][Hp]   BN partial sums of (u - b), (u - b)^2 per row group
  unsigned long long* xh0;     // [2][nblk][16][Hp]
  unsigned long long* xh1;     // [2][nblk][16][Hp]
  unsigned* xcc;               // [nblk][nt]: the XCC every workgroup runs on (cx_cluster_on_one_xcd)
  unsigned* fault;
  int T, B, D, H, n_pre, conditioned, training, nt, nblk;
  float p_drop;
};

__device__ __forceinline__ void dcl_publish4(__amdgpu_buffer_rsrc_t rr, unsigned granule, const float* v, unsigned tag) {
  u32x4 a, b;
  a[0] = __float_as_uint(v[0]); a[1] = tag; a[2] = __float_as_uint(v[1]); a[3] = tag;
  b[0] = __float_as_uint(v[2]); b[1] = tag; b[2] = __float_as_uint(v[3]); b[3] = tag;
  px_st(rr, granule * 8u, a);
  px_st(rr, granule * 8u + 16u, b);
}
// column f of the per-row-group partial sums, from the granule records [nblk][2][Hp]: the order of sum_partials()
__device__ __forceinline__ void dcl_sum_partials(const unsigned long long* rec, int nblk, int Hp, int f, unsigned tag, unsigned* fault,
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

// GRU cell epilogue of the lane's 4 units (the arithmetic of dec_cell_split_kernel); bs: [b_ih r z n, b_hh r z n][q] in LDS
__device__ __forceinline__ void dcl_cell_epilogue(const float4* xcx, const float4* xch, const float4 (*bs)[4], int lane, int q,
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

template <int KS>      // k-steps over H the kernel is built for: 13 (H <= 208: the shipped configurations' H = 200)
__global__ __launch_bounds__(256) void dec_cluster_fwd_kernel(DecClFwdArgs a) {
  __shared__ float st[2 * 16 * DSPLIT_KS];                                        // mean[H], invstd[H] (offsets as in the split kernels)
  __shared__ __attribute__((aligned(16))) float4 xch2[2][3 * 64];                // [cell][gate] hidden-side accumulators
  __shared__ __attribute__((aligned(16))) float4 xs_u[KS][64];                   // u_t rows as B fragments ([k-step][lane]); then a_t
  __shared__ __attribute__((aligned(16))) float4 xs_h0[KS][64];                  // h0 rows
  __shared__ __attribute__((aligned(16))) float4 xs_h1[KS][64];                  // h1 rows
  __shared__ __attribute__((aligned(16))) float4 xfs[DSPLIT_DT][64];             // xin_{t+1} as B fragments, one D tile per wave
  __shared__ __attribute__((aligned(16))) float4 bias_s[2][6][4];                // [cell][b_ih r z n, b_hh r z n][q]: the tile's units
  __shared__ __attribute__((aligned(16))) float4 wp_s[DSPLIT_DT + 1][64];        // pre_linear fragments of the tile + its bias
  __shared__ __attribute__((aligned(16))) float bnw_s[2 * 16 * DSPLIT_KS];       // BatchNorm weight[H], bias[H] (same offsets as st)
  extern __shared__ __attribute__((aligned(16))) float4 wo_s[];                  // [D tile][KS][64]: W_out fragments; [wave][D tile][64]: the
                                                                                  // waves' partial out-layer products; [KS][64]: Dropout(h0_t)
                                                                                  // rows; [gate][64]: input-side accumulators
  const int T = a.T, B = a.B, D = a.D, H = a.H;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, i = lane & 15, q = lane >> 4;
  // wave 1 multiplies the hidden side of cell 0 (all three gates of W_hh0); the waves 0, 2, 3 own one GATE each: its rows of W_ih0,
  // W_ih1 (the two input sides, on the critical path: 52 MFMAs per wave instead of 156 on one) and of W_hh1
  const bool xw = wave != 1;
  const int gx = wave == 0 ? 0 : wave - 1;
  // 1-D grid.  With nblk % 8 == 0 the row group is the FAST index of the linear workgroup id: under the dispatcher's round-robin
  // placement (workgroup id % 8 = XCD; observed, verified per launch below) the nt tile workgroups of a row group then share an XCD
  const int nt = a.nt, nblk = a.nblk;
  const bool rg_fast = (nblk & 7) == 0;
  const int ft = rg_fast ? (int)blockIdx.x / nblk : (int)blockIdx.x % nt, rg = rg_fast ? (int)blockIdx.x % nblk : (int)blockIdx.x / nt;
  const int b0 = rg * 16;
  const int nrows = min(16, B - b0);
  const int Hp = nt << 4, ndt = (D + 15) >> 4;
  const bool rvalid = i < nrows, wrow_ok = 16 * ft + i < H;
  const int b = b0 + (rvalid ? i : 0);
  const int f0 = 16 * ft + 4 * q;
  const bool fok = f0 < H;
  const int64_t BH = (int64_t)B * H, BD = (int64_t)B * D;
  const g2v_dec_weights& w = a.w;
  const g2v_dec_saved& sv = a.sv;
  // ---- resident operands (zero beyond H: the products below run over all KS k-steps unconditionally) ---------------------------
  float4 wreg[3][KS];      // wave 1: W_hh0 gates r, z, n; the others: W_ih0[gx], W_ih1[gx], W_hh1[gx]
  const int drow = 16 * wave + i;                               // this wave's D tile of the out layer: dt = wave
  const bool dok = wave < ndt && drow < D;
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
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    const int k = 16 * ks + 4 * q;
    const bool kok = k < H;
    wo_s[(wave * KS + ks) * 64 + lane] = ld4_or_zero(w.w_out + (int64_t)(dok ? drow : 0) * H + (kok ? k : 0), kok && dok);
  }
  // operands that are read once per step live in LDS (the weight fragments leave no registers for them)
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
  // out layer: lane (i, q) of wave dt ends up with y[row i][16 dt + 4 q + r]
  float bo[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int d = 16 * wave + 4 * q + r;
    bo[r] = (wave < ndt && d < D) ? w.b_out[d] : 0.f;
  }
  if (wave == 0) {      // pre_linear fragments of the tile's 16 features: W_pre[16 ft + i][d], d along k
#pragma unroll
    for (int dt = 0; dt < DSPLIT_DT; ++dt) {
      float v[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int d = 16 * dt + 4 * q + e;
        v[e] = (wrow_ok && dt < ndt && d < D) ? w.w_pre[(int64_t)(16 * ft + i) * D + d] : 0.f;
      }
      wp_s[dt][lane] = make_float4(v[0], v[1], v[2], v[3]);
    }
    wp_s[DSPLIT_DT][lane] = ld4_or_zero(w.b_pre + (fok ? f0 : 0), fok);
  }
  for (int f = tid; f < H; f += 256) {
    bnw_s[f] = w.bn_w[f];
    bnw_s[16 * DSPLIT_KS + f] = w.bn_b[f];
  }
  float4* yp_s = wo_s + (size_t)4 * KS * 64;        // [wave][D tile][64]
  float4* xs_x1 = yp_s + (size_t)4 * DSPLIT_DT * 64; // [KS][64]
  float4* xcx = xs_x1 + (size_t)KS * 64;             // [gate][64]
  // the state rows entering step 1 = the initial states (the quantised latent): as fragments, zeros in rows / columns that do not
  // exist (the sweeps never touch those entries)
  for (int ks = wave; ks < KS; ks += 4) {
    const int k = 16 * ks + 4 * q;
    const bool ok = k < H && rvalid;
    xs_u[ks][lane] = make_float4(0.f, 0.f, 0.f, 0.f);
    xs_h0[ks][lane] = ld4_or_zero(a.h_init + (int64_t)b * H + (ok ? k : 0), ok);
    xs_h1[ks][lane] = ld4_or_zero(a.h_init + BH + (int64_t)b * H + (ok ? k : 0), ok);
    xs_x1[ks * 64 + lane] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  const bool drop = a.training && a.keep_l0 && a.p_drop > 0.f;
  const float scale_l0 = 1.0f / (1.0f - a.p_drop);
  const unsigned rowrec = 256u * (unsigned)nt, prec = 2u * (unsigned)Hp;
  __amdgpu_buffer_rsrc_t r_u = __builtin_amdgcn_make_buffer_rsrc(a.xu, 0, (int)(2u * (unsigned)nblk * rowrec * 8u), 0x00020000);
  __amdgpu_buffer_rsrc_t r_p = __builtin_amdgcn_make_buffer_rsrc(a.xp, 0, (int)(2u * (unsigned)nblk * prec * 8u), 0x00020000);
  __amdgpu_buffer_rsrc_t r_h0 = __builtin_amdgcn_make_buffer_rsrc(a.xh0, 0, (int)(2u * (unsigned)nblk * rowrec * 8u), 0x00020000);
  __amdgpu_buffer_rsrc_t r_h1 = __builtin_amdgcn_make_buffer_rsrc(a.xh1, 0, (int)(2u * (unsigned)nblk * rowrec * 8u), 0x00020000);
  float hown[4] = {0.f, 0.f, 0.f, 0.f};      // waves 0 / 2: this tile's own h0 / h1 entering the step (the lane's 4 units)
  if ((wave == 0 || wave == 2) && rvalid && fok) {
    const float4 v = *reinterpret_cast<const float4*>(a.h_init + (wave == 0 ? 0 : BH) + (int64_t)b * H + f0);
    hown[0] = v.x; hown[1] = v.y; hown[2] = v.z; hown[3] = v.w;
    *reinterpret_cast<float4*>((wave == 0 ? sv.h0 : sv.h1) + (int64_t)b * H + f0) = v;      // index 0 of the state arrays
  }
  __shared__ int xcd_flag;
  const bool l2x = cx_cluster_on_one_xcd(a.xcc + (size_t)rg * nt, nt, ft, &xcd_flag, tid, a.fault);      // (also: the LDS operands above are complete)
  for (int t = 0; t < T; ++t) {
    // (t = 0: y_0 = the target's first frame, no cells: only the out / pre_linear stage below, which publishes u_1 with tag 1)
    const unsigned par_prev = (unsigned)((t + 1) & 1), par = (unsigned)(t & 1), tag = (unsigned)t;
    DCL_STAMP(0);
   if (t > 0) {
    // ---- the hidden sides first (their operands have been in LDS since the previous step): wave 1 all of cell 0's, the gate waves
    // their gate of cell 1's; then the gate waves sweep the u_t row and add up the BatchNorm sums ------------------------------------
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
      cx_sweep_tiles<(KS + 2) / 3>(r_u, (par_prev * (unsigned)nblk + (unsigned)rg) * rowrec, gx, 3, nt, nrows, H, tag, &xs_u[0][0], lane, a.fault);
    }
    // ---- BatchNorm statistics of step t: every feature (each workgroup needs the whole input row) -------------------------------
    for (int f = gx * 64 + lane; f < H && xw; f += 192) {
      float mean, var;
      if (a.training) {
        float s1, s2;
        dcl_sum_partials(a.xp + (size_t)par_prev * nblk * prec, nblk, Hp, f, tag, a.fault, s1, s2);
        const float mv = s1 / (float)B;
        var = fmaxf(s2 / (float)B - mv * mv, 0.f);     // biased batch variance
        mean = mv + w.b_pre[f];
        if (sv.bn_stats && ft == 0 && rg == 0) {
          sv.bn_stats[(int64_t)(t - 1) * 2 * H + f] = mean;
          sv.bn_stats[(int64_t)(t - 1) * 2 * H + H + f] = var;
        }
      } else {
        mean = w.bn_running_mean[f];
        var = w.bn_running_var[f];
      }
      st[f] = mean;
      st[16 * DSPLIT_KS + f] = bn_invstd_(var);
    }
    DCL_STAMP(1);
    lds_barrier();
    DCL_STAMP(2);
    // ---- a_t = ReLU(BN(u_t)) in place, a third of the k-steps per gate wave; the workgroup's own 16 columns (k-step ft) are written out ----
    if (xw) {
      if (wave == 0 && drop && fok && rvalid) kp = *reinterpret_cast<const uint32_t*>(a.keep_l0 + (int64_t)(t - 1) * BH + (int64_t)b * H + f0);
#pragma unroll
      for (int j = 0; j < (KS + 2) / 3; ++j) {
        const int ks = gx + 3 * j;
        if (ks < KS) {      // (uniform)
          const int k = 16 * ks + 4 * q;
          const bool kok = k < H;
          const int kk = kok ? k : 0;
          const float4 g4 = *reinterpret_cast<const float4*>(bnw_s + kk), b4 = *reinterpret_cast<const float4*>(bnw_s + 16 * DSPLIT_KS + kk);
          const float4 m4 = *reinterpret_cast<const float4*>(st + kk), i4 = *reinterpret_cast<const float4*>(st + 16 * DSPLIT_KS + kk);
          float4 v = xs_u[ks][lane];
          v.x = fmaxf((v.x - m4.x) * i4.x * g4.x + b4.x, 0.f);
          v.y = fmaxf((v.y - m4.y) * i4.y * g4.y + b4.y, 0.f);
          v.z = fmaxf((v.z - m4.z) * i4.z * g4.z + b4.z, 0.f);
          v.w = fmaxf((v.w - m4.w) * i4.w * g4.w + b4.w, 0.f);
          xs_u[ks][lane] = (kok && rvalid) ? v : make_float4(0.f, 0.f, 0.f, 0.f);
          if (ks == ft && sv.a && rvalid && kok) *reinterpret_cast<float4*>(sv.a + (int64_t)(t - 1) * BH + (int64_t)b * H + k) = v;
        }
      }
    }
    lds_barrier();
    DCL_STAMP(3);
    // ---- cell 0, input side: gate gx of a_t W_ih0 per gate wave ------------------------------------------------------------------------
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
    DCL_STAMP(4);
    lds_barrier();
    DCL_STAMP(5);
    if (wave == 0 && rvalid && fok) {
      float hn[4], xd[4], gr_[4], gz_[4], gn_[4], gh_[4];
      dcl_cell_epilogue(xcx, xch2[0], bias_s[0], lane, q, hown, hn, gr_, gz_, gn_, gh_);
#pragma unroll
      for (int r = 0; r < 4; ++r) xd[r] = drop ? (((kp >> (8 * r)) & 0xffu) ? hn[r] * scale_l0 : 0.f) : hn[r];
      cx_publish4(r_h0, (par * (unsigned)nblk + (unsigned)rg) * rowrec, ft, i, q, hn, tag, l2x);
      *reinterpret_cast<float4*>(sv.h0 + (int64_t)t * BH + (int64_t)b * H + f0) = make_float4(hn[0], hn[1], hn[2], hn[3]);
      if (drop && sv.x1) *reinterpret_cast<float4*>(sv.x1 + (int64_t)(t - 1) * BH + (int64_t)b * H + f0) = make_float4(xd[0], xd[1], xd[2], xd[3]);
      if (sv.gates0) {
        float* go = sv.gates0 + (int64_t)(t - 1) * 4 * BH + (int64_t)b * 4 * H + f0;
        *reinterpret_cast<float4*>(go) = make_float4(gr_[0], gr_[1], gr_[2], gr_[3]);
        *reinterpret_cast<float4*>(go + H) = make_float4(gz_[0], gz_[1], gz_[2], gz_[3]);
        *reinterpret_cast<float4*>(go + 2 * H) = make_float4(gn_[0], gn_[1], gn_[2], gn_[3]);
        *reinterpret_cast<float4*>(go + 3 * H) = make_float4(gh_[0], gh_[1], gh_[2], gh_[3]);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) hown[r] = hn[r];
    }
    DCL_STAMP(6);
    // ---- the h0_t row: swept by the waves 1, 2, 3 (wave 0 has just published its tile and only stores), each of which also leaves
    // Dropout(h0_t) of its tiles for cell 1's input side (xs_h0 keeps the undropped row: cell 0's hidden side at step t + 1) --------------
    if (wave != 0) {
      uint32_t km[(KS + 2) / 3];
#pragma unroll
      for (int j = 0; j < (KS + 2) / 3; ++j) {
        const int k = 16 * (wave - 1 + 3 * j) + 4 * q;
        const bool ok = drop && wave - 1 + 3 * j < KS && k < H && rvalid;
        km[j] = ok ? *reinterpret_cast<const uint32_t*>(a.keep_l0 + (int64_t)(t - 1) * BH + (int64_t)b * H + k) : 0u;
      }
#ifdef G2V_STAMPS
      cx_sweep_tiles<(KS + 2) / 3>(r_h0, (par * (unsigned)nblk + (unsigned)rg) * rowrec, wave - 1, 3, nt, nrows, H, tag, &xs_h0[0][0], lane, a.fault,
                                   (t == 5 && blockIdx.x == 0 && blockIdx.y == 0 && wave == 2) ? g2v_stamps + 128 : nullptr);
#else
      cx_sweep_tiles<(KS + 2) / 3>(r_h0, (par * (unsigned)nblk + (unsigned)rg) * rowrec, wave - 1, 3, nt, nrows, H, tag, &xs_h0[0][0], lane, a.fault);
#endif
      if (drop) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // (one wave: its LDS operations execute in order)
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
    DCL_STAMP(7);
    lds_barrier();
    // ---- cell 1, input side: gate gx of Dropout(h0_t) W_ih1 per gate wave ---------------------------------------------------------------
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
    DCL_STAMP(8);
    lds_barrier();
    DCL_STAMP(9);
    if (wave == 2 && rvalid && fok) {
      float hn[4], gr_[4], gz_[4], gn_[4], gh_[4];
      dcl_cell_epilogue(xcx, xch2[1], bias_s[1], lane, q, hown, hn, gr_, gz_, gn_, gh_);
      cx_publish4(r_h1, (par * (unsigned)nblk + (unsigned)rg) * rowrec, ft, i, q, hn, tag, l2x);
      *reinterpret_cast<float4*>(sv.h1 + (int64_t)t * BH + (int64_t)b * H + f0) = make_float4(hn[0], hn[1], hn[2], hn[3]);
      if (sv.gates1) {
        float* go = sv.gates1 + (int64_t)(t - 1) * 4 * BH + (int64_t)b * 4 * H + f0;
        *reinterpret_cast<float4*>(go) = make_float4(gr_[0], gr_[1], gr_[2], gr_[3]);
        *reinterpret_cast<float4*>(go + H) = make_float4(gz_[0], gz_[1], gz_[2], gz_[3]);
        *reinterpret_cast<float4*>(go + 2 * H) = make_float4(gn_[0], gn_[1], gn_[2], gn_[3]);
        *reinterpret_cast<float4*>(go + 3 * H) = make_float4(gh_[0], gh_[1], gh_[2], gh_[3]);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) hown[r] = hn[r];
    }
    DCL_STAMP(10);
   }
    // ---- y_t = out_layer(h1_t): wave 3 sweeps the row, every wave multiplies its D tile ------------------------------------------------
    const bool has_next = t < T - 1, teacher = has_next && t < a.n_pre;
    float tgv[4];      // teacher-forced inputs / Dropout(0.95) flags of the lane's four y elements: requested in front of the exchange
    uint8_t k95v[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int d = 16 * wave + 4 * q + r;
      const bool ok = wave < ndt && d < D && rvalid;
      tgv[r] = (ok && (teacher || t == 0)) ? a.target[((int64_t)b * T + t) * D + d] : 0.f;
      k95v[r] = (ok && has_next && a.conditioned) ? a.keep95[(int64_t)t * BD + (int64_t)b * D + d] : 0;
    }
    if (wave != 2 && t > 0)      // (wave 2 has just published its tile: the other three sweep)
      cx_sweep_tiles<(KS + 2) / 3>(r_h1, (par * (unsigned)nblk + (unsigned)rg) * rowrec, wave == 3 ? 2 : wave, 3, nt, nrows, H, tag, &xs_h1[0][0], lane,
                                   a.fault);
    DCL_STAMP(11);
    lds_barrier();
    DCL_STAMP(12);
    {
      // every wave: the k-steps wave, wave + 4, ... of all D tiles (independent chains); the four partial products of a D tile are
      // added by wave dt in the order 0, 1, 2, 3 (the split kernels run one chain over all k-steps: equal to summation order)
      f32x4 yq[DSPLIT_DT];
#pragma unroll
      for (int dt = 0; dt < DSPLIT_DT; ++dt) yq[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int j = 0; j < (KS + 3) / 4; ++j) {
        const int ks = wave + 4 * j;
        if (ks < KS) {
          const float4 x4 = xs_h1[ks][lane];
#pragma unroll
          for (int dt = 0; dt < DSPLIT_DT; ++dt) {
            if (dt < ndt) {
              const float4 w4 = wo_s[(dt * KS + ks) * 64 + lane];
              yq[dt] = mfma16(w4.x, x4.x, yq[dt]);
              yq[dt] = mfma16(w4.y, x4.y, yq[dt]);
              yq[dt] = mfma16(w4.z, x4.z, yq[dt]);
              yq[dt] = mfma16(w4.w, x4.w, yq[dt]);
            }
          }
        }
      }
#pragma unroll
      for (int dt = 0; dt < DSPLIT_DT; ++dt)
        if (dt < ndt) yp_s[(wave * DSPLIT_DT + dt) * 64 + lane] = make_float4(yq[dt][0], yq[dt][1], yq[dt][2], yq[dt][3]);
      lds_barrier();
      float ya[4] = {0.f, 0.f, 0.f, 0.f};
      if (wave < ndt) {
#pragma unroll
        for (int pw = 0; pw < 4; ++pw) {
          const float4 v = yp_s[(pw * DSPLIT_DT + wave) * 64 + lane];
          ya[0] += v.x; ya[1] += v.y; ya[2] += v.z; ya[3] += v.w;
        }
      }
      // y, the next decoder input xin = Dropout(0.95)(teacher ? target : y) (zeros when !conditioned), kept as MFMA B fragments
      float xv[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int d = 16 * wave + 4 * q + r;
        const bool ok = wave < ndt && d < D && rvalid;
        const float y = t == 0 ? tgv[r] : ya[r] + bo[r];      // (y_0 = the target's first frame)
        const float tg = tgv[r];
        const uint8_t k95 = k95v[r];
        const float src = teacher ? tg : y;                                           // :1049-1052
        xv[r] = k95 ? src * 20.0f : 0.f;                                              // Dropout(0.95): 1/(1-0.95)
        if (ft == 0 && ok) {
          sv.y[(int64_t)t * BD + (int64_t)b * D + d] = y;
          if (has_next && sv.xin) sv.xin[(int64_t)t * BD + (int64_t)b * D + d] = xv[r];
        }
      }
      xfs[wave][lane] = make_float4(xv[0], xv[1], xv[2], xv[3]);
      DCL_STAMP(13);
    }
    lds_barrier();
    DCL_STAMP(14);
    if (has_next && wave == 0) {
      // ---- u_{t+1} tile = pre_linear.0(xin_{t+1}) and the BN partial sums of (u - b) over this workgroup's rows ----------------------
      f32x4 ua = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int dt = 0; dt < DSPLIT_DT; ++dt) {
        if (dt < ndt) {
          const float4 x4 = xfs[dt][lane], w4 = wp_s[dt][lane];
          ua = mfma16(w4.x, x4.x, ua);
          ua = mfma16(w4.y, x4.y, ua);
          ua = mfma16(w4.z, x4.z, ua);
          ua = mfma16(w4.w, x4.w, ua);
        }
      }
      float s1[4], s2[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float v = (rvalid && fok) ? ua[r] : 0.f;
        s1[r] = reduce16(v);
        s2[r] = reduce16(v * v);
      }
      if (fok) {
        if (rvalid) {
          const float4 bp = wp_s[DSPLIT_DT][lane];
          const float un[4] = {ua[0] + bp.x, ua[1] + bp.y, ua[2] + bp.z, ua[3] + bp.w};
          cx_publish4(r_u, (par * (unsigned)nblk + (unsigned)rg) * rowrec, ft, i, q, un, tag + 1u, l2x);
          *reinterpret_cast<float4*>(sv.u + (int64_t)t * BH + (int64_t)b * H + f0) = make_float4(un[0], un[1], un[2], un[3]);
        }
        if (i == 0) {
          const unsigned g0 = (par * (unsigned)nblk + (unsigned)rg) * prec + (unsigned)f0;
          dcl_publish4(r_p, g0, s1, tag + 1u);
          dcl_publish4(r_p, g0 + (unsigned)Hp, s2, tag + 1u);
        }
      }
    }
    DCL_STAMP(15);
  }
}

// ---- the backward of the cluster (round 5): the four split launches per step (below) as ONE persistent launch -------------------------
// Same tile workgroups, four waves.  The two products of a stage that contract over all 3H gate columns -- carry' = direct + dgh W_hh
// and dgi W_ih -- are formed the other way round: a workgroup multiplies ITS 48 gate columns (its own gate-gradient tile, straight
// from LDS as the B fragments of three k-steps) with its 48 rows of W_hh / W_ih for ALL hidden units, publishes the two (16 x H)
// partial products as row records, and the owner of a hidden-unit tile adds the NT partial products of its tile in a fixed order
// (gru.hip's backward cluster has the argument: the exchange is a third of handing every workgroup the whole 3H-wide rows).
// Exchanges per step: the dbn rows + the BatchNorm-backward partial sums of every row group (-> du, dy feedback), the partial
// products of cell 1, the partial products of cell 0.  Wave 0 owns the element-wise stages and keeps both carries in registers.
// The pair records are single-buffered (a workgroup publishes the next stage's records only after it has swept this stage's, and
// sweeping needs every workgroup of the row group to have published, i.e. to be done with the stage before); dbn rows / partial sums
// are double-buffered by step parity as in the forward.
struct DecClBwdArgs {
  const uint8_t* keep95; const uint8_t* keep_l0;
  g2v_dec_weights w; g2v_dec_saved sv; g2v_dec_grads g;
  unsigned long long* xd;      // [2][nblk] row records: dbn rows
  unsigned long long* xp;      // [2][nblk][2][Hp]: BatchNorm-backward partial sums
  unsigned long long* xq;      // [nblk][hh1, ih1, hh0, ih0][producer tile] row records: partial products
  unsigned* xcc;               // [nblk][nt]
  unsigned* fault;
  int T, B, D, H, n_pre, conditioned, nt, nblk;
  float p_drop;
};

// gate gradients of one GRU cell for the lane's (row, 4 units): the arithmetic of gru_cell_bwd_lane on values in registers
__device__ __forceinline__ void dcl_cell_bwd(const float (&dh)[4], const float4 (&gt)[4], const float4& hp4, float (&g_r)[4],
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

template <int KS>
__global__ __launch_bounds__(256) void dec_cluster_bwd_kernel(DecClBwdArgs a) {
  constexpr int NU = (2 * KS + 3) / 4;                                           // (matrix, output tile) units of a pair per wave
  constexpr int NPW = (KS + 1) / 2;                                              // producers a wave adds up
  __shared__ float st[4 * 16 * DSPLIT_KS];                                        // s1[H], s2[H] of the step, mean[H], invstd[H]
  __shared__ float bnw_s[16 * DSPLIT_KS];
  __shared__ __attribute__((aligned(16))) float4 xs_dbn[KS][64];                 // dbn rows as B fragments
  __shared__ __attribute__((aligned(16))) float4 xs_du[KS][64];                  // du rows
  __shared__ __attribute__((aligned(16))) float4 xs_dy[DSPLIT_DT][64];           // dy rows (D tiles as k-steps)
  __shared__ __attribute__((aligned(16))) float4 xs_g[6][64];                    // the stage's gate gradients: dgh r z hn, dgi r z n
  __shared__ __attribute__((aligned(16))) float4 dsum[4][64];                    // [wave] its share of the partial products, summed
  __shared__ __attribute__((aligned(16))) float4 wot_s[DSPLIT_DT][64];           // W_out^T fragments of the tile
  extern __shared__ __attribute__((aligned(16))) float4 wpt_s[];                 // [D tile][KS][64]: W_pre^T fragments
  const int T = a.T, B = a.B, D = a.D, H = a.H;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, i = lane & 15, q = lane >> 4;
  const int nt = a.nt, nblk = a.nblk;      // (1-D grid, the mapping of dec_cluster_fwd_kernel)
  const bool rg_fast = (nblk & 7) == 0;
  const int ft = rg_fast ? (int)blockIdx.x / nblk : (int)blockIdx.x % nt, rg = rg_fast ? (int)blockIdx.x % nblk : (int)blockIdx.x / nt;
  const int b0 = rg * 16;
  const int nrows = min(16, B - b0);
  const int Hp = nt << 4, ndt = (D + 15) >> 4, G = 3 * H;
  const bool rvalid = i < nrows, wrow_ok = 16 * ft + i < H;
  const int b = b0 + (rvalid ? i : 0);
  const int f0 = 16 * ft + 4 * q;
  const bool fok = f0 < H, own = rvalid && fok;
  const int64_t BH = (int64_t)B * H, BD = (int64_t)B * D, BG = 3 * BH;
  const g2v_dec_weights& w = a.w;
  const g2v_dec_saved& sv = a.sv;
  const g2v_dec_grads& gr = a.g;
  // ---- resident: this wave's (matrix, output tile) units of both pairs: rows g H + f0 + e of the matrix at columns 16 ot + i ----------
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
  // W_out^T fragments of the tile (wave 0's dy W_out product): W_out[16 dt + 4 q + e][16 ft + i] -- in LDS (registers are short)
  if (wave == 0) {
#pragma unroll
    for (int dt = 0; dt < DSPLIT_DT; ++dt) {
      float v[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int d = 16 * dt + 4 * q + e;
        v[e] = (wrow_ok && d < D) ? w.w_out[(int64_t)d * H + 16 * ft + i] : 0.f;
      }
      wot_s[dt][lane] = make_float4(v[0], v[1], v[2], v[3]);
    }
  }
  // W_pre^T fragments (rows d, contraction over the H features): wave dt fills its D tile
  for (int ks = 0; ks < KS; ++ks) {
    const int d = 16 * wave + i;
    float v[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int k = 16 * ks + 4 * q + e;
      v[e] = (wave < ndt && d < D && k < H) ? w.w_pre[(int64_t)k * D + d] : 0.f;
    }
    wpt_s[(wave * KS + ks) * 64 + lane] = make_float4(v[0], v[1], v[2], v[3]);
  }
  for (int f = tid; f < 16 * DSPLIT_KS; f += 256) bnw_s[f] = f < H ? w.bn_w[f] : 0.f;
  for (int ks = wave; ks < KS; ks += 4) {
    xs_dbn[ks][lane] = make_float4(0.f, 0.f, 0.f, 0.f);      // (the sweeps leave zeros where nothing exists; before the first one: zeros)
    xs_du[ks][lane] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  const bool drop = a.keep_l0 && a.p_drop > 0.f;
  const float scale_l0 = 1.0f / (1.0f - a.p_drop);
  const unsigned rowrec = 256u * (unsigned)nt, prec = 2u * (unsigned)Hp;
  __amdgpu_buffer_rsrc_t r_d = __builtin_amdgcn_make_buffer_rsrc(a.xd, 0, (int)(2u * (unsigned)nblk * rowrec * 8u), 0x00020000);
  __amdgpu_buffer_rsrc_t r_p = __builtin_amdgcn_make_buffer_rsrc(a.xp, 0, (int)(2u * (unsigned)nblk * prec * 8u), 0x00020000);
  __amdgpu_buffer_rsrc_t r_q = __builtin_amdgcn_make_buffer_rsrc(a.xq, 0, (int)((unsigned)nblk * 4u * (unsigned)nt * rowrec * 8u), 0x00020000);
  const unsigned q_rg = (unsigned)rg * 4u * (unsigned)nt * rowrec;      // this row group's pair records
  float carry1[4] = {0.f, 0.f, 0.f, 0.f}, carry0[4] = {0.f, 0.f, 0.f, 0.f};      // (wave 0)
  float acc_bw = 0.f, acc_bb = 0.f;                                                // d bn weight / bias of feature tid (workgroup (0, 0))
  __shared__ int xcd_flag;
  const bool l2x = cx_cluster_on_one_xcd(a.xcc + (size_t)rg * nt, nt, ft, &xcd_flag, tid, a.fault);
  for (int t = T - 1; t >= 0; --t) {
    const bool last = t == T - 1;
    const unsigned tag = (unsigned)(T - t), par = (unsigned)(t & 1), par_next = (unsigned)((t + 1) & 1);
    const bool feedback = a.conditioned && t >= a.n_pre;
    DCB_STAMP(0);
    // ---- requests that do not depend on this step's exchanges ------------------------------------------------------------------
    // every wave: its k-steps of u_t (ks = wave, wave + 4, ...) for the du stage; waves dt < ndt: dy_t / keep flags of their D tile
    float4 uq[(KS + 3) / 4];
    float dyo[4] = {0.f, 0.f, 0.f, 0.f};
    uint8_t k95[4] = {0, 0, 0, 0};
    if (!last) {
#pragma unroll
      for (int j = 0; j < (KS + 3) / 4; ++j) {
        const int k = 16 * (wave + 4 * j) + 4 * q;
        const bool ok = wave + 4 * j < KS && k < H && rvalid;
        uq[j] = ld4_or_zero(sv.u + (int64_t)t * BH + (int64_t)b * H + (ok ? k : 0), ok);
      }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int d = 16 * wave + 4 * q + r;
      const bool ok = wave < ndt && d < D && rvalid;
      dyo[r] = ok ? gr.dy[(int64_t)t * BD + (int64_t)b * D + d] : 0.f;
      k95[r] = (ok && !last && feedback && t > 0) ? a.keep95[(int64_t)t * BD + (int64_t)b * D + d] : 0;
    }
    // wave 0: the saved values of the two cells' tiles (step index t - 1) and of the BatchNorm stage
    float4 gt1[4], gt0[4], hp1 = make_float4(0.f, 0.f, 0.f, 0.f), hp0 = hp1, a4 = hp1, u4 = hp1, m4 = hp1, v4 = hp1;
    uint32_t kp = 0x01010101u;
    if (wave == 0 && t > 0) {
      const int64_t o4 = (int64_t)(t - 1) * 4 * BH + (int64_t)b * 4 * H + (fok ? f0 : 0), o1 = (int64_t)(t - 1) * BH + (int64_t)b * H + (fok ? f0 : 0);
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        gt1[g] = ld4_or_zero(sv.gates1 + o4 + g * H, own);
        gt0[g] = ld4_or_zero(sv.gates0 + o4 + g * H, own);
      }
      hp1 = ld4_or_zero(sv.h1 + o1, own);
      hp0 = ld4_or_zero(sv.h0 + o1, own);
      a4 = ld4_or_zero(sv.a + o1, own);
      u4 = ld4_or_zero(sv.u + o1, own);
      m4 = ld4_or_zero(sv.bn_stats + (int64_t)(t - 1) * 2 * H + (fok ? f0 : 0), fok);
      v4 = ld4_or_zero(sv.bn_stats + (int64_t)(t - 1) * 2 * H + H + (fok ? f0 : 0), fok);
      if (drop && own) kp = *reinterpret_cast<const uint32_t*>(a.keep_l0 + o1);
    }
    if (!last) {
      // ---- du_t: finish the BatchNorm backward of step index t from the dbn rows and the partial sums the stage before published ----
      cx_sweep_tiles<(KS + 3) / 4>(r_d, (par_next * (unsigned)nblk + (unsigned)rg) * rowrec, wave, 4, nt, nrows, H, tag - 1u, &xs_dbn[0][0], lane,
                                   a.fault);
      if (tid < H) {
        float s1, s2;
        dcl_sum_partials(a.xp + (size_t)par_next * nblk * prec, nblk, Hp, tid, tag - 1u, a.fault, s1, s2);
        st[tid] = s1;
        st[16 * DSPLIT_KS + tid] = s2;
        st[2 * 16 * DSPLIT_KS + tid] = sv.bn_stats[(int64_t)t * 2 * H + tid];
        st[3 * 16 * DSPLIT_KS + tid] = bn_invstd_(sv.bn_stats[(int64_t)t * 2 * H + H + tid]);
        acc_bw += s2;      // (d gamma / d beta accumulate over the steps, in step order: workgroup (0, 0) writes them at the end)
        acc_bb += s1;
      }
      DCB_STAMP(1);
      lds_barrier();
      DCB_STAMP(2);
      const float invB = 1.0f / (float)B;
#pragma unroll
      for (int j = 0; j < (KS + 3) / 4; ++j) {
        const int ks = wave + 4 * j;
        if (ks < KS) {      // (uniform)
          const int k = 16 * ks + 4 * q;
          const bool kok = k < H;
          const int kk = kok ? k : 0;
          const float4 db4 = xs_dbn[ks][lane];
          const float4 s14 = *reinterpret_cast<const float4*>(st + kk), s24 = *reinterpret_cast<const float4*>(st + 16 * DSPLIT_KS + kk);
          const float4 mm4 = *reinterpret_cast<const float4*>(st + 2 * 16 * DSPLIT_KS + kk), ii4 = *reinterpret_cast<const float4*>(st + 3 * 16 * DSPLIT_KS + kk);
          const float4 w4 = *reinterpret_cast<const float4*>(bnw_s + kk);
          const float uu[4] = {uq[j].x, uq[j].y, uq[j].z, uq[j].w}, db[4] = {db4.x, db4.y, db4.z, db4.w};
          const float mm[4] = {mm4.x, mm4.y, mm4.z, mm4.w}, is[4] = {ii4.x, ii4.y, ii4.z, ii4.w}, gg[4] = {w4.x, w4.y, w4.z, w4.w};
          const float a1[4] = {s14.x, s14.y, s14.z, s14.w}, a2[4] = {s24.x, s24.y, s24.z, s24.w};
          float du[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) {      // (the arithmetic of dec_bwd_dy_split_kernel)
            const float xhat = (uu[r] - mm[r]) * is[r];
            du[r] = (rvalid && kok) ? gg[r] * is[r] * (db[r] - a1[r] * invB - xhat * a2[r] * invB) : 0.f;
          }
          const float4 d4 = make_float4(du[0], du[1], du[2], du[3]);
          xs_du[ks][lane] = d4;
          if (ft == 0 && rvalid && kok) *reinterpret_cast<float4*>(gr.du + (int64_t)t * BH + (int64_t)b * H + k) = d4;
        }
      }
      lds_barrier();
      DCB_STAMP(3);
    }
    if (t == 0) break;      // (the last dy stage forms du_0 only)
    // ---- dy_t: + keep95 * 20 * (du_t W_pre^T) where the step's input was its own previous output; as the B fragments of S2 ----------
    {
      float dyv[4] = {dyo[0], dyo[1], dyo[2], dyo[3]};
      if (!last && feedback && wave < ndt) {
        f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          const float4 x4 = xs_du[ks][lane], w4 = wpt_s[(wave * KS + ks) * 64 + lane];
          acc = mfma16(w4.x, x4.x, acc);
          acc = mfma16(w4.y, x4.y, acc);
          acc = mfma16(w4.z, x4.z, acc);
          acc = mfma16(w4.w, x4.w, acc);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          if (k95[r]) dyv[r] += acc[r] * 20.0f;
          const int d = 16 * wave + 4 * q + r;
          if (ft == 0 && rvalid && d < D && k95[r]) gr.dy[(int64_t)t * BD + (int64_t)b * D + d] = dyv[r];
        }
      }
      if (wave < DSPLIT_DT) xs_dy[wave][lane] = make_float4(dyv[0], dyv[1], dyv[2], dyv[3]);
    }
    DCB_STAMP(4);
    lds_barrier();
    DCB_STAMP(5);
    // ---- cell 1 (wave 0): dh1 = carry1 + dy_t W_out restricted to the tile; its gate gradients --------------------------------------
    float direct[4] = {0.f, 0.f, 0.f, 0.f};
    if (wave == 0) {
      f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int dt = 0; dt < DSPLIT_DT; ++dt) {
        if (dt < ndt) {
          const float4 x4 = xs_dy[dt][lane], w4 = wot_s[dt][lane];
          acc = mfma16(w4.x, x4.x, acc);
          acc = mfma16(w4.y, x4.y, acc);
          acc = mfma16(w4.z, x4.z, acc);
          acc = mfma16(w4.w, x4.w, acc);
        }
      }
      float dh[4], g_r[4], g_z[4], g_n[4], g_hn[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) dh[r] = acc[r] + (last ? 0.f : carry1[r]);
      dcl_cell_bwd(dh, gt1, hp1, g_r, g_z, g_n, g_hn, direct);
      const float4 vr = own ? make_float4(g_r[0], g_r[1], g_r[2], g_r[3]) : make_float4(0.f, 0.f, 0.f, 0.f);
      const float4 vz = own ? make_float4(g_z[0], g_z[1], g_z[2], g_z[3]) : make_float4(0.f, 0.f, 0.f, 0.f);
      const float4 vn = own ? make_float4(g_n[0], g_n[1], g_n[2], g_n[3]) : make_float4(0.f, 0.f, 0.f, 0.f);
      const float4 vh = own ? make_float4(g_hn[0], g_hn[1], g_hn[2], g_hn[3]) : make_float4(0.f, 0.f, 0.f, 0.f);
      xs_g[0][lane] = vr; xs_g[1][lane] = vz; xs_g[2][lane] = vh;
      xs_g[3][lane] = vr; xs_g[4][lane] = vz; xs_g[5][lane] = vn;
      if (own) {
        float* gi = gr.dgi1 + (int64_t)(t - 1) * BG + (int64_t)b * G + f0;
        float* gh = gr.dgh1 + (int64_t)(t - 1) * BG + (int64_t)b * G + f0;
        *reinterpret_cast<float4*>(gi) = vr; *reinterpret_cast<float4*>(gi + H) = vz; *reinterpret_cast<float4*>(gi + 2 * H) = vn;
        *reinterpret_cast<float4*>(gh) = vr; *reinterpret_cast<float4*>(gh + H) = vz; *reinterpret_cast<float4*>(gh + 2 * H) = vh;
      }
    }
    DCB_STAMP(6);
    // ---- the two stages with a partial-product exchange: c = 1 (cell 1's W_hh1 / W_ih1), then c = 0 -----------------------------------
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
      DCB_STAMP(c == 1 ? 7 : 11);
      // wave w adds the partial products of matrix w & 1 from the producers w >> 1, (w >> 1) + 2, ... for this tile
      cx_sweep_tile_sum<NPW>(r_q, q_rg + (unsigned)((1 - c) * 2 + (wave & 1)) * (unsigned)nt * rowrec, rowrec, wave >> 1, 2, nt, ft, nrows, H, tag,
                             &dsum[wave][0], lane, a.fault);
      DCB_STAMP(c == 1 ? 8 : 12);
      lds_barrier();
      DCB_STAMP(c == 1 ? 9 : 13);
      if (wave == 0) {
        const float4 h_a = dsum[0][lane], h_b = dsum[2][lane], i_a = dsum[1][lane], i_b = dsum[3][lane];
        const float shh[4] = {h_a.x + h_b.x, h_a.y + h_b.y, h_a.z + h_b.z, h_a.w + h_b.w};
        const float sih[4] = {i_a.x + i_b.x, i_a.y + i_b.y, i_a.z + i_b.z, i_a.w + i_b.w};
        if (c == 1) {
          // carry1' = dh1 z + dgh1 W_hh1 | dh0 = Dropout'(dgi1 W_ih1) + carry0; cell 0's gate gradients
          float dh[4], g_r[4], g_z[4], g_n[4], g_hn[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            carry1[r] = direct[r] + shh[r];
            float v = sih[r];
            if (drop) v = ((kp >> (8 * r)) & 0xffu) ? v * scale_l0 : 0.f;
            dh[r] = v + (last ? 0.f : carry0[r]);
          }
          dcl_cell_bwd(dh, gt0, hp0, g_r, g_z, g_n, g_hn, direct);
          const float4 vr = own ? make_float4(g_r[0], g_r[1], g_r[2], g_r[3]) : make_float4(0.f, 0.f, 0.f, 0.f);
          const float4 vz = own ? make_float4(g_z[0], g_z[1], g_z[2], g_z[3]) : make_float4(0.f, 0.f, 0.f, 0.f);
          const float4 vn = own ? make_float4(g_n[0], g_n[1], g_n[2], g_n[3]) : make_float4(0.f, 0.f, 0.f, 0.f);
          const float4 vh = own ? make_float4(g_hn[0], g_hn[1], g_hn[2], g_hn[3]) : make_float4(0.f, 0.f, 0.f, 0.f);
          xs_g[0][lane] = vr; xs_g[1][lane] = vz; xs_g[2][lane] = vh;
          xs_g[3][lane] = vr; xs_g[4][lane] = vz; xs_g[5][lane] = vn;
          if (own) {
            float* gi = gr.dgi0 + (int64_t)(t - 1) * BG + (int64_t)b * G + f0;
            float* gh = gr.dgh0 + (int64_t)(t - 1) * BG + (int64_t)b * G + f0;
            *reinterpret_cast<float4*>(gi) = vr; *reinterpret_cast<float4*>(gi + H) = vz; *reinterpret_cast<float4*>(gi + 2 * H) = vn;
            *reinterpret_cast<float4*>(gh) = vr; *reinterpret_cast<float4*>(gh + H) = vz; *reinterpret_cast<float4*>(gh + 2 * H) = vh;
          }
          DCB_STAMP(10);
        } else {
          // carry0' = dh0 z + dgh0 W_hh0 | da = dgi0 W_ih0 -> ReLU backward -> dbn of step index t - 1 and its BatchNorm sums
          float dbn[4] = {0.f, 0.f, 0.f, 0.f}, dbx[4] = {0.f, 0.f, 0.f, 0.f}, s1[4], s2[4];
          const float av[4] = {a4.x, a4.y, a4.z, a4.w}, uv[4] = {u4.x, u4.y, u4.z, u4.w}, mv[4] = {m4.x, m4.y, m4.z, m4.w},
                      vv[4] = {v4.x, v4.y, v4.z, v4.w};
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            carry0[r] = direct[r] + shh[r];
            if (own) {
              dbn[r] = av[r] > 0.f ? sih[r] : 0.f;
              dbx[r] = dbn[r] * ((uv[r] - mv[r]) * bn_invstd_(vv[r]));
            }
          }
          if (own) {
            cx_publish4(r_d, (par * (unsigned)nblk + (unsigned)rg) * rowrec, ft, i, q, dbn, tag, l2x);
            *reinterpret_cast<float4*>(gr.dbn + (int64_t)(t - 1) * BH + (int64_t)b * H + f0) = make_float4(dbn[0], dbn[1], dbn[2], dbn[3]);
          }
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            s1[r] = reduce16(dbn[r]);
            s2[r] = reduce16(dbx[r]);
          }
          if (i == 0 && fok) {
            const unsigned g0 = (par * (unsigned)nblk + (unsigned)rg) * prec + (unsigned)f0;
            dcl_publish4(r_p, g0, s1, tag);
            dcl_publish4(r_p, g0 + (unsigned)Hp, s2, tag);
          }
        }
      }
    }
    DCB_STAMP(14);
    lds_barrier();      // (xs_g / dsum / st of this step are done with)
    DCB_STAMP(15);
  }
  // ---- the gradient of the initial states, d gamma / d beta ---------------------------------------------------------------------
  if (wave == 0 && own) {
    *reinterpret_cast<float4*>(gr.dh_init + (int64_t)b * H + f0) = make_float4(carry0[0], carry0[1], carry0[2], carry0[3]);
    *reinterpret_cast<float4*>(gr.dh_init + BH + (int64_t)b * H + f0) = make_float4(carry1[0], carry1[1], carry1[2], carry1[3]);
  }
  if (ft == 0 && rg == 0 && tid < H) {
    gr.d_bn_w[tid] = acc_bw;
    gr.d_bn_b[tid] = acc_bb;
  }
}

// ---- backward of the same split: step t as up to four launches (the seams are where a product contracts over a whole
// row of gate gradients that every tile workgroup of the previous launch contributed to) ------------------------------------
//   dec_bwd_dy_split_kernel    (rows x D tiles): finish BN-backward of step t+1 -> du_{t+1}; dy_t += keep95 * 20 * (du W_pre)
//   dec_bwd_cell1_split_kernel (rows x H tiles): dh1 = carry1 + dy_t W_out; GRU cell 1 gate gradients of the tile
//   dec_bwd_pair_split_kernel<false>           : carry1' = dh1 z + dgh1 W_hh1 | dh0 = drop(dgi1 W_ih1) + carry0; cell 0 gates
//   dec_bwd_pair_split_kernel<true>            : carry0' = dh0 z + dgh0 W_hh0 | da = dgi0 W_ih0 -> ReLU bwd -> dbn_t, BN sums
// The transposed weights are plain row-major copies (output unit = row, contraction along the row).
constexpr int DSPLIT_KSG = 48;      // k-steps over 3H (3H <= 768)

struct DecBwdDyArgs {
  const float* part;      // (nblk, 2H) BN-backward sums of step t+1
  const float* stats;     // (2H) batch mean / var of step t+1
  const float* u;         // (B,H) u_{t+1}
  const float* dbn;       // (B,H) d(BN output) of step t+1
  const float* bn_w;
  float* du;              // (B,H) out: d u_{t+1}
  float* d_bn_w; float* d_bn_b;   // (H) accumulated over the steps
  const float* w_pre_t;   // (D,H) = W_pre^T
  float* dy;              // (B,D) dy_t, updated in place with the feedback term
  const uint8_t* keep95;  // (B,D)
  int nblk, first_acc, feedback, only_a;
};
__global__ __launch_bounds__(64) void dec_bwd_dy_split_kernel(DecBwdDyArgs a, int B, int D, int H) {
  __shared__ float st[2 * 16 * DSPLIT_KS];
  const int lane = threadIdx.x, i = lane & 15, q = lane >> 4;
  const int b0 = blockIdx.x * 16, dt = blockIdx.y;
  const int nrows = min(16, B - b0);
  const int nks = (H + 15) >> 4;
  const bool rvalid = i < nrows;
  const int b = b0 + (rvalid ? i : 0);
  // weight fragments of this D tile and the row's inputs first (independent of the sums)
  float4 wa[DSPLIT_KS], u4[DSPLIT_KS], g4[DSPLIT_KS];
  const int drow = 16 * dt + i;
  const bool dok = drow < D && a.feedback && !a.only_a;
#pragma unroll
  for (int ks = 0; ks < DSPLIT_KS; ++ks) {
    const int k = 16 * ks + 4 * q;
    const bool kok = ks < nks && k < H;
    wa[ks] = ld4_or_zero(a.w_pre_t + (int64_t)(dok ? drow : 0) * H + (kok ? k : 0), kok && dok);
    u4[ks] = ld4_or_zero(a.u + (int64_t)b * H + (kok ? k : 0), kok && rvalid);
    g4[ks] = ld4_or_zero(a.dbn + (int64_t)b * H + (kok ? k : 0), kok && rvalid);
  }
  for (int f = lane; f < H; f += 64) {
    float s1, s2;
    sum_partials(a.part, a.nblk, H, f, s1, s2);
    st[f] = s1;
    st[16 * DSPLIT_KS + f] = s2;
    if (blockIdx.x == 0 && blockIdx.y == 0) {      // d gamma / d beta accumulate over the steps (one writer, stream ordered)
      a.d_bn_w[f] = (a.first_acc ? 0.f : a.d_bn_w[f]) + s2;
      a.d_bn_b[f] = (a.first_acc ? 0.f : a.d_bn_b[f]) + s1;
    }
  }
  __syncthreads();
  const float invB = 1.0f / (float)B;
  float4 xb[DSPLIT_KS];
#pragma unroll
  for (int ks = 0; ks < DSPLIT_KS; ++ks) {
    const int k = 16 * ks + 4 * q;
    xb[ks] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (ks < nks && k < H) {
      const float4 m4 = *reinterpret_cast<const float4*>(a.stats + k), v4 = *reinterpret_cast<const float4*>(a.stats + H + k);
      const float4 w4 = *reinterpret_cast<const float4*>(a.bn_w + k);
      const float4 s14 = *reinterpret_cast<const float4*>(st + k), s24 = *reinterpret_cast<const float4*>(st + 16 * DSPLIT_KS + k);
      const float uu[4] = {u4[ks].x, u4[ks].y, u4[ks].z, u4[ks].w}, db[4] = {g4[ks].x, g4[ks].y, g4[ks].z, g4[ks].w};
      const float mm[4] = {m4.x, m4.y, m4.z, m4.w}, vv[4] = {v4.x, v4.y, v4.z, v4.w}, gg[4] = {w4.x, w4.y, w4.z, w4.w};
      const float a1[4] = {s14.x, s14.y, s14.z, s14.w}, a2[4] = {s24.x, s24.y, s24.z, s24.w};
      float du[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float invstd = bn_invstd_(vv[r]);
        const float xhat = (uu[r] - mm[r]) * invstd;
        du[r] = rvalid ? gg[r] * invstd * (db[r] - a1[r] * invB - xhat * a2[r] * invB) : 0.f;
      }
      xb[ks] = make_float4(du[0], du[1], du[2], du[3]);
      if (dt == 0 && rvalid) *reinterpret_cast<float4*>(a.du + (int64_t)b * H + k) = xb[ks];
    }
  }
  if (a.only_a || !a.feedback) return;              // without feedback dy_t is the loss gradient as it stands
  f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int ks = 0; ks < DSPLIT_KS; ++ks) {
    if (ks < nks) {
      acc = mfma16(wa[ks].x, xb[ks].x, acc);
      acc = mfma16(wa[ks].y, xb[ks].y, acc);
      acc = mfma16(wa[ks].z, xb[ks].z, acc);
      acc = mfma16(wa[ks].w, xb[ks].w, acc);
    }
  }
  if (!rvalid) return;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int d = 16 * dt + 4 * q + r;
    if (d < D && a.keep95[(int64_t)b * D + d]) a.dy[(int64_t)b * D + d] += acc[r] * 20.0f;
  }
}

// gate gradients of one GRU cell for (row, 4 hidden units) held by a lane; returns dh * z (or dh for nothing: rows are valid)
__device__ __forceinline__ void gru_cell_bwd_lane(const float (&dh)[4], const float* __restrict__ gates, const float* __restrict__ hprev,
                                                  float* __restrict__ dgi, float* __restrict__ dgh, float* __restrict__ direct_out,
                                                  int H, int f0) {
  const float4 r4 = *reinterpret_cast<const float4*>(gates + f0), z4 = *reinterpret_cast<const float4*>(gates + H + f0),
               n4 = *reinterpret_cast<const float4*>(gates + 2 * H + f0), h4 = *reinterpret_cast<const float4*>(gates + 3 * H + f0);
  const float4 p4 = *reinterpret_cast<const float4*>(hprev + f0);
  const float rr[4] = {r4.x, r4.y, r4.z, r4.w}, zz[4] = {z4.x, z4.y, z4.z, z4.w}, nn[4] = {n4.x, n4.y, n4.z, n4.w},
              gh[4] = {h4.x, h4.y, h4.z, h4.w}, hp[4] = {p4.x, p4.y, p4.z, p4.w};
  float g_r[4], g_z[4], g_n[4], g_hn[4], direct[4];
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
  const float4 vr = make_float4(g_r[0], g_r[1], g_r[2], g_r[3]), vz = make_float4(g_z[0], g_z[1], g_z[2], g_z[3]),
               vn = make_float4(g_n[0], g_n[1], g_n[2], g_n[3]), vh = make_float4(g_hn[0], g_hn[1], g_hn[2], g_hn[3]);
  *reinterpret_cast<float4*>(dgi + f0) = vr; *reinterpret_cast<float4*>(dgi + H + f0) = vz; *reinterpret_cast<float4*>(dgi + 2 * H + f0) = vn;
  *reinterpret_cast<float4*>(dgh + f0) = vr; *reinterpret_cast<float4*>(dgh + H + f0) = vz; *reinterpret_cast<float4*>(dgh + 2 * H + f0) = vh;
  *reinterpret_cast<float4*>(direct_out + f0) = make_float4(direct[0], direct[1], direct[2], direct[3]);
}

struct DecBwdCell1Args {
  const float* dy;        // (B,D) dy_t
  const float* w_out_t;   // (H,D) = W_out^T
  const float* carry;     // (B,H) carry1 or NULL (last step)
  const float* gates; const float* hprev;   // (B,4H), (B,H) of step t
  float* dgi; float* dgh; // (B,3H)
  float* direct;          // (B,H) scratch: dh1 * z
};
__global__ __launch_bounds__(64) void dec_bwd_cell1_split_kernel(DecBwdCell1Args a, int B, int D, int H) {
  const int lane = threadIdx.x, i = lane & 15, q = lane >> 4;
  const int b0 = blockIdx.x * 16, ft = blockIdx.y;
  const int nrows = min(16, B - b0);
  const int ndt = (D + 15) >> 4;
  const bool rvalid = i < nrows, wrow_ok = 16 * ft + i < H;
  const int b = b0 + (rvalid ? i : 0);
  f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int ks = 0; ks < DSPLIT_DT; ++ks) {
    if (ks < ndt) {
      float wv[4], xv[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int d = 16 * ks + 4 * q + e;
        wv[e] = (d < D && wrow_ok) ? a.w_out_t[(int64_t)(16 * ft + i) * D + d] : 0.f;
        xv[e] = (d < D && rvalid) ? a.dy[(int64_t)b * D + d] : 0.f;
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) acc = mfma16(wv[e], xv[e], acc);
    }
  }
  const int f0 = 16 * ft + 4 * q;
  if (!rvalid || f0 >= H) return;
  float dh[4] = {acc[0], acc[1], acc[2], acc[3]};
  if (a.carry) {
    const float4 c4 = *reinterpret_cast<const float4*>(a.carry + (int64_t)b * H + f0);
    dh[0] += c4.x; dh[1] += c4.y; dh[2] += c4.z; dh[3] += c4.w;
  }
  gru_cell_bwd_lane(dh, a.gates + (int64_t)b * 4 * H, a.hprev + (int64_t)b * H, a.dgi + (int64_t)b * 3 * H, a.dgh + (int64_t)b * 3 * H,
                    a.direct + (int64_t)b * H, H, f0);
}

struct DecBwdPairArgs {
  // wave 0: carry' = direct + dgh W_hh
  const float* dgh; const float* w_hh_t;      // (B,3H), (H,3H)
  const float* direct;                        // (B,H)
  float* carry_out;                           // (B,H)
  // wave 1: v = dgi W_ih, then the stage-specific epilogue
  const float* dgi; const float* w_ih_t;
  // BNSTAGE == false: GRU cell 0 backward
  const uint8_t* keep; float keep_scale;      // inter-layer dropout of the forward (NULL: none)
  const float* carry0;                        // (B,H) or NULL (last step)
  const float* gates0; const float* hprev0;
  float* dgi0; float* dgh0; float* direct0;
  // BNSTAGE == true: ReLU backward + BN-backward sums
  const float* a_act; const float* u; const float* stats;   // (B,H), (B,H), (2H)
  float* dbn; float* part;                    // (B,H), (nblk, 2H)
};
template <bool BNSTAGE>
__global__ __launch_bounds__(128) void dec_bwd_pair_split_kernel(DecBwdPairArgs a, int B, int H) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i = lane & 15, q = lane >> 4;
  const int b0 = blockIdx.x * 16, ft = blockIdx.y;
  const int nrows = min(16, B - b0);
  const int G = 3 * H, nks = (G + 15) >> 4;
  const bool rvalid = i < nrows, wrow_ok = 16 * ft + i < H;
  const int b = b0 + (rvalid ? i : 0);
  const float* W = (wave == 0 ? a.w_hh_t : a.w_ih_t) + (int64_t)(16 * ft + (wrow_ok ? i : 0)) * G;
  const float* X = (wave == 0 ? a.dgh : a.dgi) + (int64_t)b * G;
  float4 wa[DSPLIT_KSG], xb[DSPLIT_KSG];
#pragma unroll
  for (int ks = 0; ks < DSPLIT_KSG; ++ks) {
    const int k = 16 * ks + 4 * q;
    const bool kok = ks < nks && k < G;
    wa[ks] = ld4_or_zero(W + (kok ? k : 0), kok && wrow_ok);
    xb[ks] = ld4_or_zero(X + (kok ? k : 0), kok && rvalid);
  }
  f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int ks = 0; ks < DSPLIT_KSG; ++ks) {
    if (ks < nks) {
      acc = mfma16(wa[ks].x, xb[ks].x, acc);
      acc = mfma16(wa[ks].y, xb[ks].y, acc);
      acc = mfma16(wa[ks].z, xb[ks].z, acc);
      acc = mfma16(wa[ks].w, xb[ks].w, acc);
    }
  }
  const int f0 = 16 * ft + 4 * q;
  const bool fok = f0 < H;
  if (wave == 0) {
    if (rvalid && fok) {
      const float4 d4 = *reinterpret_cast<const float4*>(a.direct + (int64_t)b * H + f0);
      *reinterpret_cast<float4*>(a.carry_out + (int64_t)b * H + f0) = make_float4(d4.x + acc[0], d4.y + acc[1], d4.z + acc[2], d4.w + acc[3]);
    }
    return;
  }
  if constexpr (!BNSTAGE) {
    if (!rvalid || !fok) return;
    float dh[4] = {acc[0], acc[1], acc[2], acc[3]};
    if (a.keep) {
      const uint32_t kp = *reinterpret_cast<const uint32_t*>(a.keep + (int64_t)b * H + f0);
#pragma unroll
      for (int r = 0; r < 4; ++r) dh[r] = ((kp >> (8 * r)) & 0xffu) ? dh[r] * a.keep_scale : 0.f;
    }
    if (a.carry0) {
      const float4 c4 = *reinterpret_cast<const float4*>(a.carry0 + (int64_t)b * H + f0);
      dh[0] += c4.x; dh[1] += c4.y; dh[2] += c4.z; dh[3] += c4.w;
    }
    gru_cell_bwd_lane(dh, a.gates0 + (int64_t)b * 4 * H, a.hprev0 + (int64_t)b * H, a.dgi0 + (int64_t)b * G, a.dgh0 + (int64_t)b * G,
                      a.direct0 + (int64_t)b * H, H, f0);
  } else {
    float dbn[4] = {0.f, 0.f, 0.f, 0.f}, s1[4], s2[4];
    float dbx[4] = {0.f, 0.f, 0.f, 0.f};
    if (rvalid && fok) {
      const float4 a4 = *reinterpret_cast<const float4*>(a.a_act + (int64_t)b * H + f0), u4 = *reinterpret_cast<const float4*>(a.u + (int64_t)b * H + f0);
      const float4 m4 = *reinterpret_cast<const float4*>(a.stats + f0), v4 = *reinterpret_cast<const float4*>(a.stats + H + f0);
      const float av[4] = {a4.x, a4.y, a4.z, a4.w}, uv[4] = {u4.x, u4.y, u4.z, u4.w}, mv[4] = {m4.x, m4.y, m4.z, m4.w},
                  vv[4] = {v4.x, v4.y, v4.z, v4.w};
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        dbn[r] = av[r] > 0.f ? acc[r] : 0.f;
        const float invstd = bn_invstd_(vv[r]);
        dbx[r] = dbn[r] * ((uv[r] - mv[r]) * invstd);
      }
      *reinterpret_cast<float4*>(a.dbn + (int64_t)b * H + f0) = make_float4(dbn[0], dbn[1], dbn[2], dbn[3]);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      s1[r] = reduce16(dbn[r]);
      s2[r] = reduce16(dbx[r]);
    }
    if (i == 0 && fok) {
      float* part = a.part + (int64_t)blockIdx.x * 2 * H;
      *reinterpret_cast<float4*>(part + f0) = make_float4(s1[0], s1[1], s1[2], s1[3]);
      *reinterpret_cast<float4*>(part + H + f0) = make_float4(s2[0], s2[1], s2[2], s2[3]);
    }
  }
}

// The weight fragments of the fused kernels (per-step and persistent alike), in workspace order.
static DecPackF dec_fwd_pack_layout(const g2v_dec_weights* w, int D, int H, void* workspace, PackBatch& pb) {
  float* p = (float*)workspace;
  DecPackF pk;
  pb.n = 6;
  pb.d[0] = PackDesc{w->w_pre, p, H, 1, 0, D, D, 0, 0}; pk.pre = p; p += pack_floats(H, 1, D);
  pb.d[1] = PackDesc{w->w_ih0, p, H, 3, H, H, H, 0, 0}; pk.ih0 = p; p += pack_floats(H, 3, H);
  pb.d[2] = PackDesc{w->w_hh0, p, H, 3, H, H, H, 0, 0}; pk.hh0 = p; p += pack_floats(H, 3, H);
  pb.d[3] = PackDesc{w->w_ih1, p, H, 3, H, H, H, 0, 0}; pk.ih1 = p; p += pack_floats(H, 3, H);
  pb.d[4] = PackDesc{w->w_hh1, p, H, 3, H, H, H, 0, 0}; pk.hh1 = p; p += pack_floats(H, 3, H);
  pb.d[5] = PackDesc{w->w_out, p, D, 1, 0, H, H, 0, dtiles_pad(D)}; pk.out = p;
  return pk;
}
static DecTW dec_bwd_pack_layout(const g2v_dec_weights* w, int D, int H, void* workspace, PackBatch& pb) {
  float* p = (float*)workspace;
  DecTW tw;
  pb.n = 6;
  const int G = 3 * H;
  pb.d[0] = PackDesc{w->w_pre, p, D, 1, 0, H, D, 1, dtiles_pad(D)}; tw.w_pre_t = p; p += (size_t)dtiles_pad(D) * pack_ks(H) * 256;   // rows d, k = f: W_pre[f][d]
  pb.d[1] = PackDesc{w->w_out, p, H, 1, 0, D, H, 1, 0}; tw.w_out_t = p; p += pack_floats(H, 1, D);   // rows f, k = d: W_out[d][f]
  pb.d[2] = PackDesc{w->w_ih0, p, H, 1, 0, G, H, 1, 0}; tw.w_ih0_t = p; p += pack_floats(H, 1, G);   // rows k, contraction g: W[g][k]
  pb.d[3] = PackDesc{w->w_hh0, p, H, 1, 0, G, H, 1, 0}; tw.w_hh0_t = p; p += pack_floats(H, 1, G);
  pb.d[4] = PackDesc{w->w_ih1, p, H, 1, 0, G, H, 1, 0}; tw.w_ih1_t = p; p += pack_floats(H, 1, G);
  pb.d[5] = PackDesc{w->w_hh1, p, H, 1, 0, G, H, 1, 0}; tw.w_hh1_t = p;
  return tw;
}
// prepared workspaces are honoured by the fused H = 64, D = 135 kernels only (the split kernels of other shapes transpose)
static bool dec_prepared_shape(int D, int H) { return H == 64 && D == 135; }

static int dec_rollout_fwd_impl(const float* target, const float* h_init, const g2v_dec_weights* w,
                                const g2v_dec_saved* s, const uint8_t* keep95, const uint8_t* keep_l0, float p_drop,
                                int n_pre_poses, int conditioned, int training, int T, int B, int D, int H,
                                void* workspace, size_t workspace_bytes, g2v_stream_t stream, bool prepared_req) {
  const bool prepared = prepared_req && dec_prepared_shape(D, H);
  G2V_REQUIRE(target && h_init && w && s && keep95 && workspace, "null pointer");
  G2V_REQUIRE(T >= 2 && B > 0 && D > 0 && H > 0, "bad size");
  G2V_REQUIRE(s->y && s->u && s->h0 && s->h1 && s->bn_partial, "missing state buffer");
  // (training: xin / a / x1 / gates0 / gates1 left NULL are simply not saved -- a training-mode forward without a backward, and
  //  how gpurun_tools/r05_saved_diet_probe.py prices the stores; g2v_dec_rollout_bwd requires all of them)
  G2V_REQUIRE(!training || s->bn_stats, "missing saved buffer");
  G2V_REQUIRE(training || (w->bn_running_mean && w->bn_running_var), "eval mode needs the running statistics");
  G2V_REQUIRE((w->bn_running_mean == nullptr) == (w->bn_running_var == nullptr), "running mean / var: both or neither");
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
  // ---- pack the weights into MFMA fragment order (one launch; the cluster kernel below reads them in place) ----
  PackBatch pb;
  const DecPackF pk = dec_fwd_pack_layout(w, D, H, workspace, pb);
  const bool maybe_cluster = g2v_dec_rollout_cluster_ok(B, D, H) && T >= 3;
  if (!prepared && !maybe_cluster) {
    launch_pack(pb, st);
    G2V_CHECK_LAUNCH();
  }
  int scratch = 1024;
  (void)dec_fwd_lds(D, H, &scratch);
  DecDims dm{T, B, D, H, p_drop, n_pre_poses, conditioned, training, cdiv(B, 16), (H == 64 && D == 135) ? dec_wt_stores() : 0, scratch};
  const bool fast = (H == 64) && (D == 135);   // the BASELINE shape: dims are compile-time constants
  {
    auto a16 = [](const void* q_) { return (reinterpret_cast<uintptr_t>(q_) & 15) == 0; };
    const int ptiles = persist_tiles_per_wg(dm.nblk);
    const bool persist = fast && ptiles >= 1 && (B % 4) == 0 &&      // (B % 16 != 0: a ragged last tile, the multi-tile kernels)
                        
                         a16(h_init) && a16(s->y) && a16(s->u) && a16(s->h0) && a16(s->h1) && a16(s->xin) && a16(s->a) &&
                         a16(s->x1) && a16(s->gates0) && a16(s->gates1) && a16(keep95) && a16(keep_l0) && a16(workspace);
    if (s->loss_code && !(persist && training && s->loss_coef && s->loss_partial && s->loss_terms &&
                          g2v_dec_rollout_fuses_loss(B, D, H, T))) {
      set_error("g2v_dec_rollout_fwd: the loss_* fields are set where g2v_dec_rollout_fuses_loss(B, D, H, T) does not hold "
                "(or not training, or one of them is NULL)");
      return G2V_ERR_UNSUPPORTED;
    }
    if (persist) {
      void* xbase = (char*)workspace + fwd_pack_bytes_aligned(D, H);
      const int rc = dec_persist_fwd_launch(target, h_init, w, s, keep95, keep_l0, p_drop, n_pre_poses, conditioned, training, T,
                                            B, pk.pre, pk.ih0, pk.hh0, pk.ih1, pk.hh1, pk.out, xbase, st, !prepared, ptiles);
      return rc;       // (the running statistics are updated by the kernel itself)
    }
  }
  if (lds > 48 * 1024) {
    (void)hipFuncSetAttribute((const void*)dec_step_fwd_kernel<0, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    (void)hipFuncSetAttribute((const void*)dec_step_fwd_kernel<64, 135>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  }
  auto al16 = [](const void* q_) { return (reinterpret_cast<uintptr_t>(q_) & 15) == 0; };
  // small batch, generic dims: steps t >= 1 as three launches over (rows x 16-unit tiles) workgroups (see the kernels)
  // crossover measured at the VQ-VAE.yml dims: split 6.9 ms vs fused 9.2 ms per train step at B = 1024, 12.5 vs 11.1 at 2048
  const bool split = !fast && dm.nblk <= DSPLIT_MAX_NBLK && (H & 3) == 0 && H <= 16 * DSPLIT_KS && D <= 16 * DSPLIT_DT &&
                     al16(s->u) && al16(s->h0) && al16(s->h1) && al16(s->a) && al16(s->x1) && al16(s->gates0) &&
                     al16(s->gates1) && al16(s->bn_partial) && al16(w->w_ih0) && al16(w->w_hh0) && al16(w->w_ih1) &&
                     al16(w->w_hh1) && al16(w->w_out) && al16(w->b_ih0) && al16(w->b_hh0) && al16(w->b_ih1) && al16(w->b_hh1) &&
                     al16(w->b_pre) && al16(w->bn_w) && al16(w->bn_b);
  // ... or ONE persistent launch for all steps t >= 1 while the tile grid has a CU per workgroup (dec_cluster_fwd_kernel)
  bool cluster = split && persist_enabled() && T >= 3 && dec_cluster_shape(D, H) &&
                 (int64_t)dm.nblk * ((H + 15) >> 4) <= device_cu_count() && (!training || s->bn_stats) &&
                 (!keep_l0 || al16(keep_l0)) &&
                 fwd_pack_bytes_aligned(D, H) + dec_cluster_fwd_xch_bytes(dm.nblk, H) <= workspace_bytes;
  if (cluster) {
    int nocc = 0;
    const void* fn = (const void*)dec_cluster_fwd_kernel<DCL_KS>;
    const size_t dyn = dec_cluster_fwd_dyn_lds();      // W_out fragments + partial out-layer products (the static arrays take ~60 KB more)
    static bool attr_set = false;
    if (!attr_set) attr_set = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn) == hipSuccess;
    cluster = attr_set && hipOccupancyMaxActiveBlocksPerMultiprocessor(&nocc, fn, 256, dyn) == hipSuccess && nocc >= 1;
  }
  if (!cluster) g2v_internal_preclear_drop(workspace, workspace_bytes);      // (a pre-cleared note for this workspace is void)
  if (maybe_cluster && !cluster) {      // (the cluster path was declined after all: the per-step kernels need the packed weights)
    launch_pack(pb, st);
    G2V_CHECK_LAUNCH();
  }
  for (int t = 0; t < T; ++t) {
    if (cluster && t == 0) {
      const size_t Hp = (size_t)((H + 15) & ~15), rowrec = (size_t)2 * dm.nblk * 16 * Hp;
      unsigned long long* x0 = reinterpret_cast<unsigned long long*>((char*)workspace + fwd_pack_bytes_aligned(D, H));
      DecClFwdArgs ca;
      ca.target = target; ca.h_init = h_init; ca.keep95 = keep95; ca.keep_l0 = keep_l0; ca.w = *w; ca.sv = *s;
      ca.xu = x0; ca.xh0 = x0 + rowrec; ca.xh1 = x0 + 2 * rowrec; ca.xp = x0 + 3 * rowrec;
      ca.xcc = reinterpret_cast<unsigned*>(ca.xp + (size_t)2 * dm.nblk * 2 * Hp);
      ca.fault = const_cast<unsigned*>(g2v_internal_persist_fault_ptr());
      ca.T = T; ca.B = B; ca.D = D; ca.H = H; ca.n_pre = n_pre_poses; ca.conditioned = conditioned; ca.training = training;
      ca.nt = (H + 15) >> 4; ca.nblk = dm.nblk;
      ca.p_drop = p_drop;
      if (!g2v_internal_preclear_take(x0, dec_cluster_fwd_xch_bytes(dm.nblk, H)) &&
          hipMemsetAsync(x0, 0, dec_cluster_fwd_xch_bytes(dm.nblk, H), st) != hipSuccess) {
        set_error("g2v_dec_rollout_fwd: clearing the exchange records failed");
        return G2V_ERR_LAUNCH;
      }
      hipLaunchKernelGGL(dec_cluster_fwd_kernel<DCL_KS>, dim3(((H + 15) >> 4) * dm.nblk), dim3(256), dec_cluster_fwd_dyn_lds(), st, ca);
      break;
    }
    if (fast) {
      hipLaunchKernelGGL((dec_step_fwd_kernel<64, 135>), dim3(dm.nblk), dim3(256), lds, st, target, h_init, *w, pk, *s, keep95,
                         keep_l0, dm, t);
    } else if (!split || t == 0) {
      hipLaunchKernelGGL((dec_step_fwd_kernel<0, 0>), dim3(dm.nblk), dim3(512), lds, st, target, h_init, *w, pk, *s, keep95,
                         keep_l0, dm, t);
    } else {
      const int64_t BH = (int64_t)B * H, BD = (int64_t)B * D;
      const bool drop = training && keep_l0 && p_drop > 0.f;
      const dim3 grid(dm.nblk, (H + 15) >> 4);
      DecCellArgs c0{};
      c0.x = s->u + (t - 1) * BH; c0.h_prev = s->h0 + (t - 1) * BH;
      c0.w_ih = w->w_ih0; c0.w_hh = w->w_hh0; c0.b_ih = w->b_ih0; c0.b_hh = w->b_hh0;
      c0.h_out = s->h0 + t * BH; c0.gates = s->gates0 ? s->gates0 + (t - 1) * 4 * BH : nullptr;
      c0.keep = drop ? keep_l0 + (t - 1) * BH : nullptr; c0.keep_scale = 1.0f / (1.0f - p_drop);
      c0.xdrop_out = (drop && s->x1) ? s->x1 + (t - 1) * BH : nullptr;
      c0.bn_partial = s->bn_partial + (int64_t)((t - 1) & 1) * dm.nblk * 2 * H;
      c0.b_pre = w->b_pre; c0.bn_w = w->bn_w; c0.bn_b = w->bn_b; c0.run_mean = w->bn_running_mean; c0.run_var = w->bn_running_var;
      c0.a_out = s->a ? s->a + (t - 1) * BH : nullptr;
      c0.bn_stats = (training && s->bn_stats) ? s->bn_stats + (int64_t)(t - 1) * 2 * H : nullptr;
      c0.nblk = dm.nblk; c0.training = training;
      hipLaunchKernelGGL(dec_cell_split_kernel<true>, grid, dim3(128), 0, st, c0, B, H);
      DecCellArgs c1{};
      c1.x = (drop && s->x1) ? s->x1 + (t - 1) * BH : s->h0 + t * BH; c1.h_prev = s->h1 + (t - 1) * BH;
      c1.w_ih = w->w_ih1; c1.w_hh = w->w_hh1; c1.b_ih = w->b_ih1; c1.b_hh = w->b_hh1;
      c1.h_out = s->h1 + t * BH; c1.gates = s->gates1 ? s->gates1 + (t - 1) * 4 * BH : nullptr;
      c1.keep = nullptr; c1.keep_scale = 1.0f; c1.xdrop_out = nullptr;
      hipLaunchKernelGGL(dec_cell_split_kernel<false>, grid, dim3(128), 0, st, c1, B, H);
      DecOutPreArgs o{};
      o.h1 = s->h1 + t * BH; o.w_out = w->w_out; o.b_out = w->b_out; o.w_pre = w->w_pre; o.b_pre = w->b_pre;
      o.target = target; o.keep95 = keep95 + t * BD; o.y = s->y + t * BD; o.xin = s->xin ? s->xin + t * BD : nullptr;
      o.u_next = s->u + t * BH; o.part = s->bn_partial + (int64_t)(t & 1) * dm.nblk * 2 * H;
      o.t = t; o.T = T; o.has_next = t < T - 1 ? 1 : 0; o.teacher = (t < T - 1 && t < n_pre_poses) ? 1 : 0; o.conditioned = conditioned;
      hipLaunchKernelGGL(dec_out_pre_split_kernel, grid, dim3(64), 0, st, o, B, D, H);
    }
  }
  G2V_CHECK_LAUNCH();
  if (training && w->bn_running_mean) {      // (NULL: the caller commits them later, g2v_bn_running_update)
    hipLaunchKernelGGL(bn_running_update_kernel, dim3(cdiv(H, 256)), dim3(256), 0, st, s->bn_stats,
                       w->bn_running_mean, w->bn_running_var, T - 1, H, B, (const unsigned*)nullptr);
    G2V_CHECK_LAUNCH();
  }
  return G2V_OK;
}

// BatchNorm1d's running statistics from the per-step batch statistics a TRAINING rollout saved (g2v_dec_saved.bn_stats), for a
// caller that passed g2v_dec_weights.bn_running_mean / _var = NULL to that rollout: the commit then happens HERE, where the
// caller knows the whole step is valid -- the kernel reads the persistent rollouts' fault latch and leaves the statistics
// untouched when it is set (round 5, advisor finding: the forward rollout used to commit them at its end, before a fault of the
// backward rollout or of the loss chaser could be known).
extern "C" int g2v_bn_running_update(const float* bn_stats, float* running_mean, float* running_var, int steps, int H, int B,
                                     g2v_stream_t stream) {
  G2V_REQUIRE(bn_stats && running_mean && running_var, "null pointer");
  G2V_REQUIRE(steps > 0 && H > 0 && B > 0, "bad size");
  hipLaunchKernelGGL(bn_running_update_kernel, dim3(cdiv(H, 256)), dim3(256), 0, (hipStream_t)stream, bn_stats, running_mean,
                     running_var, steps, H, B, g2v_internal_persist_fault_ptr());
  G2V_CHECK_LAUNCH();
  return G2V_OK;
}

extern "C" int g2v_dec_rollout_fwd(const float* target, const float* h_init, const g2v_dec_weights* w,
                                   const g2v_dec_saved* s, const uint8_t* keep95, const uint8_t* keep_l0, float p_drop,
                                   int n_pre_poses, int conditioned, int training, int T, int B, int D, int H,
                                   void* workspace, size_t workspace_bytes, g2v_stream_t stream) {
  return dec_rollout_fwd_impl(target, h_init, w, s, keep95, keep_l0, p_drop, n_pre_poses, conditioned, training, T, B, D, H,
                              workspace, workspace_bytes, stream, false);
}
// as g2v_dec_rollout_fwd, on a workspace that g2v_dec_rollout_prepare has filled for THIS call (no pack, no clearing launch)
extern "C" int g2v_dec_rollout_fwd_prepared(const float* target, const float* h_init, const g2v_dec_weights* w,
                                            const g2v_dec_saved* s, const uint8_t* keep95, const uint8_t* keep_l0,
                                            float p_drop, int n_pre_poses, int conditioned, int training, int T, int B,
                                            int D, int H, void* workspace, size_t workspace_bytes, g2v_stream_t stream) {
  return dec_rollout_fwd_impl(target, h_init, w, s, keep95, keep_l0, p_drop, n_pre_poses, conditioned, training, T, B, D, H,
                              workspace, workspace_bytes, stream, true);
}

// split path (small batch): six plain transposes + two (B <= 512, H) scratch arrays
static size_t split_bwd_total(int D, int H) { return (size_t)2 * D * H + (size_t)12 * H * H + (size_t)2 * 16 * DSPLIT_MAX_NBLK * H; }
static size_t bwd_pack_bytes_aligned(int D, int H) { return (pack_bwd_total(D, H) * sizeof(float) + 255) / 256 * 256; }
// the fused-weight-gradient persistent backward appends its per-workgroup partial dW / db (PX_MAX_NBLK workgroups)
static size_t bwd_wslab_bytes(int D, int H) {
  return (H == 64 && D == 135) ? (size_t)PX_MAX_NBLK * dec_persist_bwd_wgrad_slab_floats() * sizeof(float) : 0;
}
extern "C" size_t g2v_dec_rollout_bwd_workspace(int D, int H) {
  const size_t a = bwd_pack_bytes_aligned(D, H) + PX_BYTES + bwd_wslab_bytes(D, H), b = split_bwd_total(D, H) * sizeof(float);
  size_t c = 0;
  if (dec_cluster_shape(D, H)) {      // the cluster kernel's exchange records at the largest grid it is admitted for
    const int nt = (H + 15) >> 4, cus = device_cu_count() > 0 ? device_cu_count() : 256;
    c = dec_cluster_bwd_xch_bytes(cus / nt + 1, H);
  }
  const size_t ab = a > b ? a : b;
  return ab > c ? ab : c;
}
// 1: g2v_dec_rollout_fwd / _bwd take the small-batch cluster kernels for this shape (T >= 3) under the current setting
extern "C" int g2v_dec_rollout_cluster_ok(int B, int D, int H) {
  if (B <= 0 || !persist_enabled() || !dec_cluster_shape(D, H)) return 0;
  const int nblk = cdiv(B, 16);
  return (nblk <= DSPLIT_MAX_NBLK && (int64_t)nblk * ((H + 15) >> 4) <= device_cu_count()) ? 1 : 0;
}
size_t g2v_internal_gru_cluster_region(int T, int B, int H, int ndir, int bwd);      // gru.hip
// Clear the exchange records of the NEXT persistent cluster launch of this kind over `workspace` now, on `stream`, and note it: that
// launch then starts with its kernel instead of a memset node (see g2v.h).  kind: 0 / 1 g2v_gru_seq_fwd / _bwd (T, B, H, ndir),
// 2 / 3 g2v_dec_rollout_fwd / _bwd (B, D, H).  Shapes that do not run as a cluster: nothing happens.
extern "C" int g2v_cluster_exchange_preclear(int kind, int T, int B, int D, int H, int ndir, void* workspace, size_t workspace_bytes,
                                             g2v_stream_t stream) {
  G2V_REQUIRE(workspace, "null pointer");
  G2V_REQUIRE(kind >= 0 && kind <= 3, "kind: 0 gru fwd, 1 gru bwd, 2 decoder fwd, 3 decoder bwd");
  size_t off = 0, bytes = 0;
  if (kind <= 1) {
    bytes = g2v_internal_gru_cluster_region(T, B, H, ndir, kind);
  } else if (g2v_dec_rollout_cluster_ok(B, D, H) && T >= 3) {
    off = kind == 2 ? fwd_pack_bytes_aligned(D, H) : 0;
    bytes = kind == 2 ? dec_cluster_fwd_xch_bytes(cdiv(B, 16), H) : dec_cluster_bwd_xch_bytes(cdiv(B, 16), H);
  }
  if (bytes == 0 || off + bytes > workspace_bytes) return G2V_OK;
  if (hipMemsetAsync((char*)workspace + off, 0, bytes, (hipStream_t)stream) != hipSuccess) {
    set_error("g2v_cluster_exchange_preclear: memset failed");
    return G2V_ERR_LAUNCH;
  }
  g2v_internal_preclear_note((char*)workspace + off, bytes);
  return G2V_OK;
}
// 0: one launch per time step; R >= 1: ONE persistent launch each way with R row tiles per workgroup (see g2v.h)
extern "C" int g2v_dec_rollout_tiles_per_workgroup(int B, int D, int H) {
  return (H == 64 && D == 135 && B > 0 && (B % 4) == 0) ? persist_tiles_per_wg(cdiv(B, 16)) : 0;
}
extern "C" int g2v_dec_rollout_fuses_loss(int B, int D, int H, int T) {
  return (H == 64 && D == 135 && B > 0 && (B % 16) == 0 && persist_tiles_per_wg(B / 16) == 1 && T >= 2 && T <= 256) ? 1 : 0;
}
extern "C" int g2v_dec_rollout_bwd_fuses_wgrad(int B, int D, int H) {
  return (H == 64 && D == 135 && B > 0 && (B % 16) == 0 && persist_tiles_per_wg(B / 16) == 1) ? 8 : 0;      // bit m <-> matrix m of (ih0, hh0, ih1, hh1): W_hh1
}

// custom_loss as a chaser of the persistent forward rollout (dec_persist.hip, loss_chase_kernel; include/g2v.h).
extern "C" int g2v_custom_loss_chase(const float* target, const g2v_dec_saved* s, const uint8_t* keep95, int T, int B, int D, int H,
                                     void* fwd_workspace, size_t fwd_bytes, g2v_stream_t stream) {
  G2V_REQUIRE(target && s && keep95 && fwd_workspace, "null pointer");
  G2V_REQUIRE(s->y && s->loss_code && s->loss_coef && s->loss_partial, "missing loss_* buffer");
  if (!g2v_dec_rollout_fuses_loss(B, D, H, T)) {
    set_error("g2v_custom_loss_chase: g2v_dec_rollout_fuses_loss(B, D, H, T) does not hold");
    return G2V_ERR_UNSUPPORTED;
  }
  auto a16 = [](const void* q_) { return (reinterpret_cast<uintptr_t>(q_) & 15) == 0; };
  G2V_REQUIRE(a16(s->y) && a16(s->loss_code) && a16(s->loss_coef) && a16(keep95) && a16(fwd_workspace), "16-byte alignment");
  if (fwd_bytes < g2v_dec_rollout_fwd_workspace(D, H)) {
    set_error("g2v_custom_loss_chase: workspace too small");
    return G2V_ERR_WORKSPACE;
  }
  return dec_persist_loss_chase_launch(target, s, keep95, T, B, (char*)fwd_workspace + fwd_pack_bytes_aligned(D, H),
                                       (hipStream_t)stream);
}

// Everything of a rollout pair that depends on the WEIGHTS only -- the fragment packs of the forward and of the backward -- and
// the clearing of the two exchange regions, launched ahead of time (the engine runs it as a parallel branch at the start of the
// step): g2v_dec_rollout_fwd_prepared / _bwd_prepared then start with their first real kernel.  The two workspaces must not be
// touched between this call and the prepared calls.  bwd_workspace may be NULL (inference).  Other shapes: a no-op.
extern "C" int g2v_dec_rollout_prepare(const g2v_dec_weights* w, int D, int H, void* fwd_workspace, size_t fwd_bytes,
                                       void* bwd_workspace, size_t bwd_bytes, g2v_stream_t stream) {
  G2V_REQUIRE(w && fwd_workspace, "null pointer");
  G2V_REQUIRE(D > 0 && H > 0, "bad size");
  if (!dec_prepared_shape(D, H)) return G2V_OK;
  if (fwd_bytes < g2v_dec_rollout_fwd_workspace(D, H) || (bwd_workspace && bwd_bytes < g2v_dec_rollout_bwd_workspace(D, H))) {
    set_error("g2v_dec_rollout_prepare: workspace too small");
    return G2V_ERR_WORKSPACE;
  }
  hipStream_t st = (hipStream_t)stream;
  PackBatch pb;
  (void)dec_fwd_pack_layout(w, D, H, fwd_workspace, pb);
  launch_pack(pb, st);
  (void)hipMemsetAsync((char*)fwd_workspace + fwd_pack_bytes_aligned(D, H), 0, PX_BYTES, st);
  if (bwd_workspace) {
    (void)dec_bwd_pack_layout(w, D, H, bwd_workspace, pb);
    launch_pack(pb, st);
    (void)hipMemsetAsync((char*)bwd_workspace + bwd_pack_bytes_aligned(D, H), 0, PX_BYTES, st);
  }
  G2V_CHECK_LAUNCH();
  return G2V_OK;
}

// gru.hip
namespace g2v {
int gru_bwd_prepare_descs(const float* const* w_hh, const float* const* w_ih, int ndir, int H, bool fused, void* bwd_workspace,
                          size_t bwd_bytes, PackDesc* out, int max_out);
}
// g2v_dec_rollout_prepare (both workspaces) + g2v_gru_seq_prepare (backward workspace) of one fused train step as ONE launch
// (round 5: they were six -- three packs, two memset nodes, one pack -- at ~6 us each in the step's side branch, which had
// become longer than the encoder GRU it runs beside: the quantiser waited for it).  Shapes the prepared entry points do not
// honour: G2V_ERR_UNSUPPORTED, nothing launched (call the two functions instead).
extern "C" int g2v_train_step_prepare(const g2v_dec_weights* w, int D, int H, void* dec_fwd_workspace, size_t dec_fwd_bytes,
                                      void* dec_bwd_workspace, size_t dec_bwd_bytes, const float* const* gru_w_hh,
                                      const float* const* gru_w_ih, int gru_ndir, int gru_fused, void* gru_bwd_workspace,
                                      size_t gru_bwd_bytes, g2v_stream_t stream) {
  G2V_REQUIRE(w && dec_fwd_workspace && dec_bwd_workspace && gru_w_hh && gru_bwd_workspace, "null pointer");
  G2V_REQUIRE(D > 0 && H > 0 && gru_ndir >= 1 && gru_ndir <= 2, "bad size");
  G2V_REQUIRE(!gru_fused || gru_w_ih, "fused needs w_ih");
  if (!dec_prepared_shape(D, H)) {
    set_error("g2v_train_step_prepare: H == 64, D == 135 only");
    return G2V_ERR_UNSUPPORTED;
  }
  if (dec_fwd_bytes < g2v_dec_rollout_fwd_workspace(D, H) || dec_bwd_bytes < g2v_dec_rollout_bwd_workspace(D, H)) {
    set_error("g2v_train_step_prepare: workspace too small");
    return G2V_ERR_WORKSPACE;
  }
  PackBatchL L;
  PackBatch pb;
  L.n = 0;
  (void)dec_fwd_pack_layout(w, D, H, dec_fwd_workspace, pb);
  for (int k = 0; k < pb.n; ++k) L.d[L.n++] = pb.d[k];
  (void)dec_bwd_pack_layout(w, D, H, dec_bwd_workspace, pb);
  for (int k = 0; k < pb.n; ++k) L.d[L.n++] = pb.d[k];
  const int ng = gru_bwd_prepare_descs(gru_w_hh, gru_w_ih, gru_ndir, H, gru_fused != 0, gru_bwd_workspace, gru_bwd_bytes,
                                       L.d + L.n, MAX_PACK_L - L.n);
  if (ng <= 0) {
    set_error("g2v_train_step_prepare: encoder GRU workspace too small");
    return G2V_ERR_WORKSPACE;
  }
  L.n += ng;
  static_assert(PX_BYTES % 16 == 0, "exchange region: whole float4s");
  L.nfill = 2;
  L.fill_dst[0] = (float4*)((char*)dec_fwd_workspace + fwd_pack_bytes_aligned(D, H));
  L.fill_dst[1] = (float4*)((char*)dec_bwd_workspace + bwd_pack_bytes_aligned(D, H));
  L.fill_n4[0] = L.fill_n4[1] = (int64_t)(PX_BYTES / 16);
  launch_pack_fill(L, (hipStream_t)stream);
  G2V_CHECK_LAUNCH();
  return G2V_OK;
}

static int dec_rollout_bwd_impl(const g2v_dec_weights* w, const g2v_dec_saved* s, const g2v_dec_grads* g,
                                const uint8_t* keep95, const uint8_t* keep_l0, float p_drop, int n_pre_poses,
                                int conditioned, int T, int B, int D, int H, void* workspace, size_t workspace_bytes,
                                g2v_stream_t stream, bool prepared_req) {
  const bool prepared = prepared_req && dec_prepared_shape(D, H);
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
  {
    auto al16 = [](const void* q_) { return (reinterpret_cast<uintptr_t>(q_) & 15) == 0; };
    const int nblk = cdiv(B, 16);
    const bool split = !((H == 64) && (D == 135)) && nblk <= DSPLIT_MAX_NBLK && (H & 3) == 0 && H <= 16 * DSPLIT_KS && 3 * H <= 16 * DSPLIT_KSG &&
                       D <= 16 * DSPLIT_DT && al16(workspace) && al16(s->u) && al16(s->a) && al16(s->h0) && al16(s->h1) &&
                       al16(s->gates0) && al16(s->gates1) && al16(s->bn_stats) && al16(g->du) && al16(g->dbn) && al16(g->dgi0) &&
                       al16(g->dgh0) && al16(g->dgi1) && al16(g->dgh1) && al16(g->dh_init) && al16(g->bn_bwd_partial) &&
                       al16(w->bn_w);
    bool cluster = split && persist_enabled() && T >= 3 && dec_cluster_shape(D, H) && (int64_t)nblk * ((H + 15) >> 4) <= device_cu_count() &&
                   (!keep_l0 || al16(keep_l0)) && dec_cluster_bwd_xch_bytes(nblk, H) <= workspace_bytes && !g->dw_gru[0] && !g->dw_gru[1] &&
                   !g->dw_gru[2] && !g->dw_gru[3] && !s->loss_code;
    if (cluster) {
      const void* fn = (const void*)dec_cluster_bwd_kernel<DCL_KS>;
      static bool attr_set = false;
      int nocc = 0;
      if (!attr_set) attr_set = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dec_cluster_bwd_dyn_lds()) == hipSuccess;
      cluster = attr_set && hipOccupancyMaxActiveBlocksPerMultiprocessor(&nocc, fn, 256, dec_cluster_bwd_dyn_lds()) == hipSuccess && nocc >= 1;
    }
    if (!cluster) g2v_internal_preclear_drop(workspace, workspace_bytes);
    if (cluster) {
      // ONE persistent launch for the whole BPTT (dec_cluster_bwd_kernel): weights read in place, no transposes
      const size_t Hp = (size_t)((H + 15) & ~15);
      unsigned long long* x0 = reinterpret_cast<unsigned long long*>(workspace);
      DecClBwdArgs ca;
      ca.keep95 = keep95; ca.keep_l0 = keep_l0; ca.w = *w; ca.sv = *s; ca.g = *g;
      ca.xd = x0; ca.xp = x0 + (size_t)2 * nblk * 16 * Hp; ca.xq = ca.xp + (size_t)2 * nblk * 2 * Hp;
      ca.xcc = reinterpret_cast<unsigned*>(ca.xq + (size_t)nblk * 4 * (Hp / 16) * 16 * Hp);
      ca.fault = const_cast<unsigned*>(g2v_internal_persist_fault_ptr());
      ca.T = T; ca.B = B; ca.D = D; ca.H = H; ca.n_pre = n_pre_poses; ca.conditioned = conditioned; ca.p_drop = p_drop;
      ca.nt = (H + 15) >> 4; ca.nblk = nblk;
      if (!g2v_internal_preclear_take(x0, dec_cluster_bwd_xch_bytes(nblk, H)) &&
          hipMemsetAsync(x0, 0, dec_cluster_bwd_xch_bytes(nblk, H), st) != hipSuccess) {
        set_error("g2v_dec_rollout_bwd: clearing the exchange records failed");
        return G2V_ERR_LAUNCH;
      }
      hipLaunchKernelGGL(dec_cluster_bwd_kernel<DCL_KS>, dim3(((H + 15) >> 4) * nblk), dim3(256), dec_cluster_bwd_dyn_lds(), st, ca);
      G2V_CHECK_LAUNCH();
      return G2V_OK;
    }
    if (split) {
      const int64_t BH = (int64_t)B * H, BD = (int64_t)B * D, BG = 3 * BH;
      float* w_pre_t = p;                 p += (size_t)D * H;
      float* w_out_t = p;                 p += (size_t)H * D;
      float* w_hh1_t = p;                 p += (size_t)3 * H * H;
      float* w_ih1_t = p;                 p += (size_t)3 * H * H;
      float* w_hh0_t = p;                 p += (size_t)3 * H * H;
      float* w_ih0_t = p;                 p += (size_t)3 * H * H;
      float* direct1 = p;                 p += (size_t)BH;
      float* direct0 = p;
      launch_transpose(w->w_pre, w_pre_t, H, D, st);          // (H,D) -> (D,H)
      launch_transpose(w->w_out, w_out_t, D, H, st);          // (D,H) -> (H,D)
      launch_transpose(w->w_hh1, w_hh1_t, 3 * H, H, st);      // (3H,H) -> (H,3H)
      launch_transpose(w->w_ih1, w_ih1_t, 3 * H, H, st);
      launch_transpose(w->w_hh0, w_hh0_t, 3 * H, H, st);
      launch_transpose(w->w_ih0, w_ih0_t, 3 * H, H, st);
      const bool drop = keep_l0 && p_drop > 0.f;
      const int nht = (H + 15) >> 4, ndt = (D + 15) >> 4;
      for (int t = T - 1; t >= 0; --t) {
        const bool last = (t == T - 1);
        if (!last) {
          DecBwdDyArgs a{};
          a.part = g->bn_bwd_partial + (int64_t)((t + 1) & 1) * nblk * 2 * H;
          a.stats = s->bn_stats + (int64_t)t * 2 * H; a.u = s->u + t * BH; a.dbn = g->dbn + t * BH; a.bn_w = w->bn_w;
          a.du = g->du + t * BH; a.d_bn_w = g->d_bn_w; a.d_bn_b = g->d_bn_b; a.w_pre_t = w_pre_t;
          a.dy = g->dy + t * BD; a.keep95 = keep95 + t * BD;
          a.nblk = nblk; a.first_acc = (t == T - 2) ? 1 : 0; a.feedback = (conditioned && t >= n_pre_poses) ? 1 : 0; a.only_a = (t == 0) ? 1 : 0;
          hipLaunchKernelGGL(dec_bwd_dy_split_kernel, dim3(nblk, (a.feedback && !a.only_a) ? ndt : 1), dim3(64), 0, st, a, B, D, H);
        }
        if (t == 0) break;
        DecBwdCell1Args c{};
        c.dy = g->dy + t * BD; c.w_out_t = w_out_t; c.carry = last ? nullptr : g->dh_init + BH;
        c.gates = s->gates1 + (t - 1) * 4 * BH; c.hprev = s->h1 + (t - 1) * BH;
        c.dgi = g->dgi1 + (t - 1) * BG; c.dgh = g->dgh1 + (t - 1) * BG; c.direct = direct1;
        hipLaunchKernelGGL(dec_bwd_cell1_split_kernel, dim3(nblk, nht), dim3(64), 0, st, c, B, D, H);
        DecBwdPairArgs e{};
        e.dgh = g->dgh1 + (t - 1) * BG; e.w_hh_t = w_hh1_t; e.direct = direct1; e.carry_out = g->dh_init + BH;
        e.dgi = g->dgi1 + (t - 1) * BG; e.w_ih_t = w_ih1_t;
        e.keep = drop ? keep_l0 + (t - 1) * BH : nullptr; e.keep_scale = 1.0f / (1.0f - p_drop);
        e.carry0 = last ? nullptr : g->dh_init; e.gates0 = s->gates0 + (t - 1) * 4 * BH; e.hprev0 = s->h0 + (t - 1) * BH;
        e.dgi0 = g->dgi0 + (t - 1) * BG; e.dgh0 = g->dgh0 + (t - 1) * BG; e.direct0 = direct0;
        hipLaunchKernelGGL(dec_bwd_pair_split_kernel<false>, dim3(nblk, nht), dim3(128), 0, st, e, B, H);
        DecBwdPairArgs f{};
        f.dgh = g->dgh0 + (t - 1) * BG; f.w_hh_t = w_hh0_t; f.direct = direct0; f.carry_out = g->dh_init;
        f.dgi = g->dgi0 + (t - 1) * BG; f.w_ih_t = w_ih0_t;
        f.a_act = s->a + (t - 1) * BH; f.u = s->u + (t - 1) * BH; f.stats = s->bn_stats + (int64_t)(t - 1) * 2 * H;
        f.dbn = g->dbn + (t - 1) * BH; f.part = g->bn_bwd_partial + (int64_t)(t & 1) * nblk * 2 * H;
        hipLaunchKernelGGL(dec_bwd_pair_split_kernel<true>, dim3(nblk, nht), dim3(128), 0, st, f, B, H);
      }
      G2V_CHECK_LAUNCH();
      return G2V_OK;
    }
  }
  PackBatch pb;
  const DecTW tw = dec_bwd_pack_layout(w, D, H, workspace, pb);
  if (!prepared) {
    launch_pack(pb, st);
    G2V_CHECK_LAUNCH();
  }
  if (lds > 48 * 1024) {
    (void)hipFuncSetAttribute((const void*)dec_step_bwd_kernel<0, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    (void)hipFuncSetAttribute((const void*)dec_step_bwd_kernel<64, 135>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  }
  int scratch = 1024;
  (void)dec_bwd_lds(D, H, &scratch);
  DecDims dm{T, B, D, H, p_drop, n_pre_poses, conditioned, 1, cdiv(B, 16), (H == 64 && D == 135) ? dec_wt_stores() : 0, scratch};
  const bool fast = (H == 64) && (D == 135);
  {
    auto a16 = [](const void* q_) { return (reinterpret_cast<uintptr_t>(q_) & 15) == 0; };
    const int ptiles = persist_tiles_per_wg(dm.nblk);
    const bool persist = fast && ptiles >= 1 && (B % 4) == 0 &&      // (B % 16 != 0: a ragged last tile, the multi-tile kernels)
                        
                         a16(s->u) && a16(s->a) && a16(s->h0) && a16(s->h1) && a16(s->gates0) && a16(s->gates1) &&
                         a16(s->bn_stats) && a16(g->dy) && a16(g->du) && a16(g->dgi0) && a16(g->dgh0) && a16(g->dgi1) &&
                         a16(g->dgh1) && a16(g->dh_init) && a16(keep95) && a16(keep_l0) && a16(workspace);
    bool want_w = false, ok_w = true;
    const int fmask = g2v_dec_rollout_bwd_fuses_wgrad(B, D, H);
    for (int m = 0; m < 4; ++m) {
      const bool has = g->dw_gru[m] || g->db_gru[m];
      want_w = want_w || has;
      ok_w = ok_w && (((fmask >> m) & 1) ? (g->dw_gru[m] && g->db_gru[m]) : !has);      // exactly the fused matrices, both pointers
    }
    if (want_w && !(ok_w && persist && fmask != 0)) {
      set_error("g2v_dec_rollout_bwd: dw_gru / db_gru must name exactly the matrices of g2v_dec_rollout_bwd_fuses_wgrad(B, D, H)");
      return G2V_ERR_UNSUPPORTED;
    }
    if (s->loss_code && !(persist && s->loss_coef && s->loss_partial && s->loss_terms && s->y && a16(s->y) &&
                          g2v_dec_rollout_fuses_loss(B, D, H, T))) {
      set_error("g2v_dec_rollout_bwd: the loss_* fields are set where g2v_dec_rollout_fuses_loss(B, D, H, T) does not hold");
      return G2V_ERR_UNSUPPORTED;
    }
    if (persist)
      return dec_persist_bwd_launch(w, s, g, keep95, keep_l0, p_drop, n_pre_poses, conditioned, T, B, tw.w_pre_t, tw.w_out_t,
                                    tw.w_ih0_t, tw.w_hh0_t, tw.w_ih1_t, tw.w_hh1_t, (char*)workspace + bwd_pack_bytes_aligned(D, H), st,
                                    !prepared,
                                    want_w ? (float*)((char*)workspace + bwd_pack_bytes_aligned(D, H) + PX_BYTES) : nullptr, ptiles);
  }
  for (int t = T - 1; t >= 0; --t) {
    if (fast)
      hipLaunchKernelGGL((dec_step_bwd_kernel<64, 135>), dim3(dm.nblk), dim3(256), lds, st, *w, tw, *s, *g, keep95, keep_l0, dm, t);
    else
      hipLaunchKernelGGL((dec_step_bwd_kernel<0, 0>), dim3(dm.nblk), dim3(512), lds, st, *w, tw, *s, *g, keep95, keep_l0, dm, t);
  }
  G2V_CHECK_LAUNCH();
  return G2V_OK;
}

extern "C" int g2v_dec_rollout_bwd(const g2v_dec_weights* w, const g2v_dec_saved* s, const g2v_dec_grads* g,
                                   const uint8_t* keep95, const uint8_t* keep_l0, float p_drop, int n_pre_poses,
                                   int conditioned, int T, int B, int D, int H, void* workspace, size_t workspace_bytes,
                                   g2v_stream_t stream) {
  return dec_rollout_bwd_impl(w, s, g, keep95, keep_l0, p_drop, n_pre_poses, conditioned, T, B, D, H, workspace, workspace_bytes,
                              stream, false);
}
extern "C" int g2v_dec_rollout_bwd_prepared(const g2v_dec_weights* w, const g2v_dec_saved* s, const g2v_dec_grads* g,
                                            const uint8_t* keep95, const uint8_t* keep_l0, float p_drop, int n_pre_poses,
                                            int conditioned, int T, int B, int D, int H, void* workspace,
                                            size_t workspace_bytes, g2v_stream_t stream) {
  return dec_rollout_bwd_impl(w, s, g, keep95, keep_l0, p_drop, n_pre_poses, conditioned, T, B, D, H, workspace, workspace_bytes,
                              stream, true);
}
