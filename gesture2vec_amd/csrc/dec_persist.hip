// dec_persist.hip -- PERSISTENT pose-decoder rollout for the BASELINE shape (H = 64, D = 135, 2 GRU layers, B % 16 == 0,
// B / 16 <= CU count): ONE launch for all T steps, forward; ONE launch for the whole BPTT, backward.
//
// Same arithmetic, same saved-for-backward arrays and the same C-ABI entry points (g2v_dec_rollout_fwd / _bwd pick this
// path when the shape allows) as the one-launch-per-step kernels of dec_rollout.hip, which replace
// model/Autoencoder_VQVAE_model.py:1039-1054 (the T-1 step loop) over Generator.forward (:646-683) ->
// BahdanauAttnDecoderRNN.forward (:499-592).  See dec_persist.hpp for the design and the exchange protocol.
//
// Forward, per workgroup (16 batch rows, 4 waves; wave w owns hidden features 16w..16w+15):
//   registers : W_ih0, W_hh0, W_ih1, W_hh1 as MFMA A fragments (4 x 48 VGPRs), u_{t+1} tile, gate pre-activations
//   LDS       : W_out / W_pre fragments (48 + 36 KB), all biases, a_t, h0, h1, x1 tiles, the dense y tile, xin tile
//   per step  : [hidden-side products W_hh0 h0, W_hh1 h1 -- independent of BatchNorm, they run while the exchange of the
//               previous step's partial sums is in flight] -> exchange -> BN + ReLU -> cell 0 -> cell 1 -> out layer ->
//               Dropout(0.95) -> pre_linear -> partial sums -> publish
#define G2V_PERSIST_DEVICE_CODE      // this translation unit owns the fault latch (dec_persist.hpp)
#include "dec_persist.hpp"
#include <utility>
#ifndef G2V_BWD_EPI
#define G2V_BWD_EPI 1
#endif

#ifdef G2V_PSTAMPS      // diagnostic build only (gpurun_tools/pstamps.py): shader-clock stamps of one step of four workgroups
__device__ unsigned long long g2v_pstamps[2 * 4 * 24];
#define PSTAMP(dir, k)                                                                                                   \
  do {                                                                                                                   \
    const int sb_ = blockIdx.x == 0 ? 0 : (blockIdx.x == 5 ? 1 : (blockIdx.x == 128 ? 2 : (blockIdx.x == 255 ? 3 : -1))); \
    if (threadIdx.x == 0 && sb_ >= 0 && t == 10) g2v_pstamps[((dir) * 4 + sb_) * 24 + (k)] = __builtin_amdgcn_s_memtime();  \
  } while (0)
extern "C" int g2v_read_pstamps(unsigned long long* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g2v_pstamps), sizeof(unsigned long long) * 2 * 4 * 24);
}
#else
#define PSTAMP(dir, k)
#endif

namespace g2v {

namespace {
constexpr int H = 64, D = 135, Dp = 144, LDH = 68, LDD = 148;
constexpr int KSH = 4, KSD = 9;               // k-steps over H and over the padded D
constexpr int OUT_TILES = 12;                 // D tiles padded to 3 per wave (tiles 9..11 are zero and never multiplied)
// LDS layout (floats)
constexpr int L_XA = 0, L_XH0 = L_XA + 16 * LDH, L_XH1 = L_XH0 + 16 * LDH, L_XX1 = L_XH1 + 16 * LDH;
constexpr int L_XY = L_XX1 + 16 * LDH;                    // xin_{t+1} tile [16][LDD]
constexpr int L_YT = L_XY + 16 * LDD;                     // dense y tile [16 * D] (+ pad)
constexpr int L_XT = L_YT + 2176;                         // dense xin tile [16 * D] (+ pad)
constexpr int L_KT = L_XT + 2176;                         // keep95 bytes of the tile (16 * D bytes, padded)
constexpr int L_POUT = L_KT + 544;                        // packed W_out: the 9 real tiles x 4 k-steps x 256 (the packed image's
constexpr int L_PPRE = L_POUT + 9 * KSH * 256;            //   padding tiles 9..11 are never multiplied); packed W_pre: 4 tiles x 9 x 256
constexpr int L_BIAS = L_PPRE + 4 * KSD * 256;            // b_ih0 b_hh0 b_ih1 b_hh1 (192 each) b_out (144) b_pre bn_w bn_b (64 each)
constexpr int L_ST = L_BIAS + 4 * 192 + 144 + 3 * 64;     // mean[64], invstd[64]
constexpr int L_RED = L_ST + 128;                         // [16][128]
constexpr int L_TOT = L_RED + 16 * 128;                   // [128]
constexpr int L_END = L_TOT + 128;
static_assert(L_END * 4 <= 160 * 1024, "LDS budget of the persistent rollout forward");
constexpr int B_IH0 = 0, B_HH0 = 192, B_IH1 = 384, B_HH1 = 576, B_OUT = 768, B_PRE = 912, B_BNW = 976, B_BNB = 1040;
}  // namespace

size_t dec_persist_fwd_lds_bytes(int tiles_per_wg) { return (size_t)(L_END + (tiles_per_wg - 1) * 2 * 16 * LDH) * sizeof(float); }

struct DecPersistArgs {
  const float* target;      // (B,T,D)
  const float* h_init;      // (2,B,H)
  g2v_dec_weights w;
  // fragment-major packed weights (pack_kernel): pre (4 tiles x 9), ih0/hh0/ih1/hh1 (3 gates x 4 tiles x 4), out (12 tiles x 4)
  const float* p_pre; const float* p_ih0; const float* p_hh0; const float* p_ih1; const float* p_hh1; const float* p_out;
  g2v_dec_saved sv;
  const uint8_t* keep95;    // (T-1,B,D)
  const uint8_t* keep_l0;   // (T-1,B,H) or null
  PersistX x;
  int T, B, nblk, n_pre, conditioned, training;
  float p_drop;
};

// A-operand fragments from LDS (conflict-free ds_read_b128: consecutive lanes read consecutive 16 bytes)
template <int NT, int KS_T>
__device__ __forceinline__ void lds_frag_mma(f32x4 (&acc)[NT], const float* P, int tile0, int tile_stride, const float* Xs,
                                             int ldx, int lane) {
  const float* xrow = Xs + (lane & 15) * ldx + 4 * (lane >> 4);
  // software pipeline: the fragments of k-step s+1 are requested before the MFMAs of k-step s (one wave per SIMD: an LDS
  // round trip in front of every k-step would otherwise idle the matrix pipe ~150 cycles per step)
  float4 wn[NT], xn;
#pragma unroll
  for (int t = 0; t < NT; ++t) wn[t] = *reinterpret_cast<const float4*>(P + ((tile0 + t * tile_stride) * KS_T) * 256 + lane * 4);
  xn = *reinterpret_cast<const float4*>(xrow);
#pragma unroll
  for (int s = 0; s < KS_T; ++s) {
    const float4 xb = xn;
    float4 wv[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) wv[t] = wn[t];
    if (s + 1 < KS_T) {
#pragma unroll
      for (int t = 0; t < NT; ++t)
        wn[t] = *reinterpret_cast<const float4*>(P + ((tile0 + t * tile_stride) * KS_T + s + 1) * 256 + lane * 4);
      xn = *reinterpret_cast<const float4*>(xrow + 16 * (s + 1));
    }
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = mfma16(wv[t].x, xb.x, acc[t]);
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = mfma16(wv[t].y, xb.y, acc[t]);
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = mfma16(wv[t].z, xb.z, acc[t]);
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = mfma16(wv[t].w, xb.w, acc[t]);
  }
}

// frag_mma with every operand fragment of the activation tile requested before the first MFMA: one wave per SIMD, so a
// ds_read + s_waitcnt lgkmcnt(0) in front of each k-step's 12 MFMAs leaves the matrix pipe idle for most of an LDS round
// trip, four times per product
template <int NT, int KS_T>
__device__ __forceinline__ void frag_mma_x1st(f32x4 (&acc)[NT], const WFrag<NT, KS_T>& f, const float* Xs, int ldx, int lane) {
  const float* xrow = Xs + (lane & 15) * ldx + 4 * (lane >> 4);
  float4 xb[KS_T];
#pragma unroll
  for (int s = 0; s < KS_T; ++s) xb[s] = *reinterpret_cast<const float4*>(xrow + 16 * s);
#pragma unroll
  for (int s = 0; s < KS_T; ++s) {
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = mfma16(f.w[t][s].x, xb[s].x, acc[t]);
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = mfma16(f.w[t][s].y, xb[s].y, acc[t]);
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = mfma16(f.w[t][s].z, xb[s].z, acc[t]);
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = mfma16(f.w[t][s].w, xb[s].w, acc[t]);
  }
}

// compile-time loop: f(integral_constant<int, 0>) ... f(integral_constant<int, N-1>) -- register arrays indexed by the counter
template <class F, int... Ks>
__device__ __forceinline__ void static_for_(F&& f, std::integer_sequence<int, Ks...>) {
  (f(std::integral_constant<int, Ks>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {      // f(integral_constant<int, 0>) ... f(integral_constant<int, N-1>)
  static_for_(static_cast<F&&>(f), std::make_integer_sequence<int, N>{});
}
// One output tile, contraction over KS_T k-steps, as TWO accumulator chains (even / odd k-steps) that alternate in the
// matrix pipe and are added at the end: a single 36-long dependent chain issues one MFMA per ~42 cycles at one wave per SIMD,
// two alternating chains one per 32.
template <int KS_T>
__device__ __forceinline__ f32x4 lds_frag_mma_2chain(const float* P, int tile, const float* Xs, int ldx, int lane) {
  const float* xrow = Xs + (lane & 15) * ldx + 4 * (lane >> 4);
  const float* wp = P + (tile * KS_T) * 256 + lane * 4;
  f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f};
  float4 w0 = *reinterpret_cast<const float4*>(wp), x0 = *reinterpret_cast<const float4*>(xrow);
  float4 w1 = *reinterpret_cast<const float4*>(wp + 256), x1 = *reinterpret_cast<const float4*>(xrow + 16);
#pragma unroll
  for (int s = 0; s < KS_T; s += 2) {
    const float4 wa = w0, xa = x0, wb = w1, xb = x1;
    const bool two = s + 1 < KS_T;
    if (s + 2 < KS_T) {
      w0 = *reinterpret_cast<const float4*>(wp + (s + 2) * 256);
      x0 = *reinterpret_cast<const float4*>(xrow + 16 * (s + 2));
    }
    if (s + 3 < KS_T) {
      w1 = *reinterpret_cast<const float4*>(wp + (s + 3) * 256);
      x1 = *reinterpret_cast<const float4*>(xrow + 16 * (s + 3));
    }
    a0 = mfma16(wa.x, xa.x, a0);
    if (two) a1 = mfma16(wb.x, xb.x, a1);
    a0 = mfma16(wa.y, xa.y, a0);
    if (two) a1 = mfma16(wb.y, xb.y, a1);
    a0 = mfma16(wa.z, xa.z, a0);
    if (two) a1 = mfma16(wb.z, xb.z, a1);
    a0 = mfma16(wa.w, xa.w, a0);
    if (two) a1 = mfma16(wb.w, xb.w, a1);
  }
  return (f32x4){a0[0] + a1[0], a0[1] + a1[1], a0[2] + a1[2], a0[3] + a1[3]};
}

// Gate math of one GRU cell for the 4 consecutive features [f0, f0+4) of batch row i held by this lane.
// Reads h_prev from Xh (own columns) and overwrites it IN PLACE with h_new (the next step's hidden-side operand);
// Xnext gets the (optionally dropped) value the next layer consumes.  Global: h_out, gates (r,z,n,ghn), dropped copy.
__device__ __forceinline__ void cell_epilogue(const f32x4 (&ai)[3], const f32x4 (&ah)[3], const float* bias_i,
                                              const float* bias_h, uint32_t kp, bool drop, float keep_scale, float* Xh,
                                              float* Xnext, float* __restrict__ h_out, float* __restrict__ gates,
                                              float* __restrict__ xdrop_out, int i, int f0) {
  const float4 hp4 = *reinterpret_cast<const float4*>(Xh + i * LDH + f0);
  const float hp[4] = {hp4.x, hp4.y, hp4.z, hp4.w};
  float4 bi[3], bh[3];
#pragma unroll
  for (int g = 0; g < 3; ++g) {
    bi[g] = *reinterpret_cast<const float4*>(bias_i + g * H + f0);
    bh[g] = *reinterpret_cast<const float4*>(bias_h + g * H + f0);
  }
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
    xd[r] = drop ? (((kp >> (8 * r)) & 0xffu) ? hn[r] * keep_scale : 0.f) : hn[r];
    gr_[r] = rr; gz_[r] = zz; gn_[r] = nn; gh_[r] = ghn;
  }
  *reinterpret_cast<float4*>(Xh + i * LDH + f0) = make_float4(hn[0], hn[1], hn[2], hn[3]);
  if (Xnext) *reinterpret_cast<float4*>(Xnext + i * LDH + f0) = make_float4(xd[0], xd[1], xd[2], xd[3]);
  if (h_out) *reinterpret_cast<float4*>(h_out + (int64_t)i * H + f0) = make_float4(hn[0], hn[1], hn[2], hn[3]);
  if (xdrop_out) *reinterpret_cast<float4*>(xdrop_out + (int64_t)i * H + f0) = make_float4(xd[0], xd[1], xd[2], xd[3]);
  if (gates) {
    float* go = gates + (int64_t)i * 4 * H + f0;
    *reinterpret_cast<float4*>(go) = make_float4(gr_[0], gr_[1], gr_[2], gr_[3]);
    *reinterpret_cast<float4*>(go + H) = make_float4(gz_[0], gz_[1], gz_[2], gz_[3]);
    *reinterpret_cast<float4*>(go + 2 * H) = make_float4(gn_[0], gn_[1], gn_[2], gn_[3]);
    *reinterpret_cast<float4*>(go + 3 * H) = make_float4(gh_[0], gh_[1], gh_[2], gh_[3]);
  }
}

// Loads of the rollout's RARE paths (y_0 = target frame 0, teacher-forced steps), issued and waited for on the spot inside one asm
// statement, invisible to hipcc's wait-count pass: as ordinary loads their target registers stay "possibly pending" in the pass's
// view of the time-step loop, and it waits vmcnt(0) at the top of EVERY step -- behind the keep-flag requests and in front of the
// hidden-side products that are there to hide the exchange (ISA, round 4).  The five dwords one thread needs per 4-element group
// (four target values, the group's keep flags): five requests in flight, ONE wait.
__device__ __forceinline__ void ldg_sync5(const float* p0, const float* p1, const float* p2, const float* p3, const float* p4,
                                          float (&v)[4], uint32_t& w) {
  asm volatile(
      "global_load_dword %0, %5, off\n\tglobal_load_dword %1, %6, off\n\tglobal_load_dword %2, %7, off\n\t"
      "global_load_dword %3, %8, off\n\tglobal_load_dword %4, %9, off\n\ts_waitcnt vmcnt(0)"
      : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(w)
      : "v"(p0), "v"(p1), "v"(p2), "v"(p3), "v"(p4)
      : "memory");
}

__global__ __launch_bounds__(256, 1) void dec_persist_fwd_kernel(DecPersistArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Xa = smem + L_XA;
  float* Xh0 = smem + L_XH0;
  float* Xh1 = smem + L_XH1;
  float* Xx1 = smem + L_XX1;
  float* Xy = smem + L_XY;
  float* Xt = smem + L_XT;
  uint32_t* Kt = reinterpret_cast<uint32_t*>(smem + L_KT);
  float* Pout = smem + L_POUT;
  float* Ppre = smem + L_PPRE;
  float* Bs = smem + L_BIAS;
  float* st = smem + L_ST;
  float* red = smem + L_RED;
  float* tot = smem + L_TOT;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i = lane & 15, q = lane >> 4;
  const int b = blockIdx.x, b0 = b * 16;
  const int T = a.T, B = a.B;
  const int f0 = 16 * wave + 4 * q;                  // this lane's 4 hidden features
  const bool training = a.training != 0;
  // custom_loss rides beside this kernel, not inside it (round 3 measured the fold: every VALU instruction costs its full issue
  // time in a one-wave-per-SIMD kernel on a serial chain, 55-60 us for the 68 us it removed).  With sv.loss_code set the kernel only
  // HANDS y_t OVER to the co-resident chaser (loss_chase_kernel below): y tiles are stored write-through (sc1) and, once every wave
  // has waited out its stores (the sweep of the next step's exchange does, vmcnt counts in order) and the workgroup has met at a
  // barrier, one lane publishes the step number in this workgroup's progress word (MI355X_MICROARCH.md, "Valid forms": sc1
  // payload, drained, barrier, sc1 flag by one lane).  Cost here: one 4-byte store per step.
  const bool chase = training && a.sv.loss_code != nullptr;
  unsigned* const yflag = a.x.yflag + (size_t)b * PX_FLAG_STRIDE;
  const bool drop = training && a.keep_l0 && a.p_drop > 0.f;
  const float keep_scale = 1.0f / (1.0f - a.p_drop);
  const int64_t BH = (int64_t)B * H;
  const float unbias = (B > 1) ? (float)B / (float)(B - 1) : 1.0f;
  float run_m = 0.f, run_v = 0.f;                    // BatchNorm running statistics (workgroup 0, tid < H)
  // (w.bn_running_mean == NULL while training: the caller commits the running statistics itself, g2v_bn_running_update)
  if (training && b == 0 && tid < H && a.w.bn_running_mean) {
    run_m = a.w.bn_running_mean[tid];
    run_v = a.w.bn_running_var[tid];
  }

  // ---- prologue: weights into registers / LDS ---------------------------------------------------------------------
  WFrag<3, KSH> f_ih0, f_hh0, f_ih1, f_hh1;
  frag_load(f_hh0, a.p_hh0, wave, 4, lane);
  frag_load(f_hh1, a.p_hh1, wave, 4, lane);
  frag_load(f_ih0, a.p_ih0, wave, 4, lane);
  frag_load(f_ih1, a.p_ih1, wave, 4, lane);
  for (int e = tid; e < 9 * KSH * 64; e += 256)              // tiles 9..11 of the packed image are padding
    reinterpret_cast<float4*>(Pout)[e] = reinterpret_cast<const float4*>(a.p_out)[e];
  for (int e = tid; e < 4 * KSD * 64; e += 256) reinterpret_cast<float4*>(Ppre)[e] = reinterpret_cast<const float4*>(a.p_pre)[e];
  for (int e = tid; e < 192; e += 256) {
    Bs[B_IH0 + e] = a.w.b_ih0[e]; Bs[B_HH0 + e] = a.w.b_hh0[e];
    Bs[B_IH1 + e] = a.w.b_ih1[e]; Bs[B_HH1 + e] = a.w.b_hh1[e];
  }
  for (int e = tid; e < 144; e += 256) Bs[B_OUT + e] = e < D ? a.w.b_out[e] : 0.f;
  if (tid < H) {
    Bs[B_PRE + tid] = a.w.b_pre[tid];
    Bs[B_BNW + tid] = a.w.bn_w[tid];
    Bs[B_BNB + tid] = a.w.bn_b[tid];
    if (!training) {
      st[tid] = a.w.bn_running_mean[tid];
      st[H + tid] = bn_invstd_(a.w.bn_running_var[tid]);
    }
  }
  // zero the padding columns of the operand tiles once (the live columns are rewritten every step)
  for (int e = tid; e < 16 * (LDD - D); e += 256) Xy[(e / (LDD - D)) * LDD + D + (e % (LDD - D))] = 0.f;
  for (int e = tid; e < 4 * 16 * LDH; e += 256) smem[L_XA + e] = 0.f;
  lds_barrier();
  // initial state: h0_0, h1_0 = the quantised latent
  {
    const int r = tid >> 4, c = (tid & 15) * 4;
    const float4 v0 = *reinterpret_cast<const float4*>(a.h_init + (int64_t)(b0 + r) * H + c);
    const float4 v1 = *reinterpret_cast<const float4*>(a.h_init + BH + (int64_t)(b0 + r) * H + c);
    *reinterpret_cast<float4*>(Xh0 + r * LDH + c) = v0;
    *reinterpret_cast<float4*>(Xh1 + r * LDH + c) = v1;
    if (a.sv.h0) *reinterpret_cast<float4*>(a.sv.h0 + (int64_t)(b0 + r) * H + c) = v0;
    if (a.sv.h1) *reinterpret_cast<float4*>(a.sv.h1 + (int64_t)(b0 + r) * H + c) = v1;
  }

  f32x4 u_acc = {0.f, 0.f, 0.f, 0.f};      // u_{t+1}[row i][f0..f0+3] WITHOUT the bias (what the BN partial sums are taken of)
  // fast path of the outputs: the out-layer epilogue leaves y_t (and xin_{t+1}) as dense LDS tiles -> coalesced 16-byte copies
  bool pend_dense = false;
  auto dense_stores = [&](int ts, bool nxt) {
    const int64_t tile = ((int64_t)ts * B + b0) * D;
    const float* Ys = smem + L_YT;
    for (int e4 = tid; e4 < (16 * D) / 4; e4 += 256) {
      st4(a.sv.y + tile + 4 * e4, reinterpret_cast<const float4*>(Ys)[e4], chase);      // (chase: write-through, the chaser reads it)
      if (nxt && a.sv.xin) *reinterpret_cast<float4*>(a.sv.xin + tile + 4 * e4) = reinterpret_cast<const float4*>(Xt)[e4];
    }
  };

  for (int t = 0; t < T; ++t) {
    const bool has_next = t < T - 1;
    float* Yt = smem + L_YT;                                  // y_t's dense tile
    if (t > 0) {
      PSTAMP(0, 0);
      // ---- hidden-side products (independent of this step's BatchNorm): they fill the exchange's latency -----------
      f32x4 gh0[3], gh1[3];
#pragma unroll
      for (int g = 0; g < 3; ++g) {
        gh0[g] = (f32x4){0.f, 0.f, 0.f, 0.f};
        gh1[g] = (f32x4){0.f, 0.f, 0.f, 0.f};
      }
      // keep flags of this step's Dropout(0.95) (consumed by the out-layer epilogue): requested now, landed in LDS below
      uint32_t kreq[3] = {0u, 0u, 0u};
      const bool fast_y = has_next && !(t < a.n_pre);      // the common case: feedback of the model's own output
      if (fast_y && a.conditioned) {
        const uint32_t* kp4 = reinterpret_cast<const uint32_t*>(a.keep95 + ((int64_t)t * B + b0) * D);
#pragma unroll
        for (int j = 0; j < 3; ++j) {
          const int e4 = tid + 256 * j;
          kreq[j] = kp4[e4 < (16 * D) / 4 ? e4 : 0];
        }
      }
      // The exchange is threaded through the two products: the row's records, published at the end of the previous step,
      // have arrived when the first product is done (hop 1: row sum, published again), and the row sums of the other rows
      // travel while the second product runs (hop 2).
      frag_mma_x1st(gh0, f_hh0, Xh0, LDH, lane);
      PSTAMP(0, 9);
      if (training) {
        // (a wave whose lanes all sit out the sweep -- batches of fewer than 256 rows -- has no vmcnt(0) of its own in there)
        // chase: every wave waits out its stores here -- y_{t-2} among them, issued a whole step ago (below, behind hop 2)
        if (chase) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        px_hop1(a.x, t & 1, (unsigned)t, a.nblk, b, red, tot, tid);
      }
      PSTAMP(0, 10);
      frag_mma_x1st(gh1, f_hh1, Xh1, LDH, lane);
      PSTAMP(0, 1);
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const int e4 = tid + 256 * j;
        if (e4 < (16 * D) / 4) Kt[e4] = kreq[j];
      }
      // Every load of this step has been consumed (kreq just now; hop 1 drained everything older): say so.  hipcc's wait-count pass
      // loses track across the loop's paths and otherwise puts an `s_waitcnt vmcnt(0)` behind the BatchNorm barrier below (a
      // register it re-uses there MIGHT still be the target of a load) -- right behind the chase mode's progress-flag store, whose
      // write-through acknowledgement (1-2 us) wave 0 then sat out every step, with the workgroup waiting at the next barrier.
      __builtin_amdgcn_s_waitcnt(0x0F70);      // vmcnt(0), nothing else
      // ---- BatchNorm statistics of u_t -----------------------------------------------------------------------------
      if (training) {
        if (px_two_hops(a.nblk)) px_hop2(a.x, t & 1, (unsigned)t, a.nblk, b, red, tot, tid);      // (one row of workgroups: hop 1's row sum is the total)
        PSTAMP(0, 11);
        if (tid < H) {
          const float s1 = tot[tid], s2 = tot[H + tid];
          const float mv = s1 / (float)B;
          const float var = fmaxf(s2 / (float)B - mv * mv, 0.f);       // biased batch variance
          const float mean = mv + Bs[B_PRE + tid];
          st[tid] = mean;
          st[H + tid] = bn_invstd_(var);
          if (b == 0 && a.sv.bn_stats) {
            a.sv.bn_stats[(int64_t)(t - 1) * 2 * H + tid] = mean;
            a.sv.bn_stats[(int64_t)(t - 1) * 2 * H + H + tid] = var;
          }
          // running statistics (momentum 0.1, unbiased variance), one update per step in step order: workgroup 0 carries
          // them in registers and writes them back once (this was a launch of its own behind the rollout)
          run_m = 0.9f * run_m + 0.1f * mean;
          run_v = 0.9f * run_v + 0.1f * (var * unbias);
        }
        lds_barrier();
        // chase: y_0 .. y_{t-2} of these 16 rows are in memory (every wave drained its stores in front of hop 1, the workgroup
        // has met at barriers since): tell the chaser.  HERE, not earlier: its burst of loads shares this CU's in-order memory
        // pipeline with the exchange's sweeps; from here to the next step's hop 1 the rollout only stores.
        if (chase && t >= 2 && tid == 0) __hip_atomic_store(yflag, (unsigned)(t - 2), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      // y_{t-1} (and xin_{t-1}) out, from the dense tiles the previous step's out-layer epilogue left (valid until this step's):
      // issued HERE so that the next wait on vector memory -- the sweep of the NEXT step's hop 1 -- is a whole step away.  Right
      // behind the previous step's publish they sat in front of this step's sweep: vmcnt counts in order, and the write-through
      // stores of the chase mode are acknowledged by the fabric, not by the L2 (+1.2 us per step, measured).
      if (pend_dense) {
        dense_stores(t - 1, true);
        pend_dense = false;
      }
      PSTAMP(0, 2);
      // ---- a_t = ReLU(BN(u_t)) from the register-resident u tile ---------------------------------------------------
      {
        const float4 m4 = *reinterpret_cast<const float4*>(st + f0), i4 = *reinterpret_cast<const float4*>(st + H + f0);
        const float4 g4 = *reinterpret_cast<const float4*>(Bs + B_BNW + f0), b4 = *reinterpret_cast<const float4*>(Bs + B_BNB + f0);
        const float4 p4 = *reinterpret_cast<const float4*>(Bs + B_PRE + f0);
        float4 a4;
        a4.x = fmaxf(((u_acc[0] + p4.x) - m4.x) * i4.x * g4.x + b4.x, 0.f);
        a4.y = fmaxf(((u_acc[1] + p4.y) - m4.y) * i4.y * g4.y + b4.y, 0.f);
        a4.z = fmaxf(((u_acc[2] + p4.z) - m4.z) * i4.z * g4.z + b4.z, 0.f);
        a4.w = fmaxf(((u_acc[3] + p4.w) - m4.w) * i4.w * g4.w + b4.w, 0.f);
        *reinterpret_cast<float4*>(Xa + i * LDH + f0) = a4;
        if (a.sv.a) *reinterpret_cast<float4*>(a.sv.a + ((int64_t)(t - 1) * B + b0 + i) * H + f0) = a4;
      }
      lds_barrier();
      PSTAMP(0, 3);
      // ---- GRU layer 0 --------------------------------------------------------------------------------------------
      {
        f32x4 ai[3];
#pragma unroll
        for (int g = 0; g < 3; ++g) ai[g] = (f32x4){0.f, 0.f, 0.f, 0.f};
        // Two code paths, not one with a selected keep word: with the load of the keep flags on a conditional path hipcc waits for
        // it (vmcnt(0)) in front of the common epilogue whether or not it was issued, and vector-memory operations retire in
        // order -- without inter-layer dropout (the bench shape) that wait sat out the write-through y / xin / a stores issued one
        // product earlier, every step (ISA, round 4).
        if (drop) {
          const uint32_t kp = *reinterpret_cast<const uint32_t*>(a.keep_l0 + ((int64_t)(t - 1) * B + b0 + i) * H + f0);
          frag_mma_x1st(ai, f_ih0, Xa, LDH, lane);
          cell_epilogue(ai, gh0, Bs + B_IH0, Bs + B_HH0, kp, true, keep_scale, Xh0, Xx1,
                        a.sv.h0 ? a.sv.h0 + ((int64_t)t * B + b0) * H : nullptr,
                        a.sv.gates0 ? a.sv.gates0 + ((int64_t)(t - 1) * B + b0) * 4 * H : nullptr,
                        a.sv.x1 ? a.sv.x1 + ((int64_t)(t - 1) * B + b0) * H : nullptr, i, f0);
        } else {
          frag_mma_x1st(ai, f_ih0, Xa, LDH, lane);
          cell_epilogue(ai, gh0, Bs + B_IH0, Bs + B_HH0, 0x01010101u, false, 1.0f, Xh0, Xx1,
                        a.sv.h0 ? a.sv.h0 + ((int64_t)t * B + b0) * H : nullptr,
                        a.sv.gates0 ? a.sv.gates0 + ((int64_t)(t - 1) * B + b0) * 4 * H : nullptr, nullptr, i, f0);
        }
      }
      lds_barrier();
      PSTAMP(0, 4);
      // ---- GRU layer 1 --------------------------------------------------------------------------------------------
      {
        f32x4 ai[3];
#pragma unroll
        for (int g = 0; g < 3; ++g) ai[g] = (f32x4){0.f, 0.f, 0.f, 0.f};
        frag_mma_x1st(ai, f_ih1, Xx1, LDH, lane);
        cell_epilogue(ai, gh1, Bs + B_IH1, Bs + B_HH1, 0x01010101u, false, 1.0f, Xh1, nullptr,
                      a.sv.h1 ? a.sv.h1 + ((int64_t)t * B + b0) * H : nullptr,
                      a.sv.gates1 ? a.sv.gates1 + ((int64_t)(t - 1) * B + b0) * 4 * H : nullptr, nullptr, i, f0);
      }
      // chase, last step: y_{T-2} (stored behind hop 2 above, two products ago) is waited out and announced now, so that only
      // y_{T-1} is left for the chaser when the rollout ends
      if (chase && !has_next) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      lds_barrier();
      if (chase && !has_next && t >= 2 && tid == 0)
        __hip_atomic_store(yflag, (unsigned)(t - 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      PSTAMP(0, 5);
      // ---- y_t = out_layer(h1_t) -> dense tile --------------------------------------------------------------------
      {
        f32x4 acc[3];
#pragma unroll
        for (int j = 0; j < 3; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (wave == 0) {
          lds_frag_mma<3, KSH>(acc, Pout, wave, 4, Xh1, LDH, lane);
        } else {                               // tiles 9..11 are padding: two tiles for waves 1..3
          f32x4 a2[2] = {acc[0], acc[1]};
          lds_frag_mma<2, KSH>(a2, Pout, wave, 4, Xh1, LDH, lane);
          acc[0] = a2[0]; acc[1] = a2[1];
        }
        const uint8_t* Kb = reinterpret_cast<const uint8_t*>(Kt);
        const bool fast_y = has_next && !(t < a.n_pre);
        // Every LDS read of the epilogue is issued up front and unconditionally (columns past D read in-bounds padding;
        // the padding tiles 9..11 re-read tile 0): reads under the per-element `d < D` test were compiled into 24
        // dependent LDS round trips, each behind its own s_waitcnt (~3000 cycles per step).  Only the stores are predicated.
        float bo[3][4], yv[3][4], xv[3][4];
        uint32_t kb[3][4];
#pragma unroll
        for (int j = 0; j < 3; ++j) {
          const int d0 = 16 * (wave + 4 * j) + 4 * q, d0c = d0 < Dp ? d0 : 4 * q;
          const float4 b4 = *reinterpret_cast<const float4*>(Bs + B_OUT + d0c);
          bo[j][0] = b4.x; bo[j][1] = b4.y; bo[j][2] = b4.z; bo[j][3] = b4.w;
#pragma unroll
          for (int r = 0; r < 4; ++r) kb[j][r] = Kb[i * D + d0c + r];
        }
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            yv[j][r] = acc[j][r] + bo[j][r];
            xv[j][r] = kb[j][r] ? yv[j][r] * 20.0f : 0.f;      // next decoder input = Dropout(0.95)(y_t): 1 / (1 - 0.95) = 20  (:568-570)
          }
#pragma unroll
        for (int j = 0; j < 3; ++j) {
          const int d0 = 16 * (wave + 4 * j) + 4 * q;
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (d0 + r < D) {
              Yt[i * D + d0 + r] = yv[j][r];
              if (fast_y) {
                Xt[i * D + d0 + r] = xv[j][r];
                Xy[i * LDD + d0 + r] = xv[j][r];
              }
            }
        }
      }
      lds_barrier();
      PSTAMP(0, 6);
    }
    // ---- y_t out, next decoder input xin_{t+1} = Dropout(0.95)(y_t | target_t)  (:1049-1052, :568-570) --------------
    // fast path: the out-layer epilogue left y (and xin) as dense tiles -> coalesced 16-byte copies, nothing else; they are
    // issued in the NEXT step behind its exchange (dense_stores above; the tiles stay valid until the next step's out-layer
    // epilogue), the last step's at once
    const bool fast_dense = t > 0 && (!has_next || !(t < a.n_pre));
    if (fast_dense) {
      if (!has_next) dense_stores(t, false);
    } else {
      const int64_t tile = ((int64_t)t * B + b0) * D;        // the block's 16 x D tile is one dense run of the (T,B,D) arrays
      const bool teacher = has_next && (t < a.n_pre);
      const uint32_t* kp4 = reinterpret_cast<const uint32_t*>(a.keep95 + tile);
      for (int e4 = tid; e4 < (16 * D) / 4; e4 += 256) {
        const int e = 4 * e4;
        float yv[4], sv_[4], tv[4];
        uint32_t kraw;
        {   // target frame t of the group's four elements (frame 0 IS y_0, :1039-1040) and its keep flags: one round trip
          const float* tp[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int r = (e + j) / D, c = (e + j) - r * D;
            tp[j] = a.target + ((int64_t)(b0 + r) * T + t) * D + c;
          }
          const bool needk = has_next && a.conditioned;
          ldg_sync5(tp[0], tp[1], tp[2], tp[3], needk ? reinterpret_cast<const float*>(kp4 + e4) : tp[0], tv, kraw);
          if (!needk) kraw = 0u;
        }
        if (t == 0) {
#pragma unroll
          for (int j = 0; j < 4; ++j) yv[j] = tv[j];
        } else {
          const float4 y4 = reinterpret_cast<const float4*>(Yt)[e4];
          yv[0] = y4.x; yv[1] = y4.y; yv[2] = y4.z; yv[3] = y4.w;
        }
        st4(a.sv.y + tile + e, make_float4(yv[0], yv[1], yv[2], yv[3]), chase);
        if (!has_next) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          sv_[j] = (teacher && t > 0) ? tv[j] : yv[j];
        }
        const uint32_t k4 = kraw;
        float xv[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) xv[j] = ((k4 >> (8 * j)) & 0xffu) ? sv_[j] * 20.0f : 0.f;     // 1 / (1 - 0.95)
        if (a.sv.xin) *reinterpret_cast<float4*>(a.sv.xin + tile + e) = make_float4(xv[0], xv[1], xv[2], xv[3]);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int r = (e + j) / D, c = (e + j) - r * D;
          Xy[r * LDD + c] = xv[j];
        }
      }
    }
    if (!has_next) break;
    if (!fast_dense) lds_barrier();      // fast path: Xy was completed by the out-layer epilogue, in front of its barrier
    PSTAMP(0, 7);
    // ---- u_{t+1} = pre_linear.0(xin_{t+1}); partial sums of (u - b) over this block's 16 rows; publish -----------------
    {
      u_acc = lds_frag_mma_2chain<KSD>(Ppre, wave, Xy, LDD, lane);
      const float4 p4 = *reinterpret_cast<const float4*>(Bs + B_PRE + f0);
      if (a.sv.u)
        *reinterpret_cast<float4*>(a.sv.u + ((int64_t)t * B + b0 + i) * H + f0) =
            make_float4(u_acc[0] + p4.x, u_acc[1] + p4.y, u_acc[2] + p4.z, u_acc[3] + p4.w);
      if (training) {
        float s1[4], s2[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          s1[r] = reduce16(u_acc[r]);
          s2[r] = reduce16(u_acc[r] * u_acc[r]);
        }
        if (i == 0) {
          const int par = (t + 1) & 1;
          __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc(
              a.x.rec1 + ((size_t)par * PX_MAX_NBLK + b) * PX_COLS, 0, PX_COLS * 8, 0x00020000);
          const unsigned tag = (unsigned)(t + 1);
          px_publish2(rr, (unsigned)f0, s1[0], s1[1], tag);
          px_publish2(rr, (unsigned)f0 + 2, s1[2], s1[3], tag);
          px_publish2(rr, (unsigned)(H + f0), s2[0], s2[1], tag);
          px_publish2(rr, (unsigned)(H + f0) + 2, s2[2], s2[3], tag);
        }
      }
    }
    if (fast_dense) pend_dense = true;      // (stored behind hop 2 of the next step)
    PSTAMP(0, 8);
    // (no barrier needed here: the next writers of Xy / Yt / Xt sit behind the barriers of step t+1)
  }
  // (a latched fault -- a bounded wait of the exchange ran out -- means garbage statistics: the model's state stays as it was)
  if (training && b == 0 && tid < H && a.w.bn_running_mean &&
      __hip_atomic_load(&g2v_persist_fault, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) {
    a.w.bn_running_mean[tid] = run_m;
    a.w.bn_running_var[tid] = run_v;
  }
  if (chase) {      // y_{T-1}: drained by every wave, then published
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    lds_barrier();
    if (tid == 0) __hip_atomic_store(yflag, (unsigned)(T - 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}


// =====================================================================================================================
// The same forward rollout for MORE row tiles than CUs (round 4): R tiles per workgroup, B / 16 <= R x CU count.  Until now
// a batch of more than 256 row tiles fell back to one launch per time step (B = 8192: 4.0 ms per train step, BELOW the rate at
// B = 4096).  Workgroup b owns tiles b, b + nwg, b + 2 nwg ... (< ntiles) and walks them one after the other inside every time
// step: the four GRU matrices stay in its registers and W_out / W_pre in its LDS, a tile's two hidden states live in LDS across
// the steps (8.7 KB per extra tile), its pre-BatchNorm u tile in 4 registers, and the BatchNorm partial sums of all its tiles are
// added up and published as ONE record per step -- the exchange (dec_persist.hpp) runs once per step over the nwg workgroups,
// threaded through the first tile's hidden-side products exactly as in the one-tile kernel above.  What the one-tile kernel
// hides and this one does not: the other tiles' hidden-side products (no exchange to hide them in) and the deferral of the y /
// xin stores.  The loss chaser is not offered here (one chaser workgroup per tile would not be co-resident).
// =====================================================================================================================
template <int R>
__global__ __launch_bounds__(256, 1) void dec_persist_fwd_mt_kernel(DecPersistArgs a, int nwg) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Xa = smem + L_XA;
  float* Xx1 = smem + L_XX1;
  float* Xy = smem + L_XY;
  float* Xt = smem + L_XT;
  float* Yt = smem + L_YT;
  uint32_t* Kt = reinterpret_cast<uint32_t*>(smem + L_KT);
  float* Pout = smem + L_POUT;
  float* Ppre = smem + L_PPRE;
  float* Bs = smem + L_BIAS;
  float* st = smem + L_ST;
  float* red = smem + L_RED;
  float* tot = smem + L_TOT;
  auto Xh0_of = [&](int r) { return r == 0 ? smem + L_XH0 : smem + L_END + (r - 1) * 2 * 16 * LDH; };
  auto Xh1_of = [&](int r) { return r == 0 ? smem + L_XH1 : smem + L_END + (r - 1) * 2 * 16 * LDH + 16 * LDH; };

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i = lane & 15, q = lane >> 4;
  const int b = blockIdx.x;
  const int T = a.T, B = a.B, ntiles = a.nblk;
  const int f0 = 16 * wave + 4 * q;
  const bool training = a.training != 0;
  const bool drop = training && a.keep_l0 && a.p_drop > 0.f;
  const float keep_scale = 1.0f / (1.0f - a.p_drop);
  const int64_t BH = (int64_t)B * H;
  const float unbias = (B > 1) ? (float)B / (float)(B - 1) : 1.0f;
  float run_m = 0.f, run_v = 0.f;
  // (w.bn_running_mean == NULL while training: the caller commits the running statistics itself, g2v_bn_running_update)
  if (training && b == 0 && tid < H && a.w.bn_running_mean) {
    run_m = a.w.bn_running_mean[tid];
    run_v = a.w.bn_running_var[tid];
  }
  WFrag<3, KSH> f_ih0, f_hh0, f_ih1, f_hh1;
  frag_load(f_hh0, a.p_hh0, wave, 4, lane);
  frag_load(f_hh1, a.p_hh1, wave, 4, lane);
  frag_load(f_ih0, a.p_ih0, wave, 4, lane);
  frag_load(f_ih1, a.p_ih1, wave, 4, lane);
  for (int e = tid; e < 9 * KSH * 64; e += 256)
    reinterpret_cast<float4*>(Pout)[e] = reinterpret_cast<const float4*>(a.p_out)[e];
  for (int e = tid; e < 4 * KSD * 64; e += 256) reinterpret_cast<float4*>(Ppre)[e] = reinterpret_cast<const float4*>(a.p_pre)[e];
  for (int e = tid; e < 192; e += 256) {
    Bs[B_IH0 + e] = a.w.b_ih0[e]; Bs[B_HH0 + e] = a.w.b_hh0[e];
    Bs[B_IH1 + e] = a.w.b_ih1[e]; Bs[B_HH1 + e] = a.w.b_hh1[e];
  }
  for (int e = tid; e < 144; e += 256) Bs[B_OUT + e] = e < D ? a.w.b_out[e] : 0.f;
  if (tid < H) {
    Bs[B_PRE + tid] = a.w.b_pre[tid];
    Bs[B_BNW + tid] = a.w.bn_w[tid];
    Bs[B_BNB + tid] = a.w.bn_b[tid];
    if (!training) {
      st[tid] = a.w.bn_running_mean[tid];
      st[H + tid] = bn_invstd_(a.w.bn_running_var[tid]);
    }
  }
  for (int e = tid; e < 16 * (LDD - D); e += 256) Xy[(e / (LDD - D)) * LDD + D + (e % (LDD - D))] = 0.f;
  for (int e = tid; e < 4 * 16 * LDH; e += 256) smem[L_XA + e] = 0.f;
  for (int e = tid; e < (R - 1) * 2 * 16 * LDH; e += 256) smem[L_END + e] = 0.f;
  lds_barrier();
  // initial state of every tile: h0_0, h1_0 = the quantised latent
  static_for<R>([&](auto rc) {
    constexpr int r = decltype(rc)::value;
    const int tb = b + r * nwg;
    if (tb < ntiles && (tid >> 4) < min(16, B - 16 * tb)) {      // (rows past B: the LDS rows stay zero)
      const int rr = tid >> 4, c = (tid & 15) * 4, b0 = 16 * tb;
      const float4 v0 = *reinterpret_cast<const float4*>(a.h_init + (int64_t)(b0 + rr) * H + c);
      const float4 v1 = *reinterpret_cast<const float4*>(a.h_init + BH + (int64_t)(b0 + rr) * H + c);
      *reinterpret_cast<float4*>(Xh0_of(r) + rr * LDH + c) = v0;
      *reinterpret_cast<float4*>(Xh1_of(r) + rr * LDH + c) = v1;
      if (a.sv.h0) *reinterpret_cast<float4*>(a.sv.h0 + (int64_t)(b0 + rr) * H + c) = v0;
      if (a.sv.h1) *reinterpret_cast<float4*>(a.sv.h1 + (int64_t)(b0 + rr) * H + c) = v1;
    }
  });
  lds_barrier();

  f32x4 u_acc[R];
#pragma unroll
  for (int r = 0; r < R; ++r) u_acc[r] = (f32x4){0.f, 0.f, 0.f, 0.f};

  for (int t = 0; t < T; ++t) {
    const bool has_next = t < T - 1;
    float s1a[4] = {0.f, 0.f, 0.f, 0.f}, s2a[4] = {0.f, 0.f, 0.f, 0.f};      // BatchNorm partial sums over this workgroup's tiles
    static_for<R>([&](auto rc) {
      constexpr int r = decltype(rc)::value;
      const int tb = b + r * nwg;
      if (tb >= ntiles) return;                       // (uniform per workgroup)
      const int b0 = 16 * tb;
      // ragged last tile (B % 16 != 0, B % 4 == 0): rows >= nrows are computed like the others from whatever their LDS rows hold
      // (finite values; an MFMA row never mixes with another row), read clamped addresses, store nothing and are left out of
      // the BatchNorm sums
      const int nrows = min(16, B - b0);
      const bool rowok = i < nrows;
      const int ic = rowok ? i : nrows - 1;
      const int n4v = (nrows * D) / 4;                // float4s of the tile's dense (nrows x D) run
      float* Xh0 = Xh0_of(r);
      float* Xh1 = Xh1_of(r);
      if (t > 0) {
        f32x4 gh0[3], gh1[3];
#pragma unroll
        for (int g = 0; g < 3; ++g) {
          gh0[g] = (f32x4){0.f, 0.f, 0.f, 0.f};
          gh1[g] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
        uint32_t kreq[3] = {0u, 0u, 0u};
        const bool fast_y = has_next && !(t < a.n_pre);
        if (fast_y && a.conditioned) {
          const uint32_t* kp4 = reinterpret_cast<const uint32_t*>(a.keep95 + ((int64_t)t * B + b0) * D);
#pragma unroll
          for (int j = 0; j < 3; ++j) {
            const int e4 = tid + 256 * j;
            kreq[j] = kp4[e4 < n4v ? e4 : 0];
          }
        }
        // the exchange of this step's BatchNorm sums is threaded through the FIRST tile's hidden-side products
        frag_mma_x1st(gh0, f_hh0, Xh0, LDH, lane);
        if (r == 0 && training) px_hop1(a.x, t & 1, (unsigned)t, nwg, b, red, tot, tid);
        frag_mma_x1st(gh1, f_hh1, Xh1, LDH, lane);
#pragma unroll
        for (int j = 0; j < 3; ++j) {
          const int e4 = tid + 256 * j;
          if (e4 < (16 * D) / 4) Kt[e4] = kreq[j];
        }
        if (r == 0 && training) {
          if (px_two_hops(nwg)) px_hop2(a.x, t & 1, (unsigned)t, nwg, b, red, tot, tid);
          if (tid < H) {
            const float s1 = tot[tid], s2 = tot[H + tid];
            const float mv = s1 / (float)B;
            const float var = fmaxf(s2 / (float)B - mv * mv, 0.f);
            const float mean = mv + Bs[B_PRE + tid];
            st[tid] = mean;
            st[H + tid] = bn_invstd_(var);
            if (b == 0 && a.sv.bn_stats) {
              a.sv.bn_stats[(int64_t)(t - 1) * 2 * H + tid] = mean;
              a.sv.bn_stats[(int64_t)(t - 1) * 2 * H + H + tid] = var;
            }
            run_m = 0.9f * run_m + 0.1f * mean;
            run_v = 0.9f * run_v + 0.1f * (var * unbias);
          }
          lds_barrier();
        }
        // a_t = ReLU(BN(u_t)) from the tile's register-resident u
        {
          const float4 m4 = *reinterpret_cast<const float4*>(st + f0), i4 = *reinterpret_cast<const float4*>(st + H + f0);
          const float4 g4 = *reinterpret_cast<const float4*>(Bs + B_BNW + f0), b4 = *reinterpret_cast<const float4*>(Bs + B_BNB + f0);
          const float4 p4 = *reinterpret_cast<const float4*>(Bs + B_PRE + f0);
          float4 a4;
          a4.x = fmaxf(((u_acc[r][0] + p4.x) - m4.x) * i4.x * g4.x + b4.x, 0.f);
          a4.y = fmaxf(((u_acc[r][1] + p4.y) - m4.y) * i4.y * g4.y + b4.y, 0.f);
          a4.z = fmaxf(((u_acc[r][2] + p4.z) - m4.z) * i4.z * g4.z + b4.z, 0.f);
          a4.w = fmaxf(((u_acc[r][3] + p4.w) - m4.w) * i4.w * g4.w + b4.w, 0.f);
          *reinterpret_cast<float4*>(Xa + i * LDH + f0) = a4;
          if (a.sv.a && rowok) *reinterpret_cast<float4*>(a.sv.a + ((int64_t)(t - 1) * B + b0 + i) * H + f0) = a4;
        }
        lds_barrier();
        {
          f32x4 ai[3];
#pragma unroll
          for (int g = 0; g < 3; ++g) ai[g] = (f32x4){0.f, 0.f, 0.f, 0.f};
          if (drop) {      // (two code paths: see the one-tile kernel)
            const uint32_t kp = *reinterpret_cast<const uint32_t*>(a.keep_l0 + ((int64_t)(t - 1) * B + b0 + ic) * H + f0);
            frag_mma_x1st(ai, f_ih0, Xa, LDH, lane);
            cell_epilogue(ai, gh0, Bs + B_IH0, Bs + B_HH0, kp, true, keep_scale, Xh0, Xx1,
                          (a.sv.h0 && rowok) ? a.sv.h0 + ((int64_t)t * B + b0) * H : nullptr,
                          (a.sv.gates0 && rowok) ? a.sv.gates0 + ((int64_t)(t - 1) * B + b0) * 4 * H : nullptr,
                          (a.sv.x1 && rowok) ? a.sv.x1 + ((int64_t)(t - 1) * B + b0) * H : nullptr, i, f0);
          } else {
            frag_mma_x1st(ai, f_ih0, Xa, LDH, lane);
            cell_epilogue(ai, gh0, Bs + B_IH0, Bs + B_HH0, 0x01010101u, false, 1.0f, Xh0, Xx1,
                          (a.sv.h0 && rowok) ? a.sv.h0 + ((int64_t)t * B + b0) * H : nullptr,
                          (a.sv.gates0 && rowok) ? a.sv.gates0 + ((int64_t)(t - 1) * B + b0) * 4 * H : nullptr, nullptr, i, f0);
          }
        }
        lds_barrier();
        {
          f32x4 ai[3];
#pragma unroll
          for (int g = 0; g < 3; ++g) ai[g] = (f32x4){0.f, 0.f, 0.f, 0.f};
          frag_mma_x1st(ai, f_ih1, Xx1, LDH, lane);
          cell_epilogue(ai, gh1, Bs + B_IH1, Bs + B_HH1, 0x01010101u, false, 1.0f, Xh1, nullptr,
                        (a.sv.h1 && rowok) ? a.sv.h1 + ((int64_t)t * B + b0) * H : nullptr,
                        (a.sv.gates1 && rowok) ? a.sv.gates1 + ((int64_t)(t - 1) * B + b0) * 4 * H : nullptr, nullptr, i, f0);
        }
        lds_barrier();
        {      // y_t = out_layer(h1_t) -> dense tile (see the one-tile kernel for the shape of this epilogue)
          f32x4 acc[3];
#pragma unroll
          for (int j = 0; j < 3; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
          if (wave == 0) {
            lds_frag_mma<3, KSH>(acc, Pout, wave, 4, Xh1, LDH, lane);
          } else {
            f32x4 a2[2] = {acc[0], acc[1]};
            lds_frag_mma<2, KSH>(a2, Pout, wave, 4, Xh1, LDH, lane);
            acc[0] = a2[0]; acc[1] = a2[1];
          }
          const uint8_t* Kb = reinterpret_cast<const uint8_t*>(Kt);
          float bo[3][4], yv[3][4], xv[3][4];
          uint32_t kb[3][4];
#pragma unroll
          for (int j = 0; j < 3; ++j) {
            const int d0 = 16 * (wave + 4 * j) + 4 * q, d0c = d0 < Dp ? d0 : 4 * q;
            const float4 b4 = *reinterpret_cast<const float4*>(Bs + B_OUT + d0c);
            bo[j][0] = b4.x; bo[j][1] = b4.y; bo[j][2] = b4.z; bo[j][3] = b4.w;
#pragma unroll
            for (int e = 0; e < 4; ++e) kb[j][e] = Kb[i * D + d0c + e];
          }
#pragma unroll
          for (int j = 0; j < 3; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              yv[j][e] = acc[j][e] + bo[j][e];
              xv[j][e] = kb[j][e] ? yv[j][e] * 20.0f : 0.f;
            }
#pragma unroll
          for (int j = 0; j < 3; ++j) {
            const int d0 = 16 * (wave + 4 * j) + 4 * q;
#pragma unroll
            for (int e = 0; e < 4; ++e)
              if (d0 + e < D) {
                Yt[i * D + d0 + e] = yv[j][e];
                if (fast_y) {
                  Xt[i * D + d0 + e] = xv[j][e];
                  Xy[i * LDD + d0 + e] = xv[j][e];
                }
              }
          }
        }
        lds_barrier();
      }
      // ---- y_t out, next decoder input xin_{t+1} = Dropout(0.95)(y_t | target_t) ------------------------------------------
      const bool fast_dense = t > 0 && (!has_next || !(t < a.n_pre));
      const int64_t tile = ((int64_t)t * B + b0) * D;
      if (fast_dense) {
        for (int e4 = tid; e4 < n4v; e4 += 256) {
          *reinterpret_cast<float4*>(a.sv.y + tile + 4 * e4) = reinterpret_cast<const float4*>(Yt)[e4];
          if (has_next && a.sv.xin) *reinterpret_cast<float4*>(a.sv.xin + tile + 4 * e4) = reinterpret_cast<const float4*>(Xt)[e4];
        }
      } else {
        const bool teacher = has_next && (t < a.n_pre);
        const uint32_t* kp4 = reinterpret_cast<const uint32_t*>(a.keep95 + tile);
        for (int e4 = tid; e4 < n4v; e4 += 256) {
          const int e = 4 * e4;
          float yv[4], sv_[4], tv[4];
          uint32_t kraw;
          {   // (as in the one-tile kernel: the group's target frame t and keep flags in one round trip)
            const float* tp[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const int rr = (e + j) / D, c = (e + j) - rr * D;
              tp[j] = a.target + ((int64_t)(b0 + rr) * T + t) * D + c;
            }
            const bool needk = has_next && a.conditioned;
            ldg_sync5(tp[0], tp[1], tp[2], tp[3], needk ? reinterpret_cast<const float*>(kp4 + e4) : tp[0], tv, kraw);
            if (!needk) kraw = 0u;
          }
          if (t == 0) {
#pragma unroll
            for (int j = 0; j < 4; ++j) yv[j] = tv[j];
          } else {
            const float4 y4 = reinterpret_cast<const float4*>(Yt)[e4];
            yv[0] = y4.x; yv[1] = y4.y; yv[2] = y4.z; yv[3] = y4.w;
          }
          *reinterpret_cast<float4*>(a.sv.y + tile + e) = make_float4(yv[0], yv[1], yv[2], yv[3]);
          if (!has_next) continue;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            sv_[j] = (teacher && t > 0) ? tv[j] : yv[j];
          }
          const uint32_t k4 = kraw;
          float xv[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) xv[j] = ((k4 >> (8 * j)) & 0xffu) ? sv_[j] * 20.0f : 0.f;
          if (a.sv.xin) *reinterpret_cast<float4*>(a.sv.xin + tile + e) = make_float4(xv[0], xv[1], xv[2], xv[3]);
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int rr = (e + j) / D, c = (e + j) - rr * D;
            Xy[rr * LDD + c] = xv[j];
          }
        }
        lds_barrier();
      }
      if (!has_next) return;
      // ---- u_{t+1} = pre_linear.0(xin_{t+1}) of this tile; its partial sums join the workgroup's ---------------------------
      u_acc[r] = lds_frag_mma_2chain<KSD>(Ppre, wave, Xy, LDD, lane);
      const float4 p4 = *reinterpret_cast<const float4*>(Bs + B_PRE + f0);
      if (a.sv.u && rowok)
        *reinterpret_cast<float4*>(a.sv.u + ((int64_t)t * B + b0 + i) * H + f0) =
            make_float4(u_acc[r][0] + p4.x, u_acc[r][1] + p4.y, u_acc[r][2] + p4.z, u_acc[r][3] + p4.w);
      if (training) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float uv = rowok ? u_acc[r][e] : 0.f;
          s1a[e] += reduce16(uv);
          s2a[e] += reduce16(uv * uv);
        }
      }
      // (the dense y / xin copies above read Yt / Xt and the product read Xy: the next tile rewrites them three barriers on;
      //  Kt is rewritten behind its hidden-side products -- its last reader, this tile's out-layer epilogue, sits behind a barrier)
    });
    if (!has_next) break;
    if (training && i == 0) {
      const int par = (t + 1) & 1;
      __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc(
          a.x.rec1 + ((size_t)par * PX_MAX_NBLK + b) * PX_COLS, 0, PX_COLS * 8, 0x00020000);
      const unsigned tag = (unsigned)(t + 1);
      px_publish2(rr, (unsigned)f0, s1a[0], s1a[1], tag);
      px_publish2(rr, (unsigned)f0 + 2, s1a[2], s1a[3], tag);
      px_publish2(rr, (unsigned)(H + f0), s2a[0], s2a[1], tag);
      px_publish2(rr, (unsigned)(H + f0) + 2, s2a[2], s2a[3], tag);
    }
  }
  if (training && b == 0 && tid < H && a.w.bn_running_mean &&
      __hip_atomic_load(&g2v_persist_fault, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) {
    a.w.bn_running_mean[tid] = run_m;
    a.w.bn_running_var[tid] = run_v;
  }
}

// =====================================================================================================================
// custom_loss (train_eval/train_seq2seq.py:40-88) as a CHASER of the forward rollout: a second, light kernel (one 256-thread
// workgroup per row tile, <= 128 registers per lane, 64 bytes of LDS) that is co-resident with dec_persist_fwd_kernel -- which
// runs one wave per SIMD and leaves ~140 registers per lane and ~29 KB of LDS of every CU unused -- and consumes each y_t tile as
// the rollout hands it over (progress word per workgroup, dec_persist.hpp PX_FLAG_*; sc1 payload, sc1 polls and loads).  It
// accumulates the four loss sums and the per-column sum of squares as the steps arrive and writes, per element, ONE code byte
// (common.hpp: the three signs of the |.| terms + the Dropout(0.95) keep flag in bit 6) and, after the last step, the column
// coefficient c3 / ||y[:,b,d]||.  dec_persist_bwd_kernel then forms dLoss/dy_t = loss_grad(table[code], cn, y_t) where it used to
// read dy_t: the separate custom_loss launch (62 + 6 us alone between the two rollouts, 225 MB of HBM traffic) is gone, and the
// loss costs the chain ~3 us behind the forward instead.  Same device functions as misc.hip's kernels: dy is bitwise theirs.
//   Round 3 had the same arithmetic INSIDE the forward kernel: +55-60 us, because a one-wave-per-SIMD kernel on a serial chain
//   pays every VALU instruction at its issue time.  On other wave slots of the same SIMDs it is free.
// Residency: the chaser depends on the rollout, never the other way round; it must be dispatched BEHIND the rollout (the engine
// launches it behind the quantiser's statistics kernels on the side stream) -- a CU that already holds two chaser workgroups has
// no room for a rollout workgroup.  Every wait is bounded and ends in the fault latch, like the exchange's.
// =====================================================================================================================
struct LossChaseArgs {
  const float* target;      // (B,T,D)
  const float* y;           // (T,B,D): the forward's output, stored write-through
  const uint8_t* keep95;    // (T-1,B,D)
  uint8_t* code;            // (T,B,D)
  float* coef;              // (B,D)
  float* partial;           // (nblk,4): sum |y - tgt|, sum |y_t - y_{t-1}|, sum of column norms, sum (y - tgt)^2
  const unsigned* yflag;
  int T, B;
  float lc3;                // w_var / (T B D)
};

__global__ __launch_bounds__(256, 4) void loss_chase_kernel(LossChaseArgs a) {
  // Target rows of one step, fetched as 16-byte ALIGNED spans and staged as a dense tile: row r of the tile sits in the (B,T,D)
  // target at a float offset that is only 4-byte aligned (D = 135), so each row's 540 bytes are fetched as the <= 35 aligned
  // 16-byte pieces that cover them and written to LDS shifted back by the row's misalignment (0..3 floats): Lt[buf] is then the
  // step's target tile in the same tile-linear order as y, read back as three aligned 16-byte vectors per thread.  Why through
  // LDS at all: read element-wise in the tile-linear mapping (12 dwords per thread, lanes 16 bytes apart, two or three rows per
  // wave-instruction) the target was 48 wave-instructions and ~480 line requests per step and CU; now 12 and ~110.
  // What the chaser costs the co-resident rollout (measured by switching its parts off in turn, 300-step runs on one box): its
  // arithmetic nothing, its polls nothing, its write-through y loads ~5 us, the target's 75 MB from HBM 35-45 us with default-policy
  // loads in either mapping, ~12 us when the same instructions hit in cache -- the rollout writes 17 MB of saved tensors per step
  // at ~7 B / clock / CU in its cell phases, about what a CU can stream, so every byte of HBM traffic beside it is paid for; with
  // non-temporal loads ~30 us.
  __shared__ __attribute__((aligned(16))) float Lt[2][16 * D + 16];
  __shared__ float red[16];
  constexpr int NQ = (16 * D) / 4;                   // 540 float4 per 16 x D tile: threads 0..27 own a third one
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = blockIdx.x, b0 = b * 16;
  const int T = a.T, B = a.B;
  const bool has3 = tid + 512 < NQ;
  const int q4[3] = {tid, tid + 256, has3 ? tid + 512 : 0};      // tile-linear element e = 4 q + r <-> (row e / D, column e % D)
  // staging role: thread (row tid >> 4, piece column tid & 15) fetches pieces c, c + 16, c + 32 (< 35) of its row
  const int srow = tid >> 4, scol = tid & 15;
  const int64_t srow_off = (int64_t)(b0 + srow) * T * D;          // float offset of the row's frame 0
  const int64_t tgt_floats = (int64_t)B * T * D;
  float4 stg[3];
  auto stage_request = [&](int t) {
    const int64_t base = (srow_off + (int64_t)t * D) & ~(int64_t)3;      // aligned span start
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const int pc = scol + 16 * k;
      const int64_t o = base + 4 * pc;
      const bool ok = pc < 35 && o + 3 < tgt_floats;                 // (the last row's span may end past the tensor)
      // non-temporal: each byte is read once; with default-policy loads the same burst cost the co-resident rollout 13 us more
      const f32x4 v = ok ? __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(a.target + o)) : (f32x4){0.f, 0.f, 0.f, 0.f};
      stg[k] = make_float4(v[0], v[1], v[2], v[3]);
    }
  };
  auto stage_commit = [&](int buf, int t) {
    const int sh = (int)((srow_off + (int64_t)t * D) & 3);            // the row's misalignment at this step
    float* dst = &Lt[buf][srow * D] - sh;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const int c0 = 4 * (scol + 16 * k) - sh;                        // column of the piece's first float
      const float v[4] = {stg[k].x, stg[k].y, stg[k].z, stg[k].w};
#pragma unroll
      for (int i2 = 0; i2 < 4; ++i2)
        if (c0 + i2 >= 0 && c0 + i2 < D) dst[4 * (scol + 16 * k) + i2] = v[i2];
    }
  };
  const unsigned* flag = a.yflag + (size_t)b * PX_FLAG_STRIDE;
  float yprev[3][4], ss[3][4];
  uint32_t pend[3];                                  // step t-1's code bytes, sign(y_t - y_{t-1}) still missing
  float l1 = 0.f, cont = 0.f, sq = 0.f, nrm_sum = 0.f;
  auto keep_bits = [&](int t, int j) -> uint32_t {   // bit 6 of each byte: the element's Dropout(0.95) keep flag of step t
    if (t >= T - 1) return 0u;                       // (row T-1 does not exist: nothing is fed back from the last step)
    const uint32_t k = reinterpret_cast<const uint32_t*>(a.keep95 + ((int64_t)t * B + b0) * D)[q4[j]];
    return ((k & 0xffu) ? 0x40u : 0u) | ((k & 0xff00u) ? 0x4000u : 0u) | ((k & 0xff0000u) ? 0x400000u : 0u) |
           ((k & 0xff000000u) ? 0x40000000u : 0u);
  };
  // step 0: y_0 = target frame 0 (:1039-1040): sign(y - tgt) = 0, no y_{-1}
  stage_request(0);
  stage_commit(0, 0);
  if (T > 1) stage_request(1);
  __syncthreads();
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    const float4 t4 = *reinterpret_cast<const float4*>(&Lt[0][4 * q4[j]]);
    const float tv[4] = {t4.x, t4.y, t4.z, t4.w};
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      yprev[j][r] = tv[r];
      ss[j][r] = fmaf(tv[r], tv[r], 0.f);
    }
    pend[j] = 0x05050505u | keep_bits(0, j);         // (0 + 1) | (0 + 1) << 2
  }
  for (int t = 1; t < T; ++t) {
    const int buf = t & 1;
    // what does not depend on the rollout first: the target rows of step t (requested an iteration ago) and its keep flags
    stage_commit(buf, t);                            // (its previous readers passed the barrier of the iteration before)
    uint32_t kb[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) kb[j] = keep_bits(t, j);
    if (tid == 0) {
      unsigned spins = 0;
      while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)t) {
        __builtin_amdgcn_s_sleep(8);
        ++spins;
        if (spins > 4000000u || ((spins & 1023u) == 0 &&
                                 __hip_atomic_load(&g2v_persist_fault, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u)) {
          __hip_atomic_store(&g2v_persist_fault, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // (2: the chaser's wait) the rollout is not running
          break;
        }
      }
    }
    __syncthreads();                                 // the flag has been seen; the staged tile is complete
    __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.y) + ((int64_t)t * B + b0) * D, 0,
                                                                  16 * D * 4, 0x00020000);
    u32x4 yv4[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) yv4[j] = px_ld(yr, (unsigned)q4[j] * 16u);      // sc1: past this CU's L1, never a stale line
    if (t + 1 < T) stage_request(t + 1);             // all of this step's vector-memory loads leave together, right at the flag
    uint32_t done[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const bool live = j < 2 || has3;
      uint32_t nb = 0u, cur = 0u;
      const float4 t4 = *reinterpret_cast<const float4*>(&Lt[buf][4 * q4[j]]);
      const float tv[4] = {t4.x, t4.y, t4.z, t4.w};
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const unsigned bits = yv4[j][r];
        const float v = __uint_as_float(bits);
        const float dlt = v - tv[r], stp = v - yprev[j][r];
        if (live) {
          l1 += fabsf(dlt);
          sq += dlt * dlt;
          cont += fabsf(stp);
        }
        ss[j][r] = fmaf(v, v, ss[j][r]);
        const uint32_t sa = (uint32_t)loss_sign_code(stp);
        nb |= (sa << 4) << (8 * r);                  // sign(y_t - y_{t-1}) completes step t-1's byte ...
        cur |= ((uint32_t)loss_sign_code(dlt) | (sa << 2)) << (8 * r);      // ... and opens step t's
        yprev[j][r] = v;
      }
      done[j] = pend[j] | nb;
      pend[j] = cur | kb[j];
    }
    uint32_t* cp = reinterpret_cast<uint32_t*>(a.code + ((int64_t)(t - 1) * B + b0) * D);
    cp[q4[0]] = done[0];
    cp[q4[1]] = done[1];
    if (has3) cp[q4[2]] = done[2];
  }
  {      // the last step has no y_{t+1}: digit 1
    uint32_t* cp = reinterpret_cast<uint32_t*>(a.code + ((int64_t)(T - 1) * B + b0) * D);
    cp[q4[0]] = pend[0] | 0x10101010u;
    cp[q4[1]] = pend[1] | 0x10101010u;
    if (has3) cp[q4[2]] = pend[2] | 0x10101010u;
  }
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    float cn[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float nrm;
      cn[r] = loss_col_coef(a.lc3, ss[j][r], nrm);
      if (j < 2 || has3) nrm_sum += nrm;
    }
    if (j < 2 || has3) reinterpret_cast<float4*>(a.coef + (int64_t)b0 * D)[q4[j]] = make_float4(cn[0], cn[1], cn[2], cn[3]);
  }
  const float s4[4] = {wave_sum(l1), wave_sum(cont), wave_sum(nrm_sum), wave_sum(sq)};
  if (lane == 0) {
#pragma unroll
    for (int k = 0; k < 4; ++k) red[wave * 4 + k] = s4[k];
  }
  __syncthreads();
  if (tid < 4) a.partial[(int64_t)b * 4 + tid] = (red[tid] + red[4 + tid]) + (red[8 + tid] + red[12 + tid]);
}

}  // namespace g2v

using namespace g2v;

// The fault latch of the persistent kernels (dec_persist.hpp): 1 if a bounded wait of the grid-wide exchange ever ran out since the
// last clear.  SYNCHRONOUS (a device-to-host copy of one word): call it where the host synchronises anyway.
extern "C" int g2v_dec_rollout_persist_fault(int clear) {
  if (clear < 0) {      // test hook: LATCH a fault of value -clear (what a bounded wait running out does on the device)
    const unsigned inj = (unsigned)(-clear);
    return hipMemcpyToSymbol(HIP_SYMBOL(g2v_persist_fault), &inj, sizeof(inj)) == hipSuccess ? (int)inj : -1;
  }
  unsigned v = 0;
  if (hipMemcpyFromSymbol(&v, HIP_SYMBOL(g2v_persist_fault), sizeof(v)) != hipSuccess) return -1;
  if (v && clear) {
    const unsigned zero = 0;
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g2v_persist_fault), &zero, sizeof(zero));
  }
  return (int)v;
}

// Data parallelism: the latch is per process, but a rank whose rollout faulted has already fed garbage gradients into the
// all-reduce -- every rank must skip the step.  to_flag: flag[0] = 1 if this rank's latch is set else 0 (in front of the SUM
// all-reduce, in a slot of the communication buffer); from_flag: a non-zero sum latches THIS rank too (value 3 where it was clear),
// so its commit kernels -- EMA update, BatchNorm statistics, clip + Adam -- leave the state alone like the faulting rank's.
__global__ void fault_flag_kernel(float* __restrict__ flag, int from_flag) {
  if (!from_flag) flag[0] = __hip_atomic_load(&g2v_persist_fault, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u ? 1.0f : 0.0f;
  else if (flag[0] != 0.0f) {
    unsigned expected = 0u;
    (void)__hip_atomic_compare_exchange_strong(&g2v_persist_fault, &expected, 3u, __ATOMIC_RELAXED, __ATOMIC_RELAXED,
                                               __HIP_MEMORY_SCOPE_AGENT);
  }
}
extern "C" int g2v_dec_rollout_fault_flag(float* flag, int from_flag, g2v_stream_t stream) {
  G2V_REQUIRE(flag, "null pointer");
  hipLaunchKernelGGL(fault_flag_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, flag, from_flag);
  G2V_CHECK_LAUNCH();
  return G2V_OK;
}

// (the "this exchange region is already clear" notes of g2v_cluster_exchange_preclear live in the caller's context: misc.hip)
const unsigned* g2v_internal_persist_fault_ptr() {
  static const unsigned* p = [] {
    void* q = nullptr;
    return hipGetSymbolAddress(&q, HIP_SYMBOL(g2v_persist_fault)) == hipSuccess ? (const unsigned*)q : (const unsigned*)nullptr;
  }();
  return p;
}

// Residency check of a persistent kernel (once per kernel): at least one workgroup of its shape must fit a CU; together with
// nblk <= CU count (dec_rollout.hip) this is what a plain launch can know.  It cannot see CU masks or other tenants of the
// device: that is what the fault latch is for.
static bool persist_fits(const void* fn, size_t lds) {
  int n = 0;
  return hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, fn, 256, lds) == hipSuccess && n >= 1;
}

// Host side: called by g2v_dec_rollout_fwd (dec_rollout.hip) when the persistent path applies.  `packed` points at the
// six fragment-major matrices in the order pre, ih0, hh0, ih1, hh1, out; `xbase` at PX_BYTES of exchange state.
int dec_persist_fwd_launch(const float* target, const float* h_init, const g2v_dec_weights* w, const g2v_dec_saved* s,
                           const uint8_t* keep95, const uint8_t* keep_l0, float p_drop, int n_pre, int conditioned,
                           int training, int T, int B, const float* p_pre, const float* p_ih0, const float* p_hh0,
                           const float* p_ih1, const float* p_hh1, const float* p_out, void* xbase, hipStream_t st,
                           bool clear, int tiles_per_wg) {
  DecPersistArgs a;
  a.target = target; a.h_init = h_init; a.w = *w;
  a.p_pre = p_pre; a.p_ih0 = p_ih0; a.p_hh0 = p_hh0; a.p_ih1 = p_ih1; a.p_hh1 = p_hh1; a.p_out = p_out;
  a.sv = *s; a.keep95 = keep95; a.keep_l0 = keep_l0;
  a.x = persist_x_at(xbase);
  a.T = T; a.B = B; a.nblk = (B + 15) / 16; a.n_pre = n_pre; a.conditioned = conditioned; a.training = training;
  a.p_drop = p_drop;
  if (tiles_per_wg > 1 || (B & 15)) {      // (a ragged last tile is the multi-tile kernel's job too, also with one tile each)
    // more row tiles than CUs: R tiles per workgroup (dec_persist_fwd_mt_kernel); no chaser there
    const int R = tiles_per_wg, nwg = (a.nblk + R - 1) / R;
    const size_t lds = dec_persist_fwd_lds_bytes(R);
    const void* fn = R == 1 ? (const void*)dec_persist_fwd_mt_kernel<1>
                            : (R == 2 ? (const void*)dec_persist_fwd_mt_kernel<2> : (const void*)dec_persist_fwd_mt_kernel<3>);
    static bool mt_set[3] = {false, false, false};
    if (R < 1 || R > 3 || s->loss_code) {
      set_error("dec_persist_fwd: %d tiles per workgroup / a chased rollout is not offered", R);
      return G2V_ERR_ARG;
    }
    if (!mt_set[R - 1]) {
      if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess || !persist_fits(fn, lds)) {
        set_error("dec_persist_fwd: the %d-tile kernel does not fit a CU (%zu bytes of LDS)", R, lds);
        return G2V_ERR_LAUNCH;
      }
      mt_set[R - 1] = true;
    }
    if (training && clear) (void)hipMemsetAsync(xbase, 0, PX_BYTES, st);
    if (R == 1) hipLaunchKernelGGL(dec_persist_fwd_mt_kernel<1>, dim3(nwg), dim3(256), lds, st, a, nwg);
    else if (R == 2) hipLaunchKernelGGL(dec_persist_fwd_mt_kernel<2>, dim3(nwg), dim3(256), lds, st, a, nwg);
    else hipLaunchKernelGGL(dec_persist_fwd_mt_kernel<3>, dim3(nwg), dim3(256), lds, st, a, nwg);
    if (hipGetLastError() != hipSuccess) {
      set_error("dec_persist_fwd: launch failed");
      return G2V_ERR_LAUNCH;
    }
    return G2V_OK;
  }
  const size_t lds = dec_persist_fwd_lds_bytes(1);
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute((const void*)dec_persist_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
      set_error("dec_persist_fwd: cannot reserve %zu bytes of LDS", lds);
      return G2V_ERR_LAUNCH;
    }
    if (!persist_fits((const void*)dec_persist_fwd_kernel, lds)) {
      set_error("dec_persist_fwd: the kernel does not fit a CU (occupancy query)");
      return G2V_ERR_LAUNCH;
    }
    attr_set = true;
  }
  // every polled word, every call (tags count steps 1..T); !clear: g2v_dec_rollout_prepare has done it for this call
  if (training && clear) (void)hipMemsetAsync(xbase, 0, PX_BYTES, st);
  hipLaunchKernelGGL(dec_persist_fwd_kernel, dim3(a.nblk), dim3(256), lds, st, a);
  if (hipGetLastError() != hipSuccess) {
    set_error("dec_persist_fwd: launch failed");
    return G2V_ERR_LAUNCH;
  }
  return G2V_OK;
}

// Host side of the chaser: called by g2v_custom_loss_chase (dec_rollout.hip).  `xbase`: the exchange state of the forward
// workspace this call's forward rollout runs on (its progress words).
int dec_persist_loss_chase_launch(const float* target, const g2v_dec_saved* s, const uint8_t* keep95, int T, int B, void* xbase,
                                  hipStream_t st) {
  LossChaseArgs a;
  a.target = target; a.y = s->y; a.keep95 = keep95;
  a.code = s->loss_code; a.coef = s->loss_coef; a.partial = s->loss_partial;
  a.yflag = persist_x_at(xbase).yflag;
  a.T = T; a.B = B;
  a.lc3 = s->loss_w[2] / ((float)T * (float)B * (float)D);      // as g2v_custom_loss_fwd_bwd forms it
  hipLaunchKernelGGL(loss_chase_kernel, dim3(B / 16), dim3(256), 0, st, a);
  if (hipGetLastError() != hipSuccess) {
    set_error("custom_loss chaser: launch failed");
    return G2V_ERR_LAUNCH;
  }
  return G2V_OK;
}

// =====================================================================================================================
// backward (BPTT of the rollout), one persistent launch.  Per workgroup (16 rows; wave w owns hidden features 16w..16w+15):
//   registers : W_ih0^T, W_hh0^T, W_ih1^T, W_hh1^T fragments (contraction over the 3H gate axis: 4 x 48 VGPRs), the two
//               hidden-state gradient carries, dbn_t / xhat_t of the step whose BatchNorm-backward sums are in flight
//   LDS       : W_pre^T (D rows) and W_out^T (K = D) fragments, du, dy, the gate-gradient tile [g_r | g_z | g_n | g_hn]
//   per step t: [exchange of the BatchNorm-backward sums of step t+1] -> du_{t+1} -> feedback dxin = du W_pre through
//               Dropout(0.95) into dy_t -> out_layer^T -> cell 1 backward -> cell 0 backward -> ReLU backward ->
//               partial sums (sum dbn, sum dbn * xhat) over the block's rows -> publish
// Produces exactly the arrays the per-step kernels produce (du, dy total, dgi/dgh of both cells, d_bn_w/b, dh_init);
// the dbn scratch array is not needed (the values stay in registers).
// =====================================================================================================================
int g2v_internal_slab_reduce4(const float* const* slab_w, float* const* out_w, const float* const* slab_b, float* const* out_b,
                              int nprob, int64_t n, int64_t nb, int nsplit, hipStream_t st);      // linear.hip

namespace g2v {
namespace {
constexpr int LDG = 4 * H + 4;                             // gate-gradient tile row stride
constexpr int KSG = 12;                                    // k-steps over 3H
constexpr int R_XDU = 0, R_XDY = R_XDU + 16 * LDH, R_G = R_XDY + 16 * LDD, R_DT = R_G + 16 * LDG, R_KT = R_DT + 2176;
constexpr int R_PPRET = R_KT + 544;                        // packed W_pre^T: 12 tiles (d) x 4 k-steps (f)
constexpr int R_POUTT = R_PPRET + OUT_TILES * KSH * 256;   // packed W_out^T: 4 tiles (f) x 9 k-steps (d)
constexpr int R_BNW = R_POUTT + 4 * KSD * 256;
constexpr int R_RED = R_BNW + 64;
constexpr int R_TOT = R_RED + 16 * 128;
constexpr int R_END = R_TOT + 128;
// fused weight gradient (FW): the cell-1 gate-gradient tile keeps its own LDS image (the cell-0 one stays in R_G), so that it
// is still there when its weight-gradient MFMAs run in the shadow of the NEXT iteration's exchange; plus one 16 x 16
// transpose scratch per wave (an operand's rows from accumulator layout to B-fragment layout)
constexpr int R_G1 = R_END, TRW = 16 * 17, R_TR = R_G1 + 16 * LDG, R_END_FW = R_TR + 4 * TRW;
// loss fold: the column coefficients of the workgroup's tile (16 * D floats) and the 64 values of loss_grad_const live in the
// three padding tiles at the end of the packed W_pre^T image (tiles 9..11: zero, never multiplied, never copied in)
constexpr int R_LCN = R_PPRET + 9 * KSH * 256, R_LTAB = R_LCN + 2176;
static_assert(R_LTAB + 64 <= R_POUTT, "the loss-fold tiles fit the padding of the packed W_pre^T image");
static_assert(R_END_FW * 4 <= 160 * 1024, "LDS budget of the fused-weight-gradient rollout backward");
}  // namespace

size_t dec_persist_bwd_lds_bytes(bool fw) { return (size_t)(fw ? R_END_FW : R_END) * sizeof(float); }
size_t dec_persist_bwd_wgrad_slab_floats() { return (size_t)(3 * H * H + 3 * H); }      // per workgroup: dW_hh1 192 x 64, db_hh1 192

struct DecPersistBwdArgs {
  g2v_dec_weights w;
  const float* p_pre_t; const float* p_out_t; const float* p_ih0_t; const float* p_hh0_t; const float* p_ih1_t; const float* p_hh1_t;
  g2v_dec_saved sv;
  g2v_dec_grads gr;
  const uint8_t* keep95; const uint8_t* keep_l0;
  PersistX x;
  int T, B, nblk, n_pre, conditioned;
  float p_drop;
  float* wslab;      // FW: [nblk][3H x H] partial dW_hh1, then [nblk][3H] partial db_hh1
  float lc1, lc2, lc3, linv_n;      // loss fold (sv.loss_code set): w_l1, w_cont, w_var over T B D, and 1 / (T B D)
};

// one 16-feature output tile, contraction over the gate axis of the merged tile [g_r | g_z | g_n | g_hn]:
// HH == false: columns 0..191 (g_r, g_z, g_n);  HH == true: g_r, g_z, g_hn (k-steps 8..11 read columns 192..255)
template <bool HH>
__device__ __forceinline__ void gate_frag_mma(f32x4& acc, const WFrag<1, KSG>& f, const float* G, int lane) {
  const float* xrow = G + (lane & 15) * LDG + 4 * (lane >> 4);
#pragma unroll
  for (int s = 0; s < KSG; ++s) {
    const int col = 16 * s + ((HH && s >= 8) ? 64 : 0);
    const float4 xb = *reinterpret_cast<const float4*>(xrow + col);
    acc = mfma16(f.w[0][s].x, xb.x, acc);
    acc = mfma16(f.w[0][s].y, xb.y, acc);
    acc = mfma16(f.w[0][s].z, xb.z, acc);
    acc = mfma16(f.w[0][s].w, xb.w, acc);
  }
}

// Both products of a cell's gate tile in one pass: a_hh += W_hh^T-tile x [g_r | g_z | g_hn], a_ih += W_ih^T-tile x [g_r | g_z | g_n].
// Each accumulator is a 48-long dependent MFMA chain; issued one after the other the matrix pipe waits out the
// result latency between every pair (measured: 42 cycles per MFMA instead of 32).  Alternating the two chains fills those
// slots, the summation order of each chain is unchanged.  The operand fragment of the next k-step is requested before the
// MFMAs of the current one.
__device__ __forceinline__ void gate_frag_mma2(f32x4& a_hh, const WFrag<1, KSG>& f_hh, f32x4& a_ih, const WFrag<1, KSG>& f_ih,
                                               const float* G, int lane) {
  const float* xrow = G + (lane & 15) * LDG + 4 * (lane >> 4);
  float4 xn = *reinterpret_cast<const float4*>(xrow);
#pragma unroll
  for (int s = 0; s < KSG; ++s) {
    const float4 xi = xn;
    float4 xh = xi;
    if (s >= 8) xh = *reinterpret_cast<const float4*>(xrow + 16 * s + 64);
    if (s + 1 < KSG) xn = *reinterpret_cast<const float4*>(xrow + 16 * (s + 1));
    a_hh = mfma16(f_hh.w[0][s].x, xh.x, a_hh);
    a_ih = mfma16(f_ih.w[0][s].x, xi.x, a_ih);
    a_hh = mfma16(f_hh.w[0][s].y, xh.y, a_hh);
    a_ih = mfma16(f_ih.w[0][s].y, xi.y, a_ih);
    a_hh = mfma16(f_hh.w[0][s].z, xh.z, a_hh);
    a_ih = mfma16(f_ih.w[0][s].z, xi.z, a_ih);
    a_hh = mfma16(f_hh.w[0][s].w, xh.w, a_hh);
    a_ih = mfma16(f_ih.w[0][s].w, xi.w, a_ih);
  }
}

struct CellSaved {      // this lane's slice of what the forward saved for one cell and step
  float4 r, z, n, hn, hp;
};
__device__ __forceinline__ void load_cell(CellSaved& c, const float* __restrict__ gates, const float* __restrict__ hprev,
                                          int64_t row, int f0) {
  const float* go = gates + row * 4 * H + f0;
  c.r = *reinterpret_cast<const float4*>(go);
  c.z = *reinterpret_cast<const float4*>(go + H);
  c.n = *reinterpret_cast<const float4*>(go + 2 * H);
  c.hn = *reinterpret_cast<const float4*>(go + 3 * H);
  c.hp = *reinterpret_cast<const float4*>(hprev + row * H + f0);
}

// GRU cell backward for this lane's (row, 4 features): dh = incoming gradient (already incl. carry).  Writes the gate
// gradients to the global dgi / dgh rows and to the LDS tile; returns direct = dh * z (the path to h_{t-1}).
template <bool STORE_GH>
__device__ __forceinline__ float4 cell_bwd(const float (&dh)[4], const CellSaved& c, float* __restrict__ dgi,
                                           float* __restrict__ dgh, float* G, int i, int f0, bool store = true) {
  const float rr[4] = {c.r.x, c.r.y, c.r.z, c.r.w}, zz[4] = {c.z.x, c.z.y, c.z.z, c.z.w}, nn[4] = {c.n.x, c.n.y, c.n.z, c.n.w},
              gh[4] = {c.hn.x, c.hn.y, c.hn.z, c.hn.w}, hp[4] = {c.hp.x, c.hp.y, c.hp.z, c.hp.w};
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
  if (store) {
    *reinterpret_cast<float4*>(dgi) = vr; *reinterpret_cast<float4*>(dgi + H) = vz; *reinterpret_cast<float4*>(dgi + 2 * H) = vn;
  }
  if (STORE_GH && store) {    // (the fused kernel consumes dgh1 from the LDS tile: that (T-1,B,3H) array is never written)
    *reinterpret_cast<float4*>(dgh) = vr; *reinterpret_cast<float4*>(dgh + H) = vz; *reinterpret_cast<float4*>(dgh + 2 * H) = vh;
  }
  float* g = G + i * LDG + f0;
  *reinterpret_cast<float4*>(g) = vr;
  *reinterpret_cast<float4*>(g + H) = vz;
  *reinterpret_cast<float4*>(g + 2 * H) = vn;
  *reinterpret_cast<float4*>(g + 3 * H) = vh;
  return make_float4(direct[0], direct[1], direct[2], direct[3]);
}

// FW: the weight (and bias) gradient of the decoder's W_hh of layer 1 -- dW = sum over steps and rows of dgh1^T h1_prev -- is
// accumulated IN this kernel: each wave keeps the 12 accumulator tiles of its own 16 hidden-unit columns (48 registers: what
// the 512-register budget has left next to the four register-resident weight-fragment sets; two matrices spill and cost
// 0.2 ms, all four 0.65 ms: measured) and issues the step's 48 MFMAs in front of the next iteration's exchange, i.e. in the
// fabric round trip the wave otherwise spins through.  dgh1 (104 MB per step at B = 4096) is neither written nor read back,
// and the batched weight-gradient launch that follows has three matrices left instead of four.
template <bool FW>
__global__ __launch_bounds__(256, 1) void dec_persist_bwd_kernel(DecPersistBwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Xdu = smem + R_XDU;
  float* Xdy = smem + R_XDY;
  float* Gt = smem + R_G;
  float* Gt1 = FW ? smem + R_G1 : Gt;
  float* Dt = smem + R_DT;
  uint32_t* Kt = reinterpret_cast<uint32_t*>(smem + R_KT);
  float* Ppre_t = smem + R_PPRET;
  float* Pout_t = smem + R_POUTT;
  float* bnw = smem + R_BNW;
  float* red = smem + R_RED;
  float* tot = smem + R_TOT;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i = lane & 15, q = lane >> 4;
  const int b = blockIdx.x, b0 = b * 16;
  const int T = a.T, B = a.B, G3 = 3 * H;
  const int f0 = 16 * wave + 4 * q;
  const bool drop = a.keep_l0 && a.p_drop > 0.f;
  const float keep_scale = 1.0f / (1.0f - a.p_drop);
  const float invB = 1.0f / (float)B;
  const int64_t row_i = b0 + i;

  WFrag<1, KSG> f_ih0, f_hh0, f_ih1, f_hh1;
  frag_load(f_hh1, a.p_hh1_t, wave, 0, lane);
  frag_load(f_ih1, a.p_ih1_t, wave, 0, lane);
  frag_load(f_hh0, a.p_hh0_t, wave, 0, lane);
  frag_load(f_ih0, a.p_ih0_t, wave, 0, lane);
  for (int e = tid; e < 9 * KSH * 64; e += 256)            // tiles 9..11 are padding (see R_LCN)
    reinterpret_cast<float4*>(Ppre_t)[e] = reinterpret_cast<const float4*>(a.p_pre_t)[e];
  for (int e = tid; e < 4 * KSD * 64; e += 256) reinterpret_cast<float4*>(Pout_t)[e] = reinterpret_cast<const float4*>(a.p_out_t)[e];
  if (tid < H) bnw[tid] = a.w.bn_w[tid];
  // custom_loss folded in (g2v.h, g2v_dec_saved.loss_*): dLoss/dy_t is formed in tile_commit below from y_t, the code byte the
  // forward left and the column coefficient -- `dy` is never read.  Workgroup 0 also adds up the forward's loss sums.
  const bool fold = a.sv.loss_code != nullptr;
  float* Lcn = smem + R_LCN;
  float* Ltab = smem + R_LTAB;
  if (fold) {
    for (int e = tid; e < (16 * D) / 4; e += 256)
      reinterpret_cast<float4*>(Lcn)[e] = reinterpret_cast<const float4*>(a.sv.loss_coef + (int64_t)b0 * D)[e];
    if (tid < 64) Ltab[tid] = loss_grad_const(a.lc1, a.lc2, tid);
    if (b == 0) {
      float s4[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) s4[k] = wave_sum(tid < a.nblk ? a.sv.loss_partial[(int64_t)tid * 4 + k] : 0.f);
      if (lane == 0) {
#pragma unroll
        for (int k = 0; k < 4; ++k) red[wave * 4 + k] = s4[k];
      }
      lds_barrier();
      if (tid == 0)
        loss_terms_write(a.sv.loss_terms, (red[0] + red[4]) + (red[8] + red[12]), (red[1] + red[5]) + (red[9] + red[13]),
                         (red[2] + red[6]) + (red[10] + red[14]), (red[3] + red[7]) + (red[11] + red[15]), a.lc1, a.lc2, a.lc3,
                         a.linv_n);
    }
  }
  for (int e = tid; e < 16 * LDH; e += 256) Xdu[e] = 0.f;
  for (int e = tid; e < 16 * (LDD - D); e += 256) Xdy[(e / (LDD - D)) * LDD + D + (e % (LDD - D))] = 0.f;
  lds_barrier();

  // dy (loss gradient) and keep95 tiles of step `ts` -> LDS, as two phases so that the requests can sit in front of a
  // product and the LDS writes behind it.  Clamped, always-valid addresses: no branch around a load, no stack array.
  float4 dy_a, dy_b, dy_c;
  uint32_t k_a, k_b, k_c;
  const uint32_t kmask = fold ? 0x40u : 0xffu;      // the Dropout(0.95) flag inside a Kt byte
  auto tile_request = [&](int ts) {
    const int e4c = tid + 512 < (16 * D) / 4 ? tid + 512 : 0;
    // fold: y_t and the code bytes (which carry the keep flag) take the places of dy_t and the keep95 bytes
    const float* dyp = (fold ? a.sv.y : a.gr.dy) + ((int64_t)ts * B + b0) * D;
    const uint32_t* kp = fold ? reinterpret_cast<const uint32_t*>(a.sv.loss_code + ((int64_t)ts * B + b0) * D)
                              : reinterpret_cast<const uint32_t*>(a.keep95 + ((int64_t)min(ts, T - 2) * B + b0) * D);   // row T-1 does not exist
    dy_a = *reinterpret_cast<const float4*>(dyp + 4 * tid);
    dy_b = *reinterpret_cast<const float4*>(dyp + 4 * (tid + 256));
    dy_c = *reinterpret_cast<const float4*>(dyp + 4 * e4c);
    k_a = kp[tid]; k_b = kp[tid + 256]; k_c = kp[e4c];
  };
  auto loss_dy = [&](float4& v, uint32_t code, int e4) {      // common.hpp: loss_grad(loss_grad_const(code), cn, y)
    const float4 cn = reinterpret_cast<const float4*>(Lcn)[e4];
    v.x = loss_grad(Ltab[code & 63u], cn.x, v.x);
    v.y = loss_grad(Ltab[(code >> 8) & 63u], cn.y, v.y);
    v.z = loss_grad(Ltab[(code >> 16) & 63u], cn.z, v.z);
    v.w = loss_grad(Ltab[(code >> 24) & 63u], cn.w, v.w);
  };
  auto tile_commit = [&](int ts) {
    const bool fb = (ts != T - 1) && a.conditioned && (ts >= a.n_pre);
    if (fold) {
      loss_dy(dy_a, k_a, tid);
      loss_dy(dy_b, k_b, tid + 256);
      loss_dy(dy_c, k_c, tid + 512 < (16 * D) / 4 ? tid + 512 : 0);
    }
    reinterpret_cast<float4*>(Dt)[tid] = dy_a;
    reinterpret_cast<float4*>(Dt)[tid + 256] = dy_b;
    Kt[tid] = fb ? k_a : 0u;
    Kt[tid + 256] = fb ? k_b : 0u;
    if (tid + 512 < (16 * D) / 4) {          // 16 * D / 4 = 540 float4: threads 0..27 own a third element
      reinterpret_cast<float4*>(Dt)[tid + 512] = dy_c;
      Kt[tid + 512] = fb ? k_c : 0u;
    }
  };
  tile_request(T - 1);
  tile_commit(T - 1);

  // ---- fused weight gradient: accumulators, the pending step's B fragment, column sums of the cell-1 tile ---------------
  f32x4 wacc[FW ? 12 : 1];
  float bfr[4] = {0.f, 0.f, 0.f, 0.f};
  float dbc1 = 0.f;                      // thread tid <-> column tid of the merged tile [g_r | g_z | g_n | g_hn]
  bool pending = false;
#pragma unroll
  for (int g = 0; g < (FW ? 12 : 1); ++g) wacc[g] = (f32x4){0.f, 0.f, 0.f, 0.f};
  // the pending step's MFMAs: A = tile^T (gate column on the lane, 4 rows per MFMA; g_r, g_z, g_hn), B = h1_prev fragment
  auto wgrad_hh1 = [&]() {
    const float* gp = Gt1 + q * LDG + i;
#pragma unroll
    for (int gt = 0; gt < 12; ++gt) {
      const int col = 16 * gt + (gt >= 8 ? 64 : 0);
      float av[4];
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) av[ks] = gp[4 * ks * LDG + col];
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) wacc[FW ? gt : 0] = mfma16(av[ks], bfr[ks], wacc[FW ? gt : 0]);
    }
    float sum = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) sum += Gt1[r * LDG + tid];
    dbc1 += sum;
  };

  float4 carry0 = make_float4(0.f, 0.f, 0.f, 0.f), carry1 = carry0;   // d h0 / d h1 flowing to the earlier step
  float dbn[4] = {0.f, 0.f, 0.f, 0.f}, xhat[4] = {0.f, 0.f, 0.f, 0.f}, gis[4] = {0.f, 0.f, 0.f, 0.f};   // of the step in flight
  float acc_w = 0.f, acc_b = 0.f;                                       // d gamma / d beta (workgroup 0, tid < H)

  for (int t = T - 1; t >= 0; --t) {
    const bool last = (t == T - 1);
    const bool feedback = !last && a.conditioned && (t >= a.n_pre);
    // ---- this step's inputs.  The dy / keep95 tiles of step t were staged into LDS during the previous iteration (see
    // "stage the next step's tiles" below; the first iteration stages its own in the prologue), so the loop opens with the
    // exchange.  Holding every saved value of the step in registers across the exchange made hipcc park them in AGPRs
    // behind four `s_waitcnt vmcnt(0)` (+6 us per step): only the five cell-1 vectors are requested in front of the exchange
    // (they travel while it is in flight), the cell-0 values after the cell-1 epilogue, behind the hh1 / ih1 products.
    const int64_t tile = ((int64_t)t * B + b0) * D;
    const int64_t srow = (int64_t)(t - 1) * B + row_i;      // row of the saved arrays this step reads (t >= 1)
    CellSaved c1;
    load_cell(c1, a.sv.gates1, a.sv.h1, t > 0 ? srow : row_i, f0);      // t == 0: a valid, unused row
    __builtin_amdgcn_sched_barrier(0);
    PSTAMP(1, 0);
    // ---- Part A: finish BatchNorm backward of step t+1 -> du_{t+1} --------------------------------------------------------
    if (!last) {
      if (FW && pending) wgrad_hh1();      // in the shadow of the exchange: everybody's partial sums are still on the fabric
      px_exchange(a.x, (t + 1) & 1, (unsigned)(T - 1 - t), a.nblk, b, red, tot, tid);
      const float4 s14 = *reinterpret_cast<const float4*>(tot + f0), s24 = *reinterpret_cast<const float4*>(tot + H + f0);
      const float a1[4] = {s14.x, s14.y, s14.z, s14.w}, a2[4] = {s24.x, s24.y, s24.z, s24.w};
      float du[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) du[r] = gis[r] * (dbn[r] - a1[r] * invB - xhat[r] * a2[r] * invB);
      const float4 du4 = make_float4(du[0], du[1], du[2], du[3]);
      *reinterpret_cast<float4*>(a.gr.du + ((int64_t)t * B + row_i) * H + f0) = du4;
      *reinterpret_cast<float4*>(Xdu + i * LDH + f0) = du4;
      if (b == 0 && tid < H) {           // d gamma / d beta accumulate over the steps in step order
        acc_w += tot[H + tid];
        acc_b += tot[tid];
      }
    }
    PSTAMP(1, 1);
    if (t == 0) break;                   // only the BatchNorm finish of step 1 was left (y_0 is data)
    lds_barrier();                       // Xdu (and, first iteration, Dt / Kt) complete
    PSTAMP(1, 2);
    // ---- Part B: dy_t (loss gradient + feedback through Dropout(0.95) and pre_linear) -------------------------------------
    {
      f32x4 acc[3];
#pragma unroll
      for (int j = 0; j < 3; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (feedback) {
        // the LDS addresses of these fragment reads are recomputed every step (the compiler cannot hoist past the empty
        // asm): as loop invariants they were spilled to scratch and reloaded here behind an s_waitcnt vmcnt(0)
        int lane_r = lane;
        asm volatile("" : "+v"(lane_r));
        if (wave == 0) {
          lds_frag_mma<3, KSH>(acc, Ppre_t, wave, 4, Xdu, LDH, lane_r);
        } else {
          f32x4 a2[2] = {acc[0], acc[1]};
          lds_frag_mma<2, KSH>(a2, Ppre_t, wave, 4, Xdu, LDH, lane_r);
          acc[0] = a2[0]; acc[1] = a2[1];
        }
      }
#if G2V_BWD_EPI
      // LDS reads of a tile issued together and unconditionally, stores predicated (see the forward kernel's out-layer
      // epilogue); one tile at a time: all three at once cost 24 live registers this kernel does not have (scratch spills)
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const int d0 = 16 * (wave + 4 * j) + 4 * q, d0c = d0 < Dp ? d0 : 4 * q;
        float dyv[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) dyv[r] = Dt[i * D + d0c + r];
        // the four keep bytes of the (unaligned, D is odd) run: two aligned words through v_alignbyte
        const int kbyte = i * D + d0c;
        const uint32_t klo = Kt[kbyte >> 2], khi = Kt[(kbyte >> 2) + 1];
        const uint32_t kb4 = __builtin_amdgcn_alignbyte(khi, klo, (uint32_t)(kbyte & 3));
#pragma unroll
        for (int r = 0; r < 4; ++r) dyv[r] = (feedback && ((kb4 >> (8 * r)) & kmask)) ? dyv[r] + acc[j][r] * 20.0f : dyv[r];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int d = d0 + r;
          if (d < D) {
            Dt[i * D + d] = dyv[r];
            Xdy[i * LDD + d] = dyv[r];
          }
        }
      }
#else
      const uint8_t* Kb = reinterpret_cast<const uint8_t*>(Kt);
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const int d0 = 16 * (wave + 4 * j) + 4 * q;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int d = d0 + r;
          if (d < D) {
            float dy = Dt[i * D + d];
            if (feedback && (Kb[i * D + d] & kmask)) dy += acc[j][r] * 20.0f;
            Dt[i * D + d] = dy;
            Xdy[i * LDD + d] = dy;
          }
        }
      }
#endif
    }
    lds_barrier();
    if (feedback || fold)      // (fold: nobody has written the loss part of dy_t to memory; out_layer's weight gradient reads it)
      for (int e4 = tid; e4 < (16 * D) / 4; e4 += 256)
        *reinterpret_cast<float4*>(a.gr.dy + tile + 4 * (int64_t)e4) = reinterpret_cast<const float4*>(Dt)[e4];
    PSTAMP(1, 3);
    // ---- dh1 = carry1 + dy W_out ; GRU cell 1 backward ---------------------------------------------------------------------
    float4 direct1;
    {
      const f32x4 acc0 = lds_frag_mma_2chain<KSD>(Pout_t, wave, Xdy, LDD, lane);
      const float dh[4] = {acc0[0] + carry1.x, acc0[1] + carry1.y, acc0[2] + carry1.z, acc0[3] + carry1.w};
      direct1 = cell_bwd<!FW>(dh, c1, a.gr.dgi1 + srow * G3 + f0, a.gr.dgh1 + srow * G3 + f0, Gt1, i, f0);
    }
    // cell-0 / BatchNorm inputs of this step: in flight during the next two products (96 MFMAs)
    CellSaved c0;
    load_cell(c0, a.sv.gates0, a.sv.h0, srow, f0);
    const float4 a4 = *reinterpret_cast<const float4*>(a.sv.a + srow * H + f0);
    const float4 u4 = *reinterpret_cast<const float4*>(a.sv.u + srow * H + f0);
    const float4 mean4 = *reinterpret_cast<const float4*>(a.sv.bn_stats + (int64_t)(t - 1) * 2 * H + f0);
    const float4 var4 = *reinterpret_cast<const float4*>(a.sv.bn_stats + (int64_t)(t - 1) * 2 * H + H + f0);
    uint32_t kl0 = 0x01010101u;
    if (drop) kl0 = *reinterpret_cast<const uint32_t*>(a.keep_l0 + srow * H + f0);
    // stage the next step's tiles: Dt / Kt were last read by the feedback epilogue and the dy write-back above
    if (t > 1) tile_request(t - 1);
    __builtin_amdgcn_sched_barrier(0);
    lds_barrier();
    PSTAMP(1, 4);
    // ---- carry1' = dh1 * z + dgh1 W_hh1 ;  dx1 = dgi1 W_ih1 -> dh0 (inter-layer dropout backward) -------------------------
    float dh0[4];
    {
      f32x4 a1 = {0.f, 0.f, 0.f, 0.f}, a2 = {0.f, 0.f, 0.f, 0.f};
      gate_frag_mma2(a1, f_hh1, a2, f_ih1, Gt1, lane);
      carry1 = make_float4(direct1.x + a1[0], direct1.y + a1[1], direct1.z + a1[2], direct1.w + a1[3]);
      const float c0v[4] = {carry0.x, carry0.y, carry0.z, carry0.w};
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float v = a2[r];
        if (drop) v = ((kl0 >> (8 * r)) & 0xffu) ? v * keep_scale : 0.f;
        dh0[r] = v + c0v[r];
      }
    }
    if (t > 1) tile_commit(t - 1);
    lds_barrier();                       // every wave is done reading the cell-1 tile
    PSTAMP(1, 5);
    // ---- GRU cell 0 backward ---------------------------------------------------------------------------------------------
    float4 direct0;
    direct0 = cell_bwd<true>(dh0, c0, a.gr.dgi0 + srow * G3 + f0, a.gr.dgh0 + srow * G3 + f0, Gt, i, f0);
    lds_barrier();
    PSTAMP(1, 6);
    // ---- carry0' = dh0 * z + dgh0 W_hh0 ;  da = dgi0 W_ih0 -> ReLU backward -> dbn_t, partial sums, publish -----------------
    {
      f32x4 a1 = {0.f, 0.f, 0.f, 0.f}, a2 = {0.f, 0.f, 0.f, 0.f};
      gate_frag_mma2(a1, f_hh0, a2, f_ih0, Gt, lane);
      carry0 = make_float4(direct0.x + a1[0], direct0.y + a1[1], direct0.z + a1[2], direct0.w + a1[3]);
      const float av[4] = {a4.x, a4.y, a4.z, a4.w}, uv[4] = {u4.x, u4.y, u4.z, u4.w}, mv[4] = {mean4.x, mean4.y, mean4.z, mean4.w},
                  vv[4] = {var4.x, var4.y, var4.z, var4.w};
      const float4 g4 = *reinterpret_cast<const float4*>(bnw + f0);
      const float gg[4] = {g4.x, g4.y, g4.z, g4.w};
      float s1[4], s2[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float invstd = bn_invstd_(vv[r]);
        dbn[r] = (av[r] > 0.f) ? a2[r] : 0.f;
        xhat[r] = (uv[r] - mv[r]) * invstd;
        gis[r] = gg[r] * invstd;
        s1[r] = reduce16(dbn[r]);
        s2[r] = reduce16(dbn[r] * xhat[r]);
      }
      if (i == 0) {
        const int par = t & 1;
        __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc(
            a.x.rec1 + ((size_t)par * PX_MAX_NBLK + b) * PX_COLS, 0, PX_COLS * 8, 0x00020000);
        const unsigned tag = (unsigned)(T - t);
        px_publish2(rr, (unsigned)f0, s1[0], s1[1], tag);
        px_publish2(rr, (unsigned)f0 + 2, s1[2], s1[3], tag);
        px_publish2(rr, (unsigned)(H + f0), s2[0], s2[1], tag);
        px_publish2(rr, (unsigned)(H + f0) + 2, s2[2], s2[3], tag);
      }
    }
    if (FW) {          // h1_prev of this step as a B fragment (lane (row i, q) holds columns f0 .. f0 + 3); its MFMAs run at the
      float* tr = smem + R_TR + wave * TRW;                       // top of the next iteration.  Same wave, LDS ops in order.
      tr[i * 17 + 4 * q + 0] = c1.hp.x; tr[i * 17 + 4 * q + 1] = c1.hp.y; tr[i * 17 + 4 * q + 2] = c1.hp.z; tr[i * 17 + 4 * q + 3] = c1.hp.w;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) bfr[ks] = tr[(4 * ks + q) * 17 + i];
      pending = true;
    }
    PSTAMP(1, 7);
  }
  if (FW) {
    // partial dW_hh1 of this workgroup's 16 rows over all steps: lane holds dW[16 gt + 4 q + r][16 wave + i]; partial db_hh1:
    // column tid of the cell-1 tile (g_r, g_z <- columns 0..127, g_hn <- 192..255)
    const int64_t nw = (int64_t)G3 * H;
    float* sl = a.wslab + (int64_t)b * nw + 16 * wave + i;
#pragma unroll
    for (int gt = 0; gt < 12; ++gt)
#pragma unroll
      for (int r = 0; r < 4; ++r) sl[(int64_t)(16 * gt + 4 * q + r) * H] = wacc[FW ? gt : 0][r];
    float* dbs = a.wslab + (int64_t)a.nblk * nw + (int64_t)b * G3;
    if (tid < 128) dbs[tid] = dbc1;
    else if (tid >= 192) dbs[tid - 64] = dbc1;
  }
  // gradient wrt the initial hidden state (the quantised latent) and the BatchNorm affine parameters
  *reinterpret_cast<float4*>(a.gr.dh_init + row_i * H + f0) = carry0;
  *reinterpret_cast<float4*>(a.gr.dh_init + ((int64_t)B + row_i) * H + f0) = carry1;
  if (b == 0 && tid < H) {
    a.gr.d_bn_w[tid] = acc_w;
    a.gr.d_bn_b[tid] = acc_b;
  }
}


// The backward rollout for more row tiles than CUs: R tiles per workgroup (see dec_persist_fwd_mt_kernel).  The four transposed
// GRU matrices stay in registers and W_pre^T / W_out^T in LDS; per tile the workgroup carries the two hidden-state gradients and
// dbn / xhat / gamma * invstd of the step whose BatchNorm-backward sums are in flight (20 registers), and publishes ONE record of
// partial sums per step.  The exchange runs once per step; du of every tile is finished from the same totals.  The dy / keep95
// tiles and the five cell-1 vectors of the NEXT unit of work -- (t, next tile) or (t-1, first tile) -- are requested behind the
// cell-1 epilogue of the current one.  Neither the fused W_hh1 weight gradient nor the loss fold is offered here.
template <int R>
__global__ __launch_bounds__(256, 1) void dec_persist_bwd_mt_kernel(DecPersistBwdArgs a, int nwg) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Xdu = smem + R_XDU;
  float* Xdy = smem + R_XDY;
  float* Gt = smem + R_G;
  float* Dt = smem + R_DT;
  uint32_t* Kt = reinterpret_cast<uint32_t*>(smem + R_KT);
  float* Ppre_t = smem + R_PPRET;
  float* Pout_t = smem + R_POUTT;
  float* bnw = smem + R_BNW;
  float* red = smem + R_RED;
  float* tot = smem + R_TOT;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i = lane & 15, q = lane >> 4;
  const int b = blockIdx.x;
  const int T = a.T, B = a.B, G3 = 3 * H, ntiles = a.nblk;
  const int f0 = 16 * wave + 4 * q;
  const bool drop = a.keep_l0 && a.p_drop > 0.f;
  const float keep_scale = 1.0f / (1.0f - a.p_drop);
  const float invB = 1.0f / (float)B;
  const int Rv = min(R, (ntiles - b + nwg - 1) / nwg);      // this workgroup's tiles: b, b + nwg, ... (>= 1 of them)

  WFrag<1, KSG> f_ih0, f_hh0, f_ih1, f_hh1;
  frag_load(f_hh1, a.p_hh1_t, wave, 0, lane);
  frag_load(f_ih1, a.p_ih1_t, wave, 0, lane);
  frag_load(f_hh0, a.p_hh0_t, wave, 0, lane);
  frag_load(f_ih0, a.p_ih0_t, wave, 0, lane);
  for (int e = tid; e < 9 * KSH * 64; e += 256)
    reinterpret_cast<float4*>(Ppre_t)[e] = reinterpret_cast<const float4*>(a.p_pre_t)[e];
  for (int e = tid; e < 4 * KSD * 64; e += 256) reinterpret_cast<float4*>(Pout_t)[e] = reinterpret_cast<const float4*>(a.p_out_t)[e];
  if (tid < H) bnw[tid] = a.w.bn_w[tid];
  for (int e = tid; e < 16 * LDH; e += 256) Xdu[e] = 0.f;
  for (int e = tid; e < 16 * (LDD - D); e += 256) Xdy[(e / (LDD - D)) * LDD + D + (e % (LDD - D))] = 0.f;
  lds_barrier();

  float4 dy_a, dy_b, dy_c;
  uint32_t k_a, k_b, k_c;
  CellSaved c1;
  // the unit (step ts >= 1, rows b0x ..): dy / keep95 tiles and the cell-1 saved vectors
  // (a ragged last tile, B % 16 != 0: clamped addresses for the rows past B -- see the forward kernel)
  auto unit_request = [&](int ts, int b0x) {
    const int nr = min(16, B - b0x), n4v = (nr * D) / 4;
    const int ea = tid < n4v ? tid : 0, eb = tid + 256 < n4v ? tid + 256 : 0, ec = tid + 512 < n4v ? tid + 512 : 0;
    const float* dyp = a.gr.dy + ((int64_t)ts * B + b0x) * D;
    const uint32_t* kp = reinterpret_cast<const uint32_t*>(a.keep95 + ((int64_t)min(ts, T - 2) * B + b0x) * D);
    dy_a = *reinterpret_cast<const float4*>(dyp + 4 * ea);
    dy_b = *reinterpret_cast<const float4*>(dyp + 4 * eb);
    dy_c = *reinterpret_cast<const float4*>(dyp + 4 * ec);
    k_a = kp[ea]; k_b = kp[eb]; k_c = kp[ec];
    load_cell(c1, a.sv.gates1, a.sv.h1, (int64_t)(ts - 1) * B + b0x + min(i, nr - 1), f0);
  };
  auto tile_commit = [&](int ts) {
    const bool fb = (ts != T - 1) && a.conditioned && (ts >= a.n_pre);
    reinterpret_cast<float4*>(Dt)[tid] = dy_a;
    reinterpret_cast<float4*>(Dt)[tid + 256] = dy_b;
    Kt[tid] = fb ? k_a : 0u;
    Kt[tid + 256] = fb ? k_b : 0u;
    if (tid + 512 < (16 * D) / 4) {
      reinterpret_cast<float4*>(Dt)[tid + 512] = dy_c;
      Kt[tid + 512] = fb ? k_c : 0u;
    }
  };
  unit_request(T - 1, 16 * b);
  tile_commit(T - 1);

  float4 carry0[R], carry1[R];
  float dbn[R][4], xhat[R][4], gis[R][4];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    carry0[r] = carry1[r] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int e = 0; e < 4; ++e) dbn[r][e] = xhat[r][e] = gis[r][e] = 0.f;
  }
  float acc_w = 0.f, acc_b = 0.f;

  for (int t = T - 1; t >= 0; --t) {
    const bool last = (t == T - 1);
    const bool feedback = !last && a.conditioned && (t >= a.n_pre);
    float a1[4] = {0.f, 0.f, 0.f, 0.f}, a2s[4] = {0.f, 0.f, 0.f, 0.f};
    if (!last) {      // BatchNorm-backward sums of step t+1 over ALL rows: one exchange for the workgroup's tiles
      px_exchange(a.x, (t + 1) & 1, (unsigned)(T - 1 - t), nwg, b, red, tot, tid);
      const float4 s14 = *reinterpret_cast<const float4*>(tot + f0), s24 = *reinterpret_cast<const float4*>(tot + H + f0);
      a1[0] = s14.x; a1[1] = s14.y; a1[2] = s14.z; a1[3] = s14.w;
      a2s[0] = s24.x; a2s[1] = s24.y; a2s[2] = s24.z; a2s[3] = s24.w;
      if (b == 0 && tid < H) {
        acc_w += tot[H + tid];
        acc_b += tot[tid];
      }
    }
    float s1a[4] = {0.f, 0.f, 0.f, 0.f}, s2a[4] = {0.f, 0.f, 0.f, 0.f};
    static_for<R>([&](auto rc) {
      constexpr int r = decltype(rc)::value;
      if (r >= Rv) return;
      const int b0 = 16 * (b + r * nwg);
      const int nrows = min(16, B - b0);
      const bool rowok = i < nrows;
      const int64_t row_i = b0 + (rowok ? i : nrows - 1);      // rows past B: the last row's addresses, nothing stored
      const int64_t tile = ((int64_t)t * B + b0) * D;
      const int64_t srow = (int64_t)(t - 1) * B + row_i;
      const CellSaved c1u = c1;
      // ---- Part A: du_{t+1} of this tile ----------------------------------------------------------------------------------
      if (!last) {
        float du[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) du[e] = gis[r][e] * (dbn[r][e] - a1[e] * invB - xhat[r][e] * a2s[e] * invB);
        const float4 du4 = make_float4(du[0], du[1], du[2], du[3]);
        if (rowok) *reinterpret_cast<float4*>(a.gr.du + ((int64_t)t * B + row_i) * H + f0) = du4;
        if (t > 0) *reinterpret_cast<float4*>(Xdu + i * LDH + f0) = du4;
      }
      if (t == 0) return;
      lds_barrier();
      // ---- Part B: dy_t = loss gradient + feedback through Dropout(0.95) and pre_linear ----------------------------------------
      {
        f32x4 acc[3];
#pragma unroll
        for (int j = 0; j < 3; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (feedback) {
          int lane_r = lane;
          asm volatile("" : "+v"(lane_r));
          if (wave == 0) {
            lds_frag_mma<3, KSH>(acc, Ppre_t, wave, 4, Xdu, LDH, lane_r);
          } else {
            f32x4 a2[2] = {acc[0], acc[1]};
            lds_frag_mma<2, KSH>(a2, Ppre_t, wave, 4, Xdu, LDH, lane_r);
            acc[0] = a2[0]; acc[1] = a2[1];
          }
        }
#pragma unroll
        for (int j = 0; j < 3; ++j) {
          const int d0 = 16 * (wave + 4 * j) + 4 * q, d0c = d0 < Dp ? d0 : 4 * q;
          float dyv[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) dyv[e] = Dt[i * D + d0c + e];
          const int kbyte = i * D + d0c;
          const uint32_t klo = Kt[kbyte >> 2], khi = Kt[(kbyte >> 2) + 1];
          const uint32_t kb4 = __builtin_amdgcn_alignbyte(khi, klo, (uint32_t)(kbyte & 3));
#pragma unroll
          for (int e = 0; e < 4; ++e) dyv[e] = (feedback && ((kb4 >> (8 * e)) & 0xffu)) ? dyv[e] + acc[j][e] * 20.0f : dyv[e];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int d = d0 + e;
            if (d < D) {
              Dt[i * D + d] = dyv[e];
              Xdy[i * LDD + d] = dyv[e];
            }
          }
        }
      }
      lds_barrier();
      if (feedback)
        for (int e4 = tid; e4 < (nrows * D) / 4; e4 += 256)
          *reinterpret_cast<float4*>(a.gr.dy + tile + 4 * (int64_t)e4) = reinterpret_cast<const float4*>(Dt)[e4];
      // ---- dh1 = carry1 + dy W_out ; GRU cell 1 backward -------------------------------------------------------------------
      float4 direct1;
      {
        const f32x4 acc0 = lds_frag_mma_2chain<KSD>(Pout_t, wave, Xdy, LDD, lane);
        const float dh[4] = {acc0[0] + carry1[r].x, acc0[1] + carry1[r].y, acc0[2] + carry1[r].z, acc0[3] + carry1[r].w};
        direct1 = cell_bwd<true>(dh, c1u, a.gr.dgi1 + srow * G3 + f0, a.gr.dgh1 + srow * G3 + f0, Gt, i, f0, rowok);
      }
      CellSaved c0;
      load_cell(c0, a.sv.gates0, a.sv.h0, srow, f0);
      const float4 a4 = *reinterpret_cast<const float4*>(a.sv.a + srow * H + f0);
      const float4 u4 = *reinterpret_cast<const float4*>(a.sv.u + srow * H + f0);
      const float4 mean4 = *reinterpret_cast<const float4*>(a.sv.bn_stats + (int64_t)(t - 1) * 2 * H + f0);
      const float4 var4 = *reinterpret_cast<const float4*>(a.sv.bn_stats + (int64_t)(t - 1) * 2 * H + H + f0);
      uint32_t kl0 = 0x01010101u;
      if (drop) kl0 = *reinterpret_cast<const uint32_t*>(a.keep_l0 + srow * H + f0);
      // the next unit of work of this workgroup
      const bool wrap = (r + 1 >= Rv);
      const int nt = wrap ? t - 1 : t;
      const int nb0 = wrap ? 16 * b : 16 * (b + (r + 1) * nwg);
      if (nt >= 1) unit_request(nt, nb0);
      __builtin_amdgcn_sched_barrier(0);
      lds_barrier();
      // ---- carry1' = dh1 * z + dgh1 W_hh1 ;  dx1 = dgi1 W_ih1 -> dh0 ---------------------------------------------------------
      float dh0[4];
      {
        f32x4 p1 = {0.f, 0.f, 0.f, 0.f}, p2 = {0.f, 0.f, 0.f, 0.f};
        gate_frag_mma2(p1, f_hh1, p2, f_ih1, Gt, lane);
        carry1[r] = make_float4(direct1.x + p1[0], direct1.y + p1[1], direct1.z + p1[2], direct1.w + p1[3]);
        const float c0v[4] = {carry0[r].x, carry0[r].y, carry0[r].z, carry0[r].w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float v = p2[e];
          if (drop) v = ((kl0 >> (8 * e)) & 0xffu) ? v * keep_scale : 0.f;
          dh0[e] = v + c0v[e];
        }
      }
      if (nt >= 1) tile_commit(nt);
      lds_barrier();
      // ---- GRU cell 0 backward -------------------------------------------------------------------------------------------------
      const float4 direct0 = cell_bwd<true>(dh0, c0, a.gr.dgi0 + srow * G3 + f0, a.gr.dgh0 + srow * G3 + f0, Gt, i, f0, rowok);
      lds_barrier();
      // ---- carry0' ; da -> ReLU backward -> dbn_t of this tile, partial sums --------------------------------------------------
      {
        f32x4 p1 = {0.f, 0.f, 0.f, 0.f}, p2 = {0.f, 0.f, 0.f, 0.f};
        gate_frag_mma2(p1, f_hh0, p2, f_ih0, Gt, lane);
        carry0[r] = make_float4(direct0.x + p1[0], direct0.y + p1[1], direct0.z + p1[2], direct0.w + p1[3]);
        const float av[4] = {a4.x, a4.y, a4.z, a4.w}, uv[4] = {u4.x, u4.y, u4.z, u4.w}, mv[4] = {mean4.x, mean4.y, mean4.z, mean4.w},
                    vv[4] = {var4.x, var4.y, var4.z, var4.w};
        const float4 g4 = *reinterpret_cast<const float4*>(bnw + f0);
        const float gg[4] = {g4.x, g4.y, g4.z, g4.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float invstd = bn_invstd_(vv[e]);
          dbn[r][e] = (rowok && av[e] > 0.f) ? p2[e] : 0.f;
          xhat[r][e] = (uv[e] - mv[e]) * invstd;
          gis[r][e] = gg[e] * invstd;
          s1a[e] += reduce16(dbn[r][e]);
          s2a[e] += reduce16(dbn[r][e] * xhat[r][e]);
        }
      }
    });
    if (t == 0) break;
    if (i == 0) {
      const int par = t & 1;
      __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc(
          a.x.rec1 + ((size_t)par * PX_MAX_NBLK + b) * PX_COLS, 0, PX_COLS * 8, 0x00020000);
      const unsigned tag = (unsigned)(T - t);
      px_publish2(rr, (unsigned)f0, s1a[0], s1a[1], tag);
      px_publish2(rr, (unsigned)f0 + 2, s1a[2], s1a[3], tag);
      px_publish2(rr, (unsigned)(H + f0), s2a[0], s2a[1], tag);
      px_publish2(rr, (unsigned)(H + f0) + 2, s2a[2], s2a[3], tag);
    }
  }
  static_for<R>([&](auto rc) {
    constexpr int r = decltype(rc)::value;
    if (r >= Rv) return;
    const int64_t row_i = 16 * (b + r * nwg) + i;
    if (row_i >= B) return;
    *reinterpret_cast<float4*>(a.gr.dh_init + row_i * H + f0) = carry0[r];
    *reinterpret_cast<float4*>(a.gr.dh_init + ((int64_t)B + row_i) * H + f0) = carry1[r];
  });
  if (b == 0 && tid < H) {
    a.gr.d_bn_w[tid] = acc_w;
    a.gr.d_bn_b[tid] = acc_b;
  }
}

}  // namespace g2v

int dec_persist_bwd_launch(const g2v_dec_weights* w, const g2v_dec_saved* s, const g2v_dec_grads* g, const uint8_t* keep95,
                           const uint8_t* keep_l0, float p_drop, int n_pre, int conditioned, int T, int B,
                           const float* p_pre_t, const float* p_out_t, const float* p_ih0_t, const float* p_hh0_t,
                           const float* p_ih1_t, const float* p_hh1_t, void* xbase, hipStream_t st, bool clear, float* wslab,
                           int tiles_per_wg) {
  const bool fw = wslab != nullptr;      // fused W_hh1 weight gradient requested (g->dw_gru[3] / db_gru[3] set)
  DecPersistBwdArgs a;
  a.w = *w; a.sv = *s; a.gr = *g;
  a.p_pre_t = p_pre_t; a.p_out_t = p_out_t; a.p_ih0_t = p_ih0_t; a.p_hh0_t = p_hh0_t; a.p_ih1_t = p_ih1_t; a.p_hh1_t = p_hh1_t;
  a.keep95 = keep95; a.keep_l0 = keep_l0;
  a.x = persist_x_at(xbase);
  a.T = T; a.B = B; a.nblk = (B + 15) / 16; a.n_pre = n_pre; a.conditioned = conditioned; a.p_drop = p_drop;
  a.wslab = wslab;
  {      // as g2v_custom_loss_fwd_bwd forms them
    const float n = (float)T * (float)B * (float)D;
    a.lc1 = s->loss_w[0] / n; a.lc2 = s->loss_w[1] / n; a.lc3 = s->loss_w[2] / n; a.linv_n = 1.0f / n;
  }
  if (tiles_per_wg > 1 || (B & 15)) {
    const int R = tiles_per_wg, nwg = (a.nblk + R - 1) / R;
    const size_t lds = dec_persist_bwd_lds_bytes(false);
    const void* fn = R == 1 ? (const void*)dec_persist_bwd_mt_kernel<1>
                            : (R == 2 ? (const void*)dec_persist_bwd_mt_kernel<2> : (const void*)dec_persist_bwd_mt_kernel<3>);
    static bool mt_set[3] = {false, false, false};
    if (R < 1 || R > 3 || fw || s->loss_code) {
      set_error("dec_persist_bwd: %d tiles per workgroup with a fused weight gradient / loss fold is not offered", R);
      return G2V_ERR_ARG;
    }
    if (!mt_set[R - 1]) {
      if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess || !persist_fits(fn, lds)) {
        set_error("dec_persist_bwd: the %d-tile kernel does not fit a CU", R);
        return G2V_ERR_LAUNCH;
      }
      mt_set[R - 1] = true;
    }
    if (clear) (void)hipMemsetAsync(xbase, 0, PX_BYTES, st);
    if (R == 1) hipLaunchKernelGGL(dec_persist_bwd_mt_kernel<1>, dim3(nwg), dim3(256), lds, st, a, nwg);
    else if (R == 2) hipLaunchKernelGGL(dec_persist_bwd_mt_kernel<2>, dim3(nwg), dim3(256), lds, st, a, nwg);
    else hipLaunchKernelGGL(dec_persist_bwd_mt_kernel<3>, dim3(nwg), dim3(256), lds, st, a, nwg);
    if (hipGetLastError() != hipSuccess) {
      set_error("dec_persist_bwd: launch failed");
      return G2V_ERR_LAUNCH;
    }
    return G2V_OK;
  }
  const size_t lds = dec_persist_bwd_lds_bytes(fw);
  static bool attr_set[2] = {false, false};
  if (!attr_set[fw]) {
    const void* fn = fw ? (const void*)dec_persist_bwd_kernel<true> : (const void*)dec_persist_bwd_kernel<false>;
    if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
      set_error("dec_persist_bwd: cannot reserve %zu bytes of LDS", lds);
      return G2V_ERR_LAUNCH;
    }
    if (!persist_fits(fn, lds)) {
      set_error("dec_persist_bwd: the kernel does not fit a CU (occupancy query)");
      return G2V_ERR_LAUNCH;
    }
    attr_set[fw] = true;
  }
  if (clear) (void)hipMemsetAsync(xbase, 0, PX_BYTES, st);
  if (fw) hipLaunchKernelGGL(dec_persist_bwd_kernel<true>, dim3(a.nblk), dim3(256), lds, st, a);
  else hipLaunchKernelGGL(dec_persist_bwd_kernel<false>, dim3(a.nblk), dim3(256), lds, st, a);
  if (hipGetLastError() != hipSuccess) {
    set_error("dec_persist_bwd: launch failed");
    return G2V_ERR_LAUNCH;
  }
  if (fw) {          // sum the per-workgroup partials in a fixed order
    const int64_t nw = (int64_t)3 * H * H, nb = 3 * H;
    const float* sw[1] = {wslab};
    const float* sb[1] = {wslab + (int64_t)a.nblk * nw};
    float* ow[1] = {g->dw_gru[3]};
    float* ob[1] = {g->db_gru[3]};
    if (g2v_internal_slab_reduce4(sw, ow, sb, ob, 1, nw, nb, a.nblk, st) != 0) {
      set_error("dec_persist_bwd: slab reduction launch failed");
      return G2V_ERR_LAUNCH;
    }
  }
  return G2V_OK;
}
