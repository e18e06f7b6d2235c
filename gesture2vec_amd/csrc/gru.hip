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
#include "dec_persist.hpp"      // the exchange primitives (px_ld / px_st) of the cluster kernels; not the fault latch

namespace g2v {

__device__ __forceinline__ void gru_seq_fwd_body(const float* __restrict__ gi, const float* __restrict__ w_hh,
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
      (void)nvalid; (void)wvec;
      wave_gemm_p<3, 0>(acc, w_hh, Hp >> 4, ft, ntile, cur, ldx, lane);     // w_hh: fragment-major pack (3 gate groups x H rows)
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

// The same forward for H % 4 == 0, H <= 256, 16-byte aligned operands (round 4; every shipped YAML: H = 200).  The body above
// loads a tile's gi rows and biases AFTER its product, element by element, in front of the gate math (one exposed memory round
// trip per tile, three or four tiles per wave and step).  Here the gi vectors and biases of the wave's NEXT tile are requested
// before the current tile's product, as 16-byte vectors.
// Packed input projections (round 5): with `on`, row (t, b) of gi / dgi lives at row off[t] + b of a COMPACT array that holds only
// the positions inside their sequences (lengths sorted descending: at step t the first n_t rows of the batch; off[t] = sum of
// n_t' for t' < t) -- what pack_padded_sequence does for the reference.  The dense products that make gi and consume dgi
// (input projection, its data and weight gradients, the embedding gradient) then run over sum(lengths) rows instead of T x B:
// 40 % fewer at Part d's lengths U{4..20}.  hs / gates / dgh keep the (T,B,.) layout.  By value: T <= 64.
// Global-address-space accesses (round 6).  Operands that reach a body through a SELECTED by-value struct (`blockIdx.y == 0 ? d0 : d1`
// copies the struct to private memory) or through a pointer select with nullptr are generic pointers to the compiler: flat loads /
// stores, which count in vmcnt AND lgkmcnt and behind which every wait is conservative.  In the resident forward (one wave per SIMD)
// the tile's gate stores then stood in front of the next tile's gi vectors, 7 us of a 22 us step; in the streaming BPTT the 24
// prefetched vectors per thread were waited for at the first LDS read of the product phase they were meant to travel behind.
typedef __attribute__((address_space(1))) const f32x4 res_gc4;
typedef __attribute__((address_space(1))) f32x4 res_g4;
__device__ __forceinline__ float4 res_ld4(const float* p) {
  const f32x4 v = *(res_gc4*)p;
  return make_float4(v[0], v[1], v[2], v[3]);
}
__device__ __forceinline__ void res_st4(float* p, float4 v) { *(res_g4*)p = (f32x4){v.x, v.y, v.z, v.w}; }

constexpr int GRU_MAX_OFF = 64;
struct RowOff {
  int on;
  int off[GRU_MAX_OFF];
};
__device__ __forceinline__ int64_t gi_row_base(const RowOff& ro, int t, int B) { return ro.on ? (int64_t)ro.off[t] : (int64_t)t * B; }

template <int NR>      // 16-row tiles per workgroup (256 NR threads): every W_hh fragment multiplies 16 NR rows
__device__ __forceinline__ void gru_seq_fwd_body_v4(const float* __restrict__ gi, const float* __restrict__ w_hh,
                                                    const float* __restrict__ b_hh, const float* __restrict__ h0,
                                                    const int32_t* __restrict__ lengths, int reverse,
                                                    float* __restrict__ hs, int64_t hs_ld, float* __restrict__ h_n,
                                                    float* __restrict__ gates, int T, int B, int H, const RowOff& ro) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int ROWS = 16 * NR, NTHR = 256 * NR, NWAVE = 4 * NR;
  const int Hp = (H + 15) & ~15, ldx = Hp + 4;
  float* hbuf0 = smem;
  float* hbuf1 = smem + ROWS * ldx;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b0 = blockIdx.x * ROWS;
  const int nrows = min(ROWS, B - b0);
  const int i = lane & 15, q = lane >> 4;
  bool rvalid[NR];
  int brow[NR], len[NR];
#pragma unroll
  for (int r = 0; r < NR; ++r) {
    rvalid[r] = 16 * r + i < nrows;
    brow[r] = b0 + (rvalid[r] ? 16 * r + i : 0);
    len[r] = (lengths && rvalid[r]) ? lengths[brow[r]] : T;
  }
  for (int e = tid; e < ROWS * ldx; e += NTHR) {
    const int r = e / ldx, k = e - r * ldx;
    float v = 0.f;
    if (h0 && r < nrows && k < H) v = h0[(int64_t)(b0 + r) * H + k];
    hbuf0[e] = v;
    hbuf1[e] = 0.f;
  }
  const int ntile = Hp >> 4;
  const int mt = (ntile - wave + NWAVE - 1) / NWAVE;      // this wave's tiles: wave, wave + NWAVE, ...
  const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
  // Packed lengths (round 5): a step at which NO row of this workgroup's tile is inside its sequence changes nothing (every row
  // keeps its state, its output is the zero of a padded position) -- the forward direction runs steps [0, lmax), the reverse
  // direction [T - lmax, T), lmax = the tile's longest sequence; the other steps' outputs are zero-filled up front and their W_hh
  // products (the whole cost of a step) are not executed.  Part d sorts a batch by length (pack_padded_sequence's
  // enforce_sorted): lengths U{4..20} leave ~40 % of the (tile, step) pairs out.  Same results bit for bit (the skipped steps
  // were select-masked no-ops); their `gates` rows are left unwritten -- the backward reads gates only where t < length.
  int lmax = T;
  if (lengths) {
    lmax = 0;
    for (int r = 0; r < nrows; ++r) lmax = max(lmax, (int)lengths[b0 + r]);
    lmax = min(max(lmax, 0), T);
  }
  const int s_lo = reverse ? T - lmax : 0, s_hi = reverse ? T : lmax;
  if (lmax < T) {
    const int H4z = H >> 2, nskip = T - lmax;
    for (int e = tid; e < nskip * nrows * H4z; e += NTHR) {
      const int k = e / (nrows * H4z), rem = e - k * nrows * H4z, r = rem / H4z, c = 4 * (rem - r * H4z);
      const int sk = reverse ? k : lmax + k;              // a skipped step
      const int tk = reverse ? (T - 1 - sk) : sk;
      *reinterpret_cast<float4*>(hs + ((int64_t)tk * B + b0 + r) * hs_ld + c) = z4;
    }
  }
  // one tile ahead: the gi vectors and biases of the NEXT (step, tile) of this wave are requested before the current tile's
  // product (13 k-steps x 12 NR MFMAs: more than an HBM round trip)
  float4 gin[NR][3], bhn[3];
  auto prefetch = [&](int s, int j) {
    const int t = reverse ? (T - 1 - s) : s;
    const int f0 = 16 * (wave + NWAVE * j) + 4 * q;
    const bool col = f0 + 3 < H;
#pragma unroll
    for (int g = 0; g < 3; ++g) bhn[g] = col ? *reinterpret_cast<const float4*>(b_hh + g * H + f0) : z4;
#pragma unroll
    for (int r = 0; r < NR; ++r) {
      const bool ok = rvalid[r] && s < T && t < len[r] && col;
      const float* gir = gi + ((ok ? gi_row_base(ro, t, B) + brow[r] : (int64_t)0)) * 3 * H + (ok ? f0 : 0);
#pragma unroll
      for (int g = 0; g < 3; ++g) gin[r][g] = ok ? *reinterpret_cast<const float4*>(gir + g * H) : z4;
    }
  };
  prefetch(s_lo, 0);
  __syncthreads();
  float* cur = hbuf0;
  float* nxt = hbuf1;
  for (int s = s_lo; s < s_hi; ++s) {
    const int t = reverse ? (T - 1 - s) : s;
#pragma unroll 1
    for (int j = 0; j < mt; ++j) {
      const int ft = wave + NWAVE * j;
      float4 gic[NR][3], bh[3];
#pragma unroll
      for (int g = 0; g < 3; ++g) {
        bh[g] = bhn[g];
#pragma unroll
        for (int r = 0; r < NR; ++r) gic[r][g] = gin[r][g];
      }
      if (j + 1 < mt) prefetch(s, j + 1); else prefetch(s + 1, 0);
      f32x4 acc[3][NR];
#pragma unroll
      for (int g = 0; g < 3; ++g)
#pragma unroll
        for (int r = 0; r < NR; ++r) acc[g][r] = (f32x4){0.f, 0.f, 0.f, 0.f};
      wave_gemm_p_rows<3, NR>(acc, w_hh, Hp >> 4, ft, ntile, cur, ldx, lane);
      const int f0 = 16 * ft + 4 * q;
      if (f0 + 3 >= H) continue;
      const float br[4] = {bh[0].x, bh[0].y, bh[0].z, bh[0].w}, bz[4] = {bh[1].x, bh[1].y, bh[1].z, bh[1].w},
                  bn[4] = {bh[2].x, bh[2].y, bh[2].z, bh[2].w};
#pragma unroll
      for (int r = 0; r < NR; ++r) {
        const int li = 16 * r + i;
        const bool valid = rvalid[r] && (t < len[r]);
        const int64_t row = (int64_t)t * B + brow[r];
        const float4 hp4 = *reinterpret_cast<const float4*>(cur + li * ldx + f0);
        const float hp[4] = {hp4.x, hp4.y, hp4.z, hp4.w};
        const float ir[4] = {gic[r][0].x, gic[r][0].y, gic[r][0].z, gic[r][0].w}, iz[4] = {gic[r][1].x, gic[r][1].y, gic[r][1].z, gic[r][1].w},
                    in_[4] = {gic[r][2].x, gic[r][2].y, gic[r][2].z, gic[r][2].w};
        float hn[4], gr[4], gz[4], gn[4], gh[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float rr = sigmoidf_(ir[e] + (acc[0][r][e] + br[e]));
          const float zz = sigmoidf_(iz[e] + (acc[1][r][e] + bz[e]));
          const float ghn = acc[2][r][e] + bn[e];
          const float nn = tanhf_(in_[e] + rr * ghn);
          const float hnew = (1.0f - zz) * nn + zz * hp[e];
          hn[e] = valid ? hnew : hp[e];
          gr[e] = valid ? rr : 0.f; gz[e] = valid ? zz : 0.f; gn[e] = valid ? nn : 0.f; gh[e] = valid ? ghn : 0.f;
        }
        *reinterpret_cast<float4*>(nxt + li * ldx + f0) = make_float4(hn[0], hn[1], hn[2], hn[3]);
        if (rvalid[r]) {
          *reinterpret_cast<float4*>(hs + row * hs_ld + f0) = valid ? make_float4(hn[0], hn[1], hn[2], hn[3]) : z4;   // padded positions are zero
          if (gates) {
            float* go = gates + row * 4 * H + f0;
            *reinterpret_cast<float4*>(go) = make_float4(gr[0], gr[1], gr[2], gr[3]);
            *reinterpret_cast<float4*>(go + H) = make_float4(gz[0], gz[1], gz[2], gz[3]);
            *reinterpret_cast<float4*>(go + 2 * H) = make_float4(gn[0], gn[1], gn[2], gn[3]);
            *reinterpret_cast<float4*>(go + 3 * H) = make_float4(gh[0], gh[1], gh[2], gh[3]);
          }
        }
      }
    }
    __syncthreads();
    float* tmp = cur; cur = nxt; nxt = tmp;
  }
  if (h_n)
    for (int e = tid; e < ROWS * H; e += NTHR) {
      const int r = e / H, k = e - r * H;
      if (r < nrows) h_n[(int64_t)(b0 + r) * H + k] = cur[r * ldx + k];
    }
}

// Generic hidden size: one workgroup per 16 batch rows walks all T steps; BOTH directions of a bidirectional layer run in
// one launch (blockIdx.y): at small batch a direction has only B/16 workgroups, so two launches in a row would leave the
// chip idle twice as long.
struct GruGenF {
  const float* gi; const float* w_hh; const float* b_hh; const float* h0; float* hs; float* h_n; float* gates; int reverse;
  const int64_t* gather;      // (g2v_gru_dir.gi_gather: gi is a table, row r of the layout is gi + gather[r] * 3H; resident forward only)
};
template <int V4>      // 0: scalar body; NR = 1 / 2: the vector body with NR row tiles per workgroup
__global__ __launch_bounds__(V4 == 2 ? 512 : 256) void gru_seq_fwd_kernel(GruGenF d0, GruGenF d1, const int32_t* __restrict__ lengths,
                                                                         int64_t hs_ld, int T, int B, int H, RowOff ro) {
  const GruGenF d = blockIdx.y == 0 ? d0 : d1;
  if constexpr (V4 > 0) gru_seq_fwd_body_v4<V4>(d.gi, d.w_hh, d.b_hh, d.h0, lengths, d.reverse, d.hs, hs_ld, d.h_n, d.gates, T, B, H, ro);
  else gru_seq_fwd_body(d.gi, d.w_hh, d.b_hh, d.h0, lengths, d.reverse, d.hs, hs_ld, d.h_n, d.gates, T, B, H);
}

// ---- large batch, 192 < H <= 208 (every shipped YAML: H = 200): W_hh RESIDENT in the CU (round 6) ----------------------------------
// The kernels above re-stream the packed W_hh (3 x 13 tiles x 13 k-blocks = 507 KB) from L2 at EVERY time step, every workgroup the
// same fragments in the same order at the same moment: 0.65 ms per bidirectional layer at B = 4096, T = 20, a third of the matrix
// rate, and none of the prefetch depths, tile orders or row-tile counts tried in rounds 3-4 moved it.  A CU of gfx950 has 512 KB of
// vector registers and 160 KB of LDS -- more than the matrix.  One 256-thread workgroup per 16-row tile (one wave per SIMD, so a
// wave may use all 512 registers) keeps its share of W_hh for the WHOLE sequence: RES_NREG fragments (a dword per lane each) per
// wave in registers, the rest in LDS, wave-private, read back with conflict-free ds_read_b32; the state tile is the only other
// LDS tenant (single-buffered: the new state waits in registers for the step's barrier).  Waves take hidden tiles wave, wave + 4,
// ...: wave 0 four (624 fragments), the others three (468).  Per (hidden tile, step): 156 MFMAs in the packed k order of
// wave_gemm_p_rows, then the gate math of gru_seq_fwd_body_v4 -- results BITWISE those of the streaming kernel.  The gi vectors
// and biases of the wave's next tile travel while the current tile multiplies.
#ifndef G2V_RES_DIAG
#define G2V_RES_DIAG 0      // timing experiments only (gpurun_tools/r06_build_res_variants.sh): 1 no stores, 2 no gi loads, 3 no gate math, 4 no LDS weights
#endif
constexpr int RES_NREG = 364;                  // multiple of 4 (a packed float4 = 4 consecutive fragments)
constexpr int RES_KB = 13, RES_NT = 13, RES_LDX = 16 * RES_KB + 4;
constexpr int RES_FR4 = 4 * 3 * RES_KB * 4, RES_FR3 = 3 * 3 * RES_KB * 4;
constexpr int RES_LW0 = RES_FR4 - RES_NREG, RES_LW = RES_FR3 - RES_NREG;      // fragments in LDS: wave 0, waves 1..3
constexpr size_t RES_LDS_BYTES = ((size_t)(RES_LW0 + 3 * RES_LW) * 64 + 16 * RES_LDX) * sizeof(float);
static_assert(RES_LDS_BYTES <= 160 * 1024, "W_hh share + state tile must fit the CU's LDS");

template <int NH>      // hidden tiles of this wave
__device__ __forceinline__ void gru_res_fwd_wave(const GruGenF& d, const int32_t* __restrict__ lengths, int64_t hs_ld, int T, int B, int H,
                                                 const RowOff& ro, float* __restrict__ wl, float* __restrict__ hbuf) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i = lane & 15, q = lane >> 4;
  const float* __restrict__ P = d.w_hh;
  const float* __restrict__ gi = d.gi;
  const float* __restrict__ b_hh = d.b_hh;
  float* __restrict__ hs = d.hs;
  float* __restrict__ gates = d.gates;
  const int reverse = d.reverse;
  // ---- this wave's share of W_hh: fragment f = ((j * 3 + g) * 13 + s) * 4 + c of hidden tile wave + 4 j, gate g, k-block s, sub-step c
  float wreg[RES_NREG];
#pragma unroll
  for (int j = 0; j < NH; ++j)
#pragma unroll
    for (int g = 0; g < 3; ++g)
#pragma unroll
      for (int s = 0; s < RES_KB; ++s) {
        const int f = ((j * 3 + g) * RES_KB + s) * 4;
        const int tile = wave + 4 * j + g * RES_NT;
        const float4 w4 = res_ld4(P + ((int64_t)(tile * RES_KB + s) * 64 + lane) * 4);
        if (f < RES_NREG) {
          wreg[f] = w4.x; wreg[f + 1] = w4.y; wreg[f + 2] = w4.z; wreg[f + 3] = w4.w;
        } else {
          float* o = wl + (f - RES_NREG) * 64 + lane;
          o[0] = w4.x; o[64] = w4.y; o[128] = w4.z; o[192] = w4.w;
        }
      }
  // (round 6, later: a workgroup walks row tiles blockIdx.x, blockIdx.x + gridDim.x, ... with the matrix loaded ONCE -- at B = 4096 the
  //  512 tile-workgroups ran as two rounds over the 256 CUs and each round paid ~45 us for its 130 MB of L2 reads)
  //  Serpentine: tiles x, 2 NP - 1 - x, 2 NP + x, ... (NP = gridDim.x) -- Part d sorts a batch by length, and with the strided walk the
  //  workgroups of the long sentences took 32 steps where the others took 16: 3.01 -> 3.15 ms at B = 4096.)
  const int ntiles = (B + 15) >> 4, NP = gridDim.x;
  for (int k = 0; k * NP < ntiles; ++k) {
  const int tile = (k & 1) ? (k + 1) * NP - 1 - (int)blockIdx.x : k * NP + (int)blockIdx.x;
  if (tile >= ntiles) continue;                      // (uniform over the workgroup)
  const int b0 = tile * 16;
  const int nrows = min(16, B - b0);
  const bool rvalid = i < nrows;
  const int brow = b0 + (rvalid ? i : 0);
  const int len = (lengths && rvalid) ? lengths[brow] : T;
  for (int e = tid; e < 16 * RES_LDX; e += 256) {
    const int r = e / RES_LDX, k = e - r * RES_LDX;
    hbuf[e] = (d.h0 && r < nrows && k < H) ? d.h0[(int64_t)(b0 + r) * H + k] : 0.f;
  }
  const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
  int lmax = T;
  if (lengths) {
    lmax = 0;
    for (int r = 0; r < nrows; ++r) lmax = max(lmax, (int)lengths[b0 + r]);
    lmax = min(max(lmax, 0), T);
  }
  const int s_lo = reverse ? T - lmax : 0, s_hi = reverse ? T : lmax;
  if (lmax < T) {      // (steps at which no row of the tile is inside its sequence: outputs zero, nothing computed -- as above)
    const int H4z = H >> 2, nskip = T - lmax;
    for (int e = tid; e < nskip * nrows * H4z; e += 256) {
      const int k = e / (nrows * H4z), rem = e - k * nrows * H4z, r = rem / H4z, c = 4 * (rem - r * H4z);
      const int sk = reverse ? k : lmax + k;
      const int tk = reverse ? (T - 1 - sk) : sk;
      res_st4(hs + ((int64_t)tk * B + b0 + r) * hs_ld + c, z4);
    }
  }
  float4 gin[3], bhn[3];
  // (gathered gi: the table row of this lane's batch row at step s -- one index per (row, step), fetched a step ahead of its use)
  const int64_t* __restrict__ gather = d.gather;
  auto src_row = [&](int s) -> int64_t {
    const int t = reverse ? (T - 1 - s) : s;
    const bool ok = rvalid && s < T && t < len;
    const int64_t r = ok ? gi_row_base(ro, t, B) + brow : (int64_t)0;
    return gather ? (ok ? *(__attribute__((address_space(1))) const int64_t*)(gather + r) : (int64_t)0) : r;
  };
  int64_t src_cur = src_row(s_lo), src_nxt = 0;
  auto prefetch = [&](int s, int j, int64_t src) {
    const int t = reverse ? (T - 1 - s) : s;
    const int f0 = 16 * (wave + 4 * j) + 4 * q;
    const bool col = f0 + 3 < H;
#pragma unroll
    for (int g = 0; g < 3; ++g) bhn[g] = col ? res_ld4(b_hh + g * H + f0) : z4;
    const bool ok = rvalid && s < T && t < len && col && (G2V_RES_DIAG != 2);
    const float* gir = gi + (ok ? src : (int64_t)0) * 3 * H + (ok ? f0 : 0);
#pragma unroll
    for (int g = 0; g < 3; ++g) gin[g] = ok ? res_ld4(gir + g * H) : z4;
  };
  prefetch(s_lo, 0, src_cur);
  __syncthreads();                                   // the state tile (filled by the caller) and every wave's LDS fragments are in place
  const float* xrow = hbuf + i * RES_LDX + 4 * q;
  for (int s_ = s_lo; s_ < s_hi; ++s_) {
    const int t = reverse ? (T - 1 - s_) : s_;
    float4 hnew[NH];
    src_nxt = src_row(s_ + 1);
#pragma unroll
    for (int j = 0; j < NH; ++j) {
      const int ft = wave + 4 * j;
      float4 gic[3], bh[3];
#pragma unroll
      for (int g = 0; g < 3; ++g) { bh[g] = bhn[g]; gic[g] = gin[g]; }
      if (j + 1 < NH) prefetch(s_, j + 1, src_cur); else prefetch(s_ + 1, 0, src_nxt);
      f32x4 acc[3];
#pragma unroll
      for (int g = 0; g < 3; ++g) acc[g] = (f32x4){0.f, 0.f, 0.f, 0.f};
      float4 xn4 = *reinterpret_cast<const float4*>(xrow);
#pragma unroll
      for (int s = 0; s < RES_KB; ++s) {
        const float4 xb4 = xn4;                          // (one k-block ahead, and nothing hoisted further: the scheduler
        if (s + 1 < RES_KB) xn4 = *reinterpret_cast<const float4*>(xrow + 16 * (s + 1));      //  would otherwise spill weights)
        const float xb[4] = {xb4.x, xb4.y, xb4.z, xb4.w};
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
          for (int g = 0; g < 3; ++g) {
            const int f = ((j * 3 + g) * RES_KB + s) * 4 + c;
            const float w = (f < RES_NREG || G2V_RES_DIAG == 4) ? wreg[f < RES_NREG ? f : f % RES_NREG] : wl[(f - RES_NREG) * 64 + lane];
            acc[g] = mfma16(w, xb[c], acc[g]);
          }
#ifndef RES_SCHED_MASK
#define RES_SCHED_MASK 0
#endif
        __builtin_amdgcn_sched_barrier(RES_SCHED_MASK);
      }
      const int f0 = 16 * ft + 4 * q;
      hnew[j] = z4;
      if (G2V_RES_DIAG == 3) {
        hnew[j] = make_float4(acc[0][0] + gic[0].x, acc[1][1] + bh[1].x, acc[2][2], acc[0][3]);
      } else if (f0 + 3 < H) {
        const float br[4] = {bh[0].x, bh[0].y, bh[0].z, bh[0].w}, bz[4] = {bh[1].x, bh[1].y, bh[1].z, bh[1].w},
                    bn[4] = {bh[2].x, bh[2].y, bh[2].z, bh[2].w};
        const bool valid = rvalid && (t < len);
        const int64_t row = (int64_t)t * B + brow;
        const float4 hp4 = *reinterpret_cast<const float4*>(hbuf + i * RES_LDX + f0);
        const float hp[4] = {hp4.x, hp4.y, hp4.z, hp4.w};
        const float ir[4] = {gic[0].x, gic[0].y, gic[0].z, gic[0].w}, iz[4] = {gic[1].x, gic[1].y, gic[1].z, gic[1].w},
                    in_[4] = {gic[2].x, gic[2].y, gic[2].z, gic[2].w};
        float hn[4], gr[4], gz[4], gn[4], gh[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float rr = sigmoidf_(ir[e] + (acc[0][e] + br[e]));
          const float zz = sigmoidf_(iz[e] + (acc[1][e] + bz[e]));
          const float ghn = acc[2][e] + bn[e];
          const float nn = tanhf_(in_[e] + rr * ghn);
          const float hnw = (1.0f - zz) * nn + zz * hp[e];
          hn[e] = valid ? hnw : hp[e];
          gr[e] = valid ? rr : 0.f; gz[e] = valid ? zz : 0.f; gn[e] = valid ? nn : 0.f; gh[e] = valid ? ghn : 0.f;
        }
        hnew[j] = make_float4(hn[0], hn[1], hn[2], hn[3]);
        if (rvalid && G2V_RES_DIAG != 1) {
          res_st4(hs + row * hs_ld + f0, valid ? hnew[j] : z4);      // padded positions are zero
          if (gates) {
            float* go = gates + row * 4 * H + f0;
            res_st4(go, make_float4(gr[0], gr[1], gr[2], gr[3]));
            res_st4(go + H, make_float4(gz[0], gz[1], gz[2], gz[3]));
            res_st4(go + 2 * H, make_float4(gn[0], gn[1], gn[2], gn[3]));
            res_st4(go + 3 * H, make_float4(gh[0], gh[1], gh[2], gh[3]));
          }
        }
      }
    }
    __syncthreads();                                 // every wave has read the old state for its products and its own elements
#pragma unroll
    for (int j = 0; j < NH; ++j) {
      const int f0 = 16 * (wave + 4 * j) + 4 * q;
      if (f0 + 3 < H) *reinterpret_cast<float4*>(hbuf + i * RES_LDX + f0) = hnew[j];
    }
    src_cur = src_nxt;
    __syncthreads();
  }
  if (d.h_n)
    for (int e = tid; e < 16 * H; e += 256) {
      const int r = e / H, k = e - r * H;
      if (r < nrows) d.h_n[(int64_t)(b0 + r) * H + k] = hbuf[r * RES_LDX + k];
    }
  __syncthreads();                                   // (the state tile is free for the next row tile)
  }
}

__global__ __launch_bounds__(256) void gru_res_fwd_kernel(GruGenF d0, GruGenF d1, const int32_t* __restrict__ lengths, int64_t hs_ld,
                                                          int T, int B, int H, RowOff ro) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const GruGenF d = blockIdx.y == 0 ? d0 : d1;
  const int wave = threadIdx.x >> 6;
  float* hbuf = smem + (size_t)(RES_LW0 + 3 * RES_LW) * 64;
  // (the two instantiations execute the SAME sequence of barriers -- per row tile one in front of the loop, two per step, one behind,
  //  all ranges uniform over the workgroup -- so the waves meet at s_barrier although they sit in different code: the hardware counts arrivals)
  if (wave == 0) gru_res_fwd_wave<4>(d, lengths, hs_ld, T, B, H, ro, smem, hbuf);
  else gru_res_fwd_wave<3>(d, lengths, hs_ld, T, B, H, ro, smem + (size_t)(RES_LW0 + (wave - 1) * RES_LW) * 64, hbuf);
}

// BPTT.  w_hh_t = W_hh^T, (H, 3H) row-major (so that dh_prev = dgh W_hh is again "weights contiguous along
// the contraction").  LDS: Gs [16][3H padded] (dgh tile = MFMA B operand), dhs [16][H padded] (carry).
__device__ __forceinline__ void gru_seq_bwd_body(const float* __restrict__ d_hs, int64_t d_hs_ld,
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
      (void)nvalid; (void)wvec;
      wave_gemm_p<1, 0>(acc, w_hh_t, Gp >> 4, ft, 0, Gs, ldg, lane);         // w_hh_t: fragment-major pack of W_hh^T (H rows, K = 3H)
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

// The same BPTT for H % 4 == 0, H <= 256 and 16-byte aligned operands (round 4), what every shipped YAML runs at large batch
// (H = 200).  Round 3's body above spent ~85 % of a step in phase 1: per element FIVE dependent global loads issued one after the
// other (12-13 elements per thread: an HBM round trip each, the saved gates are 262 MB per direction at B = 4096) behind
// integer divisions -- 2.06 ms per call at the native shape, 12 % of the matrix rate.  Here a thread owns up to four 4-column
// groups of ONE row (row = tid / 16), every load of a step is a 16-byte vector, and ALL the loads of step s-1 (gates r, z, n, hn,
// h_prev, the upstream gradient: 24 vectors) are requested before phase 2 of step s, so that they travel while the matrix pipe
// works through dgh W_hh.
struct GruBwdPre {        // one step's inputs of this thread: [column group][...]
  float4 r[4], z[4], n[4], hn[4], hp[4], up[4];
};
template <int NR, int NWV = 4 * NR>      // 16-row tiles per workgroup, waves per workgroup (64 NWV threads; 16 NR rows x TPR threads)
__device__ __forceinline__ void gru_seq_bwd_body_v4(const float* __restrict__ d_hs, int64_t d_hs_ld,
                                                    const float* __restrict__ d_hn, const float* __restrict__ hs,
                                                    int64_t hs_ld, const float* __restrict__ h0,
                                                    const float* __restrict__ gates, const float* __restrict__ w_hh_t,
                                                    const int32_t* __restrict__ lengths, int reverse,
                                                    float* __restrict__ dgi, float* __restrict__ dgh,
                                                    float* __restrict__ dh0, int T, int B, int H, const RowOff& ro) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int Hp = (H + 15) & ~15, G = 3 * H, Gp = (G + 15) & ~15, ldg = Gp + 4, ldh = Hp + 4;
  constexpr int ROWS = 16 * NR, NTHR = 64 * NWV, NWAVE = NWV;
  constexpr int TPR = NTHR / ROWS, KG = 64 / TPR;      // threads per row; 4-column groups per thread (covers H <= 256)
  static_assert(TPR == 16 || TPR == 32, "16 or 32 threads per row");
  float* Gs = smem;                // [ROWS][ldg]
  float* dhs = smem + ROWS * ldg;  // [ROWS][ldh]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b0 = blockIdx.x * ROWS;
  const int nrows = min(ROWS, B - b0);
  const int i = lane & 15, q = lane >> 4;
  const int row_l = tid / TPR, cg = tid % TPR;            // this thread's row of the tile, first column group
  const bool rowv = row_l < nrows;
  const int b = b0 + (rowv ? row_l : 0);
  const int len = (lengths && rowv) ? lengths[b] : T;
  const int H4 = H >> 2;

  for (int e = tid; e < ROWS * ldg; e += NTHR) Gs[e] = 0.f;
  for (int e = tid; e < ROWS * ldh; e += NTHR) {
    const int r = e / ldh, k = e - r * ldh;
    dhs[e] = (d_hn && r < nrows && k < H) ? d_hn[(int64_t)(b0 + r) * H + k] : 0.f;
  }
  const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
  GruBwdPre P;
  auto prefetch = [&](int s) {
    const int t = reverse ? (T - 1 - s) : s, tprev = reverse ? t + 1 : t - 1;
    const bool live = rowv && t < len;
    const int64_t row = (int64_t)t * B + b;
    const float* go = gates + row * 4 * H;
    // h_prev of this step: the forward's previous output, or the initial state (first step; reverse direction: first valid step)
    const bool from_h0 = (s == 0) || (tprev >= len);
    const float* hpp = from_h0 ? (h0 ? h0 + (int64_t)b * H : nullptr) : hs + ((int64_t)tprev * B + b) * hs_ld;
#pragma unroll
    for (int k = 0; k < KG; ++k) {
      const int c4 = cg + TPR * k;
      const bool ok = live && c4 < H4;
      const int c = 4 * (ok ? c4 : 0);
      P.r[k] = ok ? res_ld4(go + c) : z4;
      P.z[k] = ok ? res_ld4(go + H + c) : z4;
      P.n[k] = ok ? res_ld4(go + 2 * H + c) : z4;
      P.hn[k] = ok ? res_ld4(go + 3 * H + c) : z4;
      P.hp[k] = (ok && hpp) ? res_ld4(hpp + c) : z4;
      P.up[k] = (ok && d_hs) ? res_ld4(d_hs + row * d_hs_ld + c) : z4;
    }
  };
  // Packed lengths (round 5, see the forward body): steps at which no row of the tile is inside its sequence -- s >= lmax for the
  // forward direction (the FIRST steps of this sweep), s < T - lmax for the reverse direction (its last) -- only have zero gate
  // gradients to deliver (the weight-gradient products read every row): zero-filled up front, their dgh W_hh products skipped.
  int lmax = T;
  if (lengths) {
    lmax = 0;
    for (int r = 0; r < nrows; ++r) lmax = max(lmax, (int)lengths[b0 + r]);
    lmax = min(max(lmax, 0), T);
  }
  const int s_first = reverse ? T - 1 : lmax - 1, s_last = reverse ? T - lmax : 0;      // executed: s_first down to s_last
  if (lmax < T) {
    const int G4 = G >> 2, nskip = T - lmax;
    for (int e = tid; e < nskip * nrows * G4; e += NTHR) {
      const int k = e / (nrows * G4), rem = e - k * nrows * G4, r = rem / G4, c = 4 * (rem - r * G4);
      const int sk = reverse ? k : lmax + k;
      const int tk = reverse ? (T - 1 - sk) : sk;
      const int64_t rowk = ((int64_t)tk * B + b0 + r) * G + c;
      if (!ro.on) *reinterpret_cast<float4*>(dgi + rowk) = z4;      // (packed dgi: positions outside the sequences do not exist)
      *reinterpret_cast<float4*>(dgh + rowk) = z4;
    }
  }
  if (s_first >= s_last) prefetch(s_first);
  __syncthreads();
  const int ntile = Hp >> 4;
  for (int s = s_first; s >= s_last; --s) {
    const int t = reverse ? (T - 1 - s) : s;
    const bool live = rowv && t < len;
    const int64_t row = (int64_t)t * B + b;
    // phase 1: gate gradients of this thread's column groups (inputs arrived during the previous step's phase 2)
#pragma unroll
    for (int k = 0; k < KG; ++k) {
      const int c4 = cg + TPR * k;
      if (c4 >= H4) continue;
      const int c = 4 * c4;
      const float4 dc = *reinterpret_cast<const float4*>(dhs + row_l * ldh + c);
      const float dh_in[4] = {dc.x, dc.y, dc.z, dc.w};
      const float rr[4] = {P.r[k].x, P.r[k].y, P.r[k].z, P.r[k].w}, zz[4] = {P.z[k].x, P.z[k].y, P.z[k].z, P.z[k].w};
      const float nn[4] = {P.n[k].x, P.n[k].y, P.n[k].z, P.n[k].w}, gh[4] = {P.hn[k].x, P.hn[k].y, P.hn[k].z, P.hn[k].w};
      const float hp[4] = {P.hp[k].x, P.hp[k].y, P.hp[k].z, P.hp[k].w}, up[4] = {P.up[k].x, P.up[k].y, P.up[k].z, P.up[k].w};
      float g_r[4], g_z[4], g_n[4], g_hn[4], direct[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        if (live) {
          const float dh = dh_in[e] + up[e];
          const float dn = dh * (1.0f - zz[e]);
          const float dz = dh * (hp[e] - nn[e]);
          const float dnp = dn * (1.0f - nn[e] * nn[e]);
          g_n[e] = dnp;
          g_hn[e] = dnp * rr[e];
          g_r[e] = dnp * gh[e] * rr[e] * (1.0f - rr[e]);
          g_z[e] = dz * zz[e] * (1.0f - zz[e]);
          direct[e] = dh * zz[e];
        } else {
          g_r[e] = g_z[e] = g_n[e] = g_hn[e] = 0.f;
          direct[e] = dh_in[e];
        }
      }
      const float4 vr = make_float4(g_r[0], g_r[1], g_r[2], g_r[3]), vz = make_float4(g_z[0], g_z[1], g_z[2], g_z[3]),
                   vn = make_float4(g_n[0], g_n[1], g_n[2], g_n[3]), vh = make_float4(g_hn[0], g_hn[1], g_hn[2], g_hn[3]);
      if (rowv) {
        float* gh_o = dgh + row * G + c;
        *reinterpret_cast<float4*>(gh_o) = vr; *reinterpret_cast<float4*>(gh_o + H) = vz; *reinterpret_cast<float4*>(gh_o + 2 * H) = vh;
        if (live || !ro.on) {
          float* gi_o = dgi + (gi_row_base(ro, t, B) + b) * G + c;
          *reinterpret_cast<float4*>(gi_o) = vr; *reinterpret_cast<float4*>(gi_o + H) = vz; *reinterpret_cast<float4*>(gi_o + 2 * H) = vn;
        }
      }
      float* gs = Gs + row_l * ldg + c;
      *reinterpret_cast<float4*>(gs) = vr; *reinterpret_cast<float4*>(gs + H) = vz; *reinterpret_cast<float4*>(gs + 2 * H) = vh;
      *reinterpret_cast<float4*>(dhs + row_l * ldh + c) = make_float4(direct[0], direct[1], direct[2], direct[3]);
    }
    lds_barrier();      // (LDS hazards only: __syncthreads() would also drain this step's 19 gate-gradient stores / 24 prefetched vectors per thread)
    if (s > s_last) prefetch(s - 1);                    // requested ahead of the product below (in front of this wave's last
                                                        // tile instead: 2.05 ms against 1.64 at the native shape)
    // phase 2: dh_prev = direct + dgh W_hh
    // (this wave's feature tiles two at a time: two independent accumulator chains -- one chain issues a dependent MFMA every
    //  ~42 cycles, two alternate at 32 -- that share the LDS operand reads, and half the ring prologues)
    for (int ft = wave; ft < ntile; ft += 2 * NWAVE) {
      const bool two = ft + NWAVE < ntile;
      f32x4 acc[2][NR];
#pragma unroll
      for (int r = 0; r < NR; ++r) acc[0][r] = acc[1][r] = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (two) {
        wave_gemm_p_rows<2, NR>(acc, w_hh_t, Gp >> 4, ft, NWAVE, Gs, ldg, lane);   // w_hh_t: fragment-major pack of W_hh^T (H rows, K = 3H)
      } else {
        f32x4 a1[1][NR];
#pragma unroll
        for (int r = 0; r < NR; ++r) a1[0][r] = acc[0][r];
        wave_gemm_p_rows<1, NR>(a1, w_hh_t, Gp >> 4, ft, 0, Gs, ldg, lane);
#pragma unroll
        for (int r = 0; r < NR; ++r) acc[0][r] = a1[0][r];
      }
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int f0 = 16 * (ft + u * NWAVE) + 4 * q;
        if ((u == 0 || two) && f0 + 3 < H) {
#pragma unroll
          for (int r = 0; r < NR; ++r) {
            float4 v = *reinterpret_cast<float4*>(dhs + (16 * r + i) * ldh + f0);
            v.x += acc[u][r][0]; v.y += acc[u][r][1]; v.z += acc[u][r][2]; v.w += acc[u][r][3];
            *reinterpret_cast<float4*>(dhs + (16 * r + i) * ldh + f0) = v;
          }
        }
      }
    }
    lds_barrier();      // (LDS hazards only: __syncthreads() would also drain this step's 19 gate-gradient stores / 24 prefetched vectors per thread)
  }
  if (dh0)
    for (int e = tid; e < ROWS * H4; e += NTHR) {
      const int r = e / H4, c = 4 * (e - r * H4);
      if (r < nrows) *reinterpret_cast<float4*>(dh0 + (int64_t)(b0 + r) * H + c) = *reinterpret_cast<const float4*>(dhs + r * ldh + c);
    }
}

// (Second half of round 4, same shape, kernel time under rocprofv3 with eager launches: 1.70 ms -> 1.57 with the wave's tiles taken
// two at a time.  Ablations of that build: without the product 0.65 ms, without the gate-gradient stores 1.47, without the
// prefetch 1.39, product alone 1.34 against 0.24 for the empty loop -- the product phase is 46-55 us per workgroup and step for an
// MFMA floor of 8, and it stays there with EIGHT waves per 16-row tile (1.93 ms; 167 registers), with a 16-deep fragment ring
// (2.0 ms, scratch), and with the k-loop rotated per workgroup so that workgroups do not request the same L2 lines at the same
// moment (1.53).  The cause was in the ISA: an `s_waitcnt vmcnt(0)` at the head of every group of 8 k-steps of the streaming loop
// (wave_gemm_p_rows in common.hpp has the story); with whole branch-free groups the call is 1.44-1.47 ms.)
// (With the precise wait counts the BPTT is 1.44-1.47 ms; two row tiles per workgroup again, now on top of them: 1.53 ms.)
// (Round 4 also measured, at the native shape B = 4096, T = 20, H = 200, 1.63 ms per call for the body above: two row tiles per
// workgroup sharing every weight fragment 1.64-1.68 ms; the same with waves 0-3 only multiplying and waves 4-7 only moving the
// saved tensors, so that no weight fragment queues behind an HBM load, 1.68 ms; the next step's loads requested in front of the
// wave's last tile instead of its first 2.05 ms; the tiles visited in an order that differs from workgroup to workgroup 1.66 ms.
// Counters (profiles/r04_pmc_native.json): matrix pipe 14 % busy, 69 % of the
// wave cycles in s_waitcnt / barriers, 5 GB of L2 reads per call at 450 cycles average latency: every workgroup streams the SAME
// 480 KB of W_hh^T in the same order at the same time.  None of the three variants is kept.)
struct GruGenB {
  const float* d_hs; const float* d_hn; const float* hs; const float* h0; const float* gates; const float* w_hh_t;
  float* dgi; float* dgh; float* dh0; int reverse;
};
template <int V4>      // 0: scalar body; 1: the vector body
__global__ __launch_bounds__(256) void gru_seq_bwd_kernel(GruGenB d0, GruGenB d1, const int32_t* __restrict__ lengths,
                                                                         int64_t d_hs_ld, int64_t hs_ld, int T, int B, int H, RowOff ro) {
  // (pointer by pointer: `blockIdx.y == 0 ? d0 : d1` on the STRUCTS makes a private copy, whose pointers the compiler can no longer
  //  prove global -- the 24 prefetched vectors per thread became flat loads, which count in lgkmcnt too, so the first LDS wait of
  //  the product phase waited for the whole prefetch it was meant to hide; found in round 6 via the resident kernels)
  const bool y0 = blockIdx.y == 0;
  const float* d_hs = y0 ? d0.d_hs : d1.d_hs;
  const float* d_hn = y0 ? d0.d_hn : d1.d_hn;
  const float* hs = y0 ? d0.hs : d1.hs;
  const float* h0 = y0 ? d0.h0 : d1.h0;
  const float* gates = y0 ? d0.gates : d1.gates;
  const float* w_hh_t = y0 ? d0.w_hh_t : d1.w_hh_t;
  float* dgi = y0 ? d0.dgi : d1.dgi;
  float* dgh = y0 ? d0.dgh : d1.dgh;
  float* dh0 = y0 ? d0.dh0 : d1.dh0;
  const int reverse = y0 ? d0.reverse : d1.reverse;
  if constexpr (V4 > 0)
    gru_seq_bwd_body_v4<V4>(d_hs, d_hs_ld, d_hn, hs, hs_ld, h0, gates, w_hh_t, lengths, reverse, dgi, dgh, dh0, T, B, H, ro);
  else
    gru_seq_bwd_body(d_hs, d_hs_ld, d_hn, hs, hs_ld, h0, gates, w_hh_t, lengths, reverse, dgi, dgh, dh0, T, B, H);
}


// ---- large batch, H == 200: the BPTT with W_hh RESIDENT in the CU (round 6; the forward's scheme, gru_res_fwd_kernel) ---------------
// dh_prev = direct + dgh W_hh contracts over the 3H = 600 gate columns: 13 hidden tiles x 150 k-steps = 1950 weight fragments
// (the streaming kernel's packed W_hh^T pads the contraction to 608: 1976).  They do not fit beside that kernel's LDS tenants
// (the gate-gradient tile, the carry tile) and its 24 prefetched vectors per thread, so the step is laid out differently:
//   * a lane owns the elements the matrix pipe hands it -- (row i, hidden units 16 ft + 4 q .. + 3) of its wave's tiles ft = wave,
//     wave + 4, ... -- for BOTH phases: the carry dh never leaves its registers, the carry tile is gone from LDS;
//   * phase 1 (gate gradients, element-wise) writes dgh / dgi and the (16 x 600) gate-gradient tile, the B operand, to LDS;
//   * phase 2 multiplies all of the wave's tiles together (3-4 independent accumulator chains share every B read), weights from
//     registers (RESB_W4 / RESB_W3 fragments: the 4-tile wave prefetches 24 saved-tensor vectors, the 3-tile waves 18) and LDS;
//   * the saved tensors of step s - 1 are requested between the phases.
// Contraction order: k-steps of 4 in the packed kernel's interleaving (block of 16: sub-step c covers columns 16 s + 4 q + c), the
// last 8 columns as two plain k-steps -- NOT the streaming kernel's padded grouping: dgi / dgh equal to summation order, not bit
// for bit (tests: test_gru_resident_*).
constexpr int RESB_W4 = 344, RESB_W3 = 374;
constexpr int RESB_H = 200, RESB_G = 3 * RESB_H, RESB_KS = RESB_G / 4, RESB_BLK = RESB_G / 16, RESB_LDG = RESB_G + 4;
constexpr int RESB_L4 = 4 * RESB_KS - RESB_W4, RESB_L3 = 3 * RESB_KS - RESB_W3;      // fragments in LDS: wave 0, waves 1..3
constexpr size_t RESB_LDS_BYTES = ((size_t)(RESB_L4 + 3 * RESB_L3) * 64 + 16 * RESB_LDG) * sizeof(float);
static_assert(RESB_BLK * 16 + 8 == RESB_G, "37 blocks of 16 + two k-steps");
static_assert(RESB_LDS_BYTES <= 160 * 1024, "W_hh share + gate-gradient tile must fit the CU's LDS");
struct GruResB {
  const float* d_hs; const float* d_hn; const float* hs; const float* h0; const float* gates; const float* w_hh;      // w_hh: (3H, H) as the caller holds it
  float* dgi; float* dgh; float* dh0; int reverse;
};
__device__ __forceinline__ float res_ld1(const float* p) { return *(__attribute__((address_space(1))) const float*)p; }

template <int NH, int NW>      // hidden tiles of this wave; its weight fragments in registers
__device__ __forceinline__ void gru_res_bwd_wave(const GruResB& d, const int32_t* __restrict__ lengths, int64_t d_hs_ld, int64_t hs_ld, int T,
                                                 int B, const RowOff& ro, float* __restrict__ wl, float* __restrict__ Gs) {
  constexpr int H = RESB_H, G = RESB_G;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i = lane & 15, q = lane >> 4;
  const int reverse = d.reverse;
  const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
  // ---- this wave's share of W_hh^T: fragment (j, ks) of hidden tile wave + 4 j: lane (i, q) holds W_hh[k(ks, q)][16 ft + i]
  float wreg[NW];
#pragma unroll
  for (int j = 0; j < NH; ++j) {
    const int feat = 16 * (wave + 4 * j) + i;
    const bool fok = feat < H;
    const float* wc = d.w_hh + (fok ? feat : 0);
#pragma unroll
    for (int ks = 0; ks < RESB_KS; ++ks) {
      const int k = ks < 4 * RESB_BLK ? 16 * (ks >> 2) + 4 * q + (ks & 3) : 16 * RESB_BLK + 4 * (ks - 4 * RESB_BLK) + q;
      const float w = fok ? res_ld1(wc + (int64_t)k * H) : 0.f;
      const int f = j * RESB_KS + ks;
      if (f < NW) wreg[f < NW ? f : 0] = w;
      else wl[(f - NW) * 64 + lane] = w;
    }
  }
  int f0s[NH];
  bool col[NH];
#pragma unroll
  for (int j = 0; j < NH; ++j) {
    f0s[j] = 16 * (wave + 4 * j) + 4 * q;
    col[j] = f0s[j] + 3 < H;
  }
  const int ntiles = (B + 15) >> 4, NP = gridDim.x;
  for (int k = 0; k * NP < ntiles; ++k) {            // (row tiles in turn, serpentine, the matrix loaded once: see the forward)
  const int tile = (k & 1) ? (k + 1) * NP - 1 - (int)blockIdx.x : k * NP + (int)blockIdx.x;
  if (tile >= ntiles) continue;
  const int b0 = tile * 16;
  const int nrows = min(16, B - b0);
  const bool rvalid = i < nrows;
  const int b = b0 + (rvalid ? i : 0);
  const int len = (lengths && rvalid) ? lengths[b] : T;
  float4 dh[NH];
#pragma unroll
  for (int j = 0; j < NH; ++j) dh[j] = (d.d_hn && rvalid && col[j]) ? res_ld4(d.d_hn + (int64_t)b * H + f0s[j]) : z4;
  int lmax = T;
  if (lengths) {
    lmax = 0;
    for (int r = 0; r < nrows; ++r) lmax = max(lmax, (int)lengths[b0 + r]);
    lmax = min(max(lmax, 0), T);
  }
  const int s_first = reverse ? T - 1 : lmax - 1, s_last = reverse ? T - lmax : 0;      // executed: s_first down to s_last
  if (lmax < T) {      // (steps at which no row of the tile is inside its sequence: zero gate gradients, nothing computed)
    const int G4 = G >> 2, nskip = T - lmax;
    for (int e = tid; e < nskip * nrows * G4; e += 256) {
      const int k = e / (nrows * G4), rem = e - k * nrows * G4, r = rem / G4, c = 4 * (rem - r * G4);
      const int sk = reverse ? k : lmax + k;
      const int tk = reverse ? (T - 1 - sk) : sk;
      const int64_t rowk = ((int64_t)tk * B + b0 + r) * G + c;
      if (!ro.on) res_st4(d.dgi + rowk, z4);
      res_st4(d.dgh + rowk, z4);
    }
  }
  float4 Pr[NH], Pz[NH], Pn[NH], Phn[NH], Php[NH], Pup[NH];
  auto prefetch = [&](int s) {
    const int t = reverse ? (T - 1 - s) : s, tprev = reverse ? t + 1 : t - 1;
    const bool live = rvalid && t < len;
    const int64_t row = (int64_t)t * B + b;
    const float* go = d.gates + row * 4 * H;
    const bool from_h0 = (s == 0) || (tprev >= len);
    const float* hpp = from_h0 ? (d.h0 ? d.h0 + (int64_t)b * H : nullptr) : d.hs + ((int64_t)tprev * B + b) * hs_ld;
#pragma unroll
    for (int j = 0; j < NH; ++j) {
      const bool ok = live && col[j];
      const int c = ok ? f0s[j] : 0;
      Pr[j] = ok ? res_ld4(go + c) : z4;
      Pz[j] = ok ? res_ld4(go + H + c) : z4;
      Pn[j] = ok ? res_ld4(go + 2 * H + c) : z4;
      Phn[j] = ok ? res_ld4(go + 3 * H + c) : z4;
      Php[j] = (ok && hpp) ? res_ld4(hpp + c) : z4;
      Pup[j] = (ok && d.d_hs) ? res_ld4(d.d_hs + row * d_hs_ld + c) : z4;
    }
  };
  if (s_first >= s_last) prefetch(s_first);
  const float* xrow = Gs + i * RESB_LDG + 4 * q;
  for (int s = s_first; s >= s_last; --s) {
    const int t = reverse ? (T - 1 - s) : s;
    const bool live = rvalid && t < len;
    const int64_t row = (int64_t)t * B + b;
    float4 direct[NH];
    // phase 1: the gate gradients of this lane's elements (their inputs arrived during the previous step's product)
#pragma unroll
    for (int j = 0; j < NH; ++j) {
      direct[j] = z4;
      if (!col[j]) continue;
      const int c = f0s[j];
      const float dh_in[4] = {dh[j].x, dh[j].y, dh[j].z, dh[j].w};
      const float rr[4] = {Pr[j].x, Pr[j].y, Pr[j].z, Pr[j].w}, zz[4] = {Pz[j].x, Pz[j].y, Pz[j].z, Pz[j].w};
      const float nn[4] = {Pn[j].x, Pn[j].y, Pn[j].z, Pn[j].w}, gh[4] = {Phn[j].x, Phn[j].y, Phn[j].z, Phn[j].w};
      const float hp[4] = {Php[j].x, Php[j].y, Php[j].z, Php[j].w}, up[4] = {Pup[j].x, Pup[j].y, Pup[j].z, Pup[j].w};
      float g_r[4], g_z[4], g_n[4], g_hn[4], dir[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        if (live) {
          const float dhv = dh_in[e] + up[e];
          const float dn = dhv * (1.0f - zz[e]);
          const float dz = dhv * (hp[e] - nn[e]);
          const float dnp = dn * (1.0f - nn[e] * nn[e]);
          g_n[e] = dnp;
          g_hn[e] = dnp * rr[e];
          g_r[e] = dnp * gh[e] * rr[e] * (1.0f - rr[e]);
          g_z[e] = dz * zz[e] * (1.0f - zz[e]);
          dir[e] = dhv * zz[e];
        } else {
          g_r[e] = g_z[e] = g_n[e] = g_hn[e] = 0.f;
          dir[e] = dh_in[e];
        }
      }
      const float4 vr = make_float4(g_r[0], g_r[1], g_r[2], g_r[3]), vz = make_float4(g_z[0], g_z[1], g_z[2], g_z[3]),
                   vn = make_float4(g_n[0], g_n[1], g_n[2], g_n[3]), vh = make_float4(g_hn[0], g_hn[1], g_hn[2], g_hn[3]);
      if (rvalid) {
        float* gh_o = d.dgh + row * G + c;
        res_st4(gh_o, vr); res_st4(gh_o + H, vz); res_st4(gh_o + 2 * H, vh);
        if (live || !ro.on) {
          float* gi_o = d.dgi + (gi_row_base(ro, t, B) + b) * G + c;
          res_st4(gi_o, vr); res_st4(gi_o + H, vz); res_st4(gi_o + 2 * H, vn);
        }
      }
      float* gs = Gs + i * RESB_LDG + c;
      *reinterpret_cast<float4*>(gs) = vr; *reinterpret_cast<float4*>(gs + H) = vz; *reinterpret_cast<float4*>(gs + 2 * H) = vh;
      direct[j] = make_float4(dir[0], dir[1], dir[2], dir[3]);
    }
    lds_barrier();                                   // the gate-gradient tile is complete (LDS only: the stores above keep travelling)
    if (s > s_last) prefetch(s - 1);
    // phase 2: dh_prev = direct + dgh W_hh for all of this wave's tiles at once
    f32x4 acc[NH];
#pragma unroll
    for (int j = 0; j < NH; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float4 xn4 = *reinterpret_cast<const float4*>(xrow);
#pragma unroll
    for (int blk = 0; blk < RESB_BLK; ++blk) {
      const float4 xb4 = xn4;
      if (blk + 1 < RESB_BLK) xn4 = *reinterpret_cast<const float4*>(xrow + 16 * (blk + 1));
      const float xb[4] = {xb4.x, xb4.y, xb4.z, xb4.w};
#pragma unroll
      for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int j = 0; j < NH; ++j) {
          const int f = j * RESB_KS + 4 * blk + c;
          const float w = f < NW ? wreg[f < NW ? f : 0] : wl[(f - NW) * 64 + lane];
          acc[j] = mfma16(w, xb[c], acc[j]);
        }
      __builtin_amdgcn_sched_barrier(0);
    }
    {
      const float xa = Gs[i * RESB_LDG + 16 * RESB_BLK + q], xc = Gs[i * RESB_LDG + 16 * RESB_BLK + 4 + q];
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int j = 0; j < NH; ++j) {
          const int f = j * RESB_KS + 4 * RESB_BLK + u;
          const float w = f < NW ? wreg[f < NW ? f : 0] : wl[(f - NW) * 64 + lane];
          acc[j] = mfma16(w, u == 0 ? xa : xc, acc[j]);
        }
    }
#pragma unroll
    for (int j = 0; j < NH; ++j)
      dh[j] = make_float4(direct[j].x + acc[j][0], direct[j].y + acc[j][1], direct[j].z + acc[j][2], direct[j].w + acc[j][3]);
    lds_barrier();                                   // every wave has read the tile before the next phase 1 rewrites it
  }
  if (d.dh0 && rvalid)
#pragma unroll
    for (int j = 0; j < NH; ++j)
      if (col[j]) res_st4(d.dh0 + (int64_t)b * H + f0s[j], dh[j]);
  }
}

__global__ __launch_bounds__(256) void gru_res_bwd_kernel(GruResB d0, GruResB d1, const int32_t* __restrict__ lengths, int64_t d_hs_ld,
                                                          int64_t hs_ld, int T, int B, RowOff ro) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const GruResB d = blockIdx.y == 0 ? d0 : d1;
  const int wave = threadIdx.x >> 6;
  float* Gs = smem + (size_t)(RESB_L4 + 3 * RESB_L3) * 64;
  if (wave == 0) gru_res_bwd_wave<4, RESB_W4>(d, lengths, d_hs_ld, hs_ld, T, B, ro, smem, Gs);
  else gru_res_bwd_wave<3, RESB_W3>(d, lengths, d_hs_ld, hs_ld, T, B, ro, smem + (size_t)(RESB_L4 + (wave - 1) * RESB_L3) * 64, Gs);
}

// ---- small batch, generic hidden size: ONE LAUNCH PER TIME STEP, one wave per (16 rows, 16 hidden units, direction) -------
// The persistent kernels above give a direction only B/16 workgroups, each re-streaming the whole W_hh (3H x H) per step:
// at B = 128, H = 200 that is 8 busy CUs and ~32 us per step.  A hidden unit's new state needs only ITS three gate rows of
// W_hh, so a step splits over hidden-unit tiles without any cross-workgroup reduction, and the kernel boundary is the step
// barrier.  Each wave requests EVERYTHING it needs for the step up front (its W_hh rows and the 16 state rows over the whole
// contraction, as MFMA fragments in registers: one memory round trip), multiplies, and finishes its 16 x 16 outputs.
// Used when B/16 * ndir <= 128 workgroups (B <= 1024), H % 4 == 0, H <= 256.
constexpr int GRU_STEP_KS = 16;        // k-steps of the forward product (H <= 256)
constexpr int GRU_STEP_KSB = 48;       // k-steps of the backward product (3H <= 768)

struct GruStepF {
  const float* gi;      // (T,B,3H)
  const float* w_hh; const float* b_hh;
  const float* h_prev;  // (B,H) carried state entering the step, NULL = zeros
  float* h_next;        // (B,H) carried state leaving it
  float* hs; float* gates; float* h_n;   // outputs as in gru_seq_fwd_body; h_n only on the last step
  int t;
  int64_t gi_row;      // row of gi that holds (t, b = 0): t * B, or the packed offset of step t (RowOff)
};
__global__ __launch_bounds__(64) void gru_step_fwd_kernel(GruStepF d0, GruStepF d1, const int32_t* __restrict__ lengths,
                                                          int64_t hs_ld, int T, int B, int H, int last) {
  const GruStepF d = blockIdx.z == 0 ? d0 : d1;
  const int lane = threadIdx.x, i = lane & 15, q = lane >> 4;
  const int b0 = blockIdx.x * 16, ft = blockIdx.y, t = d.t;
  const int nrows = min(16, B - b0);
  const int nks = (H + 15) >> 4;
  const bool wrow_ok = 16 * ft + i < H, xrow_ok = (i < nrows) && d.h_prev != nullptr;
  const float* wr = d.w_hh + (int64_t)(16 * ft + (wrow_ok ? i : 0)) * H;
  const float* xr = d.h_prev ? d.h_prev + (int64_t)(b0 + (i < nrows ? i : 0)) * H : d.w_hh;
  // ---- every load of the step, issued back to back -----------------------------------------------------------------
  float4 wa[3][GRU_STEP_KS], xb[GRU_STEP_KS];
#pragma unroll
  for (int ks = 0; ks < GRU_STEP_KS; ++ks) {
    const int k = 16 * ks + 4 * q;
    const bool kok = ks < nks && k < H;
#pragma unroll
    for (int g = 0; g < 3; ++g) wa[g][ks] = ld4_or_zero(wr + (int64_t)g * H * H + (kok ? k : 0), kok && wrow_ok);
    xb[ks] = ld4_or_zero(xr + (kok ? k : 0), kok && xrow_ok);
  }
  const int b = b0 + i, f0 = 16 * ft + 4 * q;          // the lane's outputs: batch row i, hidden units f0 .. f0 + 3
  const bool rvalid = i < nrows, fvec = f0 + 3 < H;
  const int len = (lengths && rvalid) ? lengths[b] : T;
  const bool valid = rvalid && (t < len);
  const int64_t row = (int64_t)t * B + (rvalid ? b : b0);
  const float* gir = d.gi + (valid ? d.gi_row + b : (int64_t)0) * 3 * H;      // (packed gi: only rows inside their sequence exist)
  float4 gi4[3], bh4[3], hp4;
#pragma unroll
  for (int g = 0; g < 3; ++g) {
    gi4[g] = ld4_or_zero(gir + g * H + (fvec ? f0 : 0), fvec && valid);
    bh4[g] = ld4_or_zero(d.b_hh + g * H + (fvec ? f0 : 0), fvec);
  }
  hp4 = ld4_or_zero(d.h_prev ? d.h_prev + (int64_t)(rvalid ? b : b0) * H + (fvec ? f0 : 0) : d.w_hh, fvec && rvalid && d.h_prev);
  // ---- products ---------------------------------------------------------------------------------------------------------
  f32x4 acc[3];
#pragma unroll
  for (int g = 0; g < 3; ++g) acc[g] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int ks = 0; ks < GRU_STEP_KS; ++ks) {
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
  if (!rvalid || f0 >= H) return;
  // H % 4 == 0 (checked by the launcher): f0 + 3 < H whenever f0 < H
  const float gir_[3][4] = {{gi4[0].x, gi4[0].y, gi4[0].z, gi4[0].w}, {gi4[1].x, gi4[1].y, gi4[1].z, gi4[1].w},
                            {gi4[2].x, gi4[2].y, gi4[2].z, gi4[2].w}};
  const float bh_[3][4] = {{bh4[0].x, bh4[0].y, bh4[0].z, bh4[0].w}, {bh4[1].x, bh4[1].y, bh4[1].z, bh4[1].w},
                           {bh4[2].x, bh4[2].y, bh4[2].z, bh4[2].w}};
  const float hp_[4] = {hp4.x, hp4.y, hp4.z, hp4.w};
  float hn[4], gr[4], gz[4], gn[4], gh[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    hn[r] = hp_[r]; gr[r] = gz[r] = gn[r] = gh[r] = 0.f;
    if (valid) {
      gr[r] = sigmoidf_(gir_[0][r] + (acc[0][r] + bh_[0][r]));
      gz[r] = sigmoidf_(gir_[1][r] + (acc[1][r] + bh_[1][r]));
      gh[r] = acc[2][r] + bh_[2][r];
      gn[r] = tanhf_(gir_[2][r] + gr[r] * gh[r]);
      hn[r] = (1.0f - gz[r]) * gn[r] + gz[r] * hp_[r];
    }
  }
  *reinterpret_cast<float4*>(d.h_next + (int64_t)b * H + f0) = make_float4(hn[0], hn[1], hn[2], hn[3]);
  float* ho = d.hs + row * hs_ld + f0;                  // hs_ld may be unaligned: scalar stores
#pragma unroll
  for (int r = 0; r < 4; ++r) ho[r] = valid ? hn[r] : 0.f;       // padded positions of the output are zero
  if (d.gates) {
    float* go = d.gates + row * 4 * H + f0;
    *reinterpret_cast<float4*>(go) = make_float4(gr[0], gr[1], gr[2], gr[3]);
    *reinterpret_cast<float4*>(go + H) = make_float4(gz[0], gz[1], gz[2], gz[3]);
    *reinterpret_cast<float4*>(go + 2 * H) = make_float4(gn[0], gn[1], gn[2], gn[3]);
    *reinterpret_cast<float4*>(go + 3 * H) = make_float4(gh[0], gh[1], gh[2], gh[3]);
  }
  if (last && d.h_n) *reinterpret_cast<float4*>(d.h_n + (int64_t)b * H + f0) = make_float4(hn[0], hn[1], hn[2], hn[3]);
}

// Backward: launch `it` (0 .. T) of a direction does, for its (16 rows, 16 hidden units):
//   part A (it > 0): dh = carry + dgh[t_cur] W_hh restricted to its hidden units (t_cur = the step whose gate gradients the
//                    PREVIOUS launch wrote: all 3H columns, by all the tile workgroups -> the kernel boundary is the barrier)
//   part B (it < T): the gate gradients of the next step for its hidden units from that dh (element-wise, local), written
//                    to dgi / dgh, and carry = dh * z for the following launch.
// The carry is read and written only by the workgroup that owns the tile: one (B,H) buffer, no ping-pong.
struct GruStepB {
  const float* d_hs; const float* hs; const float* h0; const float* gates; const float* w_hh_t; const float* d_hn;
  float* carry;           // (B,H)
  float* dgi; float* dgh; float* dh0;
  int t_cur, t_next, tprev_next;   // t_cur: step of part A (-1: none); t_next: step of part B (-1: none); tprev_next: where
                                   // h_prev of step t_next lives in hs (-1: the initial state)
  int64_t dgi_row;                 // row of dgi that holds (t_next, b = 0); < 0 with `dgi_packed`: see below
  int dgi_packed;                  // dgi is the packed array: only rows inside their sequence are written
};
__global__ __launch_bounds__(64) void gru_step_bwd_kernel(GruStepB d0, GruStepB d1, const int32_t* __restrict__ lengths,
                                                          int64_t d_hs_ld, int64_t hs_ld, int T, int B, int H) {
  const GruStepB d = blockIdx.z == 0 ? d0 : d1;
  const int G = 3 * H;
  const int lane = threadIdx.x, i = lane & 15, q = lane >> 4;
  const int b0 = blockIdx.x * 16, ft = blockIdx.y;
  const int nrows = min(16, B - b0);
  const bool rvalid = i < nrows;
  const int b = b0 + (rvalid ? i : 0), f0 = 16 * ft + 4 * q;
  const bool fok = f0 < H;                              // H % 4 == 0: whole float4 or nothing
  float dh[4] = {0.f, 0.f, 0.f, 0.f};
  // ---- loads of part B that do not depend on part A, requested first ---------------------------------------------------
  const int tn = d.t_next;
  const int len = (lengths && rvalid) ? lengths[b] : T;
  const bool act = tn >= 0 && rvalid && fok && tn < len;
  const int64_t rown = (int64_t)(tn >= 0 ? tn : 0) * B + b;
  float4 g4[4], hp4, dhs4;
  {
    const float* go = d.gates + rown * 4 * H + (fok ? f0 : 0);
#pragma unroll
    for (int g = 0; g < 4; ++g) g4[g] = ld4_or_zero(go + g * H, act);
    const bool from_hs = act && d.tprev_next >= 0 && d.tprev_next < len;
    const float* hpp = from_hs ? d.hs + ((int64_t)d.tprev_next * B + b) * hs_ld + f0
                               : (d.h0 ? d.h0 + (int64_t)b * H + (fok ? f0 : 0) : d.gates);
    if (from_hs) hp4 = make_float4(hpp[0], hpp[1], hpp[2], hpp[3]);        // hs_ld may be unaligned
    else hp4 = ld4_or_zero(hpp, act && d.h0 != nullptr);
    const float* dp = d.d_hs ? d.d_hs + rown * d_hs_ld + f0 : d.gates;
    dhs4 = (act && d.d_hs) ? make_float4(dp[0], dp[1], dp[2], dp[3]) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  if (d.t_cur >= 0) {
    // ---- part A ---------------------------------------------------------------------------------------------------------
    const int nks = (G + 15) >> 4;
    const bool wrow_ok = 16 * ft + i < H;
    const float* wr = d.w_hh_t + (int64_t)(16 * ft + (wrow_ok ? i : 0)) * G;
    const float* xr = d.dgh + ((int64_t)d.t_cur * B + b) * G;
    float4 wa[GRU_STEP_KSB], xb[GRU_STEP_KSB];
#pragma unroll
    for (int ks = 0; ks < GRU_STEP_KSB; ++ks) {
      const int k = 16 * ks + 4 * q;
      const bool kok = ks < nks && k < G;
      wa[ks] = ld4_or_zero(wr + (kok ? k : 0), kok && wrow_ok);
      xb[ks] = ld4_or_zero(xr + (kok ? k : 0), kok && rvalid);
    }
    const float4 c4 = ld4_or_zero(d.carry + (int64_t)b * H + (fok ? f0 : 0), rvalid && fok);
    f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < GRU_STEP_KSB; ++ks) {
      if (ks < nks) {
        acc = mfma16(wa[ks].x, xb[ks].x, acc);
        acc = mfma16(wa[ks].y, xb[ks].y, acc);
        acc = mfma16(wa[ks].z, xb[ks].z, acc);
        acc = mfma16(wa[ks].w, xb[ks].w, acc);
      }
    }
    dh[0] = c4.x + acc[0]; dh[1] = c4.y + acc[1]; dh[2] = c4.z + acc[2]; dh[3] = c4.w + acc[3];
  } else if (d.d_hn) {
    const float4 v = ld4_or_zero(d.d_hn + (int64_t)b * H + (fok ? f0 : 0), rvalid && fok);
    dh[0] = v.x; dh[1] = v.y; dh[2] = v.z; dh[3] = v.w;
  }
  if (!rvalid || !fok) return;
  if (tn < 0) {                                          // after the last step: the gradient of the initial state
    if (d.dh0) *reinterpret_cast<float4*>(d.dh0 + (int64_t)b * H + f0) = make_float4(dh[0], dh[1], dh[2], dh[3]);
    return;
  }
  // ---- part B -----------------------------------------------------------------------------------------------------------
  const float rr_[4] = {g4[0].x, g4[0].y, g4[0].z, g4[0].w}, zz_[4] = {g4[1].x, g4[1].y, g4[1].z, g4[1].w};
  const float nn_[4] = {g4[2].x, g4[2].y, g4[2].z, g4[2].w}, gh_[4] = {g4[3].x, g4[3].y, g4[3].z, g4[3].w};
  const float hp_[4] = {hp4.x, hp4.y, hp4.z, hp4.w}, dd_[4] = {dhs4.x, dhs4.y, dhs4.z, dhs4.w};
  float g_r[4], g_z[4], g_n[4], g_hn[4], direct[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    g_r[r] = g_z[r] = g_n[r] = g_hn[r] = 0.f;
    direct[r] = dh[r];
    if (act) {
      const float dht = dh[r] + dd_[r];
      const float dn = dht * (1.0f - zz_[r]);
      const float dz = dht * (hp_[r] - nn_[r]);
      const float dnp = dn * (1.0f - nn_[r] * nn_[r]);
      g_n[r] = dnp;
      g_hn[r] = dnp * rr_[r];
      g_r[r] = dnp * gh_[r] * rr_[r] * (1.0f - rr_[r]);
      g_z[r] = dz * zz_[r] * (1.0f - zz_[r]);
      direct[r] = dht * zz_[r];
    }
  }
  float* gh_o = d.dgh + rown * G + f0;
  if (act || !d.dgi_packed) {
    float* gi_o = d.dgi + (d.dgi_row + b) * G + f0;
    *reinterpret_cast<float4*>(gi_o) = make_float4(g_r[0], g_r[1], g_r[2], g_r[3]);
    *reinterpret_cast<float4*>(gi_o + H) = make_float4(g_z[0], g_z[1], g_z[2], g_z[3]);
    *reinterpret_cast<float4*>(gi_o + 2 * H) = make_float4(g_n[0], g_n[1], g_n[2], g_n[3]);
  }
  *reinterpret_cast<float4*>(gh_o) = make_float4(g_r[0], g_r[1], g_r[2], g_r[3]);
  *reinterpret_cast<float4*>(gh_o + H) = make_float4(g_z[0], g_z[1], g_z[2], g_z[3]);
  *reinterpret_cast<float4*>(gh_o + 2 * H) = make_float4(g_hn[0], g_hn[1], g_hn[2], g_hn[3]);
  *reinterpret_cast<float4*>(d.carry + (int64_t)b * H + f0) = make_float4(direct[0], direct[1], direct[2], direct[3]);
}


// ---- small batch, generic hidden size, ONE LAUNCH FOR ALL STEPS (round 5): the step kernels above as a persistent cluster ----
// The per-step launches cost what a launch costs whatever it does: at B = 128, H = 200 a forward step is 6.3 us and a backward
// step 8.3 us for ~2 us of arithmetic each, and every launch re-requests its W_hh rows.  Here the same (16 rows x 16 hidden
// units x direction) workgroups stay resident for all T steps with their W_hh rows in registers; what a kernel boundary did --
// hand every workgroup of a row group the WHOLE state row (forward) / the whole row of hidden-side gate gradients (backward) --
// is an exchange of 8-byte self-validating granules {value, tag = step} through memory (write-through stores, sc1 loads: no flag,
// no fence; dec_persist.hpp has the measurements behind that protocol).  Records are double-buffered by step parity: a workgroup
// publishes step s + 1 only after it has read every record of step s, which exist only after every workgroup of the row group has
// read step s - 1.  A workgroup is THREE waves: in the forward each owns one gate's rows of W_hh (the same k-ordered chain per gate
// as gru_step_fwd_kernel: results bitwise equal), in the backward a third of the 3H-long contraction (three partial chains, summed
// in a fixed order: equal to summation order); waves 1 and 2 hand their accumulators to wave 0 through LDS.
// Residency: the workgroups of one (row group, direction) -- a CLUSTER -- wait for each other, so they must be co-resident: the
// launcher admits the path only while the whole grid has at most one workgroup per CU.  Every spin is bounded and latches the
// persistent kernels' fault word (dec_persist.hip) when it runs out.  Placement: see GRU_CL_DECODE_GRID.
struct GruClF {
  const float* gi; const float* w_hh; const float* b_hh; const float* h0;
  float* hs; float* gates; float* h_n;
  unsigned long long* xch;     // [2][nblk][16][Hp] granules {state value, tag}
  int reverse;
};
struct GruClB {
  const float* hn_z; const float* hn_q; const float* hn_gloss; float hn_coef;      // optional: quantiser backward folded into d_hn
  const float* d_hs; const float* hs; const float* h0; const float* gates; const float* w_hh; const float* d_hn;
  float* dgi; float* dgh; float* dh0;
  unsigned long long* xch;     // [2][nblk][16][Gp] granules {dgh value, tag}
  int reverse;
};

// Forward.  sc1 loads are served at the fabric (~25 GB/s per CU: dec_persist.hpp), so a record is swept ONCE per workgroup, with
// fully used lanes: the three waves take every third tile each and pass the fragments to each other through LDS.
constexpr int GRU_CL_KSW = (GRU_STEP_KS + 2) / 3;      // tiles a wave sweeps
// The grid is 1-D.  A CLUSTER is the nt tile workgroups of one (row group, direction).  With `xcc` given and a cluster count that is
// a multiple of 8 the cluster is the FAST index of the workgroup id, so that under round-robin placement a cluster's workgroups
// share an XCD (verified per launch: cx_cluster_on_one_xcd) and its records go through that XCD's L2: at B = 128, H = 200, T = 20 in
// the train step's graph forward 79 -> 74 us, backward 110 -> 97 (r05_aa; the backward on one XCD with write-through records: 116).
// Without `xcc` the tile is the fast index (a cluster spread over all XCDs, write-through records).
#define GRU_CL_DECODE_GRID                                                                              \
  const int ncl_ = nblk * ndir;                                                                         \
  const bool cl_fast_ = xcc != nullptr && (ncl_ & 7) == 0;                                              \
  const int cl = cl_fast_ ? (int)blockIdx.x % ncl_ : (int)blockIdx.x / nt;                              \
  const int ft = cl_fast_ ? (int)blockIdx.x / ncl_ : (int)blockIdx.x % nt;                              \
  const int rg = cl % nblk, dir = cl / nblk;
__global__ __launch_bounds__(192) void gru_cluster_fwd_kernel(GruClF d0, GruClF d1, const int32_t* __restrict__ lengths,
                                                              int64_t hs_ld, int T, int B, int H, RowOff ro, unsigned* xcc,
                                                              int nt, int nblk, int ndir, unsigned* fault) {
  __shared__ __attribute__((aligned(16))) float4 xacc[2][2][64];     // [step parity][wave 1 / 2][lane]
  __shared__ __attribute__((aligned(16))) float4 xs[GRU_STEP_KS + 2][64];      // the state row as B fragments, [k-step][lane]
  __shared__ int xcd_flag;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, i = lane & 15, q = lane >> 4;
  GRU_CL_DECODE_GRID
  const GruClF d = dir == 0 ? d0 : d1;
  const int b0 = rg * 16;
  const int nrows = min(16, B - b0);
  const int nks = nt;
  const bool rvalid = i < nrows, wrow_ok = 16 * ft + i < H;
  // ---- resident: this wave's gate rows of W_hh for the tile's 16 hidden units, as MFMA A fragments -----------------------
  const float* wr = d.w_hh + ((int64_t)wave * H + 16 * ft + (wrow_ok ? i : 0)) * H;
  float4 wa[GRU_STEP_KS];
#pragma unroll
  for (int ks = 0; ks < GRU_STEP_KS; ++ks) {
    const int k = 16 * ks + 4 * q;
    const bool kok = ks < nks && k < H;
    wa[ks] = ld4_or_zero(wr + (kok ? k : 0), kok && wrow_ok);
  }
  const int b = b0 + (rvalid ? i : 0), f0 = 16 * ft + 4 * q;     // wave 0's outputs: batch row i, hidden units f0 .. f0 + 3
  const bool fvec = f0 < H;                                       // (H % 4 == 0: a whole float4 or nothing)
  const int len = (lengths && rvalid) ? lengths[b] : T;
  float bh_[3][4], hp_[4] = {0.f, 0.f, 0.f, 0.f};
  if (wave == 0) {
#pragma unroll
    for (int g = 0; g < 3; ++g) {
      const float4 v = ld4_or_zero(d.b_hh + g * H + (fvec ? f0 : 0), fvec);
      bh_[g][0] = v.x; bh_[g][1] = v.y; bh_[g][2] = v.z; bh_[g][3] = v.w;
    }
    if (d.h0) {
      const float4 v = ld4_or_zero(d.h0 + (int64_t)b * H + (fvec ? f0 : 0), fvec && rvalid);
      hp_[0] = v.x; hp_[1] = v.y; hp_[2] = v.z; hp_[3] = v.w;
    }
  }
  // the initial state as fragments (zeros in rows / columns that do not exist: the sweeps never touch those entries)
#pragma unroll
  for (int j = 0; j < GRU_CL_KSW; ++j) {
    const int k = 16 * (wave + 3 * j) + 4 * q;
    const bool kok = k < H;
    xs[wave + 3 * j][lane] = ld4_or_zero(d.h0 ? d.h0 + (int64_t)b * H + (kok ? k : 0) : d.w_hh, kok && rvalid && d.h0 != nullptr);
  }
  const unsigned rec_granules = 256u * (unsigned)nt;
  __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc(d.xch, 0, (int)(2u * (unsigned)nblk * rec_granules * 8u), 0x00020000);
  const bool l2x = xcc != nullptr && cx_cluster_on_one_xcd(xcc + cl * nt, nt, ft, &xcd_flag, tid, fault);
  for (int s = 0; s < T; ++s) {
    const int t = d.reverse ? T - 1 - s : s;
    const bool valid = rvalid && t < len;
    float4 gi4[3];
    if (wave == 0) {      // the step's input projection: independent of the exchange, requested in front of it
      const float* gir = d.gi + (valid ? gi_row_base(ro, t, B) + b : (int64_t)0) * 3 * H;
#pragma unroll
      for (int g = 0; g < 3; ++g) gi4[g] = ld4_or_zero(gir + g * H + (fvec ? f0 : 0), fvec && valid);
    }
    // the state row entering the step: tiles wave, wave + 3, ... by this wave, to everybody through LDS
    if (s > 0)
      cx_sweep_tiles<GRU_CL_KSW>(rr, ((unsigned)((s - 1) & 1) * (unsigned)nblk + (unsigned)rg) * rec_granules, wave, 3, nt, nrows, H,
                                 (unsigned)s, &xs[0][0], lane, fault);
    lds_barrier();
    f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < GRU_STEP_KS; ++ks) {
      if (ks < nks) {
        const float4 x4 = xs[ks][lane];
        acc = mfma16(wa[ks].x, x4.x, acc);
        acc = mfma16(wa[ks].y, x4.y, acc);
        acc = mfma16(wa[ks].z, x4.z, acc);
        acc = mfma16(wa[ks].w, x4.w, acc);
      }
    }
    if (wave > 0) xacc[s & 1][wave - 1][lane] = make_float4(acc[0], acc[1], acc[2], acc[3]);
    lds_barrier();      // (also: every wave is done with xs before the next step's fragments are written)
    if (wave == 0 && rvalid && fvec) {
      const float4 az = xacc[s & 1][0][lane], an = xacc[s & 1][1][lane];
      const float a1[4] = {az.x, az.y, az.z, az.w}, a2[4] = {an.x, an.y, an.z, an.w};
      const float gir_[3][4] = {{gi4[0].x, gi4[0].y, gi4[0].z, gi4[0].w}, {gi4[1].x, gi4[1].y, gi4[1].z, gi4[1].w},
                                {gi4[2].x, gi4[2].y, gi4[2].z, gi4[2].w}};
      float hn[4], gr[4], gz[4], gn[4], gh[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        hn[r] = hp_[r]; gr[r] = gz[r] = gn[r] = gh[r] = 0.f;
        if (valid) {      // (the arithmetic of gru_step_fwd_kernel)
          gr[r] = sigmoidf_(gir_[0][r] + (acc[r] + bh_[0][r]));
          gz[r] = sigmoidf_(gir_[1][r] + (a1[r] + bh_[1][r]));
          gh[r] = a2[r] + bh_[2][r];
          gn[r] = tanhf_(gir_[2][r] + gr[r] * gh[r]);
          hn[r] = (1.0f - gz[r]) * gn[r] + gz[r] * hp_[r];
        }
      }
      if (s + 1 < T) cx_publish4(rr, ((unsigned)(s & 1) * (unsigned)nblk + (unsigned)rg) * rec_granules, ft, i, q, hn, (unsigned)(s + 1), l2x);
      const int64_t row = (int64_t)t * B + b;
      float* ho = d.hs + row * hs_ld + f0;                  // hs_ld may be unaligned: scalar stores
#pragma unroll
      for (int r = 0; r < 4; ++r) ho[r] = valid ? hn[r] : 0.f;       // padded positions of the output are zero
      if (d.gates) {
        float* go = d.gates + row * 4 * H + f0;
        *reinterpret_cast<float4*>(go) = make_float4(gr[0], gr[1], gr[2], gr[3]);
        *reinterpret_cast<float4*>(go + H) = make_float4(gz[0], gz[1], gz[2], gz[3]);
        *reinterpret_cast<float4*>(go + 2 * H) = make_float4(gn[0], gn[1], gn[2], gn[3]);
        *reinterpret_cast<float4*>(go + 3 * H) = make_float4(gh[0], gh[1], gh[2], gh[3]);
      }
      if (s == T - 1 && d.h_n) *reinterpret_cast<float4*>(d.h_n + (int64_t)b * H + f0) = make_float4(hn[0], hn[1], hn[2], hn[3]);
#pragma unroll
      for (int r = 0; r < 4; ++r) hp_[r] = hn[r];
    }
  }
}

// Backward cluster: iteration `it` (0 .. T) as gru_step_bwd_kernel's launch `it`; the carry never leaves wave 0's registers.
// The product dgh W_hh contracts over all 3H gate columns, which the tile workgroups of a row group hold 48 each.  Handing every
// workgroup the whole 3H-wide row would be three times the forward's exchange; instead a workgroup multiplies ITS 48 gate columns
// (straight from wave 0's registers: the lane layout of the gate gradients IS the MFMA B fragment of three k-steps) with its 48
// rows of W_hh for ALL hidden units and publishes that (16 x H) partial product as a row record; the owner of a hidden-unit tile
// then adds the NT partial products of its tile -- one contiguous 2 KiB block per producer -- in a fixed order (producer 0, 1, ...).
// Wave w multiplies the output tiles w, w + 3, ...
constexpr int GRU_CL_OT = (GRU_STEP_KS + 2) / 3;      // output tiles per wave
__global__ __launch_bounds__(192) void gru_cluster_bwd_kernel(GruClB d0, GruClB d1, const int32_t* __restrict__ lengths,
                                                              int64_t d_hs_ld, int64_t hs_ld, int T, int B, int H, RowOff ro,
                                                              int last_it, unsigned* xcc, int nt, int nblk, int ndir,
                                                              unsigned* fault) {
  __shared__ __attribute__((aligned(16))) float4 xs[3][64];      // this tile's gate gradients (r, z, hn) as B fragments
  __shared__ __attribute__((aligned(16))) float4 dsum3[3][64];   // [wave] its producers' partial products of this tile, [q 16 + row]
  __shared__ int xcd_flag;
  const int G = 3 * H;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, i = lane & 15, q = lane >> 4;
  GRU_CL_DECODE_GRID
  const GruClB d = dir == 0 ? d0 : d1;
  const int b0 = rg * 16;
  const int nrows = min(16, B - b0);
  const bool rvalid = i < nrows;
  const int b = b0 + (rvalid ? i : 0), f0 = 16 * ft + 4 * q;
  const bool fok = f0 < H;
  // ---- resident: rows g H + 16 ft + 4 q + e of W_hh (this tile's gate rows) at the columns of this wave's output tiles, as MFMA
  // A fragments: a = W_hh[g H + 16 ft + 4 q + e][16 ot + i] ------------------------------------------------------------------------
  float4 wt[GRU_CL_OT][3];
#pragma unroll
  for (int j = 0; j < GRU_CL_OT; ++j) {
    const int col = 16 * (wave + 3 * j) + i;
#pragma unroll
    for (int g = 0; g < 3; ++g) {
      float v[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = (fok && col < H) ? d.w_hh[((int64_t)g * H + f0 + e) * H + col] : 0.f;
      wt[j][g] = make_float4(v[0], v[1], v[2], v[3]);
    }
  }
  const int len = (lengths && rvalid) ? lengths[b] : T;
  float carry[4] = {0.f, 0.f, 0.f, 0.f};
  // records: [parity][row group][producer tile] row records of 256 nt granules
  const unsigned rec_granules = 256u * (unsigned)nt;
  __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc(d.xch, 0, (int)(2u * (unsigned)nblk * (unsigned)nt * rec_granules * 8u), 0x00020000);
  const bool rev = d.reverse != 0;
  const bool l2x = xcc != nullptr && cx_cluster_on_one_xcd(xcc + cl * nt, nt, ft, &xcd_flag, tid, fault);
  for (int it = 0; it <= last_it; ++it) {
    const int s_next = T - 1 - it;
    const int tn = s_next < 0 ? -1 : (rev ? T - 1 - s_next : s_next);
    const int tprev = (s_next <= 0) ? -1 : (rev ? tn + 1 : tn - 1);
    // ---- wave 0: the loads of part B that do not depend on part A, requested first ---------------------------------------------------
    const bool act = wave == 0 && tn >= 0 && rvalid && fok && tn < len;
    const int64_t rown = (int64_t)(tn >= 0 ? tn : 0) * B + b;
    float4 g4[4], hp4 = make_float4(0.f, 0.f, 0.f, 0.f), dhs4 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (wave == 0) {
      const float* go = d.gates + rown * 4 * H + (fok ? f0 : 0);
#pragma unroll
      for (int g = 0; g < 4; ++g) g4[g] = ld4_or_zero(go + g * H, act);
      const bool from_hs = act && tprev >= 0 && tprev < len;
      if (from_hs) {
        const float* hpp = d.hs + ((int64_t)tprev * B + b) * hs_ld + f0;      // hs_ld may be unaligned
        hp4 = make_float4(hpp[0], hpp[1], hpp[2], hpp[3]);
      } else if (act && d.h0) {
        hp4 = *reinterpret_cast<const float4*>(d.h0 + (int64_t)b * H + f0);
      }
      if (act && d.d_hs) {
        const float* dp = d.d_hs + rown * d_hs_ld + f0;
        dhs4 = make_float4(dp[0], dp[1], dp[2], dp[3]);
      }
    }
    if (it > 0) {
      // ---- every wave: the partial products of this tile from the producers wave, wave + 3, ... (producer p's block of the tile is
      // 128 chunks of 16 bytes: the lane takes chunks lane and 64 + lane), summed in ascending order, to LDS in accumulator layout ----
      const unsigned base = ((unsigned)((it - 1) & 1) * (unsigned)nblk + (unsigned)rg) * (unsigned)nt * rec_granules + (unsigned)(ft * 256);
      const unsigned tag = (unsigned)it;
      u32x4 gq[GRU_CL_OT][2];
      bool need[2];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int c = h * 64 + lane;
        need[h] = (c >> 3) < nrows && 16 * ft + 4 * ((c >> 1) & 3) < H;
      }
#pragma unroll
      for (int j = 0; j < GRU_CL_OT; ++j)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int p = wave + 3 * j;
          if (need[h] && p < nt) gq[j][h] = px_ld(rr, (base + (unsigned)p * rec_granules) * 8u + (unsigned)(h * 64 + lane) * 16u);
          else gq[j][h] = (u32x4){0u, tag, 0u, tag};
        }
      unsigned spins = 0;
      for (;;) {
        bool ok = true;
#pragma unroll
        for (int j = 0; j < GRU_CL_OT; ++j) ok &= gq[j][0][1] == tag && gq[j][0][3] == tag && gq[j][1][1] == tag && gq[j][1][3] == tag;
        if (ok) break;
        __builtin_amdgcn_s_sleep(1);
#pragma unroll
        for (int j = 0; j < GRU_CL_OT; ++j)
#pragma unroll
          for (int h = 0; h < 2; ++h)
            if (gq[j][h][1] != tag || gq[j][h][3] != tag)
              gq[j][h] = px_ld(rr, (base + (unsigned)(wave + 3 * j) * rec_granules) * 8u + (unsigned)(h * 64 + lane) * 16u);
        if (cx_give_up(spins, fault)) break;
      }
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        float s0 = 0.f, s1 = 0.f;
#pragma unroll
        for (int j = 0; j < GRU_CL_OT; ++j) {
          const unsigned lo = gq[j][h][0], hi = gq[j][h][2];
          s0 += __uint_as_float(lo);
          s1 += __uint_as_float(hi);
        }
        const int c = h * 64 + lane;
        float* dst = reinterpret_cast<float*>(&dsum3[wave][((c >> 1) & 3) * 16 + (c >> 3)]) + (c & 1) * 2;
        *reinterpret_cast<float2*>(dst) = make_float2(s0, s1);      // (zeros where nothing exists)
      }
    }
    lds_barrier();
    if (wave == 0) {
      // ---- part A: dh = carry + the NT partial products of this tile (summed by the three waves, see below) ------------------------
      float dh[4] = {0.f, 0.f, 0.f, 0.f};
      if (it > 0) {
        const float4 v0 = dsum3[0][lane], v1 = dsum3[1][lane], v2 = dsum3[2][lane];
        dh[0] = carry[0] + ((v0.x + v1.x) + v2.x); dh[1] = carry[1] + ((v0.y + v1.y) + v2.y);
        dh[2] = carry[2] + ((v0.z + v1.z) + v2.z); dh[3] = carry[3] + ((v0.w + v1.w) + v2.w);
      } else if (d.d_hn && rvalid && fok) {
        const float4 v = *reinterpret_cast<const float4*>(d.d_hn + (int64_t)b * H + f0);
        dh[0] = v.x; dh[1] = v.y; dh[2] = v.z; dh[3] = v.w;
        if (d.hn_z) {      // + gloss coef (z - q): vq_bwd_kernel's arithmetic (vq.hip), its launch saved
          const float c = d.hn_gloss[0] * d.hn_coef;
          const float4 zv = *reinterpret_cast<const float4*>(d.hn_z + (int64_t)b * H + f0);
          const float4 qv = *reinterpret_cast<const float4*>(d.hn_q + (int64_t)b * H + f0);
          dh[0] = fmaf(c, zv.x - qv.x, dh[0]); dh[1] = fmaf(c, zv.y - qv.y, dh[1]);
          dh[2] = fmaf(c, zv.z - qv.z, dh[2]); dh[3] = fmaf(c, zv.w - qv.w, dh[3]);
        }
      }
      float g_r[4] = {0.f, 0.f, 0.f, 0.f}, g_z[4] = {0.f, 0.f, 0.f, 0.f}, g_n[4] = {0.f, 0.f, 0.f, 0.f}, g_hn[4] = {0.f, 0.f, 0.f, 0.f};
      if (tn < 0) {                                          // after the last step: the gradient of the initial state
        if (d.dh0 && rvalid && fok) *reinterpret_cast<float4*>(d.dh0 + (int64_t)b * H + f0) = make_float4(dh[0], dh[1], dh[2], dh[3]);
      } else {
        // ---- part B (the arithmetic of gru_step_bwd_kernel) -------------------------------------------------------------------------
        const float rr_[4] = {g4[0].x, g4[0].y, g4[0].z, g4[0].w}, zz_[4] = {g4[1].x, g4[1].y, g4[1].z, g4[1].w};
        const float nn_[4] = {g4[2].x, g4[2].y, g4[2].z, g4[2].w}, gh_[4] = {g4[3].x, g4[3].y, g4[3].z, g4[3].w};
        const float hpv[4] = {hp4.x, hp4.y, hp4.z, hp4.w}, dd_[4] = {dhs4.x, dhs4.y, dhs4.z, dhs4.w};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          carry[r] = dh[r];
          if (act) {
            const float dht = dh[r] + dd_[r];
            const float dn = dht * (1.0f - zz_[r]);
            const float dz = dht * (hpv[r] - nn_[r]);
            const float dnp = dn * (1.0f - nn_[r] * nn_[r]);
            g_n[r] = dnp;
            g_hn[r] = dnp * rr_[r];
            g_r[r] = dnp * gh_[r] * rr_[r] * (1.0f - rr_[r]);
            g_z[r] = dz * zz_[r] * (1.0f - zz_[r]);
            carry[r] = dht * zz_[r];
          }
        }
        if (rvalid && fok) {
          float* gh_o = d.dgh + rown * G + f0;
          if (act || !ro.on) {
            float* gi_o = d.dgi + (gi_row_base(ro, tn, B) + b) * G + f0;
            *reinterpret_cast<float4*>(gi_o) = make_float4(g_r[0], g_r[1], g_r[2], g_r[3]);
            *reinterpret_cast<float4*>(gi_o + H) = make_float4(g_z[0], g_z[1], g_z[2], g_z[3]);
            *reinterpret_cast<float4*>(gi_o + 2 * H) = make_float4(g_n[0], g_n[1], g_n[2], g_n[3]);
          }
          *reinterpret_cast<float4*>(gh_o) = make_float4(g_r[0], g_r[1], g_r[2], g_r[3]);
          *reinterpret_cast<float4*>(gh_o + H) = make_float4(g_z[0], g_z[1], g_z[2], g_z[3]);
          *reinterpret_cast<float4*>(gh_o + 2 * H) = make_float4(g_hn[0], g_hn[1], g_hn[2], g_hn[3]);
        }
      }
      // the tile's hidden-side gate gradients as the B fragments of three k-steps (zeros in rows / units that do not exist)
      xs[0][lane] = make_float4(g_r[0], g_r[1], g_r[2], g_r[3]);
      xs[1][lane] = make_float4(g_z[0], g_z[1], g_z[2], g_z[3]);
      xs[2][lane] = make_float4(g_hn[0], g_hn[1], g_hn[2], g_hn[3]);
    }
    lds_barrier();
    if (it < last_it) {
      // ---- this workgroup's partial product for every hidden unit: (16 rows x 48 gate columns) x (48 x H) -----------------------
      const float4 x0 = xs[0][lane], x1 = xs[1][lane], x2 = xs[2][lane];
      const unsigned rec0 = (((unsigned)(it & 1) * (unsigned)nblk + (unsigned)rg) * (unsigned)nt + (unsigned)ft) * rec_granules;
      f32x4 acc[GRU_CL_OT];
#pragma unroll
      for (int j = 0; j < GRU_CL_OT; ++j) {
        acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (wave + 3 * j < nt) {
          acc[j] = mfma16(wt[j][0].x, x0.x, acc[j]); acc[j] = mfma16(wt[j][0].y, x0.y, acc[j]);
          acc[j] = mfma16(wt[j][0].z, x0.z, acc[j]); acc[j] = mfma16(wt[j][0].w, x0.w, acc[j]);
          acc[j] = mfma16(wt[j][1].x, x1.x, acc[j]); acc[j] = mfma16(wt[j][1].y, x1.y, acc[j]);
          acc[j] = mfma16(wt[j][1].z, x1.z, acc[j]); acc[j] = mfma16(wt[j][1].w, x1.w, acc[j]);
          acc[j] = mfma16(wt[j][2].x, x2.x, acc[j]); acc[j] = mfma16(wt[j][2].y, x2.y, acc[j]);
          acc[j] = mfma16(wt[j][2].z, x2.z, acc[j]); acc[j] = mfma16(wt[j][2].w, x2.w, acc[j]);
        }
      }
#pragma unroll
      for (int j = 0; j < GRU_CL_OT; ++j) {
        const int ot = wave + 3 * j;      // the lane's result: batch row i, hidden units 16 ot + 4 q .. + 3
        if (ot < nt && rvalid && 16 * ot + 4 * q < H) {
          const float v[4] = {acc[j][0], acc[j][1], acc[j][2], acc[j][3]};
          cx_publish4(rr, rec0, ot, i, q, v, (unsigned)(it + 1), l2x);
        }
      }
    }
    lds_barrier();      // xs is rewritten by wave 0 in the next iteration
  }
}


// ---- one GRU CELL step at small batch, input projection included (the code decoder of Part d: T = 1 per call) --------------
// g2v_gru_seq_fwd with T = 1 needs gi = x W_ih^T + b_ih from a dense-layer launch first; here a workgroup of TWO waves owns a
// (16 rows x 16 hidden units) tile: wave 0 multiplies the input side (x, W_ih), wave 1 the hidden side (h_prev, W_hh), each with
// its whole operand set requested up front as MFMA fragments (one memory round trip for both), wave 1 hands its three
// accumulators over through LDS and wave 0 finishes the gates.  Same arithmetic as g2v_linear_fwd + gru_step_fwd_kernel
// (gates bit-identical; h_new within half an ulp: the compiler contracts the final blend differently in the two kernels).  in_dim, H <= 256, multiples of 4; x_keep: dropout on the layer input (nn.GRU's).
__global__ __launch_bounds__(128) void gru_cell_fwd_kernel(const float* __restrict__ x, int in_dim,
                                                           const uint8_t* __restrict__ x_keep, float x_scale,
                                                           const float* __restrict__ h_prev, const float* __restrict__ w_ih,
                                                           const float* __restrict__ w_hh, const float* __restrict__ b_ih,
                                                           const float* __restrict__ b_hh, float* __restrict__ h_new,
                                                           float* __restrict__ gates, int B, int H) {
  __shared__ __attribute__((aligned(16))) float hand[3][256];
  const int lane = threadIdx.x & 63, side = threadIdx.x >> 6, i = lane & 15, q = lane >> 4;
  const int b0 = blockIdx.x * 16, ft = blockIdx.y;
  const int nrows = min(16, B - b0);
  const int C = side == 0 ? in_dim : H;
  const float* __restrict__ W = side == 0 ? w_ih : w_hh;
  const float* __restrict__ src = side == 0 ? x : h_prev;
  const int nks = (C + 15) >> 4;
  const bool wrow_ok = 16 * ft + i < H, xrow_ok = i < nrows;
  const float* wr = W + (int64_t)(16 * ft + (wrow_ok ? i : 0)) * C;
  const float* xr = src + (int64_t)(b0 + (xrow_ok ? i : 0)) * C;
  const uint8_t* kr = (side == 0 && x_keep) ? x_keep + (int64_t)(b0 + (xrow_ok ? i : 0)) * C : nullptr;
  float4 wa[3][GRU_STEP_KS], xb[GRU_STEP_KS];
  uint32_t kp[GRU_STEP_KS];
#pragma unroll
  for (int ks = 0; ks < GRU_STEP_KS; ++ks) {
    const int k = 16 * ks + 4 * q;
    const bool kok = ks < nks && k < C;
#pragma unroll
    for (int g = 0; g < 3; ++g) wa[g][ks] = ld4_or_zero(wr + (int64_t)g * H * C + (kok ? k : 0), kok && wrow_ok);
    xb[ks] = ld4_or_zero(xr + (kok ? k : 0), kok && xrow_ok);
    kp[ks] = (kr && kok) ? *reinterpret_cast<const uint32_t*>(kr + k) : 0x01010101u;
  }
  const int b = b0 + i, f0 = 16 * ft + 4 * q;
  const bool rvalid = i < nrows, fok = f0 < H;             // H % 4 == 0: whole float4 or nothing
  float4 bi4[3], bh4[3], hp4;
  if (side == 0) {
#pragma unroll
    for (int g = 0; g < 3; ++g) {
      bi4[g] = ld4_or_zero(b_ih + g * H + (fok ? f0 : 0), fok);
      bh4[g] = ld4_or_zero(b_hh + g * H + (fok ? f0 : 0), fok);
    }
    hp4 = ld4_or_zero(h_prev + (int64_t)(rvalid ? b : b0) * H + (fok ? f0 : 0), fok && rvalid);
  }
  f32x4 acc[3];
#pragma unroll
  for (int g = 0; g < 3; ++g) acc[g] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int ks = 0; ks < GRU_STEP_KS; ++ks) {
    if (ks < nks) {
      float4 xv = xb[ks];
      if (side == 0 && x_keep) {
        xv.x = (kp[ks] & 0xffu) ? xv.x * x_scale : 0.f;
        xv.y = (kp[ks] & 0xff00u) ? xv.y * x_scale : 0.f;
        xv.z = (kp[ks] & 0xff0000u) ? xv.z * x_scale : 0.f;
        xv.w = (kp[ks] & 0xff000000u) ? xv.w * x_scale : 0.f;
      }
#pragma unroll
      for (int g = 0; g < 3; ++g) {
        acc[g] = mfma16(wa[g][ks].x, xv.x, acc[g]);
        acc[g] = mfma16(wa[g][ks].y, xv.y, acc[g]);
        acc[g] = mfma16(wa[g][ks].z, xv.z, acc[g]);
        acc[g] = mfma16(wa[g][ks].w, xv.w, acc[g]);
      }
    }
  }
  if (side == 1) {
#pragma unroll
    for (int g = 0; g < 3; ++g) *reinterpret_cast<float4*>(&hand[g][lane * 4]) = make_float4(acc[g][0], acc[g][1], acc[g][2], acc[g][3]);
  }
  __syncthreads();
  if (side == 1 || !rvalid || !fok) return;
  float ah[3][4];
#pragma unroll
  for (int g = 0; g < 3; ++g) {
    const float4 v = *reinterpret_cast<const float4*>(&hand[g][lane * 4]);
    ah[g][0] = v.x; ah[g][1] = v.y; ah[g][2] = v.z; ah[g][3] = v.w;
  }
  const float bi_[3][4] = {{bi4[0].x, bi4[0].y, bi4[0].z, bi4[0].w}, {bi4[1].x, bi4[1].y, bi4[1].z, bi4[1].w},
                           {bi4[2].x, bi4[2].y, bi4[2].z, bi4[2].w}};
  const float bh_[3][4] = {{bh4[0].x, bh4[0].y, bh4[0].z, bh4[0].w}, {bh4[1].x, bh4[1].y, bh4[1].z, bh4[1].w},
                           {bh4[2].x, bh4[2].y, bh4[2].z, bh4[2].w}};
  const float hp_[4] = {hp4.x, hp4.y, hp4.z, hp4.w};
  float hn[4], gr[4], gz[4], gn[4], gh[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    gr[r] = sigmoidf_((acc[0][r] + bi_[0][r]) + (ah[0][r] + bh_[0][r]));
    gz[r] = sigmoidf_((acc[1][r] + bi_[1][r]) + (ah[1][r] + bh_[1][r]));
    gh[r] = ah[2][r] + bh_[2][r];
    gn[r] = tanhf_(__fmaf_rn(gr[r], gh[r], acc[2][r] + bi_[2][r]));      // fused like gi_n + r * gh_n in gru_step_fwd_kernel
    hn[r] = (1.0f - gz[r]) * gn[r] + gz[r] * hp_[r];
  }
  *reinterpret_cast<float4*>(h_new + (int64_t)b * H + f0) = make_float4(hn[0], hn[1], hn[2], hn[3]);
  if (gates) {
    float* go = gates + (int64_t)b * 4 * H + f0;
    *reinterpret_cast<float4*>(go) = make_float4(gr[0], gr[1], gr[2], gr[3]);
    *reinterpret_cast<float4*>(go + H) = make_float4(gz[0], gz[1], gz[2], gz[3]);
    *reinterpret_cast<float4*>(go + 2 * H) = make_float4(gn[0], gn[1], gn[2], gn[3]);
    *reinterpret_cast<float4*>(go + 3 * H) = make_float4(gh[0], gh[1], gh[2], gh[3]);
  }
}

// Backward of the cell, element-wise half: from d_h = d_h_a + d_h_b (gradient arriving at h_new: from above and from the next
// step) the gate gradients dgi = (dr, dz, dn), dgh = (dr, dz, dn * r) and the direct path d_h * z (the base of d_hprev);
// the two products d_hprev += dgh W_hh and dx = dgi W_ih follow in ONE launch of the dense small-M kernel (linear.hip).
__global__ __launch_bounds__(256) void gru_cell_gates_bwd_kernel(const float* __restrict__ d_h_a, const float* __restrict__ d_h_b,
                                                                 const float* __restrict__ gates, const float* __restrict__ h_prev,
                                                                 float* __restrict__ dgi, float* __restrict__ dgh,
                                                                 float* __restrict__ direct, int B, int H) {
  const int H4 = H >> 2;
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= B * H4) return;
  const int b = e / H4, f0 = 4 * (e - b * H4);
  const float* go = gates + (int64_t)b * 4 * H + f0;
  const float4 r4 = *reinterpret_cast<const float4*>(go), z4 = *reinterpret_cast<const float4*>(go + H);
  const float4 n4 = *reinterpret_cast<const float4*>(go + 2 * H), g4 = *reinterpret_cast<const float4*>(go + 3 * H);
  const float4 hp = *reinterpret_cast<const float4*>(h_prev + (int64_t)b * H + f0);
  float4 da = make_float4(0.f, 0.f, 0.f, 0.f), db = da;
  if (d_h_a) da = *reinterpret_cast<const float4*>(d_h_a + (int64_t)b * H + f0);
  if (d_h_b) db = *reinterpret_cast<const float4*>(d_h_b + (int64_t)b * H + f0);
  const float rr[4] = {r4.x, r4.y, r4.z, r4.w}, zz[4] = {z4.x, z4.y, z4.z, z4.w}, nn[4] = {n4.x, n4.y, n4.z, n4.w};
  const float gh[4] = {g4.x, g4.y, g4.z, g4.w}, hp_[4] = {hp.x, hp.y, hp.z, hp.w};
  const float d0[4] = {da.x, da.y, da.z, da.w}, d1[4] = {db.x, db.y, db.z, db.w};
  float g_r[4], g_z[4], g_n[4], g_hn[4], dir[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const float dht = d1[r] + d0[r];                    // carry + gradient from above (the order of gru_step_bwd_kernel)
    const float dn = dht * (1.0f - zz[r]);
    const float dz = dht * (hp_[r] - nn[r]);
    const float dnp = dn * (1.0f - nn[r] * nn[r]);
    g_n[r] = dnp;
    g_hn[r] = dnp * rr[r];
    g_r[r] = dnp * gh[r] * rr[r] * (1.0f - rr[r]);
    g_z[r] = dz * zz[r] * (1.0f - zz[r]);
    dir[r] = dht * zz[r];
  }
  float* gi_o = dgi + (int64_t)b * 3 * H + f0;
  float* gh_o = dgh + (int64_t)b * 3 * H + f0;
  *reinterpret_cast<float4*>(gi_o) = make_float4(g_r[0], g_r[1], g_r[2], g_r[3]);
  *reinterpret_cast<float4*>(gi_o + H) = make_float4(g_z[0], g_z[1], g_z[2], g_z[3]);
  *reinterpret_cast<float4*>(gi_o + 2 * H) = make_float4(g_n[0], g_n[1], g_n[2], g_n[3]);
  *reinterpret_cast<float4*>(gh_o) = make_float4(g_r[0], g_r[1], g_r[2], g_r[3]);
  *reinterpret_cast<float4*>(gh_o + H) = make_float4(g_z[0], g_z[1], g_z[2], g_z[3]);
  *reinterpret_cast<float4*>(gh_o + 2 * H) = make_float4(g_hn[0], g_hn[1], g_hn[2], g_hn[3]);
  *reinterpret_cast<float4*>(direct + (int64_t)b * H + f0) = make_float4(dir[0], dir[1], dir[2], dir[3]);
}

// =====================================================================================================
// Fast path, H == 64 (the BASELINE shape): dims are compile-time, each wave owns ONE 16-feature tile whose
// W_hh fragments (3 gates x 4 k-steps x float4 = 48 VGPRs) stay in registers for the whole sequence, the gi
// rows of step s+1 are prefetched while step s runs, epilogues are 16-byte vectors, barriers order LDS only,
// and both directions of a bidirectional layer run in ONE launch (blockIdx.y) so that 2 workgroups share a CU.
// =====================================================================================================
struct GruDirF {
  const float* gi; const float* p_hh; const float* b_hh; const float* h0;
  float* hs; float* h_n; float* gates;
  int reverse;
  const float* x; const float* p_ih; const float* b_ih;   // FUSE_IN: gi = x W_ih^T + b_ih is computed in the kernel
};
struct GruDirB {
  const float* d_hs; const float* d_hn; const float* hs; const float* h0; const float* gates; const float* p_hh_t;
  float* dgi; float* dgh; float* dh0;
  int reverse;
  const float* p_ih_t; float* dx;    // FUSE_DX: dx_t = dgi_t W_ih is produced in the kernel
  const float* x; float* wslab;      // FUSE_W: the layer input (T,B,H) and this direction's partial dW / db slabs
  const float* hn_z; const float* hn_q; const float* hn_gloss; float hn_coef;   // optional: quantiser backward folded into d_hn
};

// FUSE_IN: the input projection gi_t = x_t W_ih^T + b_ih (input width == H) is computed in the kernel, one step ahead
// of its use: the recurrence is latency-bound (one dependent MFMA chain + gate math per step), so the 48 extra,
// h-independent MFMAs per step ride in otherwise idle matrix-pipe slots, and the (T,B,3H) gi array -- its GEMM, its
// HBM write and its read -- disappears.  W_ih fragments live in registers next to W_hh for all T steps.
// (Round 3 also built the layer in FRONT of this one -- the encoder's in_layer, Linear(D -> H) -- into this kernel, two steps
// ahead of its use: parity-green and +0.1 ms per step, the 36 extra MFMAs per wave and step cost more inside a kernel whose two
// waves per SIMD already share the SIMD serially than the 37 us HBM-bound launch they replaced.  Removed in round 4.)
template <int HS, bool FUSE_IN>
__device__ __forceinline__ void gru_fwd_fast_body(const GruDirF& d0, const GruDirF& d1, const int32_t* __restrict__ lengths,
                                                  int64_t hs_ld, int T, int B) {
  constexpr int H = HS, KS = H / 16, NT = H / 16, ldx = H + 4, G = 3 * H;
  static_assert(NT == 4, "one feature tile per wave");
  __shared__ __attribute__((aligned(16))) float hb[2][16 * ldx];
  __shared__ __attribute__((aligned(16))) float xb[FUSE_IN ? 2 : 1][FUSE_IN ? 16 * ldx : 4];
  const GruDirF d = blockIdx.y ? d1 : d0;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i = lane & 15, q = lane >> 4;
  const int b0 = blockIdx.x * 16, nrows = min(16, B - b0), b = b0 + i;
  const bool rvalid = i < nrows;
  const int f0 = 16 * wave + 4 * q;
  float4 wf[3][KS];
#pragma unroll
  for (int g = 0; g < 3; ++g)
#pragma unroll
    for (int s = 0; s < KS; ++s)
      wf[g][s] = *reinterpret_cast<const float4*>(d.p_hh + ((int64_t)((g * NT + wave) * KS + s) * 64 + lane) * 4);
  float4 bh[3];
#pragma unroll
  for (int g = 0; g < 3; ++g) bh[g] = *reinterpret_cast<const float4*>(d.b_hh + g * H + f0);
  const int len = (lengths && rvalid) ? lengths[b] : T;
  for (int e = tid; e < 16 * ldx; e += 256) {
    const int r = e / ldx, k = e - r * ldx;
    hb[0][e] = (d.h0 && r < nrows && k < H) ? d.h0[(int64_t)(b0 + r) * H + k] : 0.f;
    hb[1][e] = 0.f;
  }
  float4 gin[3];
  float4 wi[3][FUSE_IN ? KS : 1], bi[3];
  // x tile staging: thread -> (row xr_, 4 features xc_) of the 16 x H tile (16 * H / 4 = 256 float4 = one per thread)
  const int xr_ = tid >> 4, xc_ = (tid & 15) * 4;
  auto step_t = [&](int s) { return d.reverse ? (T - 1 - s) : s; };
  auto load_x = [&](int s) {
    const bool ok = (s < T) && (xr_ < nrows);
    const int t = ok ? step_t(s) : 0;
    const float4 v = *reinterpret_cast<const float4*>(d.x + ((int64_t)t * B + b0 + (ok ? xr_ : 0)) * H + xc_);
    return ok ? v : make_float4(0.f, 0.f, 0.f, 0.f);
  };
  auto project = [&](const float* xs, float4 (&out)[3]) {     // out = W_ih x + b_ih for this lane's 4 features
    f32x4 a[3];
#pragma unroll
    for (int g = 0; g < 3; ++g) a[g] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const float* xx = xs + i * ldx + 4 * q;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const float4 xv = *reinterpret_cast<const float4*>(xx + 16 * ks);
#pragma unroll
      for (int g = 0; g < 3; ++g) a[g] = mfma16(wi[g][FUSE_IN ? ks : 0].x, xv.x, a[g]);
#pragma unroll
      for (int g = 0; g < 3; ++g) a[g] = mfma16(wi[g][FUSE_IN ? ks : 0].y, xv.y, a[g]);
#pragma unroll
      for (int g = 0; g < 3; ++g) a[g] = mfma16(wi[g][FUSE_IN ? ks : 0].z, xv.z, a[g]);
#pragma unroll
      for (int g = 0; g < 3; ++g) a[g] = mfma16(wi[g][FUSE_IN ? ks : 0].w, xv.w, a[g]);
    }
#pragma unroll
    for (int g = 0; g < 3; ++g) out[g] = make_float4(a[g][0] + bi[g].x, a[g][1] + bi[g].y, a[g][2] + bi[g].z, a[g][3] + bi[g].w);
  };
  if constexpr (FUSE_IN) {
#pragma unroll
    for (int g = 0; g < 3; ++g) {
#pragma unroll
      for (int s = 0; s < KS; ++s)
        wi[g][s] = *reinterpret_cast<const float4*>(d.p_ih + ((int64_t)((g * NT + wave) * KS + s) * 64 + lane) * 4);
      bi[g] = *reinterpret_cast<const float4*>(d.b_ih + g * H + f0);
    }
    *reinterpret_cast<float4*>(&xb[0][xr_ * ldx + xc_]) = load_x(0);
    *reinterpret_cast<float4*>(&xb[1][xr_ * ldx + xc_]) = load_x(1);
    __syncthreads();
    project(xb[0], gin);                                     // gi of the first step
  } else {
    const int t0 = d.reverse ? T - 1 : 0;
#pragma unroll
    for (int g = 0; g < 3; ++g)
      gin[g] = rvalid ? *reinterpret_cast<const float4*>(d.gi + ((int64_t)t0 * B + b) * G + g * H + f0)
                      : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  __syncthreads();
  int cur = 0;
  for (int s = 0; s < T; ++s) {
    const int t = d.reverse ? (T - 1 - s) : s;
    float4 gic[3];
#pragma unroll
    for (int g = 0; g < 3; ++g) gic[g] = gin[g];
    float4 xnext = make_float4(0.f, 0.f, 0.f, 0.f);
    if constexpr (FUSE_IN) {
      xnext = load_x(s + 2);                                 // consumed at the end of this iteration
    } else if (s + 1 < T && rvalid) {
      const int tn = d.reverse ? (T - 2 - s) : (s + 1);
#pragma unroll
      for (int g = 0; g < 3; ++g) gin[g] = *reinterpret_cast<const float4*>(d.gi + ((int64_t)tn * B + b) * G + g * H + f0);
    }
    f32x4 acc[3];
#pragma unroll
    for (int g = 0; g < 3; ++g) acc[g] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const float* hx = hb[cur] + i * ldx + 4 * q;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const float4 xb = *reinterpret_cast<const float4*>(hx + 16 * ks);
#pragma unroll
      for (int g = 0; g < 3; ++g) acc[g] = mfma16(wf[g][ks].x, xb.x, acc[g]);
#pragma unroll
      for (int g = 0; g < 3; ++g) acc[g] = mfma16(wf[g][ks].y, xb.y, acc[g]);
#pragma unroll
      for (int g = 0; g < 3; ++g) acc[g] = mfma16(wf[g][ks].z, xb.z, acc[g]);
#pragma unroll
      for (int g = 0; g < 3; ++g) acc[g] = mfma16(wf[g][ks].w, xb.w, acc[g]);
    }
    if constexpr (FUSE_IN) project(xb[(s + 1) & 1], gin);     // gi of step s+1: independent of h
    const float4 hp4 = *reinterpret_cast<const float4*>(hb[cur] + i * ldx + f0);
    const bool valid = rvalid && (t < len);
    const float hp[4] = {hp4.x, hp4.y, hp4.z, hp4.w};
    const float ir[4] = {gic[0].x, gic[0].y, gic[0].z, gic[0].w}, iz[4] = {gic[1].x, gic[1].y, gic[1].z, gic[1].w},
                in_[4] = {gic[2].x, gic[2].y, gic[2].z, gic[2].w};
    const float br[4] = {bh[0].x, bh[0].y, bh[0].z, bh[0].w}, bz[4] = {bh[1].x, bh[1].y, bh[1].z, bh[1].w},
                bn[4] = {bh[2].x, bh[2].y, bh[2].z, bh[2].w};
    float hn[4], gr_[4], gz_[4], gn_[4], gh_[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float rr = sigmoidf_(ir[r] + (acc[0][r] + br[r]));
      const float zz = sigmoidf_(iz[r] + (acc[1][r] + bz[r]));
      const float ghn = acc[2][r] + bn[r];
      const float nn = tanhf_(in_[r] + rr * ghn);
      const float hnew = (1.0f - zz) * nn + zz * hp[r];
      hn[r] = valid ? hnew : hp[r];        // selects, not branches: the gate math runs for every lane
      gr_[r] = valid ? rr : 0.f; gz_[r] = valid ? zz : 0.f; gn_[r] = valid ? nn : 0.f; gh_[r] = valid ? ghn : 0.f;
    }
    *reinterpret_cast<float4*>(hb[cur ^ 1] + i * ldx + f0) = make_float4(hn[0], hn[1], hn[2], hn[3]);
    // x of step s+2 into LDS BEFORE this step's global stores are issued: the wait for `xnext` is a vmcnt(0) (hipcc cannot count
    // across the loop's branches), and vector-memory operations retire in order -- behind the stores it drained the five 16-byte
    // stores just issued, a store round trip per step (seen in the ISA, round 4); in front of them only last step's, long done
    if constexpr (FUSE_IN) *reinterpret_cast<float4*>(&xb[s & 1][xr_ * ldx + xc_]) = xnext;
    if (rvalid) {
      const int64_t row = (int64_t)t * B + b;
      *reinterpret_cast<float4*>(d.hs + row * hs_ld + f0) =
          valid ? make_float4(hn[0], hn[1], hn[2], hn[3]) : make_float4(0.f, 0.f, 0.f, 0.f);
      if (d.gates) {
        float* go = d.gates + row * 4 * H + f0;
        *reinterpret_cast<float4*>(go) = make_float4(gr_[0], gr_[1], gr_[2], gr_[3]);
        *reinterpret_cast<float4*>(go + H) = make_float4(gz_[0], gz_[1], gz_[2], gz_[3]);
        *reinterpret_cast<float4*>(go + 2 * H) = make_float4(gn_[0], gn_[1], gn_[2], gn_[3]);
        *reinterpret_cast<float4*>(go + 3 * H) = make_float4(gh_[0], gh_[1], gh_[2], gh_[3]);
      }
    }
    lds_barrier();
    cur ^= 1;
  }
  if (d.h_n && rvalid) *reinterpret_cast<float4*>(d.h_n + (int64_t)b * H + f0) = *reinterpret_cast<const float4*>(hb[cur] + i * ldx + f0);
}

template <int HS, bool FUSE_IN>
__global__ __launch_bounds__(256) void gru_fwd_fast_kernel(GruDirF d0, GruDirF d1, const int32_t* __restrict__ lengths,
                                                           int64_t hs_ld, int T, int B) {
  gru_fwd_fast_body<HS, FUSE_IN>(d0, d1, lengths, hs_ld, T, B);
}

// FUSE_DX: the input gradient dx_t = dgi_t W_ih (input width == H) is produced here as a second, independent MFMA chain
// next to the dh chain (which is a 48-deep DEPENDENT chain: the extra chain fills its issue bubbles); the separate
// (T*B,3H) x (3H,H) product and its re-read of dgi disappear.  dgi = (g_r, g_z, g_n) and dgh = (g_r, g_z, g_hn) share
// their first 2H columns in LDS; the n-gate columns of dgi sit in a second small tile.
// FUSE_W (with FUSE_DX): the direction's two weight gradients dW_hh = sum dgh^T h_prev, dW_ih = sum dgi^T x (and the bias
// gradients) are accumulated here as well -- each wave its own 16 hidden-unit columns of both 192 x 64 matrices (24
// accumulator tiles; the kernel has the registers: 104 -> ~210, still two workgroups per CU), A = the gate-gradient tile
// transposed, B = h_prev / x through a per-wave 16 x 16 LDS transpose, 96 more MFMAs per wave and step.  The separate batched
// weight-gradient launch of the encoder (135 us at the BASELINE shape, ON the critical path behind this kernel) and the dgi /
// dgh arrays (4 x 107 MB written, then read back) disappear; the price is a matrix pipe that is now the limit of the step.
template <int HS, bool FUSE_DX, int FUSE_W = 0>      // FUSE_W: 0 none, 1 W_hh only (fits two workgroups per CU), 2 W_hh and W_ih
__global__ __launch_bounds__(256, FUSE_W == 2 ? 1 : 2) void gru_bwd_fast_kernel(GruDirB d0, GruDirB d1, const int32_t* __restrict__ lengths,
                                                           int64_t d_hs_ld, int64_t hs_ld, int T, int B) {
  static_assert(!FUSE_W || FUSE_DX, "fused weight gradients come with the fused input gradient");
  // (Round 4, measured and dropped: s_setprio(3) here, so that this latency chain issues ahead of the weight-gradient products
  //  co-resident on the CU -- the step got 50 us SLOWER (1.573 -> 1.624 ms): what the chain gains, the products lose twice.)
  constexpr int H = HS, G = 3 * H, KSG = G / 16, ldg = G + 4, ldn = H + 4;
  __shared__ __attribute__((aligned(16))) float Gs[2][16 * ldg];
  __shared__ __attribute__((aligned(16))) float Gn[FUSE_DX ? 2 : 1][FUSE_DX ? 16 * ldn : 4];
  __shared__ float Trs[FUSE_W ? 4 : 1][FUSE_W ? 2 * 16 * 17 : 1];       // per wave: h_prev and x, 16 x 16 each
  const GruDirB d = blockIdx.y ? d1 : d0;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i = lane & 15, q = lane >> 4;
  const int b0 = blockIdx.x * 16, nrows = min(16, B - b0), b = b0 + i;
  const bool rvalid = i < nrows;
  const int f0 = 16 * wave + 4 * q;
  float4 wf[KSG];
#pragma unroll
  for (int s = 0; s < KSG; ++s) wf[s] = *reinterpret_cast<const float4*>(d.p_hh_t + ((int64_t)(wave * KSG + s) * 64 + lane) * 4);
  float4 wx[FUSE_DX ? KSG : 1];
  if constexpr (FUSE_DX) {
#pragma unroll
    for (int s = 0; s < KSG; ++s) wx[s] = *reinterpret_cast<const float4*>(d.p_ih_t + ((int64_t)(wave * KSG + s) * 64 + lane) * 4);
  }
  const int len = (lengths && rvalid) ? lengths[b] : T;
  const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
  float4 dh = (d.d_hn && rvalid) ? *reinterpret_cast<const float4*>(d.d_hn + (int64_t)b * H + f0) : z4;
  if (d.hn_z && rvalid) {      // + gloss coef (z - q): vq_bwd_kernel's arithmetic (vq.hip), its launch saved
    const float c = d.hn_gloss[0] * d.hn_coef;
    const float4 zv = *reinterpret_cast<const float4*>(d.hn_z + (int64_t)b * H + f0);
    const float4 qv = *reinterpret_cast<const float4*>(d.hn_q + (int64_t)b * H + f0);
    dh.x = fmaf(c, zv.x - qv.x, dh.x); dh.y = fmaf(c, zv.y - qv.y, dh.y);
    dh.z = fmaf(c, zv.z - qv.z, dh.z); dh.w = fmaf(c, zv.w - qv.w, dh.w);
  }
  for (int e = tid; e < 2 * 16 * ldg; e += 256) (&Gs[0][0])[e] = 0.f;

  auto load_step = [&](int s, float4& gr4, float4& gz4, float4& gn4, float4& gh4, float4& hp4, float4& dhs4, float4& x4) {
    const int t = d.reverse ? (T - 1 - s) : s;
    const int tprev = d.reverse ? t + 1 : t - 1;
    gr4 = gz4 = gn4 = gh4 = hp4 = dhs4 = x4 = z4;
    if (rvalid && t < len) {
      const int64_t row = (int64_t)t * B + b;
      const float* go = d.gates + row * 4 * H + f0;
      if constexpr (FUSE_W == 2) x4 = *reinterpret_cast<const float4*>(d.x + row * H + f0);
      gr4 = *reinterpret_cast<const float4*>(go);
      gz4 = *reinterpret_cast<const float4*>(go + H);
      gn4 = *reinterpret_cast<const float4*>(go + 2 * H);
      gh4 = *reinterpret_cast<const float4*>(go + 3 * H);
      if (d.d_hs) dhs4 = *reinterpret_cast<const float4*>(d.d_hs + row * d_hs_ld + f0);
      if (s == 0 || tprev >= len) {
        if (d.h0) hp4 = *reinterpret_cast<const float4*>(d.h0 + (int64_t)b * H + f0);
      } else {
        hp4 = *reinterpret_cast<const float4*>(d.hs + ((int64_t)tprev * B + b) * hs_ld + f0);
      }
    }
  };
  float4 n_r, n_z, n_n, n_h, n_hp, n_dhs, n_x;
  load_step(T - 1, n_r, n_z, n_n, n_h, n_hp, n_dhs, n_x);
  f32x4 whh[FUSE_W ? 12 : 1], wih[FUSE_W == 2 ? 12 : 1];
  float dbh = 0.f, dbn = 0.f;            // column sums: thread tid < 192 <-> (g_r, g_z, g_hn) column, tid < 64 <-> g_n column
#pragma unroll
  for (int g = 0; g < (FUSE_W ? 12 : 1); ++g) whh[g] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int g = 0; g < (FUSE_W == 2 ? 12 : 1); ++g) wih[g] = (f32x4){0.f, 0.f, 0.f, 0.f};
  __syncthreads();
  int cur = 0;
  for (int s = T - 1; s >= 0; --s) {
    const int t = d.reverse ? (T - 1 - s) : s;
    const float4 c_r = n_r, c_z = n_z, c_n = n_n, c_h = n_h, c_hp = n_hp, c_dhs = n_dhs, c_x = n_x;
    if (s > 0) load_step(s - 1, n_r, n_z, n_n, n_h, n_hp, n_dhs, n_x);
    const bool valid = rvalid && (t < len);
    const float rr[4] = {c_r.x, c_r.y, c_r.z, c_r.w}, zz[4] = {c_z.x, c_z.y, c_z.z, c_z.w}, nn[4] = {c_n.x, c_n.y, c_n.z, c_n.w},
                gh[4] = {c_h.x, c_h.y, c_h.z, c_h.w}, hp[4] = {c_hp.x, c_hp.y, c_hp.z, c_hp.w},
                ds[4] = {c_dhs.x, c_dhs.y, c_dhs.z, c_dhs.w};
    float dhv[4] = {dh.x, dh.y, dh.z, dh.w};
    float g_r[4], g_z[4], g_n[4], g_hn[4], direct[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      if (valid) {
        const float dd = dhv[r] + ds[r];
        const float dn = dd * (1.0f - zz[r]);
        const float dz = dd * (hp[r] - nn[r]);
        const float dnp = dn * (1.0f - nn[r] * nn[r]);
        g_n[r] = dnp;
        g_hn[r] = dnp * rr[r];
        g_r[r] = dnp * gh[r] * rr[r] * (1.0f - rr[r]);
        g_z[r] = dz * zz[r] * (1.0f - zz[r]);
        direct[r] = dd * zz[r];
      } else {
        g_n[r] = g_hn[r] = g_r[r] = g_z[r] = 0.f;
        direct[r] = dhv[r];
      }
    }
    const float4 vr = make_float4(g_r[0], g_r[1], g_r[2], g_r[3]), vz = make_float4(g_z[0], g_z[1], g_z[2], g_z[3]),
                 vn = make_float4(g_n[0], g_n[1], g_n[2], g_n[3]), vh = make_float4(g_hn[0], g_hn[1], g_hn[2], g_hn[3]);
    if (rvalid) {                          // (a fused weight gradient consumes the tile from LDS: its (T,B,3H) array is not written)
      const int64_t row = (int64_t)t * B + b;
      if constexpr (FUSE_W < 2) {
        float* o1 = d.dgi + row * G + f0;
        *reinterpret_cast<float4*>(o1) = vr; *reinterpret_cast<float4*>(o1 + H) = vz; *reinterpret_cast<float4*>(o1 + 2 * H) = vn;
      }
      if constexpr (FUSE_W < 1) {
        float* o2 = d.dgh + row * G + f0;
        *reinterpret_cast<float4*>(o2) = vr; *reinterpret_cast<float4*>(o2 + H) = vz; *reinterpret_cast<float4*>(o2 + 2 * H) = vh;
      }
    }
    if constexpr (FUSE_W) {                // h_prev and x of this step, accumulator layout -> B-fragment layout (wave-private)
      float* tr = Trs[wave];
      tr[i * 17 + 4 * q + 0] = c_hp.x; tr[i * 17 + 4 * q + 1] = c_hp.y; tr[i * 17 + 4 * q + 2] = c_hp.z; tr[i * 17 + 4 * q + 3] = c_hp.w;
      if constexpr (FUSE_W == 2) {
        tr[272 + i * 17 + 4 * q + 0] = c_x.x; tr[272 + i * 17 + 4 * q + 1] = c_x.y; tr[272 + i * 17 + 4 * q + 2] = c_x.z; tr[272 + i * 17 + 4 * q + 3] = c_x.w;
      }
    }
    float* gs = Gs[cur] + i * ldg;
    *reinterpret_cast<float4*>(gs + f0) = vr;
    *reinterpret_cast<float4*>(gs + H + f0) = vz;
    *reinterpret_cast<float4*>(gs + 2 * H + f0) = vh;
    if constexpr (FUSE_DX) *reinterpret_cast<float4*>(&Gn[cur][i * ldn + f0]) = vn;
    lds_barrier();
    f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
    const float* gx = gs + 4 * q;
    if constexpr (FUSE_DX) {
      f32x4 accx = (f32x4){0.f, 0.f, 0.f, 0.f};
      const float* gnx = &Gn[cur][i * ldn + 4 * q];
#pragma unroll
      for (int ks = 0; ks < KSG; ++ks) {
        const float4 xb = *reinterpret_cast<const float4*>(gx + 16 * ks);
        // the r and z columns are common to dgi and dgh; the last H columns are g_hn (dh chain) / g_n (dx chain)
        const float4 xn = (ks < 2 * H / 16) ? xb : *reinterpret_cast<const float4*>(gnx + 16 * (ks - 2 * H / 16));
        acc = mfma16(wf[ks].x, xb.x, acc);
        accx = mfma16(wx[ks].x, xn.x, accx);
        acc = mfma16(wf[ks].y, xb.y, acc);
        accx = mfma16(wx[ks].y, xn.y, accx);
        acc = mfma16(wf[ks].z, xb.z, acc);
        accx = mfma16(wx[ks].z, xn.z, accx);
        acc = mfma16(wf[ks].w, xb.w, acc);
        accx = mfma16(wx[ks].w, xn.w, accx);
      }
      if (rvalid)
        *reinterpret_cast<float4*>(d.dx + ((int64_t)t * B + b) * H + f0) = make_float4(accx[0], accx[1], accx[2], accx[3]);
    } else {
#pragma unroll
    for (int ks = 0; ks < KSG; ++ks) {
      const float4 xb = *reinterpret_cast<const float4*>(gx + 16 * ks);
      acc = mfma16(wf[ks].x, xb.x, acc);
      acc = mfma16(wf[ks].y, xb.y, acc);
      acc = mfma16(wf[ks].z, xb.z, acc);
      acc = mfma16(wf[ks].w, xb.w, acc);
    }
    }
    dh = make_float4(direct[0] + acc[0], direct[1] + acc[1], direct[2] + acc[2], direct[3] + acc[3]);
    if constexpr (FUSE_W) {
      // weight gradients of this step: tile^T (gate column on the lane, 4 rows per MFMA) x this wave's 16 columns of h_prev / x
      const float* tr = Trs[wave];
      float bh[4], bx[4];
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) { bh[ks] = tr[(4 * ks + q) * 17 + i]; bx[ks] = FUSE_W == 2 ? tr[272 + (4 * ks + q) * 17 + i] : 0.f; }
      const float* ga = Gs[cur ^ 0] + q * ldg + i;
      const float* gna = &Gn[cur][q * ldn + i];
#pragma unroll
      for (int gt = 0; gt < 12; ++gt) {          // Gs columns: 0-3 g_r, 4-7 g_z, 8-11 g_hn; Gn: g_n
        float av[4], an[4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          av[ks] = ga[4 * ks * ldg + 16 * gt];
          an[ks] = (FUSE_W == 2 && gt >= 8) ? gna[4 * ks * ldn + 16 * (gt - 8)] : av[ks];
        }
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          whh[FUSE_W ? gt : 0] = mfma16(av[ks], bh[ks], whh[FUSE_W ? gt : 0]);
          if constexpr (FUSE_W == 2) wih[gt] = mfma16(an[ks], bx[ks], wih[gt]);
        }
      }
      if (tid < G) {
        float sum = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) sum += Gs[cur][r * ldg + tid];
        dbh += sum;
      }
      if (FUSE_W == 2 && tid < H) {
        float sum = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) sum += Gn[cur][r * ldn + tid];
        dbn += sum;
      }
    }
    cur ^= 1;
  }
  if (d.dh0 && rvalid) *reinterpret_cast<float4*>(d.dh0 + (int64_t)b * H + f0) = dh;
  if constexpr (FUSE_W) {
    // this workgroup's partial sums over its 16 rows and all steps: slab layout [matrix hh, ih][nblk][G x H], then
    // [matrix][nblk][G]; lane holds dW[16 gt + 4 q + r][16 wave + i]; dgh columns (r, z, hn), dgi columns (r, z, n)
    const int nblk = gridDim.x;
    const int64_t nw = (int64_t)G * H;
    float* s_hh = d.wslab + (int64_t)blockIdx.x * nw + 16 * wave + i;
    float* s_ih = d.wslab + ((int64_t)nblk + blockIdx.x) * nw + 16 * wave + i;
#pragma unroll
    for (int gt = 0; gt < 12; ++gt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        s_hh[(int64_t)(16 * gt + 4 * q + r) * H] = whh[FUSE_W ? gt : 0][r];
        if constexpr (FUSE_W == 2) s_ih[(int64_t)(16 * gt + 4 * q + r) * H] = wih[gt][r];
      }
    float* b_hh = d.wslab + (int64_t)2 * nblk * nw + (int64_t)blockIdx.x * G;
    float* b_ih = d.wslab + (int64_t)2 * nblk * nw + ((int64_t)nblk + blockIdx.x) * G;
    if (tid < G) {
      b_hh[tid] = dbh;
      if (FUSE_W == 2 && tid < 2 * H) b_ih[tid] = dbh;
    }
    if (FUSE_W == 2 && tid < H) b_ih[2 * H + tid] = dbn;
  }
}

}  // namespace g2v

using namespace g2v;

static bool gru_fast_ok(int H, int64_t hs_ld) { return H == 64 && (hs_ld & 3) == 0; }
// per-time-step launches (gru_step_*_kernel) while a direction would otherwise get only a handful of workgroups
constexpr int GRU_SPLIT_MAX_B = 1024;
static bool gru_split_ok(int B, int ndir, int H) {
  // measured at H = 200 (VQ-VAE.yml dims, whole train step): split 3.0 vs 4.0 ms at B = 512, 5.1 vs 5.7 ms at B = 1024
  return B <= GRU_SPLIT_MAX_B && cdiv(B, 16) * ndir <= 128 && (H & 3) == 0 && H <= 16 * GRU_STEP_KS && 3 * H <= 16 * GRU_STEP_KSB;
}
static bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }
static size_t gru_split_state_floats(int ndir, int H) { return (size_t)ndir * 2 * GRU_SPLIT_MAX_B * H; }

// The cluster kernels (gru_cluster_*_kernel): the split path's shapes while the whole grid is resident at once -- at most one
// 192-thread workgroup per CU (the workgroups of a row group wait for each other).  g2v_gru_seq_set_cluster(0) keeps the
// per-step launches (parity tests, A/B).
void g2v_internal_preclear_drop(const void* base, size_t bytes);
extern "C" int g2v_gru_seq_set_cluster(int enable) {             // = g2v_ctx_set_option(NULL, G2V_OPT_GRU_CLUSTER, enable)
  return g2v_ctx_set_option(nullptr, G2V_OPT_GRU_CLUSTER, enable);
}
static int gru_device_cus() {
  static int n = -1;
  if (n < 0) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) n = 0;
  }
  return n;
}
// exchange records of the cluster kernels: forward one (16 x Hp) record of granules per (parity, row group, direction), backward one
// per (parity, row group, producer tile, direction)
static size_t gru_cluster_rec_bytes(int B, int ndir, int H, bool bwd) {
  const size_t Hp = (size_t)((H + 15) & ~15);
  return (size_t)ndir * 2 * cdiv(B, 16) * (bwd ? Hp / 16 : 1) * 16 * Hp * 8;
}
// ... plus one word per workgroup (the XCC it runs on: cx_cluster_on_one_xcd), cleared with the records
static size_t gru_cluster_xch_bytes(int B, int ndir, int H, bool bwd) {
  return gru_cluster_rec_bytes(B, ndir, H, bwd) + (size_t)ndir * cdiv(B, 16) * ((H + 15) >> 4) * sizeof(unsigned) + 16;
}
// the largest exchange region a cluster launch can need at this H: the grid has at most one workgroup per CU
static size_t gru_cluster_max_xch_bytes(int H, bool bwd) {
  if ((H & 3) != 0 || H > 16 * GRU_STEP_KS || H == 64 || H < 4) return 0;
  const size_t Hp = (size_t)((H + 15) & ~15), nt = Hp / 16, cus = (size_t)gru_device_cus();
  return (bwd ? cus : cus / nt + 1) * 2 * 16 * Hp * 8 + cus * sizeof(unsigned) + 16;
}
int g2v_internal_preclear_take(const void* p, size_t need);      // dec_persist.hip
void g2v_internal_preclear_drop(const void* base, size_t bytes);
static bool gru_cluster_ok(int T, int B, int ndir, int H, const void* fn);
// 1: g2v_gru_seq_fwd / _bwd run this shape as the persistent cluster kernels (small batch; see g2v_gru_seq_set_cluster)
extern "C" int g2v_gru_seq_cluster_ok(int T, int B, int H, int ndir) {
  return (ndir >= 1 && ndir <= 2 && gru_cluster_ok(T, B, ndir, H, nullptr)) ? 1 : 0;
}
static bool gru_cluster_ok(int T, int B, int ndir, int H, const void* fn) {
  if (!g2v_internal_options().gru_cluster || !gru_split_ok(B, ndir, H) || T < 2 || T > (1 << 20)) return false;
  if ((int64_t)cdiv(B, 16) * cdiv(H, 16) * ndir > gru_device_cus()) return false;
  if (fn == nullptr) return true;      // (the shape query: the occupancy check is the launch's)
  int n = 0;
  return hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, fn, 192, 0) == hipSuccess && n >= 1;
}

// the exchange region a cluster launch of this shape clears (offset 0 of its workspace); 0: not a cluster shape
size_t g2v_internal_gru_cluster_region(int T, int B, int H, int ndir, int bwd) {
  if (ndir < 1 || ndir > 2 || !gru_cluster_ok(T, B, ndir, H, nullptr)) return 0;
  return gru_cluster_xch_bytes(B, ndir, H, bwd != 0);
}
extern "C" size_t g2v_gru_seq_fwd_workspace(int ndir, int H) {
  const size_t a = (size_t)2 * ndir * pack_floats(H, 3, H), b = gru_split_state_floats(ndir, H);
  const size_t c = gru_cluster_max_xch_bytes(H, false);
  const size_t ab = (a > b ? a : b) * sizeof(float);
  return ab > c ? ab : c;
}

int g2v_internal_slab_reduce4(const float* const* slab_w, float* const* out_w, const float* const* slab_b, float* const* out_b,
                              int nprob, int64_t n, int64_t nb, int nsplit, hipStream_t st);      // linear.hip
extern "C" size_t g2v_gru_seq_bwd_wslab_bytes(int B, int H) {
  return (B > 0 && H > 0) ? (size_t)cdiv(B, 16) * 2 * ((size_t)3 * H * H + 3 * H) * sizeof(float) : 0;
}
int g2v_internal_cell_bwd_products(const float* dgh, const float* w_hh, float* d_hprev, int H, const float* dgi,
                                   const float* w_ih, float* dx, int in_dim, const uint8_t* x_keep, float x_scale, int B,
                                   hipStream_t st);      // linear.hip

static bool cell_shape_ok(int in_dim, int H) { return (in_dim & 3) == 0 && (H & 3) == 0 && in_dim <= 16 * GRU_STEP_KS && H <= 16 * GRU_STEP_KS; }

extern "C" int g2v_gru_cell_fwd(const float* x, int in_dim, const uint8_t* x_keep, float x_scale, const float* h_prev,
                                const float* w_ih, const float* w_hh, const float* b_ih, const float* b_hh, float* h_new,
                                float* gates, int B, int H, g2v_stream_t stream) {
  G2V_REQUIRE(x && h_prev && w_ih && w_hh && b_ih && b_hh && h_new, "null pointer");
  G2V_REQUIRE(B > 0 && H > 0 && in_dim > 0, "bad size");
  if (!cell_shape_ok(in_dim, H) || !(aligned16(x) && aligned16(h_prev) && aligned16(w_ih) && aligned16(w_hh) && aligned16(b_ih) &&
                                     aligned16(b_hh) && aligned16(h_new) && aligned16(gates) &&
                                     (reinterpret_cast<uintptr_t>(x_keep) & 3) == 0)) {
    set_error("g2v_gru_cell_fwd: needs in_dim, H multiples of 4 and <= 256, 16-byte aligned operands (use g2v_linear_fwd + g2v_gru_seq_fwd)");
    return G2V_ERR_UNSUPPORTED;
  }
  hipLaunchKernelGGL(gru_cell_fwd_kernel, dim3(cdiv(B, 16), cdiv(H, 16)), dim3(128), 0, (hipStream_t)stream, x, in_dim, x_keep,
                     x_scale, h_prev, w_ih, w_hh, b_ih, b_hh, h_new, gates, B, H);
  G2V_CHECK_LAUNCH();
  return G2V_OK;
}

extern "C" int g2v_gru_cell_bwd(const float* d_h_a, const float* d_h_b, const float* gates, const float* h_prev,
                                const float* w_ih, const float* w_hh, const uint8_t* x_keep, float x_scale, float* dgi,
                                float* dgh, float* d_hprev, float* dx, int in_dim, int B, int H, g2v_stream_t stream) {
  G2V_REQUIRE(gates && h_prev && w_hh && dgi && dgh && d_hprev, "null pointer");
  G2V_REQUIRE(d_h_a || d_h_b, "no incoming gradient");
  G2V_REQUIRE(!dx || w_ih, "dx needs w_ih");
  G2V_REQUIRE(B > 0 && H > 0 && in_dim > 0, "bad size");
  if (!cell_shape_ok(in_dim, H) || !(aligned16(d_h_a) && aligned16(d_h_b) && aligned16(gates) && aligned16(h_prev) && aligned16(w_ih) &&
                                     aligned16(w_hh) && aligned16(dgi) && aligned16(dgh) && aligned16(d_hprev) && aligned16(dx))) {
    set_error("g2v_gru_cell_bwd: needs in_dim, H multiples of 4 and <= 256, 16-byte aligned operands (use g2v_gru_seq_bwd + g2v_linear_bwd_data)");
    return G2V_ERR_UNSUPPORTED;
  }
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(gru_cell_gates_bwd_kernel, dim3(cdiv((int64_t)B * (H >> 2), 256)), dim3(256), 0, st, d_h_a, d_h_b, gates, h_prev,
                     dgi, dgh, d_hprev, B, H);
  G2V_CHECK_LAUNCH();
  if (g2v_internal_cell_bwd_products(dgh, w_hh, d_hprev, H, dgi, w_ih, dx, in_dim, x_keep, x_scale, B, st) != 0) {
    set_error("g2v_gru_cell_bwd: launch failed");
    return G2V_ERR_LAUNCH;
  }
  return G2V_OK;
}

// fragment packs of the H == 64 fast kernels, in workspace order: per direction W_hh (, W_ih when the projection is fused)
static PackBatch gru_fwd_packs(const float* const* w_hh, const float* const* w_ih, int ndir, int H, bool fuse, float* p,
                               const float** p_hh, const float** p_ih) {
  PackBatch pb;
  pb.n = 0;
  for (int k = 0; k < ndir; ++k) {
    pb.d[pb.n++] = PackDesc{w_hh[k], p, H, 3, H, H, H, 0, 0};
    p_hh[k] = p;
    p += pack_floats(H, 3, H);
    p_ih[k] = nullptr;
    if (fuse) {
      pb.d[pb.n++] = PackDesc{w_ih[k], p, H, 3, H, H, H, 0, 0};
      p_ih[k] = p;
      p += pack_floats(H, 3, H);
    }
  }
  return pb;
}
static PackBatch gru_bwd_packs(const float* const* w_hh, const float* const* w_ih, int ndir, int H, bool fuse, float* p,
                               const float** p_hh_t, const float** p_ih_t) {
  PackBatch pb;
  pb.n = 0;
  for (int k = 0; k < ndir; ++k) {
    pb.d[pb.n++] = PackDesc{w_hh[k], p, H, 1, 0, 3 * H, H, 1, 0};   // rows k (hidden feature), contraction over the 3H gates
    p_hh_t[k] = p;
    p += pack_floats(H, 1, 3 * H);
    p_ih_t[k] = nullptr;
    if (fuse) {
      pb.d[pb.n++] = PackDesc{w_ih[k], p, H, 1, 0, 3 * H, H, 1, 0};
      p_ih_t[k] = p;
      p += pack_floats(H, 1, 3 * H);
    }
  }
  return pb;
}

// The weight-fragment packs of one g2v_gru_seq_fwd + one g2v_gru_seq_bwd call (H == 64 fast kernels), launched ahead of time
// into the two workspaces (the engine runs this as a parallel branch at the start of the step); the *_prepared entry points
// then skip their pack launch.  fused != 0: the calls fuse the input projection / input gradient (w_ih packed too).
// Either workspace may be NULL (that call packs for itself).  Other hidden sizes: a no-op (their kernels read the weights in place).
extern "C" int g2v_gru_seq_prepare(const float* const* w_hh, const float* const* w_ih, int ndir, int H, int fused,
                                   void* fwd_workspace, size_t fwd_bytes, void* bwd_workspace, size_t bwd_bytes,
                                   g2v_stream_t stream) {
  G2V_REQUIRE(w_hh && (fwd_workspace || bwd_workspace), "null pointer");
  G2V_REQUIRE(ndir >= 1 && ndir <= 2 && H > 0, "bad size");
  G2V_REQUIRE(!fused || w_ih, "fused needs w_ih");
  if (H != 64) return G2V_OK;
  if ((fwd_workspace && fwd_bytes < g2v_gru_seq_fwd_workspace(ndir, H)) ||
      (bwd_workspace && bwd_bytes < g2v_gru_seq_bwd_workspace(ndir, H))) {
    set_error("g2v_gru_seq_prepare: workspace too small");
    return G2V_ERR_WORKSPACE;
  }
  const float* a[2];
  const float* b[2];
  if (fwd_workspace) launch_pack(gru_fwd_packs(w_hh, w_ih, ndir, H, fused != 0, (float*)fwd_workspace, a, b), (hipStream_t)stream);
  if (bwd_workspace) launch_pack(gru_bwd_packs(w_hh, w_ih, ndir, H, fused != 0, (float*)bwd_workspace, a, b), (hipStream_t)stream);
  G2V_CHECK_LAUNCH();
  return G2V_OK;
}

// packed gi / dgi (g2v_gru_dir.gi_row_off): the generic kernels with 16-byte accesses, offsets by value in the kernel arguments
// 1: g2v_gru_seq_fwd would run the W_hh-resident forward for this shape (aligned operands assumed), the one kernel that gathers gi
extern "C" int g2v_gru_seq_gather_ok(int T, int B, int H, int ndir) {
  const int res_rows = g2v_internal_options().gru_resident_rows;
  return (T >= 1 && B >= 1 && ndir >= 1 && ndir <= 2 && (H & 3) == 0 && ((H + 15) & ~15) == 16 * RES_KB && res_rows > 0 && B >= res_rows &&
          !gru_split_ok(B, ndir, H)) ? 1 : 0;
}
extern "C" int g2v_gru_seq_packed_ok(int T, int B, int H) {
  return (T >= 1 && T <= GRU_MAX_OFF && B >= 1 && H >= 4 && (H & 3) == 0 && H <= 256 && H != 64) ? 1 : 0;
}

static int gru_seq_fwd_impl(const g2v_gru_dir* dirs, int ndir, const int32_t* lengths, int64_t hs_ld, int T, int B,
                            int H, void* workspace, size_t workspace_bytes, g2v_stream_t stream, bool prepared) {
  G2V_REQUIRE(dirs && workspace, "null pointer");
  G2V_REQUIRE(ndir >= 1 && ndir <= 2, "1 or 2 directions per call");
  G2V_REQUIRE(T > 0 && B > 0 && H > 0 && hs_ld >= H, "bad size");
  const bool gathered = dirs[0].gi_gather != nullptr;
  for (int k = 0; k < ndir; ++k) G2V_REQUIRE((dirs[k].gi_gather != nullptr) == gathered && (!gathered || dirs[k].gi), "gathered gi: every direction or none, with its table");
  if (gathered && !g2v_gru_seq_gather_ok(T, B, H, ndir)) {
    set_error("g2v_gru_seq_fwd: gi_gather is served by the W_hh-resident forward only (g2v_gru_seq_gather_ok)");
    return G2V_ERR_UNSUPPORTED;
  }
  for (int k = 0; k < ndir; ++k) G2V_REQUIRE(dirs[k].w_hh && dirs[k].b_hh && dirs[k].hs, "null pointer");
  if (dirs[0].gi_row_off && !g2v_gru_seq_packed_ok(T, B, H)) {
    set_error("g2v_gru_seq_fwd: packed gi is not served for this shape (g2v_gru_seq_packed_ok)");
    return G2V_ERR_UNSUPPORTED;
  }
  // gi == NULL: the input projection is fused (x, w_ih, b_ih given, in_dim == H == 64); all directions alike
  const bool fuse = dirs[0].gi == nullptr;
  for (int k = 0; k < ndir; ++k) {
    G2V_REQUIRE((dirs[k].gi == nullptr) == fuse, "directions must agree on gi / fused input projection");
    if (fuse) G2V_REQUIRE(dirs[k].x && dirs[k].w_ih && dirs[k].b_ih, "gi == NULL needs x, w_ih, b_ih");
  }
  hipStream_t st = (hipStream_t)stream;
  if (fuse && !(gru_fast_ok(H, hs_ld) && dirs[0].in_dim == H && (ndir == 1 || dirs[1].in_dim == H))) {
    set_error("g2v_gru_seq_fwd: fused input projection needs H == in_dim == 64 (compute gi with g2v_linear_fwd otherwise)");
    return G2V_ERR_UNSUPPORTED;
  }
  if (gru_fast_ok(H, hs_ld)) {
    if (workspace_bytes < g2v_gru_seq_fwd_workspace(ndir, H)) {
      set_error("g2v_gru_seq_fwd: workspace too small");
      return G2V_ERR_WORKSPACE;
    }
    const float* whh[2] = {dirs[0].w_hh, dirs[ndir - 1].w_hh};
    const float* wih[2] = {dirs[0].w_ih, dirs[ndir - 1].w_ih};
    const float* p_hh[2];
    const float* p_ih[2];
    const PackBatch pb = gru_fwd_packs(whh, wih, ndir, H, fuse, (float*)workspace, p_hh, p_ih);
    GruDirF f[2];
    for (int k = 0; k < ndir; ++k)
      f[k] = GruDirF{dirs[k].gi, p_hh[k], dirs[k].b_hh, dirs[k].h0, dirs[k].hs, dirs[k].h_n, dirs[k].gates, dirs[k].reverse,
                     dirs[k].x, p_ih[k], dirs[k].b_ih};
    if (ndir == 1) f[1] = f[0];
    if (!prepared) {
      launch_pack(pb, st);
      G2V_CHECK_LAUNCH();
    }
    if (fuse)
      hipLaunchKernelGGL((gru_fwd_fast_kernel<64, true>), dim3(cdiv(B, 16), ndir), dim3(256), 0, st, f[0], f[1], lengths, hs_ld, T, B);
    else
      hipLaunchKernelGGL((gru_fwd_fast_kernel<64, false>), dim3(cdiv(B, 16), ndir), dim3(256), 0, st, f[0], f[1], lengths, hs_ld, T, B);
    G2V_CHECK_LAUNCH();
    return G2V_OK;
  }
  const int Hp = (H + 15) & ~15;
  bool split = gru_split_ok(B, ndir, H);
  for (int k = 0; k < ndir && split; ++k)
    split = aligned16(dirs[k].gi) && aligned16(dirs[k].w_hh) && aligned16(dirs[k].b_hh) && aligned16(dirs[k].h0) &&
            aligned16(dirs[k].gates) && aligned16(dirs[k].h_n) && aligned16(workspace);
  RowOff ro;
  ro.on = 0;
  if (dirs[0].gi_row_off) {
    for (int k = 0; k < ndir; ++k) G2V_REQUIRE(dirs[k].gi_row_off != nullptr, "packed gi: every direction or none");
    if (!g2v_gru_seq_packed_ok(T, B, H) || lengths == nullptr) {
      set_error("g2v_gru_seq_fwd: packed gi needs lengths, T <= 64 and the generic kernels with H %% 4 == 0 (g2v_gru_seq_packed_ok)");
      return G2V_ERR_UNSUPPORTED;
    }
    ro.on = 1;
    for (int t = 0; t < GRU_MAX_OFF; ++t) ro.off[t] = t < T ? dirs[0].gi_row_off[t] : 0;
    for (int k = 1; k < ndir; ++k)
      for (int t = 0; t < T; ++t) G2V_REQUIRE(dirs[k].gi_row_off[t] == ro.off[t], "packed gi: the directions' offsets differ");
  } else {
    for (int k = 0; k < ndir; ++k) G2V_REQUIRE(dirs[k].gi_row_off == nullptr, "packed gi: every direction or none");
  }
  if (split) {
    // small batch: one launch per time step, (row groups x hidden-unit tiles x directions) single-wave workgroups
    if (workspace_bytes < g2v_gru_seq_fwd_workspace(ndir, H)) {
      set_error("g2v_gru_seq_fwd: workspace too small");
      return G2V_ERR_WORKSPACE;
    }
    const size_t xbytes = gru_cluster_xch_bytes(B, ndir, H, false), rbytes = gru_cluster_rec_bytes(B, ndir, H, false);
    if (gru_cluster_ok(T, B, ndir, H, (const void*)gru_cluster_fwd_kernel) && xbytes <= workspace_bytes) {
      // ... or ONE launch for all steps: the same workgroups resident, the state rows exchanged through tagged granules
      GruClF c[2];
      for (int k = 0; k < ndir; ++k)
        c[k] = GruClF{dirs[k].gi, dirs[k].w_hh, dirs[k].b_hh, dirs[k].h0, dirs[k].hs, dirs[k].gates, dirs[k].h_n,
                      reinterpret_cast<unsigned long long*>((char*)workspace + (size_t)k * (rbytes / ndir)), dirs[k].reverse};
      if (ndir == 1) c[1] = c[0];
      if (!g2v_internal_preclear_take(workspace, xbytes) && hipMemsetAsync(workspace, 0, xbytes, st) != hipSuccess) {
        set_error("g2v_gru_seq_fwd: clearing the exchange records failed");
        return G2V_ERR_LAUNCH;
      }
      hipLaunchKernelGGL(gru_cluster_fwd_kernel, dim3((Hp >> 4) * cdiv(B, 16) * ndir), dim3(192), 0, st, c[0], c[1], lengths, hs_ld, T,
                         B, H, ro, reinterpret_cast<unsigned*>((char*)workspace + rbytes), Hp >> 4, cdiv(B, 16), ndir,
                         const_cast<unsigned*>(g2v_internal_persist_fault_ptr()));
      G2V_CHECK_LAUNCH();
      return G2V_OK;
    }
    g2v_internal_preclear_drop(workspace, workspace_bytes);      // (per-step launches: a pre-cleared note for this workspace is void)
    float* state = (float*)workspace;                  // [dir][2][B][H]
    for (int s_ = 0; s_ < T; ++s_) {
      GruStepF g[2];
      for (int k = 0; k < ndir; ++k) {
        float* cur = state + ((size_t)k * 2 + (s_ & 1)) * B * H;
        float* nxt = state + ((size_t)k * 2 + ((s_ + 1) & 1)) * B * H;
        const int tk = dirs[k].reverse ? T - 1 - s_ : s_;
        g[k] = GruStepF{dirs[k].gi, dirs[k].w_hh, dirs[k].b_hh, s_ == 0 ? dirs[k].h0 : cur, nxt, dirs[k].hs, dirs[k].gates,
                        dirs[k].h_n, tk, ro.on ? (int64_t)ro.off[tk] : (int64_t)tk * B};
      }
      if (ndir == 1) g[1] = g[0];
      hipLaunchKernelGGL(gru_step_fwd_kernel, dim3(cdiv(B, 16), Hp >> 4, ndir), dim3(64), 0, st, g[0], g[1], lengths, hs_ld,
                         T, B, H, s_ == T - 1 ? 1 : 0);
    }
    G2V_CHECK_LAUNCH();
    return G2V_OK;
  }
  const size_t lds = (size_t)2 * 16 * (Hp + 4) * sizeof(float);
  G2V_REQUIRE(lds <= 160 * 1024, "hidden size too large for LDS");
  (void)hipFuncSetAttribute((const void*)gru_seq_fwd_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  (void)hipFuncSetAttribute((const void*)gru_seq_fwd_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (2 * lds <= 160 * 1024)
    (void)hipFuncSetAttribute((const void*)gru_seq_fwd_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(2 * lds));
  if (workspace_bytes < g2v_gru_seq_fwd_workspace(ndir, H)) {
    set_error("g2v_gru_seq_fwd: workspace too small");
    return G2V_ERR_WORKSPACE;
  }
  GruGenF g[2];
  {
    // W_hh in MFMA fragment order (coalesced 1 KiB fragment loads, eight k-steps kept in flight by wave_gemm_p)
    float* pp = (float*)workspace;
    PackBatch pb;
    pb.n = 0;
    for (int k = 0; k < ndir; ++k) {
      pb.d[pb.n++] = PackDesc{dirs[k].w_hh, pp, H, 3, H, H, H, 0, 0};
      g[k] = GruGenF{dirs[k].gi, pp, dirs[k].b_hh, dirs[k].h0, dirs[k].hs, dirs[k].h_n, dirs[k].gates, dirs[k].reverse, dirs[k].gi_gather};
      pp += pack_floats(H, 3, H);
    }
    launch_pack(pb, st);
    G2V_CHECK_LAUNCH();
  }
  if (ndir == 1) g[1] = g[0];
  bool v4 = (H & 3) == 0 && H <= 256 && (hs_ld & 3) == 0;
  for (int k = 0; k < ndir && v4; ++k)
    v4 = aligned16(dirs[k].gi) && aligned16(dirs[k].b_hh) && aligned16(dirs[k].hs) && aligned16(dirs[k].gates);
  // large batch, 192 < H <= 208: W_hh resident in the CU for the whole sequence (gru_res_fwd_kernel)
  const int res_rows = g2v_internal_options().gru_resident_rows;
  if (gathered && !(v4 && Hp == 16 * RES_KB && res_rows > 0 && B >= res_rows)) {
    set_error("g2v_gru_seq_fwd: gi_gather is served by the W_hh-resident forward only (g2v_gru_seq_gather_ok)");
    return G2V_ERR_UNSUPPORTED;
  }
  if (v4 && Hp == 16 * RES_KB && res_rows > 0 && B >= res_rows) {
    (void)hipFuncSetAttribute((const void*)gru_res_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)RES_LDS_BYTES);
    const int per_dir = max(1, (gru_device_cus() > 0 ? gru_device_cus() : 256) / ndir);      // one workgroup per CU; each walks its row tiles
    hipLaunchKernelGGL(gru_res_fwd_kernel, dim3(min(cdiv(B, 16), per_dir), ndir), dim3(256), RES_LDS_BYTES, st, g[0], g[1], lengths, hs_ld,
                       T, B, H, ro);
    G2V_CHECK_LAUNCH();
    return G2V_OK;
  }
  // two row tiles per workgroup (every weight fragment multiplies 32 rows: 0.73 against 0.80 ms at the native shape) once that
  // still leaves a workgroup for most CUs
  if (v4 && 2 * lds <= 160 * 1024 && cdiv(B, 32) * ndir >= 192)
    hipLaunchKernelGGL(gru_seq_fwd_kernel<2>, dim3(cdiv(B, 32), ndir), dim3(512), 2 * lds, st, g[0], g[1], lengths, hs_ld, T, B, H, ro);
  else if (v4)
    hipLaunchKernelGGL(gru_seq_fwd_kernel<1>, dim3(cdiv(B, 16), ndir), dim3(256), lds, st, g[0], g[1], lengths, hs_ld, T, B, H, ro);
  else
  {
    G2V_REQUIRE(!ro.on, "packed gi: the scalar generic body does not serve it");
    hipLaunchKernelGGL(gru_seq_fwd_kernel<0>, dim3(cdiv(B, 16), ndir), dim3(256), lds, st, g[0], g[1], lengths, hs_ld, T, B, H, ro);
  }
  G2V_CHECK_LAUNCH();
  return G2V_OK;
}

extern "C" int g2v_gru_seq_fwd(const g2v_gru_dir* dirs, int ndir, const int32_t* lengths, int64_t hs_ld, int T, int B,
                               int H, void* workspace, size_t workspace_bytes, g2v_stream_t stream) {
  return gru_seq_fwd_impl(dirs, ndir, lengths, hs_ld, T, B, H, workspace, workspace_bytes, stream, false);
}
// g2v_train_step_prepare (dec_rollout.hip): the pack descriptors of g2v_gru_seq_prepare's BACKWARD workspace, appended to `out`
// (returns how many; 0: this hidden size reads its weights in place, or the workspace is too small)
namespace g2v {
int gru_bwd_prepare_descs(const float* const* w_hh, const float* const* w_ih, int ndir, int H, bool fused, void* bwd_workspace,
                          size_t bwd_bytes, PackDesc* out, int max_out) {
  if (H != 64 || !bwd_workspace || bwd_bytes < g2v_gru_seq_bwd_workspace(ndir, H)) return 0;
  const float* a[2];
  const float* b[2];
  const PackBatch pb = gru_bwd_packs(w_hh, w_ih, ndir, H, fused, (float*)bwd_workspace, a, b);
  if (pb.n > max_out) return 0;
  for (int k = 0; k < pb.n; ++k) out[k] = pb.d[k];
  return pb.n;
}
}  // namespace g2v

// as g2v_gru_seq_fwd / _bwd on workspaces that g2v_gru_seq_prepare has filled for THIS call (no pack launch)
extern "C" int g2v_gru_seq_fwd_prepared(const g2v_gru_dir* dirs, int ndir, const int32_t* lengths, int64_t hs_ld, int T,
                                        int B, int H, void* workspace, size_t workspace_bytes, g2v_stream_t stream) {
  return gru_seq_fwd_impl(dirs, ndir, lengths, hs_ld, T, B, H, workspace, workspace_bytes, stream, H == 64);
}

extern "C" size_t g2v_gru_seq_bwd_workspace(int ndir, int H) {
  const size_t a = (size_t)2 * ndir * pack_floats(H, 1, 3 * H), b = (size_t)ndir * 3 * H * H + gru_split_state_floats(ndir, H);
  const size_t c = gru_cluster_max_xch_bytes(H, true);      // the cluster kernel's exchange records (whatever B it admits)
  const size_t ab = (a > b ? a : b) * sizeof(float);
  return ab > c ? ab : c;
}

static int gru_seq_bwd_impl(const g2v_gru_dir_bwd* dirs, int ndir, const int32_t* lengths, int64_t d_hs_ld,
                            int64_t hs_ld, int T, int B, int H, void* workspace, size_t workspace_bytes,
                            g2v_stream_t stream, bool prepared) {
  G2V_REQUIRE(dirs && workspace, "null pointer");
  G2V_REQUIRE(ndir >= 1 && ndir <= 2, "1 or 2 directions per call");
  G2V_REQUIRE(T > 0 && B > 0 && H > 0, "bad size");
  for (int k = 0; k < ndir; ++k)
    G2V_REQUIRE(dirs[k].hs && dirs[k].gates && dirs[k].w_hh && ((dirs[k].dgi && dirs[k].dgh) || dirs[k].dw_hh), "null pointer");
  if (workspace_bytes < g2v_gru_seq_bwd_workspace(ndir, H)) {
    set_error("g2v_gru_seq_bwd: workspace too small");
    return G2V_ERR_WORKSPACE;
  }
  hipStream_t st = (hipStream_t)stream;
  float* p = (float*)workspace;
  if (dirs[0].dgi_row_off && !g2v_gru_seq_packed_ok(T, B, H)) {
    set_error("g2v_gru_seq_bwd: packed dgi is not served for this shape (g2v_gru_seq_packed_ok)");
    return G2V_ERR_UNSUPPORTED;
  }
  // dx != NULL: the input gradient dx = dgi W_ih is fused (w_ih given, in_dim == H == 64); all directions alike
  const bool fuse = dirs[0].dx != nullptr;
  for (int k = 0; k < ndir; ++k) {
    G2V_REQUIRE((dirs[k].dx != nullptr) == fuse, "directions must agree on the fused input gradient");
    if (fuse) G2V_REQUIRE(dirs[k].w_ih, "dx != NULL needs w_ih");
  }
  const bool fast = gru_fast_ok(H, hs_ld) && (d_hs_ld & 3) == 0;
  // (the fused quantiser backward: the H == 64 kernels, or the small-batch cluster kernel -- g2v_gru_seq_cluster_ok)
  const bool hn_cluster = !fast && gru_cluster_ok(T, B, ndir, H, (const void*)gru_cluster_bwd_kernel) &&
                          gru_cluster_xch_bytes(B, ndir, H, true) <= workspace_bytes;
  bool hn_any = false;
  for (int k = 0; k < ndir; ++k) {
    hn_any = hn_any || dirs[k].hn_z != nullptr;
    if (dirs[k].hn_z && !((fast || hn_cluster) && dirs[k].d_hn && dirs[k].hn_q && dirs[k].hn_gloss && aligned16(dirs[k].hn_z) &&
                          aligned16(dirs[k].hn_q))) {
      set_error("g2v_gru_seq_bwd: the fused quantiser backward (hn_z, hn_q, hn_gloss) needs the H == 64 kernels or the cluster "
                "kernels (g2v_gru_seq_cluster_ok), d_hn and 16-byte-aligned slices");
      return G2V_ERR_UNSUPPORTED;
    }
  }
  if (fuse && !(fast && dirs[0].in_dim == H && (ndir == 1 || dirs[1].in_dim == H))) {
    set_error("g2v_gru_seq_bwd: fused input gradient needs H == in_dim == 64 (use g2v_linear_bwd_data otherwise)");
    return G2V_ERR_UNSUPPORTED;
  }
  if (fast) {
    const float* whh[2] = {dirs[0].w_hh, dirs[ndir - 1].w_hh};
    const float* wih[2] = {dirs[0].w_ih, dirs[ndir - 1].w_ih};
    const float* p_hh_t[2];
    const float* p_ih_t[2];
    const PackBatch pb = gru_bwd_packs(whh, wih, ndir, H, fuse, p, p_hh_t, p_ih_t);
    GruDirB f[2];
    // fused weight gradients: every direction names x, its four outputs and a slab buffer (or none does)
    // fused weight gradients: W_hh only (dw_hh, db_hh, wslab) or W_hh and W_ih (plus dw_ih, db_ih, x); every direction alike
    const bool fuse_w = dirs[0].dw_hh != nullptr, fuse_w2 = fuse_w && dirs[0].dw_ih != nullptr;
    for (int k = 0; k < ndir; ++k) {
      const bool hh = dirs[k].dw_hh && dirs[k].db_hh && dirs[k].wslab, ih = dirs[k].dw_ih && dirs[k].db_ih && dirs[k].x;
      const bool no_hh = !dirs[k].dw_hh && !dirs[k].db_hh && !dirs[k].wslab, no_ih = !dirs[k].dw_ih && !dirs[k].db_ih;
      G2V_REQUIRE(fuse_w ? (hh && fuse && (fuse_w2 ? (ih && aligned16(dirs[k].x)) : (no_ih && dirs[k].dgi != nullptr))) : (no_hh && no_ih),
                  "fused weight gradients: dw_hh, db_hh, wslab (+ dw_ih, db_ih, x) for every direction, with the fused input gradient");
    }
    for (int k = 0; k < ndir; ++k)
      f[k] = GruDirB{dirs[k].d_hs, dirs[k].d_hn, dirs[k].hs, dirs[k].h0, dirs[k].gates, p_hh_t[k],
                     dirs[k].dgi, dirs[k].dgh, dirs[k].dh0, dirs[k].reverse, p_ih_t[k], dirs[k].dx, dirs[k].x, dirs[k].wslab,
                     dirs[k].hn_z, dirs[k].hn_q, dirs[k].hn_gloss, dirs[k].hn_coef};
    if (ndir == 1) f[1] = f[0];
    if (!prepared) {
      launch_pack(pb, st);
      G2V_CHECK_LAUNCH();
    }
    if (fuse_w) {
      const int nblk = cdiv(B, 16);
      if (fuse_w2)
        hipLaunchKernelGGL((gru_bwd_fast_kernel<64, true, 2>), dim3(nblk, ndir), dim3(256), 0, st, f[0], f[1], lengths, d_hs_ld,
                           hs_ld, T, B);
      else
        hipLaunchKernelGGL((gru_bwd_fast_kernel<64, true, 1>), dim3(nblk, ndir), dim3(256), 0, st, f[0], f[1], lengths, d_hs_ld,
                           hs_ld, T, B);
      G2V_CHECK_LAUNCH();
      const int64_t nw = (int64_t)3 * H * H, nb = 3 * H;
      const float* sw[4]; const float* sb[4]; float* ow[4]; float* ob[4];
      int np = 0;
      for (int k = 0; k < ndir; ++k) {
        sw[np] = dirs[k].wslab; ow[np] = dirs[k].dw_hh;
        sb[np] = dirs[k].wslab + (int64_t)2 * nblk * nw; ob[np] = dirs[k].db_hh;
        ++np;
        if (fuse_w2) {
          sw[np] = dirs[k].wslab + (int64_t)nblk * nw; ow[np] = dirs[k].dw_ih;
          sb[np] = dirs[k].wslab + (int64_t)2 * nblk * nw + (int64_t)nblk * nb; ob[np] = dirs[k].db_ih;
          ++np;
        }
      }
      if (g2v_internal_slab_reduce4(sw, ow, sb, ob, np, nw, nb, nblk, st) != 0) {
        set_error("g2v_gru_seq_bwd: slab reduction launch failed");
        return G2V_ERR_LAUNCH;
      }
      return G2V_OK;
    }
    if (fuse)
      hipLaunchKernelGGL((gru_bwd_fast_kernel<64, true>), dim3(cdiv(B, 16), ndir), dim3(256), 0, st, f[0], f[1], lengths, d_hs_ld,
                         hs_ld, T, B);
    else
      hipLaunchKernelGGL((gru_bwd_fast_kernel<64, false>), dim3(cdiv(B, 16), ndir), dim3(256), 0, st, f[0], f[1], lengths, d_hs_ld,
                         hs_ld, T, B);
    G2V_CHECK_LAUNCH();
    return G2V_OK;
  }
  const int Hp = (H + 15) & ~15, Gp = (3 * H + 15) & ~15;
  const size_t lds = (size_t)16 * ((Gp + 4) + (Hp + 4)) * sizeof(float);
  G2V_REQUIRE(lds <= 160 * 1024, "hidden size too large for LDS");
  (void)hipFuncSetAttribute((const void*)gru_seq_bwd_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  (void)hipFuncSetAttribute((const void*)gru_seq_bwd_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  bool split = gru_split_ok(B, ndir, H) && aligned16(workspace);
  for (int k = 0; k < ndir && split; ++k)
    split = aligned16(dirs[k].gates) && aligned16(dirs[k].dgi) && aligned16(dirs[k].dgh) && aligned16(dirs[k].d_hn) &&
            aligned16(dirs[k].h0) && aligned16(dirs[k].dh0);
  RowOff ro;
  ro.on = 0;
  if (dirs[0].dgi_row_off) {
    for (int k = 0; k < ndir; ++k) G2V_REQUIRE(dirs[k].dgi_row_off != nullptr, "packed dgi: every direction or none");
    if (!g2v_gru_seq_packed_ok(T, B, H) || lengths == nullptr) {
      set_error("g2v_gru_seq_bwd: packed dgi needs lengths, T <= 64 and the generic kernels with H %% 4 == 0 (g2v_gru_seq_packed_ok)");
      return G2V_ERR_UNSUPPORTED;
    }
    ro.on = 1;
    for (int t = 0; t < GRU_MAX_OFF; ++t) ro.off[t] = t < T ? dirs[0].dgi_row_off[t] : 0;
    for (int k = 1; k < ndir; ++k)
      for (int t = 0; t < T; ++t) G2V_REQUIRE(dirs[k].dgi_row_off[t] == ro.off[t], "packed dgi: the directions' offsets differ");
  } else {
    for (int k = 0; k < ndir; ++k) G2V_REQUIRE(dirs[k].dgi_row_off == nullptr, "packed dgi: every direction or none");
  }
  if (split) {
    const int Hp = (H + 15) & ~15;
    const size_t xbytes = gru_cluster_xch_bytes(B, ndir, H, true), rbytes = gru_cluster_rec_bytes(B, ndir, H, true);
    if (gru_cluster_ok(T, B, ndir, H, (const void*)gru_cluster_bwd_kernel) && xbytes <= workspace_bytes) {
      GruClB c[2];
      for (int k = 0; k < ndir; ++k)
        c[k] = GruClB{dirs[k].hn_z, dirs[k].hn_q, dirs[k].hn_gloss, dirs[k].hn_coef,
                      dirs[k].d_hs, dirs[k].hs, dirs[k].h0, dirs[k].gates, dirs[k].w_hh, dirs[k].d_hn, dirs[k].dgi, dirs[k].dgh,
                      dirs[k].dh0, reinterpret_cast<unsigned long long*>((char*)workspace + (size_t)k * (rbytes / ndir)),
                      dirs[k].reverse};
      if (ndir == 1) c[1] = c[0];
      const bool want_dh0 = dirs[0].dh0 || (ndir == 2 && dirs[1].dh0);
      if (!g2v_internal_preclear_take(workspace, xbytes) && hipMemsetAsync(workspace, 0, xbytes, st) != hipSuccess) {
        set_error("g2v_gru_seq_bwd: clearing the exchange records failed");
        return G2V_ERR_LAUNCH;
      }
      hipLaunchKernelGGL(gru_cluster_bwd_kernel, dim3((Hp >> 4) * cdiv(B, 16) * ndir), dim3(192), 0, st, c[0], c[1], lengths, d_hs_ld,
                         hs_ld, T, B, H, ro, want_dh0 ? T : T - 1, reinterpret_cast<unsigned*>((char*)workspace + rbytes), Hp >> 4,
                         cdiv(B, 16), ndir, const_cast<unsigned*>(g2v_internal_persist_fault_ptr()));
      G2V_CHECK_LAUNCH();
      return G2V_OK;
    }
    g2v_internal_preclear_drop(workspace, workspace_bytes);
    if (hn_any) {
      set_error("g2v_gru_seq_bwd: the fused quantiser backward was requested but the cluster kernel does not serve this call");
      return G2V_ERR_UNSUPPORTED;
    }
    float* carry = p + (size_t)ndir * 3 * H * H;      // [dir][B][H], after the transposed weights (3 H^2 floats: 16-byte multiple)
    for (int k = 0; k < ndir; ++k) launch_transpose(dirs[k].w_hh, p + (size_t)k * 3 * H * H, 3 * H, H, st);
    // launch `it`: part A finishes the step of forward iteration T - it, part B opens forward iteration T - 1 - it
    for (int it = 0; it <= T; ++it) {
      GruStepB g[2];
      for (int k = 0; k < ndir; ++k) {
        const bool rev = dirs[k].reverse != 0;
        const int s_cur = T - it, s_next = T - 1 - it;
        const int t_cur = it == 0 ? -1 : (rev ? T - 1 - s_cur : s_cur);
        const int t_next = s_next < 0 ? -1 : (rev ? T - 1 - s_next : s_next);
        const int tprev = (s_next <= 0) ? -1 : (rev ? t_next + 1 : t_next - 1);
        g[k] = GruStepB{dirs[k].d_hs, dirs[k].hs, dirs[k].h0, dirs[k].gates, p + (size_t)k * 3 * H * H, dirs[k].d_hn,
                        carry + (size_t)k * B * H, dirs[k].dgi, dirs[k].dgh, dirs[k].dh0, t_cur, t_next, tprev,
                        t_next < 0 ? (int64_t)0 : (ro.on ? (int64_t)ro.off[t_next] : (int64_t)t_next * B), ro.on};
      }
      if (ndir == 1) g[1] = g[0];
      if (it == T && !dirs[0].dh0 && (ndir == 1 || !dirs[1].dh0)) break;      // nobody wants the initial-state gradient
      hipLaunchKernelGGL(gru_step_bwd_kernel, dim3(cdiv(B, 16), Hp >> 4, ndir), dim3(64), 0, st, g[0], g[1], lengths,
                         d_hs_ld, hs_ld, T, B, H);
    }
    G2V_CHECK_LAUNCH();
    return G2V_OK;
  }
  // large batch, H == 200: W_hh resident in the CU for the whole sequence (gru_res_bwd_kernel)
  {
    const int res_rows = g2v_internal_options().gru_resident_rows;
    bool res = H == RESB_H && res_rows > 0 && B >= res_rows && g2v_internal_options().gru_resident_bwd && (hs_ld & 3) == 0 &&
               (d_hs_ld & 3) == 0;
    for (int k = 0; k < ndir && res; ++k)
      res = aligned16(dirs[k].gates) && aligned16(dirs[k].dgi) && aligned16(dirs[k].dgh) && aligned16(dirs[k].hs) && aligned16(dirs[k].h0) &&
            aligned16(dirs[k].dh0) && aligned16(dirs[k].d_hs) && aligned16(dirs[k].d_hn);
    if (res) {
      GruResB r[2];
      for (int k = 0; k < ndir; ++k)
        r[k] = GruResB{dirs[k].d_hs, dirs[k].d_hn, dirs[k].hs, dirs[k].h0, dirs[k].gates, dirs[k].w_hh, dirs[k].dgi, dirs[k].dgh,
                       dirs[k].dh0, dirs[k].reverse};
      if (ndir == 1) r[1] = r[0];
      (void)hipFuncSetAttribute((const void*)gru_res_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)RESB_LDS_BYTES);
      // (one workgroup per row tile here: with the forward's one-per-CU walk Part d with attention at B = 4096 lost 1.4 % -- the
      //  side branch's weight gradients wait for a CU until a whole walk is through -- and without attention gained nothing)
      hipLaunchKernelGGL(gru_res_bwd_kernel, dim3(cdiv(B, 16), ndir), dim3(256), RESB_LDS_BYTES, st, r[0], r[1], lengths, d_hs_ld, hs_ld,
                         T, B, ro);
      G2V_CHECK_LAUNCH();
      return G2V_OK;
    }
  }
  GruGenB g[2];
  {
    PackBatch pb;
    pb.n = 0;
    for (int k = 0; k < ndir; ++k) {
      float* wt = p + (size_t)k * pack_floats(H, 1, 3 * H);
      pb.d[pb.n++] = PackDesc{dirs[k].w_hh, wt, H, 1, 0, 3 * H, H, 1, 0};   // rows k (hidden unit), contraction over the 3H gates
      g[k] = GruGenB{dirs[k].d_hs, dirs[k].d_hn, dirs[k].hs, dirs[k].h0, dirs[k].gates, wt, dirs[k].dgi, dirs[k].dgh,
                     dirs[k].dh0, dirs[k].reverse};
    }
    launch_pack(pb, st);
    G2V_CHECK_LAUNCH();
  }
  if (ndir == 1) g[1] = g[0];
  bool v4 = (H & 3) == 0 && H <= 256 && (hs_ld & 3) == 0 && (d_hs_ld & 3) == 0;
  for (int k = 0; k < ndir && v4; ++k)
    v4 = aligned16(dirs[k].gates) && aligned16(dirs[k].dgi) && aligned16(dirs[k].dgh) && aligned16(dirs[k].hs) &&
         aligned16(dirs[k].h0) && aligned16(dirs[k].dh0) && aligned16(dirs[k].d_hs);
  if (v4)
    hipLaunchKernelGGL(gru_seq_bwd_kernel<1>, dim3(cdiv(B, 16), ndir), dim3(256), lds, st, g[0], g[1], lengths, d_hs_ld, hs_ld, T,
                       B, H, ro);
  else {
    G2V_REQUIRE(!ro.on, "packed dgi: the scalar generic body does not serve it");
    hipLaunchKernelGGL(gru_seq_bwd_kernel<0>, dim3(cdiv(B, 16), ndir), dim3(256), lds, st, g[0], g[1], lengths, d_hs_ld, hs_ld, T,
                       B, H, ro);
  }
  G2V_CHECK_LAUNCH();
  return G2V_OK;
}

extern "C" int g2v_gru_seq_bwd(const g2v_gru_dir_bwd* dirs, int ndir, const int32_t* lengths, int64_t d_hs_ld,
                               int64_t hs_ld, int T, int B, int H, void* workspace, size_t workspace_bytes,
                               g2v_stream_t stream) {
  return gru_seq_bwd_impl(dirs, ndir, lengths, d_hs_ld, hs_ld, T, B, H, workspace, workspace_bytes, stream, false);
}
extern "C" int g2v_gru_seq_bwd_prepared(const g2v_gru_dir_bwd* dirs, int ndir, const int32_t* lengths, int64_t d_hs_ld,
                                        int64_t hs_ld, int T, int B, int H, void* workspace, size_t workspace_bytes,
                                        g2v_stream_t stream) {
  return gru_seq_bwd_impl(dirs, ndir, lengths, d_hs_ld, hs_ld, T, B, H, workspace, workspace_bytes, stream, H == 64);
}
