// seq2seq.hip -- the remaining operators of Part d (text -> gesture-code seq2seq):
//   embedding gather / scatter-add        nn.Embedding in EncoderRNN (model/text2embedding_model.py:90-92,126) and in the
//                                         code decoder (:252,340-343), with the decoder's nn.Dropout(0.5) fused in
//   BatchNorm1d (+ReLU) forward/backward  decoder.pre_linear[1:] (:286-290), one call per decode step
//   cross-entropy forward+backward        torch.nn.CrossEntropyLoss over the code logits (train_eval/train_seq2seq.py:520-530)
//   row argmax                            greedy feedback `decoder_output.argmax(1)` (:740)
// All HBM-bound element/row-wise kernels; the contractions of Part d run on the dense-layer and GRU kernels.
#include "common.hpp"

namespace g2v {

__global__ void embedding_fwd_kernel(const float* __restrict__ table, const int64_t* __restrict__ ids,
                                     const uint8_t* __restrict__ keep, float scale, float* __restrict__ out, int64_t ldo,
                                     int64_t n, int dim, int64_t V) {
  const int64_t total = n * dim;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = e / dim;
    const int c = (int)(e - r * dim);
    const int64_t id = ids[r];
    float v = (id >= 0 && id < V) ? table[id * dim + c] : 0.f;
    if (keep) v = keep[e] ? v * scale : 0.f;
    out[r * ldo + c] = v;
  }
}

// ---- embedding gradient: d_table[v, :] (+)= sum over the tokens r with ids[r] == v of d_out[r, :] * keep * scale --------------
// No float atomics (round 1 used atomicAdd here, the one kernel of the library whose output depended on arrival order) and no
// serial walk over a hot row either (a padded batch sends half of all tokens to row 0).
//   n <= EMB_SMALL_N tokens (the code decoder's per-step lookup at small batch; round 6: also the word table at the reference's
//     batch size, 1.5 k packed tokens at B = 128 -- there the sorted form's eight launches were 55 us of a 0.7 ms iteration):
//     ONE launch, a workgroup owns a table row, finds its tokens with a ballot scan over the ids and adds them in token order.
//   otherwise the tokens are counting-sorted by table row, STABLY, in O(n) work --
//       emb_rank_kernel     blocks of 1024 tokens: rank of every token among the equal ids before it in its block (ids held in
//                           LDS, broadcast compares); (block, row) counts by integer atomicMax of rank + 1
//       emb_colscan_kernel  per row: counts -> exclusive prefix over the blocks, and the row's total
//       emb_scan_kernel     exclusive prefix sum of the row totals (one workgroup)
//       emb_scatter_kernel  perm[off[row] + blockoff[block][row] + rank] = token
//     -- the sorted list is cut into chunks that one wave each sums in list order (U rows in flight), and a row whose tokens
//     straddle chunks is finished by one workgroup adding the chunk partials in chunk order:
//       emb_chunk_sum       a row that lies wholly inside the chunk is written to d_table directly, the run that started before
//                           the chunk goes to partial slot 0, the one that continues beyond it to slot 1
//       emb_row_finish      rows without tokens are zeroed (overwrite mode); rows that span chunks sum their partials
//   Either way a fixed summation tree: bitwise reproducible, with ~n / chunk waves of parallelism whatever the id distribution.
constexpr int EMB_SMALL_N = 2048;
constexpr int EMB_TB = 1024;        // tokens per ranking block
constexpr int EMB_MAX_CHUNK = 128;

// (round 6: half the chunk lengths of round 5 -- 1536 instead of 768 one-wave workgroups for Part d's 49 k packed tokens at B = 4096,
//  where a wave with 2-4 rows in flight is a chain of memory round trips: 87 -> 5x us per call)
static inline int emb_chunk_for(int64_t n) { return n <= 32768 ? 16 : (n <= 131072 ? 32 : (n <= 262144 ? 64 : EMB_MAX_CHUNK)); }

template <int NE>
__device__ __forceinline__ void emb_store_row(float* __restrict__ dst, const float (&acc)[NE], int dim, int lane, int accumulate) {
#pragma unroll
  for (int k = 0; k < NE; ++k) {
    const int c = lane + 64 * k;
    if (c < dim) dst[c] = accumulate ? dst[c] + acc[k] : acc[k];
  }
}

// Row `base` of d_out * keep * scale, elements lane + 64 k.  Every load is UNCONDITIONAL (clamped column, caller passes a valid
// row even for a dead slot) and the select comes afterwards: loads under per-element branches are issued one at a time behind
// s_waitcnt vmcnt(0) by hipcc, which is the whole cost of these kernels.
template <int NE, bool KEEP>
__device__ __forceinline__ void emb_load_row(float (&g)[NE], const float* __restrict__ d_out, const uint8_t* __restrict__ keep,
                                             float scale, int64_t base, int dim, int lane, bool valid) {
  float x[NE];
  uint8_t kp[NE];
#pragma unroll
  for (int k = 0; k < NE; ++k) {
    const int c = lane + 64 * k, cc = c < dim ? c : dim - 1;
    x[k] = d_out[base + cc];
    if (KEEP) kp[k] = keep[base + cc];
  }
#pragma unroll
  for (int k = 0; k < NE; ++k) {
    const bool live = valid && (lane + 64 * k < dim) && (!KEEP || kp[k]);
    g[k] = live ? (KEEP ? x[k] * scale : x[k]) : 0.f;
  }
}

// the next (up to) U tokens of the ballot mask m, added to acc in token order
template <int NE, bool KEEP, int U>
__device__ __forceinline__ void emb_take(unsigned long long& m, float (&acc)[NE], const float* __restrict__ d_out,
                                         const uint8_t* __restrict__ keep, float scale, int r0, int dim, int lane) {
  float g[U][NE];
  bool ok[U];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    ok[u] = m != 0;
    const int j = ok[u] ? __builtin_ctzll(m) : 0;
    m &= m - 1;
    emb_load_row<NE, KEEP>(g[u], d_out, keep, scale, (int64_t)(r0 + j) * dim, dim, lane, ok[u]);
  }
#pragma unroll
  for (int u = 0; u < U; ++u)
    if (ok[u]) {
#pragma unroll
      for (int k = 0; k < NE; ++k) acc[k] += g[u][k];
    }
}

// elements per lane: dim <= 64 * NE.  One workgroup per table row; its EMB_OW waves split the tokens.  (Round 6: 8 waves and up to
// 16 rows in flight per wave instead of 4 and 8 -- the greedy feedback of an untrained code decoder sends most of a step's tokens
// to a handful of codes, and a hot row is a serial chain of load rounds per wave: 640 tokens on one row 44 -> 14 us.)
constexpr int EMB_OW = 8;
template <int NE, bool KEEP>
__global__ __launch_bounds__(64 * EMB_OW) void emb_owner_kernel(const float* __restrict__ d_out, const int64_t* __restrict__ ids,
                                                                const uint8_t* __restrict__ keep, float scale,
                                                                float* __restrict__ d_table, int n, int dim, int64_t V,
                                                                int accumulate) {
  constexpr int U = NE <= 5 ? 16 : (NE <= 8 ? 8 : 4);
  __shared__ float comb[EMB_OW - 1][64 * NE];
  __shared__ int any_hit;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t v = blockIdx.x;
  if (threadIdx.x == 0) any_hit = 0;
  __syncthreads();
  float acc[NE];
#pragma unroll
  for (int k = 0; k < NE; ++k) acc[k] = 0.f;
  const int per = (n + EMB_OW - 1) / EMB_OW, lo = wave * per < n ? wave * per : n, hi = lo + per < n ? lo + per : n;
  bool hit = false;
  for (int r0 = lo; r0 < hi; r0 += 64) {
    const int r = r0 + lane;
    unsigned long long m = __ballot(r < hi && ids[r < hi ? r : lo] == v);
    hit |= m != 0;
    while (m) {                 // (m is wave-uniform: a few tokens take the narrow round, a hot row the wide one)
      if (__popcll(m) > 4) emb_take<NE, KEEP, U>(m, acc, d_out, keep, scale, r0, dim, lane);
      else emb_take<NE, KEEP, 4>(m, acc, d_out, keep, scale, r0, dim, lane);
    }
  }
  if (hit && lane == 0) any_hit = 1;
  __syncthreads();
  if (!any_hit) {                                          // the common case: nobody refers to this row
    if (!accumulate && wave == 0) emb_store_row<NE>(d_table + v * dim, acc, dim, lane, 0);
    return;
  }
  if (wave > 0) {
#pragma unroll
    for (int k = 0; k < NE; ++k) comb[wave - 1][lane + 64 * k] = acc[k];
  }
  __syncthreads();
  if (wave == 0) {
#pragma unroll
    for (int w = 0; w < EMB_OW - 1; ++w)
#pragma unroll
      for (int k = 0; k < NE; ++k) acc[k] += comb[w][lane + 64 * k];
    emb_store_row<NE>(d_table + v * dim, acc, dim, lane, accumulate);
  }
}

__global__ __launch_bounds__(EMB_TB) void emb_rank_kernel(const int64_t* __restrict__ ids, int64_t n, int64_t V,
                                                          int* __restrict__ blockcnt, int* __restrict__ rank) {
  __shared__ __attribute__((aligned(16))) int sid[EMB_TB];
  const int tid = threadIdx.x;
  const int64_t r = (int64_t)blockIdx.x * EMB_TB + tid;
  const int64_t id64 = r < n ? ids[r] : -1;
  const int id = (id64 >= 0 && id64 < V) ? (int)id64 : -1;
  sid[tid] = id;
  __syncthreads();
  // equal ids in the waves before this one (2 VALU ops per comparison), then this wave's own 64 with the position tests
  const int wbase = __builtin_amdgcn_readfirstlane(tid >> 6) * 64;
  int before = 0, after = 0;
#pragma unroll 4
  for (int j = 0; j < wbase; j += 4) {
    const int4 q = *reinterpret_cast<const int4*>(&sid[j]);
    before += (q.x == id) + (q.y == id) + (q.z == id) + (q.w == id);
  }
#pragma unroll 4
  for (int j = wbase; j < wbase + 64; j += 4) {
    const int4 q = *reinterpret_cast<const int4*>(&sid[j]);
    const int e0 = q.x == id, e1 = q.y == id, e2 = q.z == id, e3 = q.w == id;
    before += (e0 & (j < tid)) + (e1 & (j + 1 < tid)) + (e2 & (j + 2 < tid)) + (e3 & (j + 3 < tid));
    after += (e0 & (j > tid)) + (e1 & (j + 1 > tid)) + (e2 & (j + 2 > tid)) + (e3 & (j + 3 > tid));
  }
  if (id >= 0) {
    rank[r] = before;
    // the last token of a (wave, row) pair proposes the pair's running count; the maximum over the block's waves is the
    // (block, row) count.  Integer max: exact and order-free.
    if (after == 0) atomicMax(blockcnt + (int64_t)blockIdx.x * V + id, before + 1);
  }
}

__global__ void emb_colscan_kernel(int* __restrict__ blockcnt, int nb, int64_t V, int* __restrict__ cnt) {
  const int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= V) return;
  int run = 0;
  for (int b = 0; b < nb; ++b) {
    const int c = blockcnt[(int64_t)b * V + v];
    blockcnt[(int64_t)b * V + v] = run;
    run += c;
  }
  cnt[v] = run;
}

__global__ __launch_bounds__(1024) void emb_scan_kernel(const int* __restrict__ cnt, int* __restrict__ off, int64_t V) {
  __shared__ int part[1024];
  const int tid = threadIdx.x;
  const int64_t per = (V + 1023) / 1024, lo = tid * per < V ? tid * per : V, hi = lo + per < V ? lo + per : V;
  int s = 0;
  for (int64_t v = lo; v < hi; ++v) s += cnt[v];
  part[tid] = s;
  __syncthreads();
  for (int d = 1; d < 1024; d <<= 1) {
    const int add = tid >= d ? part[tid - d] : 0;
    __syncthreads();
    part[tid] += add;
    __syncthreads();
  }
  int run = part[tid] - s;
  for (int64_t v = lo; v < hi; ++v) {
    off[v] = run;
    run += cnt[v];
  }
  if (tid == 1023) off[V] = part[1023];
}

__global__ void emb_scatter_kernel(const int64_t* __restrict__ ids, int64_t n, int64_t V, const int* __restrict__ blockoff,
                                   const int* __restrict__ off, const int* __restrict__ rank, int* __restrict__ perm) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n) return;
  const int64_t id = ids[r];
  if (id < 0 || id >= V) return;
  perm[off[id] + blockoff[(r / EMB_TB) * V + id] + rank[r]] = (int)r;
}

template <int NE, bool KEEP>
__global__ __launch_bounds__(64) void emb_chunk_sum_kernel(const float* __restrict__ d_out, const int64_t* __restrict__ ids,
                                                           const uint8_t* __restrict__ keep, float scale,
                                                           const int* __restrict__ off, const int* __restrict__ perm,
                                                           float* __restrict__ d_table, float* __restrict__ partial, int dim,
                                                           int64_t V, int chunk, int accumulate) {
  constexpr int U = NE <= 5 ? 8 : (NE <= 10 ? 4 : 2);     // rows in flight
  constexpr int Q = EMB_MAX_CHUNK / 64;
  const int lane = threadIdx.x;
  const int total = off[V];
  const int begin = blockIdx.x * chunk;
  if (begin >= total) return;
  const int end = begin + chunk < total ? begin + chunk : total;
  // per list position (lane + 64 q): token, row, and where the row's sum goes: 2 = d_table (the row lies inside this chunk),
  // 0 / 1 = partial slot (the row started before this chunk / continues beyond it)
  int tok[Q], row[Q], dst[Q];
#pragma unroll
  for (int q = 0; q < Q; ++q) {
    const int p = begin + q * 64 + lane;
    const bool in = p < end;
    tok[q] = perm[in ? p : begin];                        // unconditional loads (see emb_load_row)
    const int rw = (int)ids[tok[q]];
    const int o0 = off[rw], o1 = off[rw + 1];
    row[q] = in ? rw : -1;
    dst[q] = (o0 >= begin && o1 <= end) ? 2 : (o0 < begin ? 0 : 1);
  }
  float acc[NE];
#pragma unroll
  for (int k = 0; k < NE; ++k) acc[k] = 0.f;
  int cur = __builtin_amdgcn_readfirstlane(row[0]), cur_dst = __builtin_amdgcn_readfirstlane(dst[0]);
  auto flush = [&]() {
    float* o = cur_dst == 2 ? d_table + (int64_t)cur * dim : partial + ((int64_t)blockIdx.x * 2 + cur_dst) * dim;
    emb_store_row<NE>(o, acc, dim, lane, cur_dst == 2 ? accumulate : 0);
#pragma unroll
    for (int k = 0; k < NE; ++k) acc[k] = 0.f;
  };
#pragma unroll
  for (int q = 0; q < Q; ++q) {
    for (int j0 = 0; j0 < 64; j0 += U) {
      if (begin + q * 64 + j0 >= end) break;
      float g[U][NE];
      int rw[U], ds[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int t = __builtin_amdgcn_readlane(tok[q], j0 + u);
        rw[u] = __builtin_amdgcn_readlane(row[q], j0 + u);          // -1 past the end of the list
        ds[u] = __builtin_amdgcn_readlane(dst[q], j0 + u);
        emb_load_row<NE, KEEP>(g[u], d_out, keep, scale, (int64_t)t * dim, dim, lane, rw[u] >= 0);
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if (rw[u] < 0) continue;
        if (rw[u] != cur) {
          flush();
          cur = rw[u];
          cur_dst = ds[u];
        }
#pragma unroll
        for (int k = 0; k < NE; ++k) acc[k] += g[u][k];
      }
    }
  }
  flush();
}

template <int NE>     // one workgroup (4 waves) per table row; the waves split the row's chunk range, wave order fixes the sum
__global__ __launch_bounds__(256) void emb_row_finish_kernel(const int* __restrict__ off, const float* __restrict__ partial,
                                                             float* __restrict__ d_table, int dim, int64_t V, int chunk,
                                                             int accumulate) {
  constexpr int U = NE <= 5 ? 8 : (NE <= 8 ? 4 : 2);
  __shared__ float comb[3][64 * NE];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t v = blockIdx.x;
  const int lo = off[v], hi = off[v + 1];
  float acc[NE];
#pragma unroll
  for (int k = 0; k < NE; ++k) acc[k] = 0.f;
  if (hi == lo) {
    if (!accumulate && wave == 0) emb_store_row<NE>(d_table + v * dim, acc, dim, lane, 0);
    return;
  }
  const int c0 = lo / chunk, c1 = (hi - 1) / chunk;
  if (c0 == c1) return;                                    // written by the chunk's wave
  const int per = (c1 - c0 + 4) / 4;
  const int wb = c0 + wave * per, we = wb + per - 1 < c1 ? wb + per - 1 : c1;
  for (int cb = wb; cb <= we; cb += U) {
    float g[U][NE];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int c = cb + u;
      const float* src = partial + ((int64_t)c * 2 + (lo < c * chunk ? 0 : 1)) * dim;
#pragma unroll
      for (int k = 0; k < NE; ++k) {
        const int col = lane + 64 * k;
        g[u][k] = (c <= we && col < dim) ? src[col] : 0.f;
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int k = 0; k < NE; ++k) acc[k] += g[u][k];
  }
  if (wave > 0) {
#pragma unroll
    for (int k = 0; k < NE; ++k) comb[wave - 1][lane + 64 * k] = acc[k];
  }
  __syncthreads();
  if (wave == 0) {
#pragma unroll
    for (int w = 0; w < 3; ++w)
#pragma unroll
      for (int k = 0; k < NE; ++k) acc[k] += comb[w][lane + 64 * k];
    emb_store_row<NE>(d_table + v * dim, acc, dim, lane, accumulate);
  }
}

// ---- BatchNorm1d over (B,H), 16 features per workgroup, 16 row lanes x 16 feature lanes -----------------------------
// A workgroup owns BN_FB = 4 features and splits the batch over BN_RL = 64 row lanes (H / 4 workgroups: enough to cover
// the chip for H >= 64 even though the op is tiny); every row loop keeps 8 independent loads in flight.
constexpr int BN_FB = 4, BN_RL = 64;

template <class F>
__device__ __forceinline__ void bn_rows(int rl, int B, F body) {      // body(r) for r = rl, rl + 64, ... (unrolled by 8)
  int r = rl;
  for (; r + 7 * BN_RL < B; r += 8 * BN_RL) {
#pragma unroll
    for (int j = 0; j < 8; ++j) body(r + j * BN_RL);
  }
  for (; r < B; r += BN_RL) body(r);
}

__device__ __forceinline__ float bn_block_sum(float v, float (*red)[BN_FB + 1], int rl, int fl) {
  __syncthreads();                 // red may still be read from the previous use
  red[rl][fl] = v;
  __syncthreads();
  float t = 0.f;
  for (int k = 0; k < BN_RL; ++k) t += red[k][fl];     // fixed order: deterministic
  return t;
}

__global__ __launch_bounds__(256) void bn_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                     const float* __restrict__ b, float* __restrict__ rm,
                                                     float* __restrict__ rv, int training, int relu,
                                                     float* __restrict__ y, float* __restrict__ save_mean,
                                                     float* __restrict__ save_invstd, int B, int H,
                                                     const unsigned* __restrict__ fault) {
  __shared__ float red[BN_RL][BN_FB + 1];
  const int fl = threadIdx.x & (BN_FB - 1), rl = threadIdx.x / BN_FB;
  const int f = blockIdx.x * BN_FB + fl;
  const bool fv = f < H;
  const int fc = fv ? f : 0;
  float mean, invstd;
  if (training) {
    float s = 0.f;
    bn_rows(rl, B, [&](int r) { s += x[(int64_t)r * H + fc]; });
    mean = bn_block_sum(s, red, rl, fl) / (float)B;
    float q = 0.f;
    bn_rows(rl, B, [&](int r) { const float d = x[(int64_t)r * H + fc] - mean; q += d * d; });
    const float var = bn_block_sum(q, red, rl, fl) / (float)B;          // biased: used for normalisation
    invstd = 1.0f / sqrtf(var + 1e-5f);
    if (rl == 0 && fv) {
      const float unb = (B > 1) ? var * (float)B / (float)(B - 1) : var;
      // (the persistent kernels' fault latch: a faulted encoder launch in front of this step must not reach the model state)
      if (rm && (fault == nullptr || *fault == 0u)) {      // (rm == NULL: the caller commits later, g2v_bn_running_update_invstd)
        rm[f] = 0.9f * rm[f] + 0.1f * mean;                             // momentum 0.1, unbiased variance
        rv[f] = 0.9f * rv[f] + 0.1f * unb;
      }
      if (save_mean) save_mean[f] = mean;
      if (save_invstd) save_invstd[f] = invstd;
    }
  } else {
    mean = rm[fc];
    invstd = 1.0f / sqrtf(rv[fc] + 1e-5f);
  }
  if (!fv) return;
  const float g = w[f], bb = b[f];
  bn_rows(rl, B, [&](int r) {
    float v = (x[(int64_t)r * H + f] - mean) * invstd * g + bb;
    if (relu) v = fmaxf(v, 0.f);
    y[(int64_t)r * H + f] = v;
  });
}

__global__ __launch_bounds__(256) void bn_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                     const float* __restrict__ y, const float* __restrict__ w,
                                                     const float* __restrict__ save_mean,
                                                     const float* __restrict__ save_invstd, int relu,
                                                     float* __restrict__ dx, float* __restrict__ dw,
                                                     float* __restrict__ db, int B, int H) {
  __shared__ float red[BN_RL][BN_FB + 1];
  const int fl = threadIdx.x & (BN_FB - 1), rl = threadIdx.x / BN_FB;
  const int f = blockIdx.x * BN_FB + fl;
  const bool fv = f < H;
  const int fc = fv ? f : 0;
  const float mean = save_mean[fc], invstd = save_invstd[fc];
  float s1 = 0.f, s2 = 0.f;
  bn_rows(rl, B, [&](int r) {
    const int64_t e = (int64_t)r * H + fc;
    float g = dy[e];
    if (relu && !(y[e] > 0.f)) g = 0.f;
    s1 += g;
    s2 += g * ((x[e] - mean) * invstd);
  });
  const float S1 = bn_block_sum(s1, red, rl, fl);
  const float S2 = bn_block_sum(s2, red, rl, fl);
  if (!fv) return;
  if (rl == 0) { db[f] = S1; dw[f] = S2; }
  const float g0 = w[f] * invstd, invB = 1.0f / (float)B;
  bn_rows(rl, B, [&](int r) {
    const int64_t e = (int64_t)r * H + f;
    float g = dy[e];
    if (relu && !(y[e] > 0.f)) g = 0.f;
    const float xhat = (x[e] - mean) * invstd;
    dx[e] = g0 * (g - S1 * invB - xhat * S2 * invB);
  });
}

// The same over the `steps` calls of a decode loop whose gradients arrive together (Part d behind the cluster BPTT: nothing of
// BatchNorm's backward feeds the recurrence): ONE launch, a workgroup walks its 4 features through the steps in call order and
// leaves the parameters' gradients SUMMED over the steps (what autograd's accumulation over the loop's T-1 nodes produces) --
// it replaces steps launches + two reductions; dx of a step is bitwise bn_bwd_kernel's.
__global__ __launch_bounds__(256) void bn_bwd_steps_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                           const float* __restrict__ y, const float* __restrict__ w,
                                                           const float* __restrict__ save_mean,
                                                           const float* __restrict__ save_invstd, int64_t stat_stride, int relu,
                                                           float* __restrict__ dx, float* __restrict__ dw,
                                                           float* __restrict__ db, int steps, int B, int H) {
  __shared__ float red[BN_RL][BN_FB + 1];
  const int fl = threadIdx.x & (BN_FB - 1), rl = threadIdx.x / BN_FB;
  const int f = blockIdx.x * BN_FB + fl;
  const bool fv = f < H;
  const int fc = fv ? f : 0;
  const float wf = w[fc], invB = 1.0f / (float)B;
  float dws = 0.f, dbs = 0.f;
  for (int s = 0; s < steps; ++s) {
    const int64_t base = (int64_t)s * B * H;
    const float mean = save_mean[s * stat_stride + fc], invstd = save_invstd[s * stat_stride + fc];
    float s1 = 0.f, s2 = 0.f;
    bn_rows(rl, B, [&](int r) {
      const int64_t e = base + (int64_t)r * H + fc;
      float g = dy[e];
      if (relu && !(y[e] > 0.f)) g = 0.f;
      s1 += g;
      s2 += g * ((x[e] - mean) * invstd);
    });
    const float S1 = bn_block_sum(s1, red, rl, fl);
    const float S2 = bn_block_sum(s2, red, rl, fl);
    dbs += S1;
    dws += S2;
    if (fv) {
      const float g0 = wf * invstd;
      bn_rows(rl, B, [&](int r) {
        const int64_t e = base + (int64_t)r * H + f;
        float g = dy[e];
        if (relu && !(y[e] > 0.f)) g = 0.f;
        const float xhat = (x[e] - mean) * invstd;
        dx[e] = g0 * (g - S1 * invB - xhat * S2 * invB);
      });
    }
  }
  if (fv && rl == 0) { db[f] = dbs; dw[f] = dws; }
}

// out[r, :] = one_hot(ids[r]) (row r at out + r * ld): slot 0 of Part d's outputs (reference :676-677) without a fill + a scatter
__global__ __launch_bounds__(256) void one_hot_rows_kernel(const int64_t* __restrict__ ids, float* __restrict__ out, int64_t ld,
                                                           int M, int K) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (int64_t)M * K) return;
  const int r = (int)(i / K), k = (int)(i - (int64_t)r * K);
  out[(int64_t)r * ld + k] = ids[r] == (int64_t)k ? 1.0f : 0.0f;
}

// Large batch (B >= 1024): the kernel above gives the whole batch to H / 4 workgroups and reads 16 bytes per row and thread
// group (81 us at B = 4096, H = 200: 50 workgroups on a 256-CU chip, a quarter of every fetched line used).  Three short launches
// instead, every load a full row segment: (1) per 16-row block, thread <-> feature: partial sums of g and g * xhat; (2) the
// partials of a feature added up in block order (deterministic) into db / dw -- which ARE the two sums; (3) dx element-wise.
// The partials live in the first B * H / 8 floats of dx (overwritten by (3) afterwards): no workspace in the C-ABI.
constexpr int BN_RB = 16;
__global__ __launch_bounds__(256) void bn_bwd_partial_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                             const float* __restrict__ y, const float* __restrict__ save_mean,
                                                             const float* __restrict__ save_invstd, int relu,
                                                             float* __restrict__ partial, int B, int H) {
  const int r0 = blockIdx.x * BN_RB, nr = min(BN_RB, B - r0);
  for (int f = threadIdx.x; f < H; f += 256) {
    const float mean = save_mean[f], invstd = save_invstd[f];
    float s1 = 0.f, s2 = 0.f;
    float gv[BN_RB], xv[BN_RB];
#pragma unroll
    for (int r = 0; r < BN_RB; ++r) {            // every load of the column slice issued before the first use
      const int64_t e = (int64_t)(r0 + (r < nr ? r : 0)) * H + f;
      float g = dy[e];
      if (relu && !(y[e] > 0.f)) g = 0.f;
      gv[r] = r < nr ? g : 0.f;
      xv[r] = x[e];
    }
#pragma unroll
    for (int r = 0; r < BN_RB; ++r) {
      s1 += gv[r];
      s2 += gv[r] * ((xv[r] - mean) * invstd);
    }
    partial[((int64_t)blockIdx.x * 2 + 0) * H + f] = s1;
    partial[((int64_t)blockIdx.x * 2 + 1) * H + f] = s2;
  }
}
__global__ __launch_bounds__(256) void bn_bwd_finish_kernel(const float* __restrict__ partial, int nrb, float* __restrict__ dw,
                                                            float* __restrict__ db, int H) {
  __shared__ float red[2][4][64];
  const int fl = threadIdx.x & 63, grp = threadIdx.x >> 6, f = blockIdx.x * 64 + fl;
  float s1 = 0.f, s2 = 0.f;
  if (f < H)
    for (int k = grp; k < nrb; k += 4) {
      s1 += partial[((int64_t)k * 2 + 0) * H + f];
      s2 += partial[((int64_t)k * 2 + 1) * H + f];
    }
  red[0][grp][fl] = s1;
  red[1][grp][fl] = s2;
  __syncthreads();
  if (grp == 0 && f < H) {
    db[f] = (red[0][0][fl] + red[0][1][fl]) + (red[0][2][fl] + red[0][3][fl]);
    dw[f] = (red[1][0][fl] + red[1][1][fl]) + (red[1][2][fl] + red[1][3][fl]);
  }
}
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                           const float* __restrict__ y, const float* __restrict__ w,
                                                           const float* __restrict__ save_mean,
                                                           const float* __restrict__ save_invstd, int relu,
                                                           const float* __restrict__ dw, const float* __restrict__ db,
                                                           float* __restrict__ dx, int B, int H) {
  const float invB = 1.0f / (float)B;
  const int r0 = blockIdx.x * BN_RB, nr = min(BN_RB, B - r0);
  for (int f = threadIdx.x; f < H; f += 256) {
    const float mean = save_mean[f], invstd = save_invstd[f], g0 = w[f] * invstd, S1 = db[f], S2 = dw[f];
    for (int r = 0; r < nr; ++r) {
      const int64_t e = (int64_t)(r0 + r) * H + f;
      float g = dy[e];
      if (relu && !(y[e] > 0.f)) g = 0.f;
      const float xhat = (x[e] - mean) * invstd;
      dx[e] = g0 * (g - S1 * invB - xhat * S2 * invB);
    }
  }
}

// ---- cross entropy: one wave per row ----------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void ce_kernel(const float* __restrict__ logits, int64_t ld, const int64_t* __restrict__ tgt,
                                                 float* __restrict__ row_loss, float* __restrict__ dlogits, int64_t ldd,
                                                 int M, int K, float gcoef) {
  const int row = (blockIdx.x * 256 + threadIdx.x) >> 6, lane = threadIdx.x & 63;
  if (row >= M) return;
  const float* z = logits + (int64_t)row * ld;
  float mx = -INFINITY;
  for (int k = lane; k < K; k += 64) mx = fmaxf(mx, z[k]);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
  float s = 0.f;
  for (int k = lane; k < K; k += 64) s += expf(z[k] - mx);
  s = wave_sum(s);
  const float lse = mx + logf(s);
  const int t = (int)tgt[row];
  if (lane == 0) row_loss[row] = lse - z[t];
  if (dlogits) {
    float* d = dlogits + (int64_t)row * ldd;
    for (int k = lane; k < K; k += 64) d[k] = gcoef * (expf(z[k] - lse) - (k == t ? 1.0f : 0.0f));
  }
}
__global__ void mean_kernel(const float* __restrict__ v, int n, float* __restrict__ out) {
  __shared__ float red[4];
  float s = 0.f;
  for (int k = threadIdx.x; k < n; k += 256) s += v[k];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) out[0] = ((red[0] + red[1]) + (red[2] + red[3])) / (float)n;
}

__global__ __launch_bounds__(256) void argmax_rows_kernel(const float* __restrict__ x, int64_t ld, int64_t* __restrict__ out,
                                                          int M, int K) {
  const int row = (blockIdx.x * 256 + threadIdx.x) >> 6, lane = threadIdx.x & 63;
  if (row >= M) return;
  const float* z = x + (int64_t)row * ld;
  float bv = -INFINITY;
  int bk = 0x7fffffff;
  for (int k = lane; k < K; k += 64) {
    const float v = z[k];
    if (v > bv) { bv = v; bk = k; }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float v2 = __shfl_xor(bv, o);
    const int k2 = __shfl_xor(bk, o);
    if (v2 > bv || (v2 == bv && k2 < bk)) { bv = v2; bk = k2; }
  }
  if (lane == 0) out[row] = (int64_t)bk;
}


// ---- Bahdanau attention of the code decoder (model/text2embedding_model.py:160-198, 353-359) -------------------------
// energy[t,b,:] = tanh(attn([h_b ; enc[t,b,:]])) is split as tanh(hp[b,:] + ep[t,b,:]) with hp = W_h h + bias (one small
// product per decode step) and ep = enc W_e^T (ONE product per sentence batch, shared by all decode steps).
// One WORKGROUP per batch row (round 1: one wave per row -- 128 waves on the whole chip at B = 128, each walking T serially
// through dependent wave reductions: 32 / 79 us per call): the four waves split the T positions for the score / d_w dot products
// (four positions in flight per wave), the softmax runs redundantly in every wave, and the feature loops run one thread per
// feature with the T loads independent.  tanh on the hardware exp / rcp like every other gate of the library.  Softmax runs
// over ALL T positions (the reference does not mask padded positions; their encoder rows are zero).
__device__ __forceinline__ void attn_fwd_row(const float* __restrict__ hp, const float* __restrict__ ep,
                                             const float* __restrict__ enc, const float* __restrict__ v,
                                             float* __restrict__ weights, float* __restrict__ ctx, int64_t ldctx, int T, int B,
                                             int H, float* sc) {      // sc: [T] floats of LDS (scores / weights)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int b = blockIdx.x;
  const float* hpr = hp + (int64_t)b * H;
  // (round 6: every load of a pass is requested before the first tanh -- the loops used to walk H in 64-wide slices and T one
  //  position at a time, a memory round trip per slice / position: 10 us per call at B = 128, T = 20, H = 200, now 5)
  for (int t0 = wave * 4; t0 < T; t0 += 16) {         // positions t0 .. t0 + 3 of this wave, together
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int f0 = 0; f0 < H; f0 += 256) {             // H <= 256: one pass
      float e[4][4], vf[4], hf[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int f = f0 + lane + 64 * k, fc = f < H ? f : H - 1;
        vf[k] = f < H ? v[fc] : 0.f;
        hf[k] = hpr[fc];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int t = t0 + u < T ? t0 + u : T - 1;
          e[u][k] = ep[((int64_t)t * B + b) * H + fc];
        }
      }
#pragma unroll
      for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int u = 0; u < 4; ++u) acc[u] += vf[k] * tanhf_(hf[k] + e[u][k]);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const float a = wave_sum(acc[u]);
      if (lane == 0 && t0 + u < T) sc[t0 + u] = a;
    }
  }
  __syncthreads();
  float mx = -INFINITY;
  for (int t = lane; t < T; t += 64) mx = fmaxf(mx, sc[t]);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
  float sum = 0.f;
  for (int t = lane; t < T; t += 64) sum += expf(sc[t] - mx);
  sum = wave_sum(sum);
  const float inv = 1.0f / sum;
  __syncthreads();                                    // every wave has read the scores
  if (wave == 0)
    for (int t = lane; t < T; t += 64) {
      const float w = expf(sc[t] - mx) * inv;
      sc[t] = w;
      weights[(int64_t)b * T + t] = w;
    }
  __syncthreads();
  for (int f = threadIdx.x; f < H; f += 256) {
    float c = 0.f;
    for (int t0 = 0; t0 < T; t0 += 8) {
      float x[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) x[u] = enc[((int64_t)(t0 + u < T ? t0 + u : T - 1) * B + b) * H + f];
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (t0 + u < T) c += sc[t0 + u] * x[u];
    }
    ctx[(int64_t)b * ldctx + f] = c;
  }
}

__global__ __launch_bounds__(256) void attn_fwd_kernel(const float* __restrict__ hp, const float* __restrict__ ep,
                                                       const float* __restrict__ enc, const float* __restrict__ v,
                                                       float* __restrict__ weights, float* __restrict__ ctx,
                                                       int64_t ldctx, int T, int B, int H) {
  extern __shared__ float smem[];
  attn_fwd_row(hp, ep, enc, v, weights, ctx, ldctx, T, B, H, smem);
}

// The row-local head of a decode step in ONE launch (round 6; at the reference's batch size a step is a chain of dependent
// launches of ~5 us each whatever they do): the greedy feedback id = argmax(previous logits row) (lowest index on ties, as
// g2v_argmax_rows; logits == NULL: the id is given), the code embedding with its Dropout(0.5) mask into the first half of the
// decoder input row, the attention context into the second half.  One workgroup per batch row.
__global__ __launch_bounds__(256) void attn_step_fwd_kernel(const float* __restrict__ logits, int64_t ldl, int K,
                                                            int64_t* __restrict__ ids, const float* __restrict__ table,
                                                            const uint8_t* __restrict__ keep, float emb_scale,
                                                            float* __restrict__ ec, int64_t ldec, const float* __restrict__ hp,
                                                            const float* __restrict__ ep, const float* __restrict__ enc,
                                                            const float* __restrict__ v, float* __restrict__ weights, int T, int B,
                                                            int H) {
  extern __shared__ float smem[];
  __shared__ float wv[4];
  __shared__ int wk[4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int b = blockIdx.x;
  int64_t id;
  if (logits) {
    const float* z = logits + (int64_t)b * ldl;
    float bv = -INFINITY;
    int bk = 0x7fffffff;
    for (int k = threadIdx.x; k < K; k += 256) {
      const float x = z[k];
      if (x > bv) { bv = x; bk = k; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float v2 = __shfl_xor(bv, o);
      const int k2 = __shfl_xor(bk, o);
      if (v2 > bv || (v2 == bv && k2 < bk)) { bv = v2; bk = k2; }
    }
    if (lane == 0) { wv[wave] = bv; wk[wave] = bk; }
    __syncthreads();
#pragma unroll
    for (int w = 0; w < 4; ++w)
      if (wv[w] > bv || (wv[w] == bv && wk[w] < bk)) { bv = wv[w]; bk = wk[w]; }
    id = (int64_t)bk;
    if (threadIdx.x == 0) ids[b] = id;
  } else {
    id = ids[b];
  }
  for (int c = threadIdx.x; c < H; c += 256) {
    float x = (id >= 0 && id < (int64_t)K) ? table[id * H + c] : 0.f;
    if (keep) x = keep[(int64_t)b * H + c] ? x * emb_scale : 0.f;
    ec[(int64_t)b * ldec + c] = x;
  }
  attn_fwd_row(hp, ep, enc, v, weights, ec + H, ldec, T, B, H, smem);
}

// Backward of the step above, one workgroup per batch row.  d_ep / d_enc rows of row b are written (or accumulated) by that
// workgroup only; d_v goes to per-row partials [B][H] (summed by the slab reduction in a fixed order).
__global__ __launch_bounds__(256) void attn_bwd_kernel(const float* __restrict__ d_ctx, int64_t ldd,
                                                       const float* __restrict__ hp, const float* __restrict__ ep,
                                                       const float* __restrict__ enc, const float* __restrict__ v,
                                                       const float* __restrict__ weights, float* __restrict__ d_hp,
                                                       float* __restrict__ d_ep, float* __restrict__ d_enc,
                                                       float* __restrict__ dv_partial, int accumulate, int T, int B,
                                                       int H) {
  extern __shared__ float smem[];                     // [T] d_w -> ds
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int b = blockIdx.x;
  float* ds = smem;
  const float* dcr = d_ctx + (int64_t)b * ldd;
  const float* wr = weights + (int64_t)b * T;
  // d_w[t] = <d_ctx, enc[t,b,:]>
  for (int t0 = wave * 4; t0 < T; t0 += 16) {
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int f0 = 0; f0 < H; f0 += 256) {             // (loads up front as in the forward)
      float x[4][4], dc[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int f = f0 + lane + 64 * k, fc = f < H ? f : H - 1;
        dc[k] = f < H ? dcr[fc] : 0.f;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int t = t0 + u < T ? t0 + u : T - 1;
          x[u][k] = enc[((int64_t)t * B + b) * H + fc];
        }
      }
#pragma unroll
      for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int u = 0; u < 4; ++u) acc[u] += dc[k] * x[u][k];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const float a = wave_sum(acc[u]);
      if (lane == 0 && t0 + u < T) ds[t0 + u] = a;
    }
  }
  __syncthreads();
  // softmax backward ds[t] = w[t] (d_w[t] - sum_t' w[t'] d_w[t']), the sum in every wave
  float dot = 0.f;
  for (int t = lane; t < T; t += 64) dot += wr[t] * ds[t];
  dot = wave_sum(dot);
  __syncthreads();
  if (wave == 0)
    for (int t = lane; t < T; t += 64) ds[t] = wr[t] * (ds[t] - dot);
  __syncthreads();
  for (int f = threadIdx.x; f < H; f += 256) {
    const float hpf = hp[(int64_t)b * H + f], vf = v[f], dc = dcr[f];
    float dh = 0.f, dvf = 0.f;
    // (round 6: five positions' operands -- ep and, when accumulating, the two running gradients -- requested together; the
    //  read-modify-write per position was a memory round trip per position: 19.8 us per call at T = 20, now 8)
    for (int t0 = 0; t0 < T; t0 += 5) {
      float x[5], a0[5], a1[5];
#pragma unroll
      for (int u = 0; u < 5; ++u) {
        const int64_t row = ((int64_t)(t0 + u < T ? t0 + u : T - 1) * B + b) * H + f;
        x[u] = ep[row];
        a0[u] = accumulate ? d_ep[row] : 0.f;
        a1[u] = accumulate ? d_enc[row] : 0.f;
      }
#pragma unroll
      for (int u = 0; u < 5; ++u)
        if (t0 + u < T) {
          const int t = t0 + u;
          const int64_t row = ((int64_t)t * B + b) * H + f;
          const float e = tanhf_(hpf + x[u]);
          const float de = ds[t] * vf * (1.0f - e * e);
          dh += de;
          dvf += ds[t] * e;
          const float dn = wr[t] * dc;
          d_ep[row] = accumulate ? a0[u] + de : de;
          d_enc[row] = accumulate ? a1[u] + dn : dn;
        }
    }
    d_hp[(int64_t)b * H + f] = dh;
    dv_partial[(int64_t)b * H + f] = dvf;
  }
}

}  // namespace g2v

using namespace g2v;

static int blocks_for(int64_t n) {
  int b = cdiv(n, 256);
  return b > 4096 ? 4096 : (b < 1 ? 1 : b);
}

extern "C" int g2v_embedding_fwd(const float* table, const int64_t* ids, const uint8_t* keep, float scale, float* out,
                                 int64_t ldo, int64_t n, int dim, int64_t V, g2v_stream_t stream) {
  G2V_REQUIRE(table && ids && out, "null pointer");
  G2V_REQUIRE(n > 0 && dim > 0 && V > 0 && ldo >= dim, "bad size");
  hipLaunchKernelGGL(embedding_fwd_kernel, dim3(blocks_for(n * dim)), dim3(256), 0, (hipStream_t)stream, table, ids, keep,
                     scale, out, ldo, n, dim, V);
  G2V_CHECK_LAUNCH();
  return G2V_OK;
}

static size_t emb_align(size_t b) { return (b + 255) & ~(size_t)255; }

extern "C" size_t g2v_embedding_bwd_ws_bytes(int64_t n, int dim, int64_t V) {
  if (n <= EMB_SMALL_N) return 256;
  const size_t nb = (size_t)cdiv(n, EMB_TB), chunks = (size_t)cdiv(n, emb_chunk_for(n));
  return emb_align(nb * V * 4) + emb_align((size_t)V * 4) + emb_align((size_t)(V + 1) * 4) + 2 * emb_align((size_t)n * 4) +
         emb_align(chunks * 2 * dim * 4);
}

extern "C" int g2v_embedding_bwd(const float* d_out, const int64_t* ids, const uint8_t* keep, float scale, float* d_table,
                                 int64_t n, int dim, int64_t V, int zero_first, void* ws, size_t ws_bytes,
                                 g2v_stream_t stream) {
  G2V_REQUIRE(d_out && ids && d_table && ws, "null pointer");
  G2V_REQUIRE(n > 0 && dim > 0 && V > 0, "bad size");
  G2V_REQUIRE(n < (int64_t)1 << 31 && V < (int64_t)1 << 31, "more than 2^31 tokens or rows");
  G2V_REQUIRE(dim <= 64 * 16, "embedding dim > 1024");
  G2V_REQUIRE(ws_bytes >= g2v_embedding_bwd_ws_bytes(n, dim, V), "workspace too small (g2v_embedding_bwd_ws_bytes)");
  hipStream_t st = (hipStream_t)stream;
  const int acc = (zero_first & 1) ? 0 : 1;   // bit 0: every row is (over)written, rows without tokens with zeros
  const bool reuse_sort = (zero_first & 2) != 0;      // bit 1: `ws` still holds the sort of THESE ids (the previous call's, same n / V)
#define G2V_EMB_DISPATCH(LAUNCH)    \
  if (dim <= 64) LAUNCH(1);         \
  else if (dim <= 128) LAUNCH(2);   \
  else if (dim <= 256) LAUNCH(4);   \
  else if (dim <= 320) LAUNCH(5);   \
  else if (dim <= 512) LAUNCH(8);   \
  else if (dim <= 640) LAUNCH(10);  \
  else LAUNCH(16)
  if (n <= EMB_SMALL_N) {
#define G2V_EMB_OWNER(NE)                                                                                                    \
  do {                                                                                                                       \
    if (keep)                                                                                                                \
      hipLaunchKernelGGL((emb_owner_kernel<NE, true>), dim3((unsigned)V), dim3(64 * EMB_OW), 0, st, d_out, ids, keep, scale, d_table, \
                         (int)n, dim, V, acc);                                                                               \
    else                                                                                                                     \
      hipLaunchKernelGGL((emb_owner_kernel<NE, false>), dim3((unsigned)V), dim3(64 * EMB_OW), 0, st, d_out, ids, keep, scale,        \
                         d_table, (int)n, dim, V, acc);                                                                      \
  } while (0)
    G2V_EMB_DISPATCH(G2V_EMB_OWNER);
#undef G2V_EMB_OWNER
    G2V_CHECK_LAUNCH();
    return G2V_OK;
  }
  const int nb = cdiv(n, EMB_TB), chunk = emb_chunk_for(n), chunks = cdiv(n, chunk);
  char* w = (char*)ws;
  int* blockcnt = (int*)w;            w += emb_align((size_t)nb * V * 4);
  int* cnt = (int*)w;                 w += emb_align((size_t)V * 4);
  int* off = (int*)w;                 w += emb_align((size_t)(V + 1) * 4);
  int* rank = (int*)w;                w += emb_align((size_t)n * 4);
  int* perm = (int*)w;                w += emb_align((size_t)n * 4);
  float* partial = (float*)w;
  if (!reuse_sort) {
    (void)hipMemsetAsync(blockcnt, 0, (size_t)nb * V * 4, st);
    hipLaunchKernelGGL(emb_rank_kernel, dim3(nb), dim3(EMB_TB), 0, st, ids, n, V, blockcnt, rank);
    hipLaunchKernelGGL(emb_colscan_kernel, dim3(cdiv(V, 64)), dim3(64), 0, st, blockcnt, nb, V, cnt);
    hipLaunchKernelGGL(emb_scan_kernel, dim3(1), dim3(1024), 0, st, cnt, off, V);
    hipLaunchKernelGGL(emb_scatter_kernel, dim3(cdiv(n, 256)), dim3(256), 0, st, ids, n, V, blockcnt, off, rank, perm);
  }
#define G2V_EMB_SORTED(NE)                                                                                                 \
  do {                                                                                                                     \
    if (keep)                                                                                                              \
      hipLaunchKernelGGL((emb_chunk_sum_kernel<NE, true>), dim3(chunks), dim3(64), 0, st, d_out, ids, keep, scale, off,    \
                         perm, d_table, partial, dim, V, chunk, acc);                                                      \
    else                                                                                                                   \
      hipLaunchKernelGGL((emb_chunk_sum_kernel<NE, false>), dim3(chunks), dim3(64), 0, st, d_out, ids, keep, scale, off,   \
                         perm, d_table, partial, dim, V, chunk, acc);                                                      \
    hipLaunchKernelGGL((emb_row_finish_kernel<NE>), dim3((unsigned)V), dim3(256), 0, st, off, partial, d_table, dim, V,    \
                       chunk, acc);                                                                                        \
  } while (0)
  G2V_EMB_DISPATCH(G2V_EMB_SORTED);
#undef G2V_EMB_SORTED
#undef G2V_EMB_DISPATCH
  G2V_CHECK_LAUNCH();
  return G2V_OK;
}

extern "C" int g2v_batchnorm_fwd(const float* x, const float* weight, const float* bias, float* running_mean,
                                 float* running_var, int training, int relu, float* y, float* save_mean,
                                 float* save_invstd, int B, int H, g2v_stream_t stream) {
  G2V_REQUIRE(x && weight && bias && y, "null pointer");
  G2V_REQUIRE((running_mean == nullptr) == (running_var == nullptr) && (training || running_mean),
              "running statistics: both, or (training only) neither");
  G2V_REQUIRE(B > 0 && H > 0, "bad size");
  hipLaunchKernelGGL(bn_fwd_kernel, dim3(cdiv(H, BN_FB)), dim3(256), 0, (hipStream_t)stream, x, weight, bias, running_mean,
                     running_var, training, relu, y, save_mean, save_invstd, B, H, g2v_internal_persist_fault_ptr());
  G2V_CHECK_LAUNCH();
  return G2V_OK;
}

// Deferred commit of the running statistics from the SAVED (mean, 1/sqrt(var + eps)) of `steps` training calls, in call order
// (momentum 0.1, unbiased variance): Part d's decoder applies its one BatchNorm1d S-1 times per iteration, and a persistent-kernel
// fault latched AFTER a step's forward (a later decode step, the cells' BPTT, the encoder's backward) must leave the model state
// as it was (round-5 advisor finding).  var = 1 / invstd^2 - eps, clamped at 0 (relative error <= 3e-7 (var + eps) / var).
__global__ void bn_running_update_invstd_kernel(const float* __restrict__ mean, const float* __restrict__ invstd, int64_t stride,
                                                float* __restrict__ rm, float* __restrict__ rv, int steps, int H, int B,
                                                const unsigned* __restrict__ fault) {
  const int f = blockIdx.x * blockDim.x + threadIdx.x;
  if (f >= H) return;
  if (fault && *fault != 0u) return;
  float m = rm[f], v = rv[f];
  const float unbias = (B > 1) ? (float)B / (float)(B - 1) : 1.0f;
  for (int s = 0; s < steps; ++s) {
    const float is = invstd[s * stride + f];
    const float var = fmaxf(1.0f / (is * is) - 1e-5f, 0.f);
    m = 0.9f * m + 0.1f * mean[s * stride + f];
    v = 0.9f * v + 0.1f * (var * unbias);
  }
  rm[f] = m;
  rv[f] = v;
}

extern "C" int g2v_bn_running_update_invstd(const float* save_mean, const float* save_invstd, int64_t step_stride,
                                            float* running_mean, float* running_var, int steps, int H, int B,
                                            g2v_stream_t stream) {
  G2V_REQUIRE(save_mean && save_invstd && running_mean && running_var, "null pointer");
  G2V_REQUIRE(steps > 0 && H > 0 && B > 0 && step_stride >= H, "bad size");
  hipLaunchKernelGGL(bn_running_update_invstd_kernel, dim3(cdiv(H, 256)), dim3(256), 0, (hipStream_t)stream, save_mean, save_invstd,
                     step_stride, running_mean, running_var, steps, H, B, g2v_internal_persist_fault_ptr());
  G2V_CHECK_LAUNCH();
  return G2V_OK;
}

extern "C" int g2v_batchnorm_bwd(const float* dy, const float* x, const float* y, const float* weight,
                                 const float* save_mean, const float* save_invstd, int relu, float* dx, float* dw,
                                 float* db, int B, int H, g2v_stream_t stream) {
  G2V_REQUIRE(dy && x && weight && save_mean && save_invstd && dx && dw && db, "null pointer");
  G2V_REQUIRE(!relu || y, "y required for the ReLU mask");
  G2V_REQUIRE(B > 0 && H > 0, "bad size");
  if (B >= 1024) {        // (see bn_bwd_partial_kernel; the partials need B * H / 8 floats of dx)
    const int nrb = cdiv(B, BN_RB);
    hipLaunchKernelGGL(bn_bwd_partial_kernel, dim3(nrb), dim3(256), 0, (hipStream_t)stream, dy, x, y, save_mean, save_invstd, relu,
                       dx, B, H);
    hipLaunchKernelGGL(bn_bwd_finish_kernel, dim3(cdiv(H, 64)), dim3(256), 0, (hipStream_t)stream, dx, nrb, dw, db, H);
    hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(nrb), dim3(256), 0, (hipStream_t)stream, dy, x, y, weight, save_mean, save_invstd,
                       relu, dw, db, dx, B, H);
    G2V_CHECK_LAUNCH();
    return G2V_OK;
  }
  hipLaunchKernelGGL(bn_bwd_kernel, dim3(cdiv(H, BN_FB)), dim3(256), 0, (hipStream_t)stream, dy, x, y, weight, save_mean,
                     save_invstd, relu, dx, dw, db, B, H);
  G2V_CHECK_LAUNCH();
  return G2V_OK;
}

extern "C" int g2v_batchnorm_bwd_steps(const float* dy, const float* x, const float* y, const float* weight,
                                       const float* save_mean, const float* save_invstd, int64_t stat_stride, int relu, float* dx,
                                       float* dw, float* db, int steps, int B, int H, g2v_stream_t stream) {
  G2V_REQUIRE(dy && x && weight && save_mean && save_invstd && dx && dw && db, "null pointer");
  G2V_REQUIRE(!relu || y, "y required for the ReLU mask");
  G2V_REQUIRE(steps > 0 && B > 0 && H > 0 && stat_stride >= H, "bad size");
  G2V_REQUIRE(B < 1024, "1024 rows or more per step: call g2v_batchnorm_bwd per step (its large-batch kernels)");
  hipLaunchKernelGGL(bn_bwd_steps_kernel, dim3(cdiv(H, BN_FB)), dim3(256), 0, (hipStream_t)stream, dy, x, y, weight, save_mean,
                     save_invstd, stat_stride, relu, dx, dw, db, steps, B, H);
  G2V_CHECK_LAUNCH();
  return G2V_OK;
}

extern "C" int g2v_one_hot_rows(const int64_t* ids, float* out, int64_t ld, int M, int K, g2v_stream_t stream) {
  G2V_REQUIRE(ids && out, "null pointer");
  G2V_REQUIRE(M > 0 && K > 0 && ld >= K, "bad size");
  hipLaunchKernelGGL(one_hot_rows_kernel, dim3(cdiv((int64_t)M * K, 256)), dim3(256), 0, (hipStream_t)stream, ids, out, ld, M, K);
  G2V_CHECK_LAUNCH();
  return G2V_OK;
}

extern "C" int g2v_cross_entropy_fwd_bwd(const float* logits, int64_t ld, const int64_t* targets, float* loss,
                                         float* row_loss, float* dlogits, int64_t ldd, int M, int K, float g_scale,
                                         g2v_stream_t stream) {
  G2V_REQUIRE(logits && targets && loss && row_loss, "null pointer");
  G2V_REQUIRE(M > 0 && K > 0 && ld >= K, "bad size");
  hipLaunchKernelGGL(ce_kernel, dim3(cdiv((int64_t)M * 64, 256)), dim3(256), 0, (hipStream_t)stream, logits, ld, targets,
                     row_loss, dlogits, ldd, M, K, g_scale / (float)M);
  G2V_CHECK_LAUNCH();
  hipLaunchKernelGGL(mean_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, row_loss, M, loss);
  G2V_CHECK_LAUNCH();
  return G2V_OK;
}

extern "C" int g2v_argmax_rows(const float* x, int64_t ld, int64_t* out, int M, int K, g2v_stream_t stream) {
  G2V_REQUIRE(x && out, "null pointer");
  G2V_REQUIRE(M > 0 && K > 0 && ld >= K, "bad size");
  hipLaunchKernelGGL(argmax_rows_kernel, dim3(cdiv((int64_t)M * 64, 256)), dim3(256), 0, (hipStream_t)stream, x, ld, out, M, K);
  G2V_CHECK_LAUNCH();
  return G2V_OK;
}

extern "C" size_t g2v_attn_bwd_workspace(int B, int H) {
  return (B > 0 && H > 0) ? (size_t)B * H * sizeof(float) : 0;
}

extern "C" int g2v_attn_fwd(const float* hp, const float* ep, const float* enc, const float* v, float* weights, float* ctx,
                            int64_t ldctx, int T, int B, int H, g2v_stream_t stream) {
  G2V_REQUIRE(hp && ep && enc && v && weights && ctx, "null pointer");
  G2V_REQUIRE(T > 0 && B > 0 && H > 0 && ldctx >= H, "bad size");
  const size_t lds = (size_t)T * sizeof(float);
  G2V_REQUIRE(lds <= 48 * 1024, "sequence too long for the attention kernel");
  hipLaunchKernelGGL(attn_fwd_kernel, dim3(B), dim3(256), lds, (hipStream_t)stream, hp, ep, enc, v, weights, ctx,
                     ldctx, T, B, H);
  G2V_CHECK_LAUNCH();
  return G2V_OK;
}

extern "C" int g2v_attn_step_fwd(const float* logits, int64_t ldl, int K, int64_t* ids, const float* table, const uint8_t* keep,
                                 float emb_scale, float* ec, int64_t ldec, const float* hp, const float* ep, const float* enc,
                                 const float* v, float* weights, int T, int B, int H, g2v_stream_t stream) {
  G2V_REQUIRE(ids && table && ec && hp && ep && enc && v && weights, "null pointer");
  G2V_REQUIRE(T > 0 && B > 0 && H > 0 && K > 0 && ldec >= 2 * (int64_t)H && (!logits || ldl >= K), "bad size");
  const size_t lds = (size_t)T * sizeof(float);
  G2V_REQUIRE(lds <= 48 * 1024, "sequence too long for the attention kernel");
  hipLaunchKernelGGL(attn_step_fwd_kernel, dim3(B), dim3(256), lds, (hipStream_t)stream, logits, ldl, K, ids, table, keep, emb_scale,
                     ec, ldec, hp, ep, enc, v, weights, T, B, H);
  G2V_CHECK_LAUNCH();
  return G2V_OK;
}

// out (+)= the sum of `n` slabs of `len` floats in slab order (the fixed-order reduction the library's split products use)
extern "C" int g2v_slab_sum(const float* slabs, int n, int64_t len, float* out, int accumulate, g2v_stream_t stream) {
  G2V_REQUIRE(slabs && out, "null pointer");
  G2V_REQUIRE(n > 0 && len > 0, "bad size");
  launch_slab_reduce(slabs, n, len, out, accumulate, (hipStream_t)stream);
  G2V_CHECK_LAUNCH();
  return G2V_OK;
}

extern "C" int g2v_attn_bwd(const float* d_ctx, int64_t ldd, const float* hp, const float* ep, const float* enc,
                            const float* v, const float* weights, float* d_hp, float* d_ep, float* d_enc, float* d_v,
                            int accumulate, int T, int B, int H, void* workspace, size_t workspace_bytes,
                            g2v_stream_t stream) {
  G2V_REQUIRE(d_ctx && hp && ep && enc && v && weights && d_hp && d_ep && d_enc && workspace, "null pointer");
  G2V_REQUIRE(T > 0 && B > 0 && H > 0 && ldd >= H, "bad size");
  if (workspace_bytes < g2v_attn_bwd_workspace(B, H)) {
    set_error("g2v_attn_bwd: workspace too small");
    return G2V_ERR_WORKSPACE;
  }
  const size_t lds = (size_t)T * sizeof(float);
  G2V_REQUIRE(lds <= 48 * 1024, "sequence too long for the attention kernel");
  const int nblk = B;
  hipLaunchKernelGGL(attn_bwd_kernel, dim3(nblk), dim3(256), lds, (hipStream_t)stream, d_ctx, ldd, hp, ep, enc, v, weights,
                     d_hp, d_ep, d_enc, (float*)workspace, accumulate, T, B, H);
  G2V_CHECK_LAUNCH();
  if (d_v) launch_slab_reduce((const float*)workspace, nblk, H, d_v, accumulate, (hipStream_t)stream);      // (NULL: the caller sums the slabs)
  G2V_CHECK_LAUNCH();
  return G2V_OK;
}
