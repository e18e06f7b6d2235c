// misc.hip -- custom_loss (K10), fused clip+Adam (K11), keep-mask generator, small helpers, error plumbing.
#include <stdarg.h>

#include "common.hpp"
#include <new>

namespace g2v {

static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

}  // namespace g2v

// ---- caller-owned contexts (include/g2v.h: g2v_ctx) ------------------------------------------------------------------------------
// "This exchange region is already clear" notes (g2v_cluster_exchange_preclear, dec_rollout.hip).  A cluster launch clears its
// exchange records in front of the kernel: a memset node of 5-12 us on the caller's chain.  A caller whose workspace nobody else
// writes can have that done ahead of time on a side branch; the memset is noted by (address, bytes) and the next cluster launch over
// exactly that region TAKES the note (one shot) instead of clearing again.  Host-side bookkeeping only (under stream capture it
// runs once, at capture: the graph then holds the early memset node and no late one -- so a note must be made and taken on the
// same side of a capture).  Round 6: the table belongs to the caller's CONTEXT (it was one process-global table of 16 entries keyed
// by raw device address: an engine's stale note could make another engine's launch skip its clear), and
// g2v_cluster_exchange_preclear_drop lets the owner void notes when a step is abandoned or a workspace is freed.
struct PreclearNote { const void* p; size_t n; };
struct g2v_ctx {
  G2vOptions opt;
  PreclearNote preclear[16] = {};
  int preclear_next = 0;
};
static g2v_ctx g_default_ctx;
static thread_local g2v_ctx* t_bound_ctx = nullptr;
G2vOptions& g2v_internal_options() { return (t_bound_ctx ? t_bound_ctx : &g_default_ctx)->opt; }
static g2v_ctx& cur_ctx() { return *(t_bound_ctx ? t_bound_ctx : &g_default_ctx); }
void g2v_internal_preclear_note(const void* p, size_t n) {
  g2v_ctx& c = cur_ctx();
  for (auto& e : c.preclear)
    if (e.p == p) { e.n = n; return; }
  c.preclear[c.preclear_next] = PreclearNote{p, n};
  c.preclear_next = (c.preclear_next + 1) % 16;
}
int g2v_internal_preclear_take(const void* p, size_t need) {
  for (auto& e : cur_ctx().preclear)
    if (e.p == p && e.p != nullptr) {
      const bool ok = e.n >= need;
      e = PreclearNote{nullptr, 0};
      return ok ? 1 : 0;
    }
  return 0;
}
// forget every note inside [base, base + bytes) (a launch over that workspace that does not consume one), or all of them
void g2v_internal_preclear_drop(const void* base, size_t bytes) {
  for (auto& e : cur_ctx().preclear)
    if (e.p != nullptr && (base == nullptr || ((const char*)e.p >= (const char*)base && (const char*)e.p < (const char*)base + bytes)))
      e = PreclearNote{nullptr, 0};
}
extern "C" int g2v_cluster_exchange_preclear_drop(const void* workspace, size_t workspace_bytes) {
  g2v_internal_preclear_drop(workspace, workspace_bytes);
  return G2V_OK;
}
extern "C" g2v_ctx* g2v_ctx_create(void) { return new (std::nothrow) g2v_ctx(); }
extern "C" void g2v_ctx_destroy(g2v_ctx* ctx) {
  if (!ctx) return;
  if (t_bound_ctx == ctx) t_bound_ctx = nullptr;
  delete ctx;
}
extern "C" g2v_ctx* g2v_ctx_bind(g2v_ctx* ctx) {
  g2v_ctx* prev = t_bound_ctx;
  t_bound_ctx = ctx;
  return prev;
}
extern "C" int g2v_ctx_set_option(g2v_ctx* ctx, int option, int value) {
  G2vOptions& o = (ctx ? ctx : (t_bound_ctx ? t_bound_ctx : &g_default_ctx))->opt;
  int prev = -1;
  switch (option) {
    case G2V_OPT_PERSISTENT:
      prev = o.persist;
      o.persist = value <= 0 ? 0 : (value > 3 ? 3 : value);
      g2v_internal_preclear_drop(nullptr, 0);      // (a kernel-family switch voids every "already clear" note)
      break;
    case G2V_OPT_GRU_CLUSTER:
      prev = o.gru_cluster;
      o.gru_cluster = value ? 1 : 0;
      g2v_internal_preclear_drop(nullptr, 0);
      break;
    case G2V_OPT_SMALLM_ROWS:
      prev = o.smallm_max_rows;
      if (value >= 0) o.smallm_max_rows = value;
      break;
    case G2V_OPT_GRU_RESIDENT_ROWS:
      prev = o.gru_resident_rows;
      if (value >= 0) o.gru_resident_rows = value;
      break;
    case G2V_OPT_GRU_RESIDENT_BWD:
      prev = o.gru_resident_bwd;
      o.gru_resident_bwd = value ? 1 : 0;
      break;
    default:
      g2v::set_error("g2v_ctx_set_option: unknown option %d", option);
      return G2V_ERR_ARG;
  }
  return prev;
}
extern "C" int g2v_ctx_get_option(const g2v_ctx* ctx, int option) {
  const G2vOptions& o = (ctx ? ctx : (t_bound_ctx ? t_bound_ctx : &g_default_ctx))->opt;
  switch (option) {
    case G2V_OPT_PERSISTENT: return o.persist;
    case G2V_OPT_GRU_CLUSTER: return o.gru_cluster;
    case G2V_OPT_SMALLM_ROWS: return o.smallm_max_rows;
    case G2V_OPT_GRU_RESIDENT_ROWS: return o.gru_resident_rows;
    case G2V_OPT_GRU_RESIDENT_BWD: return o.gru_resident_bwd;
    default: g2v::set_error("g2v_ctx_get_option: unknown option %d", option); return G2V_ERR_ARG;
  }
}

namespace g2v {

// ---- custom_loss: train_eval/train_seq2seq.py:40-88 ----------------------------------------------------
// One thread per (b,d) column walks the T frames twice (norm over TIME, :70).  y is (T,B,D), target (B,T,D).

__global__ __launch_bounds__(256) void custom_loss_kernel(const float* __restrict__ y, const float* __restrict__ tgt,
                                                          float* __restrict__ dy, float* __restrict__ partial, float c1,
                                                          float c2, float c3, float g_scale, int T, int B, int D) {
  __shared__ float red[4][4];
  const int64_t col = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t BD = (int64_t)B * D;
  float l1 = 0.f, cont = 0.f, nrm = 0.f, sq = 0.f;
  if (col < BD) {
    const int b = (int)(col / D), d = (int)(col - (int64_t)b * D);
    float ss = 0.f, prev = 0.f;
    for (int t = 0; t < T; ++t) {
      const float v = y[(int64_t)t * BD + col];
      const float tv = tgt[((int64_t)b * T + t) * D + d];
      l1 += fabsf(v - tv);
      sq += (v - tv) * (v - tv);
      if (t > 0) cont += fabsf(v - prev);
      ss = fmaf(v, v, ss);
      prev = v;
    }
    const float cn = loss_col_coef(c3, ss, nrm);
    if (dy) {
      float vm = 0.f, v = y[col], vp;
      for (int t = 0; t < T; ++t) {
        vp = (t + 1 < T) ? y[(int64_t)(t + 1) * BD + col] : 0.f;
        const float tv = tgt[((int64_t)b * T + t) * D + d];
        const int code = loss_sign_code(v - tv) | ((t > 0 ? loss_sign_code(v - vm) : 1) << 2) |
                         ((t + 1 < T ? loss_sign_code(vp - v) : 1) << 4);
        dy[(int64_t)t * BD + col] = loss_grad(loss_grad_const(c1, c2, code), cn, v) * g_scale;
        vm = v;
        v = vp;
      }
    }
  }
  l1 = wave_sum(l1);
  cont = wave_sum(cont);
  nrm = wave_sum(nrm);
  sq = wave_sum(sq);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) {
    red[0][wave] = l1;
    red[1][wave] = cont;
    red[2][wave] = nrm;
    red[3][wave] = sq;
  }
  __syncthreads();
  if (threadIdx.x < 4)
    partial[(int64_t)blockIdx.x * 4 + threadIdx.x] =
        (red[threadIdx.x][0] + red[threadIdx.x][1]) + (red[threadIdx.x][2] + red[threadIdx.x][3]);
}

// Compile-time T: the whole (b, d) column of y and of the target lives in registers (2 * TT values), every load is
// issued before the first use (one HBM round trip instead of 2 * TT dependent ones), and the gradient pass re-reads
// nothing: 2 reads + 1 write of (T,B,D), the algorithmic minimum.
template <int TT>
__global__ __launch_bounds__(256) void custom_loss_reg_kernel(const float* __restrict__ y, const float* __restrict__ tgt,
                                                              float* __restrict__ dy, float* __restrict__ partial, float c1,
                                                              float c2, float c3, float g_scale, int B, int D) {
  __shared__ float red[4][4];
  const int64_t col = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t BD = (int64_t)B * D;
  float l1 = 0.f, cont = 0.f, nrm = 0.f, sq = 0.f;
  if (col < BD) {
    const int b = (int)(col / D), d = (int)(col - (int64_t)b * D);
    float v[TT], tv[TT];
    const float* yp = y + col;
    const float* tp = tgt + (int64_t)b * TT * D + d;
#pragma unroll
    for (int t = 0; t < TT; ++t) v[t] = yp[(int64_t)t * BD];
#pragma unroll
    for (int t = 0; t < TT; ++t) tv[t] = tp[(int64_t)t * D];
    float ss = 0.f;
#pragma unroll
    for (int t = 0; t < TT; ++t) {
      l1 += fabsf(v[t] - tv[t]);
      sq += (v[t] - tv[t]) * (v[t] - tv[t]);
      if (t > 0) cont += fabsf(v[t] - v[t - 1]);
      ss = fmaf(v[t], v[t], ss);
    }
    const float cn = loss_col_coef(c3, ss, nrm);
    if (dy) {
      float* dp = dy + col;
#pragma unroll
      for (int t = 0; t < TT; ++t) {
        const int code = loss_sign_code(v[t] - tv[t]) | ((t > 0 ? loss_sign_code(v[t] - v[t - 1]) : 1) << 2) |
                         ((t + 1 < TT ? loss_sign_code(v[t + 1] - v[t]) : 1) << 4);
        dp[(int64_t)t * BD] = loss_grad(loss_grad_const(c1, c2, code), cn, v[t]) * g_scale;
      }
    }
  }
  l1 = wave_sum(l1);
  cont = wave_sum(cont);
  nrm = wave_sum(nrm);
  sq = wave_sum(sq);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) {
    red[0][wave] = l1;
    red[1][wave] = cont;
    red[2][wave] = nrm;
    red[3][wave] = sq;
  }
  __syncthreads();
  if (threadIdx.x < 4)
    partial[(int64_t)blockIdx.x * 4 + threadIdx.x] =
        (red[threadIdx.x][0] + red[threadIdx.x][1]) + (red[threadIdx.x][2] + red[threadIdx.x][3]);
}

__global__ void custom_loss_finalize_kernel(const float* __restrict__ partial, int nblk, float* __restrict__ terms,
                                            float c1, float c2, float c3, float inv_n) {
  __shared__ float red[4][4];
  float s[4] = {0.f, 0.f, 0.f, 0.f};
  for (int k = threadIdx.x; k < nblk; k += 256)
    for (int j = 0; j < 4; ++j) s[j] += partial[(int64_t)k * 4 + j];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int j = 0; j < 4; ++j) {
    s[j] = wave_sum(s[j]);
    if (lane == 0) red[j][wave] = s[j];
  }
  __syncthreads();
  if (threadIdx.x == 0)
    loss_terms_write(terms, (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]), (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]),
                     (red[2][0] + red[2][1]) + (red[2][2] + red[2][3]), (red[3][0] + red[3][1]) + (red[3][2] + red[3][3]), c1, c2,
                     c3, inv_n);
}

// ---- MSE (DAE reconstruction loss, train_eval/train_seq2seq.py:208-222) -----------------------------------------
__global__ __launch_bounds__(256) void mse_kernel(const float* __restrict__ y, const float* __restrict__ t,
                                                  float* __restrict__ dy, float* __restrict__ partial, int64_t n,
                                                  float gcoef) {
  __shared__ float red[4];
  float s = 0.f;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (int64_t)gridDim.x * 256) {
    const float d = y[e] - t[e];
    s += d * d;
    if (dy) dy[e] = gcoef * d;
  }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}
__global__ void mse_finalize_kernel(const float* __restrict__ partial, int nblk, float* __restrict__ out, float inv_n) {
  __shared__ float red[4];
  float s = 0.f;
  for (int k = threadIdx.x; k < nblk; k += 256) s += partial[k];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) out[0] = ((red[0] + red[1]) + (red[2] + red[3])) * inv_n;
}

// ---- clip_grad_norm_ + Adam -------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void sumsq_kernel(const float* __restrict__ g, int64_t n, float* __restrict__ partial,
                                                    int32_t* __restrict__ step_counter, const unsigned* __restrict__ fault) {
  __shared__ float red[4];
  float s = 0.f;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (int64_t)gridDim.x * 256) s += g[e] * g[e];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
    if (blockIdx.x == 0 && !(fault && *fault != 0u)) step_counter[0] += 1;
  }
}

__global__ __launch_bounds__(256) void clip_adam_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                        float* __restrict__ m, float* __restrict__ v, int64_t n,
                                                        const float* __restrict__ partial, int npart,
                                                        const int32_t* __restrict__ step_counter,
                                                        float* __restrict__ gnorm_out, float max_norm, float grad_scale,
                                                        float lr, float b1, float b2, float eps,
                                                        const unsigned* __restrict__ fault, const float* __restrict__ rb0,
                                                        const float* __restrict__ rb1, const float* __restrict__ rb2,
                                                        float* __restrict__ rb_out) {
  __shared__ float red[4];
  __shared__ float bc;
  if (rb_out && blockIdx.x == 0 && threadIdx.x == 0) {      // the iteration's read-back (g2v_iteration_readback), latch included
    rb_out[0] = rb0 ? rb0[0] : 0.f;
    rb_out[1] = rb1 ? rb1[0] : 0.f;
    rb_out[2] = rb2 ? rb2[0] : 0.f;
    rb_out[3] = fault ? (float)fault[0] : 0.f;
  }
  if (fault && *fault != 0u) return;      // a latched rollout fault: the gradients are garbage, the parameters and moments stay
  float s = 0.f;
  for (int k = threadIdx.x; k < npart; k += 256) s += partial[k];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) bc = sqrtf((red[0] + red[1]) + (red[2] + red[3]));
  __syncthreads();
  const float total = bc * grad_scale;       // norm of the (scaled) gradient
  float coef = max_norm / (total + 1e-6f);
  coef = fminf(coef, 1.0f) * grad_scale;
  if (blockIdx.x == 0 && threadIdx.x == 0 && gnorm_out) gnorm_out[0] = total;
  const int t = step_counter[0];
  const float bc1 = (float)(1.0 - pow((double)b1, (double)t));
  const float bc2s = (float)sqrt(1.0 - pow((double)b2, (double)t));
  const float step_size = lr / bc1;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (int64_t)gridDim.x * 256) {
    const float gg = g[e] * coef;
    const float mm = m[e] * b1 + (1.0f - b1) * gg;
    const float vv = v[e] * b2 + (1.0f - b2) * gg * gg;
    m[e] = mm;
    v[e] = vv;
    const float denom = sqrtf(vv) / bc2s + eps;
    p[e] = p[e] - step_size * (mm / denom);
  }
}

// ---- philox4x32-10 keep masks ------------------------------------------------------------------------------
__device__ __forceinline__ void philox_round(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
  const uint64_t p0 = (uint64_t)0xD2511F53u * c[0];
  const uint64_t p1 = (uint64_t)0xCD9E8D57u * c[2];
  const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0;
  const uint32_t n1 = (uint32_t)p1;
  const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
  const uint32_t n3 = (uint32_t)p0;
  c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
}

// One philox block = 4 consecutive outputs (block g <-> elements 4g .. 4g+3, counter (g, offset), key = seed): the mapping every
// version of this kernel has had.  A thread takes FOUR consecutive blocks and stores its 16 flags with one 16-byte store (the
// first version stored them byte by byte: 84 us for the 18 M flags of a B = 4096 step, store-issue bound; the arithmetic is ~10 us).
__device__ __forceinline__ uint32_t philox_keep4(int64_t g, uint64_t off, uint64_t seed, float keep_prob) {
  uint32_t c[4] = {(uint32_t)g, (uint32_t)((uint64_t)g >> 32), (uint32_t)off, (uint32_t)(off >> 32)};
  uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    philox_round(c, k0, k1);
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  uint32_t w = 0u;
#pragma unroll
  for (int j = 0; j < 4; ++j) w |= (((float)(c[j] >> 8) * (1.0f / 16777216.0f) < keep_prob) ? 1u : 0u) << (8 * j);
  return w;
}
template <bool ALIGNED16>
__global__ __launch_bounds__(256) void keep_mask_kernel(uint8_t* __restrict__ keep, int64_t n, float keep_prob, uint64_t seed,
                                                        const int64_t* __restrict__ offset_counter, int64_t offset_add) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;   // elements 16 i .. 16 i + 15
  if (i * 16 >= n) return;
  const uint64_t off = (uint64_t)(offset_counter[0] + offset_add);
  uint32_t w[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) w[k] = philox_keep4(4 * i + k, off, seed, keep_prob);
  if (ALIGNED16 && i * 16 + 16 <= n) {
    *reinterpret_cast<uint4*>(keep + i * 16) = make_uint4(w[0], w[1], w[2], w[3]);
  } else {
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const int64_t e = i * 16 + k;
      if (e < n) keep[e] = (uint8_t)((w[k >> 2] >> (8 * (k & 3))) & 0xffu);
    }
  }
}
__global__ void tick_kernel(int64_t* c, int64_t n) { c[0] += n; }

// Dropout of a (row-mapped) dense-layer input with the mask drawn in the same kernel: out[m, k] = keep(m K + k) ? x[row(m), k] *
// scale : 0, keep(e) = element e of the g2v_keep_mask stream (seed, offset_counter[0] + offset_add): bit for bit what
// g2v_keep_mask + g2v_mask_rows produce, without the mask tensor and its two passes.  One philox block (4 elements) per thread.
__global__ __launch_bounds__(256) void dropout_rows_kernel(const float* __restrict__ x, int64_t ldx, int rows_inner, int64_t so,
                                                           int64_t si, float keep_prob, float scale, uint64_t seed,
                                                           const int64_t* __restrict__ offset_counter, int64_t offset_add,
                                                           float* __restrict__ out, int64_t ldo, int M, int K) {
  const int64_t n = (int64_t)M * K;
  const uint64_t off = (uint64_t)(offset_counter[0] + offset_add);
  for (int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x; g * 4 < n; g += (int64_t)gridDim.x * 256) {
    const uint32_t w = philox_keep4(g, off, seed, keep_prob);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int64_t e = g * 4 + j;
      if (e < n) {
        const int m = (int)(e / K), k = (int)(e - (int64_t)m * K);
        const float* xr = x + (rows_inner > 0 ? (int64_t)(m / rows_inner) * so + (int64_t)(m % rows_inner) * si : (int64_t)m * ldx);
        out[(int64_t)m * ldo + k] = ((w >> (8 * j)) & 0xffu) ? xr[k] * scale : 0.f;
      }
    }
  }
}

__global__ void fill_kernel(float* __restrict__ p, float v, int64_t n) {
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) p[e] = v;
}

__global__ void scale_kernel(const float* __restrict__ in, const float* __restrict__ s, float* __restrict__ out, int64_t n) {
  const float f = s[0];
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) out[e] = in[e] * f;
}

__global__ void mask_mul_kernel(const float* __restrict__ in, const uint8_t* __restrict__ keep,
                                const float* __restrict__ pos_of, float scale, float* __restrict__ out, int64_t n) {
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) {
    const bool on = keep ? (keep[e] != 0) : (pos_of[e] > 0.f);
    out[e] = on ? in[e] * scale : 0.f;
  }
}

__global__ void add_halves_kernel(const float* __restrict__ a, int64_t lda, const float* __restrict__ b, int64_t ldb,
                                  float* __restrict__ out, int64_t ldo, int64_t M, int H) {
  const int64_t total = M * H;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = e / H;
    const int c = (int)(e - r * H);
    out[r * ldo + c] = a[r * lda + c] + b[r * ldb + c];
  }
}


// ---- calibration probes (bench.py only): what this box sustains, next to the datasheet peaks --------------------------
// fp32 MFMA: every wave runs `iters` rounds of 8 independent v_mfma_f32_16x16x4_f32 chains (no memory traffic at all).
__global__ __launch_bounds__(256) void mfma_probe_kernel(float* __restrict__ out, int iters) {
  f32x4 acc[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const float a = 1.0f + 1e-6f * (float)threadIdx.x, b = 1.0f - 1e-6f * (float)threadIdx.x;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = mfma16(a, b, acc[j]);
  }
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < 8; ++j) s += acc[j][0] + acc[j][1] + acc[j][2] + acc[j][3];
  out[(int64_t)blockIdx.x * 256 + threadIdx.x] = s;
}
// streaming copy, 16 bytes per thread per iteration
__global__ __launch_bounds__(256) void copy_probe_kernel(const float4* __restrict__ src, float4* __restrict__ dst, int64_t n4) {
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n4; e += (int64_t)gridDim.x * 256) dst[e] = src[e];
}

}  // namespace g2v

using namespace g2v;

extern "C" const char* g2v_version(void) { return "g2v-hip 0.1 (gfx950, fp32 MFMA 16x16x4)"; }
extern "C" const char* g2v_last_error(void) { return g_err; }
extern "C" int g2v_device_ok(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return 0;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, 0) != hipSuccess) return 0;
  return strstr(prop.gcnArchName, "gfx950") != nullptr ? 1 : 0;
}

extern "C" int g2v_custom_loss_blocks(int B, int D) { return (B > 0 && D > 0) ? cdiv((int64_t)B * D, 256) : 0; }

extern "C" int g2v_custom_loss_fwd_bwd(const float* y, const float* target, float* dy, float* terms, float* partial,
                                       float w_l1, float w_cont, float w_var, float g_scale, int T, int B, int D,
                                       g2v_stream_t stream) {
  G2V_REQUIRE(y && target && terms && partial, "null pointer");
  G2V_REQUIRE(T > 0 && B > 0 && D > 0, "bad size");
  const float n = (float)T * (float)B * (float)D;
  const float c1 = w_l1 / n, c2 = w_cont / n, c3 = w_var / n;
  const int nblk = g2v_custom_loss_blocks(B, D);
  if (T == 34)        // the BASELINE chunk length: whole columns in registers (see custom_loss_reg_kernel)
    hipLaunchKernelGGL(custom_loss_reg_kernel<34>, dim3(nblk), dim3(256), 0, (hipStream_t)stream, y, target, dy, partial, c1,
                       c2, c3, g_scale, B, D);
  else if (T == 20)   // config/VQ-VAE.yml n_poses
    hipLaunchKernelGGL(custom_loss_reg_kernel<20>, dim3(nblk), dim3(256), 0, (hipStream_t)stream, y, target, dy, partial, c1,
                       c2, c3, g_scale, B, D);
  else
    hipLaunchKernelGGL(custom_loss_kernel, dim3(nblk), dim3(256), 0, (hipStream_t)stream, y, target, dy, partial, c1, c2,
                       c3, g_scale, T, B, D);
  G2V_CHECK_LAUNCH();
  hipLaunchKernelGGL(custom_loss_finalize_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, partial, nblk, terms, c1,
                     c2, c3, 1.0f / n);
  G2V_CHECK_LAUNCH();
  return G2V_OK;
}

extern "C" int g2v_mse_blocks(int64_t n) {
  if (n <= 0) return 0;
  int64_t b = (n + 1023) / 1024;
  return (int)(b > 1024 ? 1024 : b);
}

extern "C" int g2v_mse_fwd_bwd(const float* y, const float* target, float* dy, float* loss, float* partial, int64_t n,
                               float g_scale, g2v_stream_t stream) {
  G2V_REQUIRE(y && target && loss && partial, "null pointer");
  G2V_REQUIRE(n > 0, "bad size");
  const int nblk = g2v_mse_blocks(n);
  hipLaunchKernelGGL(mse_kernel, dim3(nblk), dim3(256), 0, (hipStream_t)stream, y, target, dy, partial, n,
                     2.0f * g_scale / (float)n);
  G2V_CHECK_LAUNCH();
  hipLaunchKernelGGL(mse_finalize_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, partial, nblk, loss, 1.0f / (float)n);
  G2V_CHECK_LAUNCH();
  return G2V_OK;
}

extern "C" int g2v_adam_blocks(int64_t n) {
  if (n <= 0) return 0;
  int64_t b = (n + 1023) / 1024;
  return (int)(b > 1024 ? 1024 : b);
}

static int clip_adam_impl(float* param, const float* grad, float* m, float* v, int64_t n, float* partial, int32_t* step_counter,
                          float* gnorm_out, float max_norm, float grad_scale, float lr, float beta1, float beta2, float eps,
                          const float* rb0, const float* rb1, const float* rb2, float* rb_out, g2v_stream_t stream) {
  const int nblk = g2v_adam_blocks(n);
  const unsigned* fault = g2v_internal_persist_fault_ptr();
  hipLaunchKernelGGL(sumsq_kernel, dim3(nblk), dim3(256), 0, (hipStream_t)stream, grad, n, partial, step_counter, fault);
  if (hipGetLastError() != hipSuccess) return G2V_ERR_LAUNCH;
  hipLaunchKernelGGL(clip_adam_kernel, dim3(nblk), dim3(256), 0, (hipStream_t)stream, param, grad, m, v, n, partial,
                     nblk, step_counter, gnorm_out, max_norm, grad_scale, lr, beta1, beta2, eps, fault, rb0, rb1, rb2, rb_out);
  return hipGetLastError() == hipSuccess ? G2V_OK : G2V_ERR_LAUNCH;
}
extern "C" int g2v_clip_adam_step(float* param, const float* grad, float* m, float* v, int64_t n, float* partial,
                                  int32_t* step_counter, float* gnorm_out, float max_norm, float grad_scale, float lr,
                                  float beta1, float beta2, float eps, g2v_stream_t stream) {
  G2V_REQUIRE(param && grad && m && v && partial && step_counter, "null pointer");
  G2V_REQUIRE(n > 0, "bad size");
  const int rc = clip_adam_impl(param, grad, m, v, n, partial, step_counter, gnorm_out, max_norm, grad_scale, lr, beta1, beta2, eps,
                                nullptr, nullptr, nullptr, nullptr, stream);
  if (rc != G2V_OK) set_error("g2v_clip_adam_step: launch failed");
  return rc;
}
// the same step with g2v_iteration_readback folded into its second launch (no launch of its own at the end of the iteration):
// out4 is written whether or not the fault latch holds the update back
extern "C" int g2v_clip_adam_step_readback(float* param, const float* grad, float* m, float* v, int64_t n, float* partial,
                                           int32_t* step_counter, float* gnorm_out, float max_norm, float grad_scale, float lr,
                                           float beta1, float beta2, float eps, const float* s0, const float* s1,
                                           const float* s2, float* out4, g2v_stream_t stream) {
  G2V_REQUIRE(param && grad && m && v && partial && step_counter && out4, "null pointer");
  G2V_REQUIRE(n > 0, "bad size");
  const int rc = clip_adam_impl(param, grad, m, v, n, partial, step_counter, gnorm_out, max_norm, grad_scale, lr, beta1, beta2, eps,
                                s0, s1, s2, out4, stream);
  if (rc != G2V_OK) set_error("g2v_clip_adam_step_readback: launch failed");
  return rc;
}
extern "C" int g2v_keep_mask_at(uint8_t* keep, int64_t n, float keep_prob, uint64_t seed, const int64_t* offset_counter,
                                int64_t offset_add, g2v_stream_t stream) {
  G2V_REQUIRE(keep && offset_counter, "null pointer");
  G2V_REQUIRE(n > 0, "bad size");
  const int64_t nthreads = (n + 15) / 16;
  if ((reinterpret_cast<uintptr_t>(keep) & 15) == 0)
    hipLaunchKernelGGL(keep_mask_kernel<true>, dim3(cdiv(nthreads, 256)), dim3(256), 0, (hipStream_t)stream, keep, n, keep_prob,
                       seed, offset_counter, offset_add);
  else
    hipLaunchKernelGGL(keep_mask_kernel<false>, dim3(cdiv(nthreads, 256)), dim3(256), 0, (hipStream_t)stream, keep, n, keep_prob,
                       seed, offset_counter, offset_add);
  G2V_CHECK_LAUNCH();
  return G2V_OK;
}
extern "C" int g2v_counter_add(int64_t* counter, int64_t n, g2v_stream_t stream) {
  G2V_REQUIRE(counter, "null pointer");
  hipLaunchKernelGGL(tick_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, counter, n);
  G2V_CHECK_LAUNCH();
  return G2V_OK;
}
extern "C" int g2v_keep_mask(uint8_t* keep, int64_t n, float keep_prob, uint64_t seed, int64_t* offset_counter,
                             g2v_stream_t stream) {
  const int rc = g2v_keep_mask_at(keep, n, keep_prob, seed, offset_counter, 0, stream);
  return rc != G2V_OK ? rc : g2v_counter_add(offset_counter, 1, stream);
}
extern "C" int g2v_dropout_rows(const float* x, int64_t ldx, int rows_inner, int64_t stride_outer, int64_t stride_inner,
                                float keep_prob, float scale, uint64_t seed, const int64_t* offset_counter, int64_t offset_add,
                                float* out, int64_t ldo, int M, int K, g2v_stream_t stream) {
  G2V_REQUIRE(x && out && offset_counter, "null pointer");
  G2V_REQUIRE(M > 0 && K > 0 && ldo >= K, "bad size");
  int64_t blocks = cdiv(((int64_t)M * K + 3) / 4, 256);
  if (blocks > 16384) blocks = 16384;
  hipLaunchKernelGGL(dropout_rows_kernel, dim3((int)blocks), dim3(256), 0, (hipStream_t)stream, x, ldx, rows_inner, stride_outer,
                     stride_inner, keep_prob, scale, seed, offset_counter, offset_add, out, ldo, M, K);
  G2V_CHECK_LAUNCH();
  return G2V_OK;
}

// dst_k[i] = src_k[i] (src_k == NULL: 0) for up to COPY_SEGS segments per launch; blockIdx.y = segment.
constexpr int COPY_SEGS = 48;
struct CopyBatch {
  const float* src[COPY_SEGS];
  float* dst[COPY_SEGS];
  int64_t n[COPY_SEGS];
};
__global__ __launch_bounds__(256) void copy_segments_kernel(CopyBatch cb) {
  const float* __restrict__ src = cb.src[blockIdx.y];
  float* __restrict__ dst = cb.dst[blockIdx.y];
  const int64_t n = cb.n[blockIdx.y];
  const bool vec = ((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst)) & 15) == 0;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x, t0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (vec) {
    const int64_t n4 = n >> 2;
    for (int64_t i = t0; i < n4; i += stride)
      reinterpret_cast<float4*>(dst)[i] = src ? reinterpret_cast<const float4*>(src)[i] : make_float4(0.f, 0.f, 0.f, 0.f);
    for (int64_t i = (n4 << 2) + t0; i < n; i += stride) dst[i] = src ? src[i] : 0.f;
  } else {
    for (int64_t i = t0; i < n; i += stride) dst[i] = src ? src[i] : 0.f;
  }
}

extern "C" int g2v_copy_segments(const float* const* src, float* const* dst, const int64_t* n, int nseg,
                                 g2v_stream_t stream) {
  G2V_REQUIRE(src && dst && n, "null pointer");
  G2V_REQUIRE(nseg >= 0, "bad size");
  for (int s0 = 0; s0 < nseg; s0 += COPY_SEGS) {
    CopyBatch cb;
    const int cnt = nseg - s0 < COPY_SEGS ? nseg - s0 : COPY_SEGS;
    int64_t nmax = 1;
    for (int k = 0; k < COPY_SEGS; ++k) {
      const int j = k < cnt ? s0 + k : s0;
      G2V_REQUIRE(dst[j] && n[j] >= 0, "null destination / negative length");
      cb.src[k] = src[j]; cb.dst[k] = dst[j]; cb.n[k] = k < cnt ? n[j] : 0;
      if (cb.n[k] > nmax) nmax = cb.n[k];
    }
    int gx = cdiv(nmax, 1024);
    if (gx > 64) gx = 64;
    hipLaunchKernelGGL(copy_segments_kernel, dim3(gx, cnt), dim3(256), 0, (hipStream_t)stream, cb);
    G2V_CHECK_LAUNCH();
  }
  return G2V_OK;
}

extern "C" int g2v_fill_f32(float* p, float v, int64_t n, g2v_stream_t stream) {
  G2V_REQUIRE(p, "null pointer");
  if (n <= 0) return G2V_OK;
  int blocks = cdiv(n, 256);
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(fill_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, p, v, n);
  G2V_CHECK_LAUNCH();
  return G2V_OK;
}

// The scalars a training iteration reads back, gathered into ONE small array so that the host needs one device-to-host copy
// (train_eval/train_seq2seq.py: loss.item() of the reference): out[k] = *src[k] (NULL: 0) for k < 3, out[3] = the persistent
// rollouts' fault latch as a float (0 / 1 / 2).
__global__ void iteration_readback_kernel(const float* __restrict__ s0, const float* __restrict__ s1, const float* __restrict__ s2,
                                          const unsigned* __restrict__ fault, float* __restrict__ out) {
  if (threadIdx.x == 0) {
    out[0] = s0 ? s0[0] : 0.f;
    out[1] = s1 ? s1[0] : 0.f;
    out[2] = s2 ? s2[0] : 0.f;
    out[3] = fault ? (float)fault[0] : 0.f;
  }
}
extern "C" int g2v_iteration_readback(const float* s0, const float* s1, const float* s2, float* out4, g2v_stream_t stream) {
  G2V_REQUIRE(out4, "null pointer");
  hipLaunchKernelGGL(iteration_readback_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, s0, s1, s2,
                     g2v_internal_persist_fault_ptr(), out4);
  G2V_CHECK_LAUNCH();
  return G2V_OK;
}

extern "C" int g2v_scale_f32(const float* in, const float* scalar, float* out, int64_t n, g2v_stream_t stream) {
  G2V_REQUIRE(in && scalar && out, "null pointer");
  if (n <= 0) return G2V_OK;
  int blocks = cdiv(n, 256);
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(scale_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, in, scalar, out, n);
  G2V_CHECK_LAUNCH();
  return G2V_OK;
}

extern "C" int g2v_mask_mul(const float* in, const uint8_t* keep, const float* positive_of, float scale, float* out,
                            int64_t n, g2v_stream_t stream) {
  G2V_REQUIRE(in && out && (keep || positive_of), "null pointer");
  if (n <= 0) return G2V_OK;
  int blocks = cdiv(n, 256);
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(mask_mul_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, in, keep, positive_of, scale, out, n);
  G2V_CHECK_LAUNCH();
  return G2V_OK;
}

// one wave per row: K consecutive floats of the (row-mapped) input, the K keep bytes of the output row
__global__ __launch_bounds__(256) void mask_rows_kernel(const float* __restrict__ x, int64_t ldx, int rows_inner, int64_t so, int64_t si,
                                                        const uint8_t* __restrict__ keep, float scale, float* __restrict__ out,
                                                        int64_t ldo, int M, int K) {
  const int lane = threadIdx.x & 63;
  for (int m = blockIdx.x * 4 + (threadIdx.x >> 6); m < M; m += gridDim.x * 4) {
    const float* xr = x + (rows_inner > 0 ? (int64_t)(m / rows_inner) * so + (int64_t)(m % rows_inner) * si : (int64_t)m * ldx);
    const uint8_t* kr = keep + (int64_t)m * K;
    float* o = out + (int64_t)m * ldo;
    for (int k = lane; k < K; k += 64) o[k] = kr[k] ? xr[k] * scale : 0.f;
  }
}

extern "C" int g2v_mask_rows(const float* x, int64_t ldx, int rows_inner, int64_t stride_outer, int64_t stride_inner,
                             const uint8_t* keep, float scale, float* out, int64_t ldo, int M, int K, g2v_stream_t stream) {
  G2V_REQUIRE(x && keep && out, "null pointer");
  G2V_REQUIRE(M > 0 && K > 0 && ldo >= K, "bad size");
  int blocks = cdiv(M, 4);
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(mask_rows_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, ldx, rows_inner, stride_outer,
                     stride_inner, keep, scale, out, ldo, M, K);
  G2V_CHECK_LAUNCH();
  return G2V_OK;
}

extern "C" int g2v_transpose(const float* in, float* out, int rows, int cols, g2v_stream_t stream) {
  G2V_REQUIRE(in && out, "null pointer");
  G2V_REQUIRE(rows > 0 && cols > 0, "bad size");
  launch_transpose(in, out, rows, cols, (hipStream_t)stream);
  G2V_CHECK_LAUNCH();
  return G2V_OK;
}

extern "C" int g2v_add_halves(const float* a, int64_t lda, const float* b, int64_t ldb, float* out, int64_t ldo,
                              int64_t M, int H, g2v_stream_t stream) {
  G2V_REQUIRE(a && b && out, "null pointer");
  G2V_REQUIRE(M > 0 && H > 0, "bad size");
  int blocks = cdiv(M * H, 256);
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(add_halves_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a, lda, b, ldb, out, ldo, M, H);
  G2V_CHECK_LAUNCH();
  return G2V_OK;
}

extern "C" int g2v_probe_mfma_f32(float* scratch, int blocks, int iters, g2v_stream_t stream) {
  G2V_REQUIRE(scratch, "null pointer");
  G2V_REQUIRE(blocks > 0 && iters > 0, "bad size");
  hipLaunchKernelGGL(mfma_probe_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, scratch, iters);
  G2V_CHECK_LAUNCH();
  return G2V_OK;
}

extern "C" int g2v_probe_copy(const float* src, float* dst, int64_t n, g2v_stream_t stream) {
  G2V_REQUIRE(src && dst, "null pointer");
  G2V_REQUIRE(n > 0 && (n & 3) == 0, "n must be a positive multiple of 4");
  hipLaunchKernelGGL(copy_probe_kernel, dim3(256 * 8), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<const float4*>(src),
                     reinterpret_cast<float4*>(dst), n >> 2);
  G2V_CHECK_LAUNCH();
  return G2V_OK;
}
