/* g2v.h -- C-ABI of the MI355X-native Gesture2Vec hot path (libg2v_hip.so).
 *
 * The reference (pjyazdian/Gesture2Vec) is pure Python on PyTorch: it has no FFI / plugin
 * boundary of its own.  The boundary it DOES have is the Python operator
 * surface (scripts/model/ classes, scripts/train_eval/train_seq2seq.py functions); each
 * entry point below names the reference call site whose arithmetic it replaces.  The Python
 * host in gesture2vec_amd/ binds these with ctypes (INTEGRATION.md shows the stub) and exposes
 * the reference's own class / function names on top.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer into caller-owned memory (PyTorch-ROCm allocations);
 *     the library never allocates, frees, synchronises or touches the default stream;
 *   - `stream` is a hipStream_t passed as void*; all work is stream-ordered, capture-safe
 *     (hipGraph), re-entrant per stream;
 *   - return value: 0 = success, negative = G2V_ERR_*; g2v_last_error() gives a message;
 *   - all arithmetic is IEEE fp32 (MFMA v_mfma_f32_16x16x4_f32 = exact fp32 fma chains);
 *     code indices are int64 (the reference's torch.argmin dtype), masks are uint8 keep-masks;
 *   - matrices are row-major; "ld" = row stride in elements;
 *   - sequence tensors are (T,B,F) "time-major" unless stated.
 */
#ifndef G2V_H
#define G2V_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* g2v_stream_t; /* hipStream_t */

#define G2V_OK 0
#define G2V_ERR_ARG (-1)         /* bad argument (null pointer, non-positive size, misalignment) */
#define G2V_ERR_LAUNCH (-2)      /* hipLaunch / runtime error */
#define G2V_ERR_WORKSPACE (-3)   /* workspace too small */
#define G2V_ERR_UNSUPPORTED (-4) /* configuration outside what the kernels implement */

const char* g2v_version(void);
const char* g2v_last_error(void);
/* 1 when a gfx950 device is visible to this process, else 0 (never throws). */
int g2v_device_ok(void);

/* ------------------------------------------------------------------------------------------
 * Dense layers.  Replaces nn.Linear call sites on the path:
 *   EncoderRNN.in_layer            model/Autoencoder_VQVAE_model.py:93
 *   VQ_Payam_EMA.pre_linear        model/Autoencoder_VQVAE_model.py:1230
 *   GRU input projections          (nn.GRU internals, :94, text2embedding_model.py:131)
 *   DAE_Network encoder/decoder    model/DAE_model.py:107-110
 * y[m, n] = act( sum_k xin[m, k] * w[n, k] + bias[n] ),  xin = x * keep * x_scale (keep optional)
 * Row m of x is read at  x + (m / rows_inner) * stride_outer + (m % rows_inner) * stride_inner
 * when rows_inner > 0 (lets a (B,T,D) tensor be consumed in (T,B) row order without a copy),
 * else at x + m * ldx.  x_keep (uint8, may be NULL) is indexed [m * K + k].
 * act: 0 = identity, 1 = ReLU, 2 = tanh.
 * ------------------------------------------------------------------------------------------ */
/* measurement only: largest row count served by the wave-per-tile kernel of g2v_linear_fwd / g2v_linear_bwd_data
 * (0 = never); returns the previous value, rows < 0 only queries */
int g2v_linear_set_smallm_rows(int rows);
/* IMPLEMENTATION SWITCHES, complete list (everything else is per call).  The library reads NO environment variable.  All of
 * them select between implementations that produce the same results (measurements, parity tests, and the fall-back after a latched
 * residency fault of the persistent kernels):
 *   G2V_OPT_SMALLM_ROWS   row count up to which the wave-per-tile dense kernels are used (default 1024)
 *   G2V_OPT_PERSISTENT    0..3: persistent rollout kernels vs one launch per step (default 1; see g2v_dec_rollout_set_persistent)
 *   G2V_OPT_GRU_CLUSTER   0 / 1: small-batch g2v_gru_seq_fwd / _bwd as one persistent launch vs one launch per step
 *   G2V_OPT_GRU_RESIDENT_ROWS   batch rows from which g2v_gru_seq_fwd (192 < H <= 208) and g2v_gru_seq_bwd (H = 200) keep W_hh
 *                         resident in each CU's registers + LDS for the whole sequence instead of streaming it from L2 every step
 *                         (default 1025: every batch beyond the small-batch cluster kernels; 0 = never).  Forward: bitwise the
 *                         streaming kernel's results; BPTT: equal to summation order
 *   G2V_OPT_GRU_RESIDENT_BWD    0 / 1 (default 1): the BPTT too.  A resident launch takes a CU's whole LDS and register file, so
 *                         nothing co-resides with it: a caller that runs other launches BESIDE the BPTT (the VQ-VAE engine: the
 *                         decoder's weight gradients on other queues) turns it off in its context
 * They live in a CALLER-OWNED CONTEXT (round 6; until round 5 they were three process-global variables, so two engines in one
 * process shared them and a fault in one switched off the fast path of the other):
 *   g2v_ctx_create()              a context with the defaults above (host memory; NULL on allocation failure)
 *   g2v_ctx_destroy(ctx)          (unbinds it from the calling thread if bound there; never destroy a context bound elsewhere)
 *   g2v_ctx_bind(ctx)             binds it to the CALLING THREAD and returns the previous binding; NULL = the process's default
 *                                 context.  Every entry point of this library reads the switches of the calling thread's bound
 *                                 context at call time -- launches already captured in a hipGraph keep what they were captured with.
 *   g2v_ctx_set_option(ctx, option, value) / g2v_ctx_get_option(ctx, option)
 *                                 ctx = NULL: the calling thread's bound context (the default context if none).  set returns the
 *                                 previous value; both return G2V_ERR_ARG (< 0) for an unknown option.
 * g2v_linear_set_smallm_rows, g2v_dec_rollout_set_persistent and g2v_gru_seq_set_cluster are the same calls with ctx = NULL.
 * A thread that never binds a context behaves as before (one set of switches per process, in the default context).
 * What stays process-wide: one device-side error latch, g2v_dec_rollout_persist_fault (below) -- a fault of the DEVICE, not an
 * option.  (The "already clear" notes of g2v_cluster_exchange_preclear belong to the bound context too.) */
typedef struct g2v_ctx g2v_ctx;
#define G2V_OPT_PERSISTENT 1
#define G2V_OPT_GRU_CLUSTER 2
#define G2V_OPT_SMALLM_ROWS 3
#define G2V_OPT_GRU_RESIDENT_ROWS 4
#define G2V_OPT_GRU_RESIDENT_BWD 5
g2v_ctx* g2v_ctx_create(void);
void g2v_ctx_destroy(g2v_ctx* ctx);
g2v_ctx* g2v_ctx_bind(g2v_ctx* ctx);
int g2v_ctx_set_option(g2v_ctx* ctx, int option, int value);
int g2v_ctx_get_option(const g2v_ctx* ctx, int option);
int g2v_linear_fwd(const float* x, int64_t ldx, int rows_inner, int64_t stride_outer, int64_t stride_inner,
                   const uint8_t* x_keep, float x_scale,
                   const float* w, const float* bias, float* y, int64_t ldy,
                   int M, int K, int N, int act, g2v_stream_t stream);

/* Two linear layers in a row composed into one, for two second layers p = 0, 1 on the same first layer:
 *     wc_p = w_p w_in ([G][D]),  bc_p = w_p b_in + b_p ([G])     (w_p: [G][H], w_in: [H][D])
 * so that (x w_in^T + b_in) w_p^T + b_p = x wc_p^T + bc_p: EncoderRNN.in_layer :93 straight into the bidirectional nn.GRU's input
 * projections :94 at the shipped dims (pose dim < hidden_size): D / H of the arithmetic, equal to the two-layer form to rounding.
 * The caller then never forms the first layer's output (its gradients: g2v_linear_bwd_weight_fold2 / _chain2). */
int g2v_linear_compose2(const float* w0, const float* b0, const float* w1, const float* b1, const float* w_in, const float* b_in,
                        float* wc0, float* bc0, float* wc1, float* bc1, int G, int H, int D, g2v_stream_t stream);

/* y_a = act(x w_a^T + bias_a), y_b = act(x w_b^T + bias_b) (plain rows, no mask; y_a / y_b share ldy): the input projections of
 * the two directions of a bidirectional nn.GRU layer (ref Autoencoder_VQVAE_model.py:94: weight_ih_l0 / weight_ih_l0_reverse on
 * the same input) in ONE launch where that pays (small row counts); results bitwise those of two g2v_linear_fwd calls. */
int g2v_linear_fwd_pair(const float* x, int64_t ldx, const float* w_a, const float* bias_a, float* y_a, const float* w_b,
                        const float* bias_b, float* y_b, int64_t ldy, int M, int K, int N, int act, g2v_stream_t stream);

/* y_a = x w_a^T + bias_a (M x N_a, row stride ldya), y_b = x w_b^T + bias_b (M x N_b, ldyb): two g2v_linear_fwd calls on the same
 * rows -- a decode step of Part d with attention multiplies the new top state twice, for the logits (out :389-391) and for the
 * next step's attention query (attn :173-185).  One launch at small row counts (bitwise the two calls), two otherwise. */
int g2v_linear_fwd_dual(const float* x, int64_t ldx, const float* w_a, const float* bias_a, float* y_a, int64_t ldya, int N_a,
                        const float* w_b, const float* bias_b, float* y_b, int64_t ldyb, int N_b, int M, int K,
                        g2v_stream_t stream);

/* dx[m, k] (+)= sum_n dy[m, n] * w[n, k]      (w is the forward weight, [N][K] row-major) */
int g2v_linear_bwd_data(const float* dy, int64_t lddy, const float* w, float* dx, int64_t lddx,
                        int M, int K, int N, int accumulate, g2v_stream_t stream);

/* dw[n, k] (+)= sum_m dy[m, n] * xin[m, k];  db[n] (+)= sum_m dy[m, n]  (db may be NULL).
 * xin addressing / keep-mask exactly as in g2v_linear_fwd.  Deterministic (split-M slabs + ordered
 * reduction, no float atomics).  workspace >= g2v_linear_bwd_weight_workspace(M,K,N) bytes.
 * `accumulate` is a flag word: bit 0 = add into dw / db; bit 1 (G2V_WGRAD_BF16X3) = allow the products to run on
 * the bf16 matrix pipe as a 3-term split (x = hi + lo; hi*hi + hi*lo + lo*hi, fp32 accumulation, ~1.5e-5 relative per
 * product instead of fp32's 6e-8; db stays exact fp32).  Honoured by the wave-autonomous kernels (large M), ignored
 * elsewhere.  Default (bit clear) is exact fp32 MFMA. */
#define G2V_WGRAD_ACCUMULATE 1
#define G2V_WGRAD_BF16X3 2

/* Up to 4 weight-gradient problems of ONE shape (same M, K, N, row strides; plain row addressing, no keep mask) in one call:
 * the large-M kernels run them in one launch (their workgroups fill each other's ramp-up / tail) followed by one slab
 * reduction.  workspace >= nprob * g2v_linear_bwd_weight_workspace(M,K,N).  flags as above.  db may be NULL per item. */
typedef struct {
  const float* dy;   /* (M,N), row stride lddy */
  const float* x;    /* (M,K), row stride ldx  */
  float* dw;         /* (N,K) */
  float* db;         /* (N) or NULL */
} g2v_wgrad_item;
int g2v_linear_bwd_weight_batch(const g2v_wgrad_item* items, int nprob, int64_t lddy, int64_t ldx, int M, int K, int N,
                                int flags, void* workspace, size_t workspace_bytes, g2v_stream_t stream);
/* ... with xin row-mapped as in g2v_linear_fwd (rows_inner / stride_outer / stride_inner: the (B,T,D) network input read in
 * (T,B) row order), the same map for every item. */
int g2v_linear_bwd_weight_batch_mapped(const g2v_wgrad_item* items, int nprob, int64_t lddy, int64_t ldx, int rows_inner,
                                       int64_t stride_outer, int64_t stride_inner, int M, int K, int N, int flags,
                                       void* workspace, size_t workspace_bytes, g2v_stream_t stream);

/* Deferred slab reductions (round 6).  Every large-M weight-gradient call above is [products into split-M slabs] + [one ordered
 * reduction of the slabs into dw / db]; a chain of such calls on one stream makes every product wait for the reduction in front
 * of it.  g2v_linear_bwd_weight_deferred is the union of the calls above -- nprob = 1: one product; rows_inner > 0: xin
 * row-mapped; dy_b != NULL (nprob = 1): (dy + dy_b)^T xin as g2v_linear_bwd_weight_sum2 -- WITHOUT the reduction: it fills the
 * caller-owned `pending` record, and g2v_linear_bwd_weight_reduce runs the reductions of up to G2V_WGRAD_PENDING_MAX records in
 * ONE launch, wherever the caller places it.  Until then the slabs live in `workspace`: one workspace per pending call (same size
 * rule as the immediate calls).  Results are bitwise those of the immediate calls.  A shape whose path has no slab reduction
 * (small row counts) or needs dw finished at once (ragged row counts) is completed inside the call and leaves `pending` empty
 * (nprob = 0), which g2v_linear_bwd_weight_reduce skips.  No hidden state: the record is plain data owned by the caller. */
#define G2V_WGRAD_PENDING_MAX 8
typedef struct {
  const float* slab_w[4]; float* out_w[4];   /* per problem: split-M slabs of dw (nsplit x n floats) and dw itself */
  const float* slab_b[4]; float* out_b[4];   /* ... of db (nsplit x nb floats), NULL where the item had no db */
  int64_t n, nb;                             /* N * K and N */
  int nsplit, nprob, accumulate, reserved;
} g2v_wgrad_pending;
int g2v_linear_bwd_weight_deferred(const g2v_wgrad_item* items, int nprob, int64_t lddy, int64_t ldx, int rows_inner,
                                   int64_t stride_outer, int64_t stride_inner, const float* dy_b, int M, int K, int N, int flags,
                                   void* workspace, size_t workspace_bytes, g2v_wgrad_pending* pending, g2v_stream_t stream);
int g2v_linear_bwd_weight_reduce(const g2v_wgrad_pending* pending, int count, g2v_stream_t stream);
size_t g2v_linear_bwd_weight_workspace(int M, int K, int N);
/* The weight gradient of a layer y = x W_in^T + b_in (W_in: [H][D]) that feeds TWO layers g_p = y W_p^T (W_p: [G][H]; the two
 * directions' input projections of the bidirectional encoder GRU, ref Autoencoder_VQVAE_model.py:447-464 + EncoderRNN.in_layer
 * :93), from the weight-gradient-shaped products p_p = dg_p^T x ([G][D]) and c_p = column sums of dg_p ([G]) -- both are what
 * g2v_linear_bwd_weight(_batch)(dy = dg_p, x) returns as dw / db:
 *     dw[h][d] (+)= sum_g w0[g][h] p0[g][d] + sum_g w1[g][h] p1[g][d],    db[h] (+)= sum_g w0[g][h] c0[g] + sum_g w1[g][h] c1[g]
 * = autograd's dW_in = (dg_0 W_0 + dg_1 W_1)^T x re-associated (equal to summation order): the (M x H) gradient of y, which
 * nothing else reads when x is the network's input, is never formed -- D / H of its arithmetic.  Deterministic. */
int g2v_linear_bwd_weight_fold2(const float* w0, const float* w1, const float* p0, const float* p1, const float* c0,
                                const float* c1, float* dw, float* db, int G, int H, int D, int accumulate, g2v_stream_t stream);
/* ... and the weight gradients of those two layers themselves, dW_p = dg_p^T y with y = x W_in^T + b_in, from the same p_p, c_p:
 *     dw_p[g][h] (+)= sum_d p_p[g][d] w_in[h][d] + c_p[g] b_in[h]        (w_in: [H][D], b_in: [H], dw_p: [G][H])
 * = autograd's dW_ih = dgi^T in_layer(x) re-associated (ref EncoderRNN: in_layer :93 straight into the GRU :94); db_p = c_p.  A
 * (G x D)(D x H) product instead of (G x M)(M x H).  Deterministic. */
int g2v_linear_bwd_weight_chain2(const float* p0, const float* p1, const float* c0, const float* c1, const float* w_in,
                                 const float* b_in, float* dw0, float* dw1, int G, int H, int D, int accumulate, g2v_stream_t stream);
/* Both of the above (overwrite form) in ONE launch -- they read the same p, c and nothing of each other; bitwise their results. */
int g2v_linear_bwd_weight_fold_chain2(const float* w0, const float* w1, const float* p0, const float* p1, const float* c0,
                                      const float* c1, const float* w_in, const float* b_in, float* dw_in, float* db_in,
                                      float* dw0, float* dw1, int G, int H, int D, g2v_stream_t stream);
/* dw = (dy_a + dy_b)^T x (+ db = column sums of dy_a + dy_b): the addends are summed as the operand fragments are used --
 * the arithmetic of "add, then g2v_linear_bwd_weight" without the add pass.  The encoder's input layer uses it: its dx
 * arrives as one array per GRU direction (ref Autoencoder_VQVAE_model.py:447-464, the bidirectional nn.GRU's input
 * gradient).  Served for dW shapes of 4 x 9 tiles (64 x 135) with M % 16 == 0 above the small-M threshold:
 * g2v_linear_bwd_weight_sum2_ok() says so, G2V_ERR_UNSUPPORTED otherwise.  Row addressing of x and workspace as for
 * g2v_linear_bwd_weight. */
int g2v_linear_bwd_weight_sum2_ok(int M, int K, int N);
int g2v_linear_bwd_weight_sum2(const float* dy_a, const float* dy_b, int64_t lddy, const float* x, int64_t ldx, int rows_inner,
                               int64_t stride_outer, int64_t stride_inner, float* dw, float* db, int M, int K, int N,
                               int accumulate, void* workspace, size_t workspace_bytes, g2v_stream_t stream);
int g2v_linear_bwd_weight(const float* dy, int64_t lddy,
                          const float* x, int64_t ldx, int rows_inner, int64_t stride_outer, int64_t stride_inner,
                          const uint8_t* x_keep, float x_scale,
                          float* dw, float* db, int M, int K, int N, int accumulate,
                          void* workspace, size_t workspace_bytes, g2v_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Vector quantiser.  Replaces VQ_Payam_EMA.forward / its autograd
 * (model/Autoencoder_VQVAE_model.py:1217-1296), also VQ_Payam (:1114-1173) and the
 * code-assignment call sites (data_loader/lmdb_data_loader.py:1274-1281, Clustering.py:151-157).
 *
 * g2v_vq_assign_fwd   K1+K2+K5:  d = ||x||^2 + ||W||^2 - 2 x W^T  (x = flat rows, fp32 MFMA),
 *                     idx = argmin_k d (lowest index on ties), q = W[idx],
 *                     quantized = z + (q - z), sse_partial[block] = sum (q - z)^2.
 *   flat      (N,E) rows the distances are computed from (pre_linear(z) for the EMA variant, :1230)
 *   z         (N,E) raw rows used for the loss and the straight-through output (:1285,1292)
 *   codebook  (K,E);  code_sqnorm (K) = sum_e W^2  (from g2v_vq_code_sqnorm)
 *   idx (N) int64 out; quantized (N,E) out (may be NULL); dist_min (N) out (may be NULL);
 *   sse_partial (g2v_vq_assign_blocks(N)) out (may be NULL when quantized is NULL).
 * ------------------------------------------------------------------------------------------ */
int g2v_vq_assign_blocks(int N);
int g2v_vq_code_sqnorm(const float* codebook, float* code_sqnorm, int K, int E, g2v_stream_t stream);
int g2v_vq_assign_fwd(const float* flat, const float* z, const float* codebook, const float* code_sqnorm,
                      int64_t* idx, float* quantized, float* dist_min, float* sse_partial,
                      int N, int E, int K, g2v_stream_t stream);

/* pre_linear + assign in ONE launch (E == 128, K % 128 == 0; G2V_ERR_UNSUPPORTED otherwise -> g2v_linear_fwd +
 * g2v_vq_assign_fwd): flat = z w_pre^T + b_pre (:1230) is written to flat_out (the code statistics read it), the
 * distances / argmin use it from LDS, quantized / sse_partial use the raw z as in g2v_vq_assign_fwd. */
/* Bulk code assignment (row f-2: latents of a whole corpus -> code indices, pipeline.chunks_to_codes): idx only, N large.
 * The -2 x W^T contraction runs on the bf16 matrix pipe as a 3-term split (xh.wh + xh.wl + xl.wh, fp32 accumulate); a row
 * whose best and second-best approximate distances are closer than 2^-11 |x| max|w| (more than twice the split's error
 * bound) is assigned again by the exact fp32 kernel of g2v_vq_assign_fwd, so the result is the fp32 argmin
 * (Autoencoder_VQVAE_model.py:1234-1259) at ~5x fewer matrix cycles.  E == 128, K % 128 == 0.  undecided (device int, may be
 * NULL): number of rows that took the exact path. */
size_t g2v_vq_assign_bulk_workspace(int N, int E, int K);
int g2v_vq_assign_bulk(const float* flat, const float* codebook, const float* code_sqnorm, int64_t* idx, int N, int E, int K,
                       void* workspace, size_t workspace_bytes, int* undecided, g2v_stream_t stream);
int g2v_vq_fused_assign_fwd(const float* z, const float* w_pre, const float* b_pre, const float* codebook,
                            const float* code_sqnorm, float* flat_out, int64_t* idx, float* quantized,
                            float* sse_partial, int N, int E, int K, g2v_stream_t stream);
/* The same launch reading the codebook's MFMA fragments from a fragment-major image (coalesced 1 KB runs instead of 16
 * half-used cache lines per wave-level load; 14.6 -> 12.8 us at N = 4096, K = 512).  g2v_vq_pack_codebook writes the image
 * ([K/16][E/16][64][4] floats, K * E in all) and has to be called whenever the codebook changed (after g2v_vq_ema_update);
 * the row-major codebook is still read for the gather of the chosen codes.  Results are bitwise those of
 * g2v_vq_fused_assign_fwd. */
int g2v_vq_pack_codebook(const float* codebook, float* codebook_frag, int K, int E, g2v_stream_t stream);
/* Round 5: g2v_vq_assign_fwd for ANY E % 16 == 0, K % 16 == 0 on that image (K * E floats) -- the reference's own quantiser
 * shapes, E = hidden_size * n_layers = 400 with K = 512 (config/VQ-VAE.yml) or 400 (VQ-VAE_GENEA.yml), which g2v_vq_assign_fwd
 * serves at 0.22 of the fp32 matrix peak.  Same outputs bit for bit (idx, dist_min, quantized, the SSE partials per 16 rows);
 * eight waves per workgroup, 1 / 2 / 4 row tiles per workgroup by N (bulk assignment: every fragment feeds all of them). */
int g2v_vq_assign_packed_ok(int N, int E, int K);
int g2v_vq_assign_packed_fwd(const float* flat, const float* z, const float* codebook, const float* codebook_frag,
                             const float* code_sqnorm, int64_t* idx, float* quantized, float* dist_min, float* sse_partial,
                             int N, int E, int K, g2v_stream_t stream);
int g2v_vq_fused_assign_packed_fwd(const float* z, const float* w_pre, const float* b_pre, const float* codebook,
                                   const float* codebook_frag, const float* code_sqnorm, float* flat_out, int64_t* idx,
                                   float* quantized, float* sse_partial, int N, int E, int K, g2v_stream_t stream);
/* The north-star form (round 3): pre_linear in fp32 MFMA and, BESIDE it in the same launch, the -2 x W^T contraction SCREENED on
 * the bf16 matrix pipe in z-space: flat.w_k = z.u_k + b.w_k with u_k = w_pre^T w_k, so e_k = (|w_k|^2 - 2 b.w_k) - 2 (zh + zl).bf16(u_k)
 * (fp32 accumulate, 8 v_mfma_f32_16x16x32_bf16 per 16 x 16 x 128 tile instead of 32 fp32 MFMAs) needs no projected row.  With the
 * per-code radius r_k = 2^-7 (1 + 2^-5) |z||u_k| + 2^-13 |flat|^ |w_k| + 2^-20 (|flat|^^2 + |w_k|^2)  (|flat|^ = |w_pre|_F |z| + |b|; an
 * upper bound of |e_k - (d_k - |flat|^2)| for the fp32 kernel's d_k, derivation in vq.hip) every code with e_k - r_k <= min_j (e_j + r_j)
 * is re-evaluated IN THE SAME LAUNCH with the exact fp32 chain of g2v_vq_fused_assign_fwd on the projected rows, the row's code
 * being the torch.argmin (:1259) of those exact distances: flat_out, idx and quantized are bitwise those of
 * g2v_vq_fused_assign_fwd (sse_partial: the same sum in another order).  A 16-row tile with a non-finite screening value or more
 * than 128 candidates takes the exact fp32 sweep over all K codes instead.  E == 128, K in {128, 256, 384, 512}
 * (g2v_vq_fused_assign_bx_ok).
 *   w_pre_frag  g2v_vq_pack_codebook(w_pre, ., E, E): pre_linear's weight as fp32 MFMA fragments (E * E floats)
 *   image       g2v_vq_bx_pack(codebook, code_sqnorm, w_pre, b_pre, ., K, E): bf16 MFMA fragments of U = W w_pre, s'_k, the radius
 *               coefficients (g2v_vq_bx_image_bytes(K, E) bytes); rewrite it whenever the codebook / code_sqnorm / pre_linear changed
 *   diag        device int[4] or NULL: [0] += tiles that took the exact sweep, [1] += (row, candidate) pairs re-evaluated
 *   flags       G2V_VQ_BX_*: explicit per-call switches (no process-global state) */
#define G2V_VQ_BX_EXACT 1      /* every tile takes the exact fp32 sweep: the A/B reference of the screened path */
size_t g2v_vq_bx_image_bytes(int K, int E);
int g2v_vq_bx_pack(const float* codebook, const float* code_sqnorm, const float* w_pre, const float* b_pre, void* image, int K, int E,
                   g2v_stream_t stream);
int g2v_vq_fused_assign_bx_ok(int N, int E, int K);
int g2v_vq_fused_assign_bx_fwd(const float* z, const float* w_pre_frag, const float* b_pre, const float* codebook,
                               const void* image, const float* code_sqnorm, float* flat_out, int64_t* idx, float* quantized,
                               float* sse_partial, int* diag, int N, int E, int K, int flags, g2v_stream_t stream);
/* K3: cnt[k] = #{i: idx[i]=k};  dw[k,:] = sum_{i: idx[i]=k} flat[i,:]   (:1265,1275).
 * Deterministic one-hot^T x flat MFMA contraction (the one-hot is generated on the fly from idx).
 * stats layout: [cnt (K) | dw (K*E)] contiguous fp32, so it can be all-reduced as one buffer. */
size_t g2v_vq_stats_workspace(int N, int E, int K);
int g2v_vq_stats(const int64_t* idx, const float* flat, float* stats /* K + K*E */, int N, int E, int K,
                 void* workspace, size_t workspace_bytes, g2v_stream_t stream);

/* K4 + K5 scalars (:1263-1282, :1285-1294).  In place on ema_cluster_size (K), ema_w (K,E), codebook (K,E):
 *   cs = cs*decay + (1-decay)*cnt; n = sum cs; cs = (cs+eps)/(n+K*eps)*n;
 *   ema_w = ema_w*decay + (1-decay)*dw; codebook = ema_w / cs[:,None]; code_sqnorm refreshed.
 * update = 0 skips the EMA part (eval mode, :1262) and only produces the scalars.
 * scalars out (2 floats): [0] loss = beta * sum(sse_partial) / (N_loss*E), [1] perplexity = exp(-sum p log(p+1e-10)),
 * p = cnt / N_cnt.  N_cnt is the number of rows behind `stats` (global batch under data parallelism). */
int g2v_vq_ema_update(const float* stats, const float* sse_partial, int n_sse_partial,
                      float* ema_cluster_size, float* ema_w, float* codebook, float* code_sqnorm,
                      float* scalars, int N_loss, int N_cnt, int E, int K,
                      float beta, float decay, float eps, int update, g2v_stream_t stream);

/* K5': gz = g_quantized + g_loss * 2*beta/(N*E) * (z - W[idx])   (autograd of :1285-1292).
 * g_loss is a device scalar (d total / d loss_vq), g_quantized may be NULL.
 * idx == NULL: `codebook` is instead the dense (N,E) forward output `quantized` (= z + (W[idx]-z)), so the
 * backward does not depend on a codebook that the EMA update has already moved. */
int g2v_vq_bwd(const float* g_quantized, const float* g_loss, const float* z, const float* codebook,
               const int64_t* idx, float* gz, int N, int E, float beta, g2v_stream_t stream);

/* Codebook gradient of the NON-EMA quantiser VQ_Payam (:1114-1173, loss = q_latent + beta*e_latent):
 *   g_codebook[k,:] = g_loss * 2/(N*E) * (cnt[k]*W[k,:] - sum_{i: idx[i]=k} z[i,:])
 * with `stats` = g2v_vq_stats(idx, z).  The gradient wrt z is g2v_vq_bwd (e_latent term only, :1158-1159). */
int g2v_vq_codebook_grad(const float* stats, const float* codebook, const float* g_loss, float* g_codebook,
                         int N, int E, int K, g2v_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Full-sequence GRU direction (K8, K12).  Replaces one direction of one layer of
 * nn.GRU in EncoderRNN (model/Autoencoder_VQVAE_model.py:94, model/text2embedding_model.py:131).
 *   gi (T,B,3H) = x W_ih^T + b_ih (from g2v_linear_fwd), gate order r,z,n (PyTorch)
 *   r = s(gi_r + W_hr h + b_hr); z = s(gi_z + W_hz h + b_hz); n = tanh(gi_n + r*(W_hn h + b_hn)); h' = (1-z) n + z h
 *   lengths (B) int32 or NULL: pack_padded_sequence semantics (rows freeze after their last valid step,
 *   padded outputs are zero, the reverse direction starts at each row's own last step).
 *   hs row (t,b) is written at hs + (t*B + b)*hs_ld  (hs_ld >= H lets both directions share a (T,B,2H) buffer).
 *   gates (T,B,4H) = r,z,n,W_hn h + b_hn  saved for the backward (NULL = inference).
 * ------------------------------------------------------------------------------------------ */
typedef struct {          /* one direction of one layer, forward */
  const float* gi;        /* (T,B,3H) input projections incl. b_ih          */
  const float* w_hh;      /* (3H,H)                                         */
  const float* b_hh;      /* (3H)                                           */
  const float* h0;        /* (B,H) or NULL = zeros                          */
  float* hs;              /* out: hidden states, row (t,b) at hs + (t*B+b)*hs_ld */
  float* h_n;             /* out: (B,H) final state (may be NULL)           */
  float* gates;           /* out: (T,B,4H) saved for backward (NULL = inference) */
  int reverse;            /* 0: t = 0..T-1, 1: t = T-1..0                   */
  /* Fused input projection (gi == NULL): gi_t = x_t W_ih^T + b_ih is computed inside the recurrent kernel, one step
   * ahead of its use (the gi array, its GEMM and its HBM round trip disappear).  Supported for in_dim == H == 64;
   * otherwise G2V_ERR_UNSUPPORTED and the caller computes gi with g2v_linear_fwd. */
  const float* x;         /* (T,B,in_dim) layer input                       */
  const float* w_ih;      /* (3H,in_dim)                                    */
  const float* b_ih;      /* (3H)                                           */
  int in_dim;
  /* Packed input projections (round 5; NULL = the (T,B,3H) layout): a HOST array of T row offsets.  gi then holds only the
   * positions inside their sequences, step-major: row (t,b) at gi + (gi_row_off[t] + b) * 3H for b < n_t = #{lengths > t}
   * (lengths sorted descending, as pack_padded_sequence(enforce_sorted) requires; gi_row_off[t] = n_0 + ... + n_{t-1}) -- the
   * dense product that makes gi runs over sum(lengths) rows instead of T x B.  T <= 64, the generic kernels with H % 4 == 0
   * (g2v_gru_seq_packed_ok); every direction of the call must carry the same offsets.  hs / gates keep the (T,B,.) layout. */
  const int32_t* gi_row_off;
  /* Gathered input projections (round 6; NULL = off): gi is a TABLE (V, 3H), and row r of the layout above -- r = t * B + b, or
   * gi_row_off[t] + b when packed -- is gi + gi_gather[r] * 3H.  The encoder the reference feeds straight from nn.Embedding
   * (model/text2embedding_model.py:126-131): the projected table G = E W_ih^T + b_ih is gathered inside the recurrent kernel
   * instead of being materialised per position.  Served by the W_hh-resident forward only: g2v_gru_seq_gather_ok(T, B, H, ndir);
   * otherwise G2V_ERR_UNSUPPORTED.  Entries of positions outside their sequences are never read. */
  const int64_t* gi_gather;
} g2v_gru_dir;
int g2v_gru_seq_packed_ok(int T, int B, int H);
int g2v_gru_seq_gather_ok(int T, int B, int H, int ndir);
/* Round 5.  At small batch (B <= 1024, H % 4 == 0, H <= 256, H != 64) g2v_gru_seq_fwd / _bwd run one launch per time step over
 * (16 rows x 16 hidden units x direction) workgroups.  While that grid fits the device with one workgroup per CU (B = 128 at
 * H = 200, both directions: 208 workgroups) the same workgroups instead stay resident for ALL steps of ONE launch, their W_hh rows
 * in registers, and hand each other the state rows (forward) / the hidden-side gate gradients (backward) through tagged 8-byte
 * granules in the call's workspace (csrc/gru.hip: gru_cluster_*_kernel).  Forward results are bitwise those of the per-step
 * launches, backward results equal to summation order.  Like the persistent rollouts these kernels need their workgroups
 * co-resident; a bounded wait that runs out latches g2v_dec_rollout_persist_fault.  This switch (default 1; 0 = one launch per
 * step) exists for parity tests, A/B measurements and the fall-back after a latched fault.  Returns the previous setting. */
int g2v_gru_seq_set_cluster(int enable);
/* 1: this shape runs as the cluster kernels under the current setting (then g2v_gru_dir_bwd.hn_z .. hn_coef, the fused quantiser
 * backward below, are honoured at H != 64 too) */
int g2v_gru_seq_cluster_ok(int T, int B, int H, int ndir);
/* A persistent cluster launch (g2v_gru_seq_fwd / _bwd, g2v_dec_rollout_fwd / _bwd at the shapes their *_cluster_ok queries
 * name) clears its exchange records in front of the kernel: a memset of 1-11 MB, 5-12 us on the caller's chain.  A caller that hands
 * such a call a workspace NOTHING ELSE WRITES can have the records cleared ahead of time, e.g. on a side stream at the start of the
 * step: this call issues the memset on `stream` now and notes it; the next cluster launch of that kind over this workspace takes
 * the note (one shot) and starts with its kernel.  The caller orders `stream` in front of that launch and keeps the workspace
 * untouched in between.  kind: 0 / 1 = g2v_gru_seq_fwd / _bwd (T, B, H, ndir; D ignored), 2 / 3 = g2v_dec_rollout_fwd / _bwd
 * (T, B, D, H; ndir ignored).  Shapes that do not run as a cluster: G2V_OK, nothing happens.  A launch over the workspace that does
 * not run as a cluster, and either of the two switches above, forget the note. */
int g2v_cluster_exchange_preclear(int kind, int T, int B, int D, int H, int ndir, void* workspace, size_t workspace_bytes,
                                  g2v_stream_t stream);
/* Void the "already clear" notes of the calling thread's context that lie inside [workspace, workspace + workspace_bytes)
 * (workspace = NULL: all of them): for an owner that abandons a step between g2v_cluster_exchange_preclear and the launch that
 * would have taken the note, or frees / re-purposes the workspace.  The notes belong to the bound context (g2v_ctx above). */
int g2v_cluster_exchange_preclear_drop(const void* workspace, size_t workspace_bytes);

/* Up to 2 directions per call run in ONE launch (the two directions of a bidirectional layer are independent). */
size_t g2v_gru_seq_fwd_workspace(int ndir, int H);   /* W_hh in MFMA fragment order */
int g2v_gru_seq_fwd(const g2v_gru_dir* dirs, int ndir, const int32_t* lengths, int64_t hs_ld,
                    int T, int B, int H, void* workspace, size_t workspace_bytes, g2v_stream_t stream);

/* BPTT of the above.  d_hs (row stride d_hs_ld) / d_hn may be NULL.  Produces dgi (T,B,3H) (grad wrt gi:
 * feeds W_ih, b_ih, x grads through g2v_linear_bwd_*), dgh (T,B,3H) (feeds W_hh, b_hh) and dh0 (B,H) (may be NULL). */
typedef struct {
  const float* d_hs; const float* d_hn;       /* incoming gradients (either may be NULL)            */
  const float* hs; const float* h0; const float* gates; const float* w_hh;   /* forward tensors */
  float* dgi; float* dgh; float* dh0;         /* outputs (dh0 may be NULL)                          */
  int reverse;
  /* Fused input gradient (dx != NULL): dx (T,B,in_dim) = dgi W_ih is produced by the recurrent kernel itself (a second,
   * independent MFMA chain next to the dh chain).  in_dim == H == 64 only; otherwise G2V_ERR_UNSUPPORTED and the caller
   * uses g2v_linear_bwd_data on dgi.  Each direction writes its own dx (the caller sums the two directions). */
  const float* w_ih;      /* (3H,in_dim) */
  float* dx;              /* out: (T,B,in_dim), overwritten */
  int in_dim;
  /* Fused weight gradients (dw_hh != NULL; with the fused input gradient, H == in_dim == 64, every direction alike): the
   * recurrent kernel also accumulates dW_hh = sum dgh^T h_prev, dW_ih = sum dgi^T x and the two bias gradients (overwritten);
   * dgi / dgh are then neither written nor needed (may be NULL).  x: the layer input (T,B,in_dim); wslab: scratch of
   * g2v_gru_seq_bwd_wslab_bytes(B, H) bytes PER DIRECTION. */
  const float* x;
  float* dw_hh; float* db_hh; float* dw_ih; float* db_ih;
  float* wslab;
  /* Optional (hn_z != NULL; the H == 64 fast kernels and the small-batch cluster kernels -- g2v_gru_seq_cluster_ok --,
   * G2V_ERR_UNSUPPORTED elsewhere): the quantiser's backward (K5', g2v_vq_bwd
   * with a dense quantised tensor) applied where d_hn is read,
   *   d_hn_effective[b,h] = d_hn[b,h] + hn_gloss[0] * hn_coef * (hn_z[b,h] - hn_q[b,h]),
   * i.e. the straight-through gradient plus the commitment term (model/Autoencoder_VQVAE_model.py:1285-1292) when the final
   * state of this direction IS the quantiser's input: hn_z / hn_q are the (B,H) slices of the encoder state and of its
   * quantised value, hn_coef = 2 beta / (N E).  Same arithmetic as g2v_vq_bwd; saves its launch on the critical chain. */
  const float* hn_z; const float* hn_q; const float* hn_gloss;
  float hn_coef;
  /* Packed dgi (NULL = (T,B,3H)): as g2v_gru_dir.gi_row_off -- dgi row (t,b) at dgi + (dgi_row_off[t] + b) * 3H, only positions
   * inside their sequences are written (sum(lengths) rows); dgh keeps the (T,B,3H) layout with zero rows at padded positions. */
  const int32_t* dgi_row_off;
} g2v_gru_dir_bwd;
size_t g2v_gru_seq_bwd_wslab_bytes(int B, int H);
size_t g2v_gru_seq_bwd_workspace(int ndir, int H);   /* room for W_hh^T (fragment order) */
int g2v_gru_seq_bwd(const g2v_gru_dir_bwd* dirs, int ndir, const int32_t* lengths, int64_t d_hs_ld, int64_t hs_ld,
                    int T, int B, int H, void* workspace, size_t workspace_bytes, g2v_stream_t stream);

/* ONE GRU cell step at small batch with the input projection included (nn.GRU cell, the code decoder of Part d,
 * model/text2embedding_model.py:372-380, where every call is one time step):
 *   gi = (x * x_keep * x_scale) W_ih^T + b_ih,  gh = h_prev W_hh^T + b_hh,  r, z, n as in g2v_gru_seq_fwd,  h_new = (1-z) n + z h_prev
 *   gates (B,4H) = r, z, n, gh_n saved for the backward (NULL = inference).  x_keep (uint8 (B,in_dim), may be NULL): the
 *   inter-layer dropout of nn.GRU on this layer's input.  One launch instead of g2v_linear_fwd + g2v_gru_seq_fwd(T = 1), same
 *   contraction order (gates bit-identical, h_new within an ulp).  in_dim, H multiples of 4 and <= 256.
 * g2v_gru_cell_bwd: d_h = d_h_a + d_h_b (gradient arriving at h_new from above / from the next step; either may be NULL) ->
 *   dgi, dgh (B,3H), d_hprev (B,H) = d_h z + dgh W_hh, dx (B,in_dim) = (dgi W_ih) * x_keep * x_scale (dx may be NULL).
 *   Two launches (gate gradients; both products) instead of four. */
int g2v_gru_cell_fwd(const float* x, int in_dim, const uint8_t* x_keep, float x_scale, const float* h_prev, const float* w_ih,
                     const float* w_hh, const float* b_ih, const float* b_hh, float* h_new, float* gates, int B, int H,
                     g2v_stream_t stream);
int g2v_gru_cell_bwd(const float* d_h_a, const float* d_h_b, const float* gates, const float* h_prev, const float* w_ih,
                     const float* w_hh, const uint8_t* x_keep, float x_scale, float* dgi, float* dgh, float* d_hprev, float* dx,
                     int in_dim, int B, int H, g2v_stream_t stream);

/* Ahead-of-time weight packs (the fused train step runs them as a parallel branch while the step's first kernels execute):
 * g2v_gru_seq_prepare launches the fragment packs of ONE g2v_gru_seq_fwd + ONE g2v_gru_seq_bwd call of the H == 64 fast
 * kernels into their two workspaces (w_hh / w_ih: one pointer per direction; fused != 0: the calls fuse the input projection /
 * input gradient, so w_ih is packed too; either workspace may be NULL); the *_prepared entry points then skip their pack launch.
 * Nothing else may touch the workspaces in between.  Other hidden sizes: prepare is a no-op and *_prepared == the plain call. */
int g2v_gru_seq_prepare(const float* const* w_hh, const float* const* w_ih, int ndir, int H, int fused, void* fwd_workspace,
                        size_t fwd_bytes, void* bwd_workspace, size_t bwd_bytes, g2v_stream_t stream);
int g2v_gru_seq_fwd_prepared(const g2v_gru_dir* dirs, int ndir, const int32_t* lengths, int64_t hs_ld, int T, int B, int H,
                             void* workspace, size_t workspace_bytes, g2v_stream_t stream);
int g2v_gru_seq_bwd_prepared(const g2v_gru_dir_bwd* dirs, int ndir, const int32_t* lengths, int64_t d_hs_ld, int64_t hs_ld,
                             int T, int B, int H, void* workspace, size_t workspace_bytes, g2v_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Autoregressive pose-decoder rollout (K9).  Replaces the T-1 step loop
 * model/Autoencoder_VQVAE_model.py:1039-1054 over Generator.forward (:646-683) ->
 * BahdanauAttnDecoderRNN.forward (:499-592) with autoencoder_att == "False", n_layers == 2:
 *   xin_t = keep95_t * y_{t-1} / 0.05   (inline nn.Dropout(0.95), ALWAYS active :570; zeros if !conditioned :568)
 *   u_t   = xin_t W_pre^T + b_pre;  a_t = ReLU(BatchNorm1d(u_t))   (batch stats if training, running stats else)
 *   h0_t  = GRUcell0(a_t, h0_{t-1});  x1_t = keep_l0_t * h0_t / (1-p)  (nn.GRU inter-layer dropout, training only)
 *   h1_t  = GRUcell1(x1_t, h1_{t-1});  y_t = h1_t W_out^T + b_out;  y_0 = target frame 0;
 *   next input = target[t] if t < n_pre_poses else y_t (:1049-1052).
 * One launch per time step (the BatchNorm batch statistics are a grid-wide reduction: the seam is a
 * kernel boundary), 16 batch rows per workgroup.
 * ------------------------------------------------------------------------------------------ */
typedef struct {
  const float* w_pre;  const float* b_pre;     /* (H,D), (H)   decoder.decoder.pre_linear.0 */
  const float* bn_w;   const float* bn_b;      /* (H), (H)     decoder.decoder.pre_linear.1 */
  float* bn_running_mean; float* bn_running_var; /* (H) each, updated in place when training; both NULL while training: not updated (g2v_bn_running_update) */
  const float* w_ih0; const float* w_hh0; const float* b_ih0; const float* b_hh0; /* (3H,H),(3H,H),(3H),(3H) */
  const float* w_ih1; const float* w_hh1; const float* b_ih1; const float* b_hh1;
  const float* w_out;  const float* b_out;     /* (D,H), (D)   decoder.decoder.out_layer */
} g2v_dec_weights;

typedef struct {           /* saved-for-backward / state arrays, all caller-owned */
  float* y;                /* (T,B,D)   outputs, y[0] = target frame 0                       */
  float* xin;              /* (T-1,B,D) dropped decoder inputs xin_t (t = 1..T-1 at index t-1) */
  float* u;                /* (T-1,B,H) pre-BatchNorm activations                             */
  float* a;                /* (T-1,B,H) post BN+ReLU                                          */
  float* h0;               /* (T,B,H)   layer-0 hidden states, h0[0] = initial state          */
  float* h1;               /* (T,B,H)   layer-1 hidden states, h1[0] = initial state          */
  float* x1;               /* (T-1,B,H) dropped layer-0 outputs (NULL => aliases h0[1:], p == 0 or eval) */
  float* gates0;           /* (T-1,B,4H) r,z,n,ghn of layer 0 (NULL in inference)             */
  float* gates1;           /* (T-1,B,4H)                                                      */
  float* bn_partial;       /* (2, nblk, 2, H) ping-pong per-block sums of (u-b), (u-b)^2      */
  float* bn_stats;         /* (T-1,2,H) batch mean / biased var per step (training)           */
  /* Optional, all NULL / 0 = off: custom_loss (K10, train_eval/train_seq2seq.py:40-88) carried by the rollout pair and a
   * CHASER kernel, with the rollout's `target` as the loss target.  Where g2v_dec_rollout_fuses_loss() says 1 (training only):
   *   g2v_dec_rollout_prepare(...)                      (clears the forward's progress words with its exchange records)
   *   g2v_dec_rollout_fwd_prepared(..., s with loss_* set, ...)   on stream A: stores y_t write-through and publishes, per
   *                                                     workgroup and step, that y_t is in memory
   *   g2v_custom_loss_chase(..., same s, same workspace) on stream B, ordered behind g2v_dec_rollout_prepare and dispatched
   *                                                     BEHIND the forward (see its comment): consumes y_t as it lands,
   *                                                     writes loss_code, loss_coef, loss_partial
   *   (join A and B)
   *   g2v_dec_rollout_bwd[_prepared](..., same s, ...)  forms dLoss/dy_t from y_t, the code byte and the column coefficient in
   *                                                     its own tile load; `dy` of g2v_dec_grads is OUTPUT only; writes loss_terms
   * g2v_custom_loss_fwd_bwd is not called.  Same dy bits as g2v_custom_loss_fwd_bwd(g_scale = 1); the four loss sums are added
   * in a different order.  Removes the 62 + 6 us launch pair that sat alone between the rollouts and 150 MB of HBM traffic per
   * step at B = 4096, T = 34. */
  uint8_t* loss_code;      /* (T,B,D)   sign codes of the three |.| terms + the Dropout(0.95) flag */
  float* loss_coef;        /* (B,D)     w_var / numel / ||y[:,b,d]||_2                             */
  float* loss_partial;     /* (nblk,4)  per-workgroup loss sums                                    */
  float* loss_terms;       /* (5)       total, l1, cont, var, mse -- as g2v_custom_loss_fwd_bwd    */
  float loss_w[3];         /* w_l1, w_cont, w_var (custom_loss's loss_regularization weights)     */
} g2v_dec_saved;

int g2v_dec_rollout_blocks(int B);
/* Two interchangeable implementations sit behind g2v_dec_rollout_fwd / _bwd (same arguments, same saved arrays):
 * one launch per time step (any shape), and -- for H == 64, D == 135, B % 16 == 0, B / 16 <= 3 x the device's CU count --
 * ONE persistent launch for the whole rollout with register/LDS-resident weights and an in-kernel two-level exchange of
 * the BatchNorm partial sums (csrc/dec_persist.hpp): one 16-row tile per workgroup while there is a CU per tile, 2 or 3 tiles
 * per workgroup beyond that (no fused weight gradient / loss chaser there: the two queries below return 0).  The persistent
 * one is used whenever it applies; this switch (default 1; 0 = one launch per step; 2 / 3 = persistent with AT LEAST that many
 * tiles per workgroup, which is how the parity tests reach the multi-tile kernels at small batches) exists for A/B
 * measurements, parity tests and the fall-back after a latched fault.  Returns the previous setting.
 * Round 5: the same switch covers the CLUSTER kernels of the generic dims (H % 4 == 0, H <= 208, D <= 64 -- every shipped YAML's
 * H = 200 -- at small batch): where the per-step path runs three / four launches per step over (hidden-unit tile x row group)
 * workgroups, and that grid fits the device with one workgroup per CU (B <= 304 at H = 200), the steps t >= 1 of the forward and the
 * whole backward are ONE launch each with those workgroups resident: weight rows in registers, the rows a kernel boundary used to
 * hand over (u / h0 / h1 and the BatchNorm sums forward; dbn rows, BatchNorm-backward sums and the partial products of the
 * 3H-long contractions backward) exchanged through tagged 8-byte granules in the call's workspace (csrc/dec_rollout.hip:
 * dec_cluster_fwd_kernel / dec_cluster_bwd_kernel).  Same saved arrays, results equal to summation order, bitwise reproducible. */
int g2v_dec_rollout_set_persistent(int enable);
/* which of the two g2v_dec_rollout_fwd / _bwd take for this shape under the current setting: 0 = one launch per time step,
 * R = 1..3 = the persistent pair with R row tiles per workgroup (B % 4 == 0; a ragged last tile where B % 16 != 0) */
int g2v_dec_rollout_tiles_per_workgroup(int B, int D, int H);
/* 1: for this shape g2v_dec_rollout_fwd / _bwd run as the small-batch cluster kernels (one launch for the steps t >= 1 of the
 * forward, one for the backward; needs T >= 3 and a workspace of the queried size) under the current setting, 0: not */
int g2v_dec_rollout_cluster_ok(int B, int D, int H);
/* The persistent kernels need every workgroup of their launch resident at once; what a plain launch can check is checked
 * (B / 16 <= CU count, the occupancy query).  What it cannot see -- a CU mask, another tenant or a second persistent launch
 * interleaved on the same device -- ends in a bounded wait running out: the kernel then LATCHES a device-side fault word and
 * stops waiting (its outputs are garbage) instead of trapping.  g2v_dec_rollout_persist_fault returns the latch (0 / 1; -1 if
 * it cannot be read) and clears it when `clear` != 0.  It is SYNCHRONOUS (a one-word device-to-host copy): call it where the
 * host synchronises anyway (bench.py at the end of the timed region; train_iter reads the latch as part of its one read-back,
 * g2v_iteration_readback); on 1 discard the step, g2v_dec_rollout_set_persistent(0), and run the step again on the per-step
 * kernels.  clear < 0 is the test hook of that path: it LATCHES the value -clear, as a bounded wait running out would. */
int g2v_dec_rollout_persist_fault(int clear);
/* Data parallelism (one process per GPU): the latch is per process.  from_flag == 0: flag[0] = 1.0f if this process' latch is set,
 * else 0.0f -- in front of the SUM all-reduce, with `flag` a slot of the communication buffer; from_flag != 0: a non-zero
 * (reduced) flag latches this process too (value 3), so that EVERY rank's commit kernels skip the step a faulting rank poisoned. */
int g2v_dec_rollout_fault_flag(float* flag, int from_flag, g2v_stream_t stream);
/* 1 where the rollout pair + chaser can carry custom_loss (the loss_* fields of g2v_dec_saved): wherever the persistent path
 * applies with ONE row tile per workgroup (H == 64, D == 135, B % 16 == 0, B / 16 <= CU count, persistent setting 1), 2 <= T <= 256.  Elsewhere leave the loss_*
 * fields NULL and call g2v_custom_loss_fwd_bwd between the two rollouts (setting them anyway is refused with
 * G2V_ERR_UNSUPPORTED, never silently ignored). */
int g2v_dec_rollout_fuses_loss(int B, int D, int H, int T);
/* The chaser (csrc/dec_persist.hip, loss_chase_kernel): one light workgroup per 16 batch rows (256 threads, <= 128 registers per
 * lane, 64 B of LDS) that is CO-RESIDENT with the persistent forward rollout -- which runs one wave per SIMD and leaves a third of
 * every CU's registers unused -- polls that workgroup's progress word and consumes y_t as it lands.  `s`: the SAME struct the
 * forward of this call runs with (y, loss_code, loss_coef, loss_partial, loss_w); `fwd_workspace`: the SAME workspace (its
 * progress words).  Launch it on a second stream, ordered behind g2v_dec_rollout_prepare, and so that it is DISPATCHED BEHIND the
 * forward (the engine puts it behind the ~30 us of quantiser-statistics kernels of its side branch): the chaser waits for the
 * rollout, never the other way round, but a CU that already holds two chaser workgroups has no room for a rollout workgroup.
 * Every wait is bounded and ends in the fault latch (g2v_dec_rollout_persist_fault), as the exchange's do. */
int g2v_custom_loss_chase(const float* target /* (B,T,D) */, const g2v_dec_saved* s, const uint8_t* keep95 /* (T-1,B,D) */,
                          int T, int B, int D, int H, void* fwd_workspace, size_t fwd_workspace_bytes, g2v_stream_t stream);
/* workspace: the weights re-laid-out in MFMA fragment order (packed once per call) + the persistent kernel's exchange
 * state (zeroed by a memset node in front of its launch). */
size_t g2v_dec_rollout_fwd_workspace(int D, int H);
int g2v_dec_rollout_fwd(const float* target /* (B,T,D) row-major */, const float* h_init /* (2,B,H) */,
                        const g2v_dec_weights* w, const g2v_dec_saved* s,
                        const uint8_t* keep95 /* (T-1,B,D) */, const uint8_t* keep_l0 /* (T-1,B,H) or NULL */,
                        float p_drop, int n_pre_poses, int conditioned, int training,
                        int T, int B, int D, int H, void* workspace, size_t workspace_bytes, g2v_stream_t stream);

typedef struct {           /* gradient outputs of the rollout backward, caller-owned */
  float* dy;               /* (T,B,D) in: dLoss/dy_t (t=0 row ignored); out: total dL/dy_t incl. feedback */
  float* du;               /* (T-1,B,H) grad wrt pre-BN activations          -> W_pre, b_pre        */
  float* dbn;              /* (T-1,B,H) scratch: grad wrt BN output after ReLU                     */
  float* dgi0; float* dgh0;/* (T-1,B,3H) each                                -> W_ih0/b_ih0, W_hh0/b_hh0 */
  float* dgi1; float* dgh1;/* (T-1,B,3H)                                                             */
  float* dh_init;          /* (2,B,H) grad wrt the initial hidden state (the quantised latent)     */
  float* d_bn_w; float* d_bn_b; /* (H) each, overwritten                                           */
  float* bn_bwd_partial;   /* (2, nblk, 2, H) ping-pong per-block sums                              */
  /* Optional: GRU weight / bias gradients dW (3H,H), db (3H), indexed ih0, hh0, ih1, hh1, overwritten.  Set exactly the
   * entries named by g2v_dec_rollout_bwd_fuses_wgrad() (or none): those are accumulated INSIDE the persistent backward kernel,
   * in the shadow of its BatchNorm exchange, and the matching dgi / dgh array (dgh1 for W_hh1) is neither written nor needed. */
  float* dw_gru[4];
  float* db_gru[4];
} g2v_dec_grads;

/* workspace: transposed weights in MFMA fragment order. */
size_t g2v_dec_rollout_bwd_workspace(int D, int H);
/* Which GRU weight gradients g2v_dec_rollout_bwd can accumulate inside its persistent kernel at this batch / shape: bit m of
 * the result <-> matrix m of (W_ih0, W_hh0, W_ih1, W_hh1).  Today 8 (W_hh1: what the kernel's register budget has room for) where
 * the persistent path applies (H == 64, D == 135, B % 16 == 0, B / 16 <= CU count), else 0.
 * The caller sets dw_gru[m] / db_gru[m] for exactly those matrices (or for none) and forms the other products itself. */
int g2v_dec_rollout_bwd_fuses_wgrad(int B, int D, int H);
int g2v_dec_rollout_bwd(const g2v_dec_weights* w, const g2v_dec_saved* s, const g2v_dec_grads* g,
                        const uint8_t* keep95, const uint8_t* keep_l0, float p_drop, int n_pre_poses,
                        int conditioned, int T, int B, int D, int H,
                        void* workspace, size_t workspace_bytes, g2v_stream_t stream);

/* The same for a rollout pair: the forward and backward fragment packs AND the clearing of the two exchange regions of the
 * persistent kernels (everything of the pair that depends on the weights only), ahead of time; g2v_dec_rollout_fwd_prepared /
 * _bwd_prepared then start with their first real kernel.  Honoured for H == 64, D == 135 (the fused kernels); other shapes:
 * prepare is a no-op and *_prepared == the plain call.  bwd_workspace may be NULL (inference). */
int g2v_dec_rollout_prepare(const g2v_dec_weights* w, int D, int H, void* fwd_workspace, size_t fwd_bytes,
                            void* bwd_workspace, size_t bwd_bytes, g2v_stream_t stream);
/* Round 5: g2v_dec_rollout_prepare (both workspaces) AND g2v_gru_seq_prepare (backward workspace only; w_hh / w_ih: one pointer
 * per direction) of one fused train step as ONE launch instead of six (three packs, two memset nodes, one pack).  H == 64,
 * D == 135 only (G2V_ERR_UNSUPPORTED otherwise, nothing launched: call the two functions). */
int g2v_train_step_prepare(const g2v_dec_weights* w, int D, int H, void* dec_fwd_workspace, size_t dec_fwd_bytes,
                           void* dec_bwd_workspace, size_t dec_bwd_bytes, const float* const* gru_w_hh,
                           const float* const* gru_w_ih, int gru_ndir, int gru_fused, void* gru_bwd_workspace,
                           size_t gru_bwd_bytes, g2v_stream_t stream);
/* Deferred commit of BatchNorm1d's running statistics (momentum 0.1, unbiased variance, `steps` = T-1 updates in step order)
 * from g2v_dec_saved.bn_stats of a TRAINING rollout that was given bn_running_mean = bn_running_var = NULL: the forward rollout
 * then leaves them alone, and this call -- placed where the whole step is known to be valid, e.g. behind the backward rollout --
 * applies them unless the persistent rollouts' fault latch is set (g2v_dec_rollout_persist_fault). */
int g2v_bn_running_update(const float* bn_stats, float* running_mean, float* running_var, int steps, int H, int B,
                          g2v_stream_t stream);
int g2v_dec_rollout_fwd_prepared(const float* target, const float* h_init, const g2v_dec_weights* w, const g2v_dec_saved* s,
                                 const uint8_t* keep95, const uint8_t* keep_l0, float p_drop, int n_pre_poses,
                                 int conditioned, int training, int T, int B, int D, int H, void* workspace,
                                 size_t workspace_bytes, g2v_stream_t stream);
int g2v_dec_rollout_bwd_prepared(const g2v_dec_weights* w, const g2v_dec_saved* s, const g2v_dec_grads* g,
                                 const uint8_t* keep95, const uint8_t* keep_l0, float p_drop, int n_pre_poses,
                                 int conditioned, int T, int B, int D, int H, void* workspace, size_t workspace_bytes,
                                 g2v_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * custom_loss forward + gradient (K10), train_eval/train_seq2seq.py:40-88.
 *   loss = w_l1*mean|y-tgt| + w_cont*sum_t|y_t-y_{t-1}|/numel - w_var*sum_{b,d}||y[b,:,d]||_2/numel
 * y is time-major (T,B,D) (the rollout's buffer), target is (B,T,D).  dy (T,B,D) = g_scale * dloss/dy
 * (may be NULL).  terms out (5 floats): total, l1, cont, var, mse (= mean (y-tgt)^2, the metric of
 * evaluate_testset train_autoencoder_VQVAE.py:350-410).  partial: >= g2v_custom_loss_blocks(B,D)*4 floats.
 * ------------------------------------------------------------------------------------------ */
int g2v_custom_loss_blocks(int B, int D);
int g2v_custom_loss_fwd_bwd(const float* y, const float* target, float* dy, float* terms, float* partial,
                            float w_l1, float w_cont, float w_var, float g_scale,
                            int T, int B, int D, g2v_stream_t stream);

/* MSE loss + gradient (train_iter_DAE, train_eval/train_seq2seq.py:208-222): loss[0] = mean((y-t)^2),
 * dy = g_scale * 2 (y-t)/n (dy may be NULL).  partial: >= g2v_mse_blocks(n) floats. */
int g2v_mse_blocks(int64_t n);
int g2v_mse_fwd_bwd(const float* y, const float* target, float* dy, float* loss, float* partial, int64_t n,
                    float g_scale, g2v_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Fused clip_grad_norm_(5) + Adam over one flat parameter buffer (K11),
 * train_eval/train_seq2seq.py:743-744, train_autoencoder_VQVAE.py:193-195.
 *   total = ||grad||_2; coef = min(1, max_norm/(total+1e-6)); g = grad*coef*grad_scale;
 *   m = b1 m + (1-b1) g; v = b2 v + (1-b2) g^2; p -= lr/(1-b1^t) * m / (sqrt(v)/sqrt(1-b2^t) + eps)
 * step_counter: device int32, incremented by this call before use (so hipGraph replays advance it).
 * gnorm_out (1 float, may be NULL).  partial: >= g2v_adam_blocks(n) floats.
 * ------------------------------------------------------------------------------------------ */
int g2v_adam_blocks(int64_t n);
/* out4[k] = *s_k (NULL: 0), k < 3; out4[3] = g2v_dec_rollout_persist_fault's latch as a float: the scalars an iteration reads
 * back (loss terms, perplexity) plus the latch in ONE device array, i.e. one device-to-host copy at the iteration's sync point. */
int g2v_iteration_readback(const float* s0, const float* s1, const float* s2, float* out4, g2v_stream_t stream);
/* g2v_clip_adam_step with g2v_iteration_readback(s0, s1, s2, out4) folded into its second launch (out4 is written also when the
 * fault latch holds the update back) */
int g2v_clip_adam_step_readback(float* param, const float* grad, float* m, float* v, int64_t n,
                                float* partial, int32_t* step_counter, float* gnorm_out,
                                float max_norm, float grad_scale, float lr, float beta1, float beta2, float eps,
                                const float* s0, const float* s1, const float* s2, float* out4, g2v_stream_t stream);
int g2v_clip_adam_step(float* param, const float* grad, float* m, float* v, int64_t n,
                       float* partial, int32_t* step_counter, float* gnorm_out,
                       float max_norm, float grad_scale, float lr, float beta1, float beta2, float eps,
                       g2v_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Soft quantiser VQ_Payam_GSSoft (model/Autoencoder_VQVAE_model.py:1304-1438; the variant Autoencoder_VQVAE ships with,
 * :816-820).  flat = mean_layer(x) and logvar = logvar_layer(flat) are g2v_linear_fwd calls, dots = flat W^T too.
 *   g2v_vq_soft_fwd   d = |flat|^2 + |W|^2 - 2 dots (written over `dots_to_dist`), smooth = 1/exp(logvar)^2,
 *                     probs = row-normalised exp(-(d/400) * 0.5 * smooth)/sqrt(smooth) (:1349-1372,1396-1411);
 *                     perplexity[0] = exp(-sum_k avg_k log(avg_k + 1e-10)), avg = mean_n probs (:1432-1433; may be NULL)
 *   g2v_vq_soft_bwd   from dprobs: dd (N,K) = dL/d distance, dlogvar (N,K), rowsum (N) = sum_k dd
 *   g2v_rowscale_combine  out[r,c] = 2 a[r,c] v[r] - 2 t[r,c]: the gradient of d wrt flat (a = flat, v = rowsum,
 *                     t = dd W) and wrt the codebook (a = W, v = column sums of dd, t = dd^T flat)
 *   g2v_ste_f32       out = z + (q - z)   (the straight-through value, :1431)
 *   g2v_vq_soft_perplexity  the same perplexity from probs as its own call over the whole device (two launches: per-row-range
 *                     column sums into `workspace`, >= g2v_vq_soft_perplexity_workspace(N, K) bytes, then one workgroup; fixed
 *                     summation order).  g2v_vq_soft_fwd's built-in one runs as ONE workgroup (it has no workspace to spread
 *                     over): 255 us at N = 4096, K = 512 against 12 here -- pass perplexity = NULL there and call this.
 * q = probs W is g2v_linear_bwd_data(probs, W); its gradients are g2v_linear_fwd / g2v_linear_bwd_weight.
 * ------------------------------------------------------------------------------------------ */
size_t g2v_vq_soft_perplexity_workspace(int N, int K);
int g2v_vq_soft_perplexity(const float* probs, float* perplexity, int N, int K, void* workspace, size_t workspace_bytes,
                           g2v_stream_t stream);
int g2v_vq_soft_fwd(const float* flat, float* dots_to_dist, const float* logvar, const float* code_sqnorm, float* probs,
                    float* perplexity, int N, int E, int K, g2v_stream_t stream);
int g2v_vq_soft_bwd(const float* probs, const float* dprobs, const float* dist, const float* logvar, float* dd,
                    float* dlogvar, float* rowsum, int N, int K, g2v_stream_t stream);
int g2v_rowscale_combine(const float* a, const float* v, const float* t, float* out, int64_t rows, int cols,
                         g2v_stream_t stream);
int g2v_ste_f32(const float* z, const float* q, float* out, int64_t n, g2v_stream_t stream);
/* The same quantiser, fused (csrc/vq_soft.hip; E == 128, K % 128 == 0, K <= 1024: g2v_vq_soft_fused_ok, else use the calls above).
 * A workgroup owns 16 rows of x (N,E); nothing (N,K)-sized travels between launches:
 *   g2v_vq_soft_fused_fwd   flat = mean_layer(x), logvar = logvar_layer(flat), dist, probs, q = probs W (all written: the
 *                           backward and the weight gradients read them), dq = 2 g_scale (q - x) / (N E) (the q_latent gradient),
 *                           quant = x + (q - x), mse_partial[blk] = sum over the workgroup's rows of (q - x)^2 and -- colsum may be
 *                           NULL -- colsum[blk][K] = column sums of probs; blk < g2v_vq_soft_fused_blocks(N)
 *   g2v_vq_soft_finish      mse = sum mse_partial / (N E) (mse may be NULL), loss_vq = mse * one_plus_beta[0] (device scalar),
 *                           perplexity = exp(-sum_k avg_k log(avg_k + 1e-10)) from colsum (both may be NULL); one workgroup
 *   g2v_vq_soft_fused_bwd   gz = [dh + g_loss[0] 2 beta (x - q) / (N E)] + dflat W_mean with dflat = (2 flat sum_k dd - 2 dd W) +
 *                           dlogvar W_logvar, (dd, dlogvar) = the backward of probs wrt (dist, logvar) at dprobs = dq W^T; dd,
 *                           dlogvar (N,K) and dflat (N,E) are written for g2v_linear_bwd_weight.  dh / g_loss may be NULL (= 0).
 * Element-wise arithmetic as in the separate kernels; the K- and E-long sums meet in another fixed order (fp32 rounding). */
int g2v_vq_soft_fused_ok(int N, int E, int K);
int g2v_vq_soft_fused_blocks(int N);
int g2v_vq_soft_fused_fwd(const float* x, const float* w_mean, const float* b_mean, const float* w_logvar, const float* b_logvar,
                          const float* codebook, const float* code_sqnorm, float* flat, float* logvar, float* dist, float* probs,
                          float* q, float* dq, float* quant, float* mse_partial, float* colsum, float g_scale, int N, int E, int K,
                          g2v_stream_t stream);
int g2v_vq_soft_finish(const float* mse_partial, const float* colsum, const float* one_plus_beta, float* mse, float* loss_vq,
                       float* perplexity, int N, int E, int K, g2v_stream_t stream);
int g2v_vq_soft_fused_bwd(const float* dh, const float* g_loss, const float* x, const float* q, const float* dq, const float* flat,
                          const float* probs, const float* dist, const float* logvar, const float* w_mean, const float* w_logvar,
                          const float* codebook, float* dd, float* dlogvar, float* dflat, float* gz, float beta, int N, int E, int K,
                          g2v_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Remaining operators of Part d (text -> gesture-code seq2seq, model/text2embedding_model.py).
 *   g2v_embedding_fwd   out[r,:] = table[ids[r],:] * keep * scale   (nn.Embedding :90-92,126 / :252,340-343 with the
 *                       decoder's nn.Dropout(0.5) fused; keep may be NULL; row r of out starts at out + r * ldo, e.g. the
 *                       first half of the attention decoder's (B,2H) input)
 *   g2v_embedding_bwd   d_table[v,:] (+)= sum_{r: ids[r]=v} d_out[r,:] * keep * scale.  No float atomics: tokens are counting-
 *                       sorted by row (stable), 128-token chunks of the sorted list are summed in order and rows that span
 *                       chunks add their chunk partials in order -- a fixed summation tree, bitwise reproducible.
 *                       zero_first bit 0: overwrite instead of accumulate; bit 1 (round 6): `ws` still holds the sort of THESE
 *                       ids from the previous call (same ids, n, V, untouched since) -- the sort is skipped (the two directions of
 *                       a bidirectional layer scatter-add by the same words).  ws: g2v_embedding_bwd_ws_bytes(n, dim, V) bytes
 *                       of device scratch (ids outside [0,V) contribute nothing)
 *   g2v_batchnorm_fwd   nn.BatchNorm1d(H) on (B,H) (+ optional fused ReLU), decoder.pre_linear[1:] :286-290; training:
 *                       batch statistics, running stats updated with momentum 0.1 / unbiased variance (both NULL while training:
 *                       left alone, the caller commits them with g2v_bn_running_update_invstd); save_* for bwd
 *   g2v_batchnorm_bwd   dx, dweight, dbias (overwritten) from dy (the ReLU mask is taken from y > 0 when relu)
 *   g2v_batchnorm_bwd_steps   the same for the `steps` calls of a decode loop in one launch (dy, x, y, dx: (steps,B,H); row s
 *                       of save_* at + s * stat_stride floats; B < 1024): dx per step bitwise g2v_batchnorm_bwd's, dw / db the
 *                       SUM over the steps in call order (autograd's accumulation over the loop's T-1 BatchNorm nodes,
 *                       model/text2embedding_model.py:286-290 under :701-744)
 *   g2v_one_hot_rows    out[r,:] = one_hot(ids[r]) as float (row r at out + r * ld; an id outside [0,K) gives a zero row):
 *                       outputs[:, 0, :] of Part d (model/text2embedding_model.py:676-677)
 *   g2v_cross_entropy_fwd_bwd   loss[0] = mean_r( logsumexp(logits[r]) - logits[r, t_r] ), dlogits = g_scale *
 *                       (softmax - onehot)/M  (torch.nn.CrossEntropyLoss, train_eval/train_seq2seq.py:520-530);
 *                       row_loss: M floats of scratch; dlogits may be NULL
 *   g2v_argmax_rows     out[r] = argmax_k x[r,k] (lowest index on ties): greedy feedback :740
 * ------------------------------------------------------------------------------------------ */
int g2v_embedding_fwd(const float* table, const int64_t* ids, const uint8_t* keep, float scale, float* out, int64_t ldo,
                      int64_t n, int dim, int64_t V, g2v_stream_t stream);
size_t g2v_embedding_bwd_ws_bytes(int64_t n, int dim, int64_t V);
int g2v_embedding_bwd(const float* d_out, const int64_t* ids, const uint8_t* keep, float scale, float* d_table,
                      int64_t n, int dim, int64_t V, int zero_first, void* ws, size_t ws_bytes, g2v_stream_t stream);
int g2v_batchnorm_fwd(const float* x, const float* weight, const float* bias, float* running_mean, float* running_var,
                      int training, int relu, float* y, float* save_mean, float* save_invstd, int B, int H,
                      g2v_stream_t stream);
/* Deferred commit of the running statistics from the saved (mean, 1/sqrt(var + eps)) of `steps` training calls that were given
 * running_mean = running_var = NULL (g2v_batchnorm_fwd, g2v_attn_code_rollout_fwd), in call order; row s of either array starts
 * at + s * step_stride floats.  Does nothing while the persistent kernels' fault latch is set (g2v_dec_rollout_persist_fault):
 * place it behind the backward, in front of the optimiser step.  Replaces the T-1 in-place updates of the reference's
 * nn.BatchNorm1d inside the decode loop (model/text2embedding_model.py:286-290 under :701-744). */
int g2v_bn_running_update_invstd(const float* save_mean, const float* save_invstd, int64_t step_stride, float* running_mean,
                                 float* running_var, int steps, int H, int B, g2v_stream_t stream);
int g2v_batchnorm_bwd(const float* dy, const float* x, const float* y, const float* weight, const float* save_mean,
                      const float* save_invstd, int relu, float* dx, float* dw, float* db, int B, int H,
                      g2v_stream_t stream);
int g2v_batchnorm_bwd_steps(const float* dy, const float* x, const float* y, const float* weight, const float* save_mean,
                            const float* save_invstd, int64_t stat_stride, int relu, float* dx, float* dw, float* db, int steps,
                            int B, int H, g2v_stream_t stream);
int g2v_one_hot_rows(const int64_t* ids, float* out, int64_t ld, int M, int K, g2v_stream_t stream);
int g2v_cross_entropy_fwd_bwd(const float* logits, int64_t ld, const int64_t* targets, float* loss, float* row_loss,
                              float* dlogits, int64_t ldd, int M, int K, float g_scale, g2v_stream_t stream);
int g2v_argmax_rows(const float* x, int64_t ld, int64_t* out, int M, int K, g2v_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Bahdanau attention of the code decoder (autoencoder_att == "True", config/seq2seqtxt.yml:37):
 * Attn.forward / Attn.score (model/text2embedding_model.py:160-198) + the context product :353-359.
 *   energy[t,b,:] = tanh(attn([h_b ; enc[t,b,:]])) = tanh(hp[b,:] + ep[t,b,:]) with
 *     hp (B,H) = h W_attn[:, :H]^T + b_attn   (g2v_linear_fwd, once per decode step)
 *     ep (T,B,H) = enc W_attn[:, H:]^T        (g2v_linear_fwd, once per batch of sentences)
 *   weights[b,:] = softmax_t( v . energy[t,b,:] )  over ALL T positions (no padding mask, as in the reference)
 *   ctx[b,:]     = sum_t weights[b,t] * enc[t,b,:]           (written with row stride ldctx, e.g. into the second
 *                                                              half of the (B,2H) decoder input)
 * g2v_attn_bwd: from d_ctx (row stride ldd) computes d_hp (B,H), d_ep (T,B,H), d_enc (T,B,H) [the direct context
 * term only; the ep path continues through g2v_linear_bwd_data] and d_v (H); accumulate != 0 adds into d_ep / d_enc /
 * d_v (the S-1 decode steps share them).  Deterministic (fixed summation order, no atomics).  d_v == NULL: the per-row
 * partials (B,H) stay in `workspace` and the caller sums them (g2v_slab_sum; one reduction behind the last step instead of one
 * per step when every step is handed its own workspace slab).
 * g2v_attn_step_fwd (round 6): the row-local head of a decode step in one launch -- id[b] = argmax_k logits[b,k] (the greedy
 * feedback :740, lowest index on ties; logits == NULL: ids[b] is given, the teacher-forced steps), written to ids;
 * ec[b, :H] = table[id[b], :] * keep[b, :] * emb_scale (Embedding + Dropout(0.5) :340-343; keep may be NULL; table is (K,H));
 * ec[b, H:2H] = the attention context of g2v_attn_fwd (row b of ec at ec + b * ldec, ldec >= 2H); weights as g2v_attn_fwd.
 * ------------------------------------------------------------------------------------------ */
int g2v_attn_fwd(const float* hp, const float* ep, const float* enc, const float* v, float* weights, float* ctx,
                 int64_t ldctx, int T, int B, int H, g2v_stream_t stream);
int g2v_attn_step_fwd(const float* logits, int64_t ldl, int K, int64_t* ids, const float* table, const uint8_t* keep,
                      float emb_scale, float* ec, int64_t ldec, const float* hp, const float* ep, const float* enc,
                      const float* v, float* weights, int T, int B, int H, g2v_stream_t stream);
/* out (+)= slab 0 + slab 1 + ... + slab n-1 (n slabs of `len` floats, added in slab order) */
int g2v_slab_sum(const float* slabs, int n, int64_t len, float* out, int accumulate, g2v_stream_t stream);
size_t g2v_attn_bwd_workspace(int B, int H);
int g2v_attn_bwd(const float* d_ctx, int64_t ldd, const float* hp, const float* ep, const float* enc, const float* v,
                 const float* weights, float* d_hp, float* d_ep, float* d_enc, float* d_v, int accumulate, int T, int B,
                 int H, void* workspace, size_t workspace_bytes, g2v_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * K13 + K14 (SURVEY.md 8(b) "attn_code_rollout_fwd/bwd"): the greedy gesture-code decoder of Part d as FUSED PER-STEP KERNELS.
 * Replaces the loop model/text2embedding_model.py:701-744 over BahdanauAttnDecoderRNN.forward (:338-395), n_layers == 2,
 * discrete_representation (code ids in, logits out):
 *   e_t   = Dropout(0.5)(Embedding(id_t))                                  (:340-343; keep_emb, scale 2)
 *   [att] w_t = softmax_tw( v . tanh(attn([h1_{t} ; enc[tw]])) ), ctx_t = sum_tw w_t[tw] enc[tw]; x_t = [e_t | ctx_t]   (:347-359)
 *   u_t   = x_t W_pre^T + b_pre;  a_t = ReLU(BatchNorm1d(u_t))              (:376, batch statistics when training)
 *   h0_{t+1}, h1_{t+1} = GRU(a_t; h0_t, h1_t)  (inter-layer dropout p on h0: keep_l0)      (:380)
 *   logits_t = h1_{t+1} W_out^T + b_out                                     (:390)
 *   id_{t+1} = codes[t+1] while t + 1 < n_pre, else argmax_k logits_t (lowest index on ties)    (:737-744)
 * for t = 0 .. S1-1 (S1 = sentence_frame_length // n_frames - 1; id_0 = codes[0]).  The step is row-local except for
 * BatchNorm's batch statistics, so the forward is S1 + 1 launches of ONE kernel (16 batch rows per 512-thread workgroup; launch j =
 * [BatchNorm finish of u_{j-1}, both GRU cells, out layer, argmax] + [embedding, attention, pre_linear of step j, per-workgroup
 * partial sums]); every contraction is v_mfma_f32_16x16x4_f32 on weights packed once per call.  Without attention nothing in
 * the BACKWARD couples batch rows between steps (the argmax feedback carries no gradient), so the whole BPTT over the S1 steps is
 * ONE launch (carries in LDS), followed by the BatchNorm backward of all steps at once and the batched products that do not
 * feed the recurrence (embedding gradient, every weight gradient over the S1 x B rows).  With attention the context gradient
 * needs BatchNorm's backward sums of the same step: the backward is S1 launches, each [BatchNorm backward finish of step t+1 ->
 * d(input) -> attention backward -> d(h1_t)] + [cells of step t].
 * H % 4 == 0, H <= 256, K <= 1024, Tw <= 64; g2v_attn_code_rollout_ok() says whether a shape is served (0: chain the operators
 * above from the host, as rounds 1-4 did).  All arrays caller-owned, time-major, fp32 unless said.
 * ------------------------------------------------------------------------------------------ */
typedef struct {
  const float* emb;              /* (K,H)    decoder.embedding.weight                         */
  const float* w_pre;            /* (H,Hin)  decoder.pre_linear.0.weight; Hin = H, or 2H with attention */
  const float* b_pre;            /* (H)                                                       */
  const float* bn_w;             /* (H)      pre_linear.1.weight                              */
  const float* bn_b;             /* (H)                                                       */
  float* bn_running_mean;        /* (H)      updated once per step when training; both NULL while training: left alone
                                  *          (the caller commits them later: g2v_bn_running_update_invstd on bn_stats)        */
  float* bn_running_var;         /* (H)                                                       */
  const float* w_ih0; const float* w_hh0; const float* b_ih0; const float* b_hh0;   /* (3H,H), (3H) */
  const float* w_ih1; const float* w_hh1; const float* b_ih1; const float* b_hh1;
  const float* w_out;            /* (K,H)    decoder.out.weight                               */
  const float* b_out;            /* (K)                                                       */
  const float* w_attn;           /* (H,2H)   attn.attn.weight: [:, :H] on the state, [:, H:] on the encoder outputs; NULL = no attention */
  const float* b_attn;           /* (H)                                                       */
  const float* v_attn;           /* (H)      attn.v                                           */
} g2v_code_dec_weights;

typedef struct {                 /* outputs + saved for backward                              */
  int64_t* ids;                  /* (S1,B)    the code fed at every step                      */
  float* ec;                     /* (S1,B,Hin) dropped embedding [| context]                  */
  float* u;                      /* (S1,B,H)                                                  */
  float* a;                      /* (S1,B,H)                                                  */
  float* bn_stats;               /* (S1,2,H)  batch mean / 1/sqrt(biased var + eps) per step  */
  float* h0;                     /* (S1+1,B,H) slot t = state in front of step t              */
  float* h1;                     /* (S1+1,B,H)                                                */
  float* x1;                     /* (S1,B,H)  dropped h0_{t+1} (layer 1's input); NULL without inter-layer dropout */
  float* gates0;                 /* (S1,B,4H) r, z, n, W_hn h + b_hn                          */
  float* gates1;                 /* (S1,B,4H)                                                 */
  float* logits;                 /* (S1,B,K)                                                  */
  float* bn_partial;             /* (2,nblk,2,H) scratch, nblk = g2v_attn_code_rollout_blocks(B) */
  float* hp;                     /* (S1,B,H)  attention: h1_t W_attn[:, :H]^T + b_attn; NULL without attention */
  float* attw;                   /* (S1,B,Tw) attention weights                               */
} g2v_code_dec_saved;

typedef struct {                 /* gradients (overwritten)                                   */
  float* d_hidden0;              /* (2,B,H)   w.r.t. the initial state                        */
  float* d_emb;                  /* (K,H)                                                     */
  float* d_w_pre; float* d_b_pre; float* d_bn_w; float* d_bn_b;
  float* d_w_ih0; float* d_w_hh0; float* d_b_ih0; float* d_b_hh0;
  float* d_w_ih1; float* d_w_hh1; float* d_b_ih1; float* d_b_hh1;
  float* d_w_out; float* d_b_out;
  float* d_w_attn; float* d_b_attn; float* d_v_attn;      /* attention only                    */
  float* d_enc;                  /* (Tw,B,H)  w.r.t. the encoder outputs (context + energies)  */
} g2v_code_dec_grads;

int g2v_attn_code_rollout_ok(int S1, int B, int H, int K, int Tw, int attention);
/* Round 5: 1 where g2v_attn_code_rollout_fwd runs the shape as ONE persistent cluster launch of (hidden-unit tile x row group)
 * workgroups (csrc/t2e_rollout.hip: code_cluster_fwd_kernel -- the scheme of the pose decoder's cluster kernels, see
 * g2v_dec_rollout_set_persistent, which also switches this one): no attention, H <= 208, K <= 48 ceil(H / 16), the grid with a CU
 * per workgroup (B <= 304 at H = 200).  It writes the same arrays as the step kernels, so g2v_attn_code_rollout_bwd or a
 * per-operator backward run on them unchanged. */
int g2v_attn_code_rollout_cluster_ok(int S1, int B, int H, int K, int attention);
/* Round 5: for the same shapes, the BPTT through the two GRU cells of the attention-free rollout as ONE persistent cluster launch
 * (code_cluster_bptt_kernel: the partial-product exchange of the pose decoder's backward cluster).  Nothing but the cells couples
 * two steps there (the greedy feedback carries no gradient), so the caller forms dh_top (S1,B,H) = dLogits W_out for all steps
 * first (g2v_linear_bwd_data) and runs BatchNorm's backward, the embedding gradient and every weight gradient afterwards over all
 * steps at once.  Writes dgi0 / dgh0 / dgi1 / dgh1 (S1,B,3H), da (S1,B,H) = the gradient w.r.t. a_t = ReLU(BN(u_t)) before the
 * ReLU mask, d_hidden0 (2,B,H); reads s->gates0 / gates1 / h0 / h1; keep_l0 (S1,B,H) or NULL: the inter-layer dropout masks. */
size_t g2v_code_cluster_bptt_workspace(int B, int H);
int g2v_code_cluster_bptt(const float* dh_top, const g2v_code_dec_weights* w, const g2v_code_dec_saved* s, const uint8_t* keep_l0,
                          float p_drop, float* dgi0, float* dgh0, float* dgi1, float* dgh1, float* da, float* d_hidden0,
                          int S1, int B, int H, void* workspace, size_t workspace_bytes, g2v_stream_t stream);
int g2v_attn_code_rollout_blocks(int B);
size_t g2v_attn_code_rollout_fwd_workspace(int H, int K, int attention);
/* codes (S,B) int64 (rows 0 .. max(1, n_pre) - 1 are read); h_init (2,B,H); enc (Tw,B,H) and enc_proj = enc W_attn[:, H:]^T
 * (Tw,B,H) with attention, else NULL; keep_emb / keep_l0 (S1,B,H) uint8 keep masks (NULL: no dropout; training == 0: eval
 * BatchNorm on the running statistics). */
int g2v_attn_code_rollout_fwd(const int64_t* codes, const float* h_init, const float* enc, const float* enc_proj,
                              const g2v_code_dec_weights* w, const g2v_code_dec_saved* s, const uint8_t* keep_emb,
                              const uint8_t* keep_l0, float p_drop, int n_pre, int training, int S1, int B, int H, int K,
                              int Tw, void* workspace, size_t workspace_bytes, g2v_stream_t stream);
size_t g2v_attn_code_rollout_bwd_workspace(int S1, int B, int H, int K, int Tw, int attention);
/* d_logits (S1,B,K); everything else as passed to the forward of the same call. */
int g2v_attn_code_rollout_bwd(const float* d_logits, const float* enc, const float* enc_proj, const g2v_code_dec_weights* w,
                              const g2v_code_dec_saved* s, const g2v_code_dec_grads* g, const uint8_t* keep_emb,
                              const uint8_t* keep_l0, float p_drop, int S1, int B, int H, int K, int Tw, void* workspace,
                              size_t workspace_bytes, g2v_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Calibration probes used by bench.py to report what the box sustains next to the datasheet peaks (no counterpart in the
 * reference): g2v_probe_mfma_f32 runs `blocks` x 4 waves x `iters` x 8 independent v_mfma_f32_16x16x4_f32 (2048 flop each)
 * without memory traffic (scratch: blocks * 256 floats); g2v_probe_copy streams n floats from src to dst.
 * ------------------------------------------------------------------------------------------ */
int g2v_probe_mfma_f32(float* scratch, int blocks, int iters, g2v_stream_t stream);
int g2v_probe_copy(const float* src, float* dst, int64_t n, g2v_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Keep-mask generator: keep[i] = (philox4x32-10(seed, offset_counter, i) uniform < keep_prob).
 * Replaces the RNG draws of nn.Dropout / nn.GRU dropout on the path (e.g. :570).  offset_counter is a
 * device int64 that the call advances by 1 (so graph replays draw fresh masks).
 * ------------------------------------------------------------------------------------------ */
int g2v_keep_mask(uint8_t* keep, int64_t n, float keep_prob, uint64_t seed, int64_t* offset_counter,
                  g2v_stream_t stream);
/* The same stream at offset_counter[0] + offset_add WITHOUT advancing the counter, and the advance as its own call: several
 * masks of one step (offset_add = 0, 1, 2 ...) then cost one counter launch instead of one each. */
int g2v_keep_mask_at(uint8_t* keep, int64_t n, float keep_prob, uint64_t seed, const int64_t* offset_counter, int64_t offset_add,
                     g2v_stream_t stream);
int g2v_counter_add(int64_t* counter, int64_t n, g2v_stream_t stream);
/* g2v_keep_mask_at + g2v_mask_rows in one kernel, without the mask tensor: out[m, k] = keep(m K + k) ? x[row(m), k] * scale : 0
 * with keep(e) = element e of the stream (seed, offset_counter[0] + offset_add) -- bit for bit what the two calls produce. */
int g2v_dropout_rows(const float* x, int64_t ldx, int rows_inner, int64_t stride_outer, int64_t stride_inner, float keep_prob,
                     float scale, uint64_t seed, const int64_t* offset_counter, int64_t offset_add, float* out, int64_t ldo,
                     int M, int K, g2v_stream_t stream);

/* small helpers used by the host */
int g2v_fill_f32(float* p, float v, int64_t n, g2v_stream_t stream);
/* dst[k][i] = src[k][i] for i < n[k], k < nseg (src[k] == NULL: zero fill), all segments in one launch per 48.  src / dst /
 * n are HOST arrays (read at launch).  FlatParams.gather_grads: every parameter's .grad into the flat gradient buffer that
 * g2v_clip_adam_step consumes (torch.nn.utils.clip_grad_norm_ + Adam.step over a parameter list, train_seq2seq.py:540-546). */
int g2v_copy_segments(const float* const* src, float* const* dst, const int64_t* n, int nseg, g2v_stream_t stream);
/* out[i] = in[i] * scalar[0]   (scalar is a DEVICE float: chains an upstream autograd scalar without a host sync) */
int g2v_scale_f32(const float* in, const float* scalar, float* out, int64_t n, g2v_stream_t stream);
/* out[i] = on(i) ? in[i]*scale : 0, on(i) = keep[i] != 0 (uint8 mask) or, when keep is NULL, positive_of[i] > 0
 * (dropout / ReLU backward of the thin dense models). */
int g2v_mask_mul(const float* in, const uint8_t* keep, const float* positive_of, float scale, float* out,
                 int64_t n, g2v_stream_t stream);
/* out[m, k] = keep[m K + k] ? x[row(m), k] * scale : 0 for m < M, k < K -- the dropped input of a dense layer as a tensor of its
 * own (row addressing of x as in g2v_linear_fwd; out rows at stride ldo).  The encoder's input dropout
 * (model/Autoencoder_VQVAE_model.py:88-93: self.do(inputs) in front of in_layer) uses it once per step, so that the forward
 * product and the weight-gradient product both run on their unmasked fast kernels. */
int g2v_mask_rows(const float* x, int64_t ldx, int rows_inner, int64_t stride_outer, int64_t stride_inner, const uint8_t* keep,
                  float scale, float* out, int64_t ldo, int M, int K, g2v_stream_t stream);
int g2v_transpose(const float* in, float* out, int rows, int cols, g2v_stream_t stream); /* out[c][r] = in[r][c] */
/* out[m, 0:H] = a[m, 0:H] + b[m, 0:H] with row strides (sum of the two GRU directions, :95-97) */
int g2v_add_halves(const float* a, int64_t lda, const float* b, int64_t ldb, float* out, int64_t ldo,
                   int64_t M, int H, g2v_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* G2V_H */
