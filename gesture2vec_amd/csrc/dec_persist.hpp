// dec_persist.hpp -- shared pieces of the PERSISTENT decoder rollout kernels (dec_persist.hip) and their launcher
// hooks in dec_rollout.hip.
//
// Why a persistent kernel: with one launch per time step (dec_rollout.hip) every workgroup re-streams 276 KB of weight
// fragments and its state rows from L2 at every step, pays a kernel boundary plus the write-back of ~18 MB of dirty
// lines, and ramps up from cold registers -- 16.4 / 21.9 us per forward / backward step against a 3.5 us MFMA floor.
// Here one workgroup per 16 batch rows stays resident for all T steps with the four GRU matrices in registers
// (192 VGPRs per lane), the out / pre_linear matrices and every bias in LDS, and the recurrent state in LDS.
// What remains per step is BatchNorm1d's batch statistics: a grid-wide sum of 128 floats per workgroup.
//
// The exchange (measured on MI355X with gpurun_tools/exchange_bench.hip before this was written):
//   * agent-scope (sc1) loads are served at the fabric, not by the XCD's L2: a flat all-to-all in which every workgroup
//     reads all 256 records (32 MB of sc1 loads per step) costs ~10 us, and so does any scheme in which all 256
//     workgroups read the SAME lines (one line serves ~1 request per 10 ns chip-wide);
//   * so the sum runs as a fixed two-level tree of groups of 16 workgroups with 8-byte self-validating granules
//     {value, tag = step} (MI355X_MICROARCH.md "R2": one aligned 8-byte sc1 store is observed untorn, no flag, no fence,
//     no release).  The 256 workgroups form a 16 x 16 grid (row = b / 16, column = b % 16): every workgroup publishes its
//     128 granules, sums the 16 records of its ROW (hop 1) and publishes that row sum as its own second record, then sums
//     the row sums published by the workgroups of its COLUMN (hop 2).  No leaders (nobody waits for a workgroup that has
//     extra serial work), every line is read by ~16 workgroups, and the summation order is fixed -> bitwise reproducible.
//     In-kernel stamps (gpurun_tools/px_test.hip, 256 workgroups): publish -> group sums visible 1.0 us, sweep (one
//     fabric round trip) 0.75 us, LDS reduction 0.24 us, second hop the same: 3.1-4.2 us per exchange; waiting on one
//     sentinel granule per record before the full sweep was measured 1.1 us SLOWER than polling with the sweep itself.
//   * records are double-buffered by step parity: a workgroup can only publish step s+1 after it has consumed every
//     group record of step s, which exist only after every workgroup has published step s, i.e. finished reading s-1.
//   * every polled word is zeroed by a memset node in front of the launch; tags count steps inside the call (1..T).
//   * all workgroups must be co-resident: the launcher admits the path only when nblk <= the device's CU count and the
//     kernel needs one workgroup per CU (256 threads, <= 512 VGPRs, ~135 KB LDS).  Every spin is bounded; a thread whose
//     bound runs out (a workgroup of the launch is not resident: CU mask, another persistent launch interleaved on the same
//     device ...) LATCHES g2v_persist_fault and stops waiting, every other spinning thread sees the latch within 4096 polls
//     and stops waiting too: the launch ends (with garbage in its outputs) instead of trapping -- a GPU fault would kill
//     the process and, under data parallelism, hang the peers at the next collective.  The host reads the latch with
//     g2v_dec_rollout_persist_fault() at its own sync points (round-2 advisor finding).
#pragma once
#include "common.hpp"

#ifndef PX_STAMP          // diagnostic builds (gpurun_tools/px_test.hip) define it to record s_memrealtime stamps
#define PX_STAMP(k)
#endif

namespace g2v {

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// Sticky fault latch of the persistent kernels.  ONE translation unit owns it -- dec_persist.hip, which holds every kernel that
// latches it, its host reader / clear and g2v_internal_persist_fault_ptr() -- and defines G2V_PERSIST_DEVICE_CODE in front of
// this header; every other includer (dec_rollout.hip: the layout constants) gets neither the latch nor the device functions that
// use it, so no kernel elsewhere can fault into a private copy nobody reads (round-4 advisor finding).  Stand-alone tools
// (gpurun_tools/px_test.hip) define the macro themselves.
#ifdef G2V_PERSIST_DEVICE_CODE
static __device__ unsigned g2v_persist_fault;
#endif

constexpr int PX_GROUP = 16;          // workgroups per exchange group
constexpr int PX_MAX_NBLK = 256;      // one workgroup per CU on MI355X
constexpr int PX_COLS = 128;          // floats per record (2 x H, H = 64)
constexpr size_t PX_REC1_BYTES = (size_t)2 * PX_MAX_NBLK * PX_COLS * 8;
constexpr int PX_MAX_GRP = PX_MAX_NBLK / PX_GROUP;
constexpr size_t PX_REC2_BYTES = (size_t)2 * PX_MAX_NBLK * PX_COLS * 8;               // [parity][workgroup]: its row sum
// progress words of the forward rollout for the co-resident custom_loss chaser (dec_persist.hip, loss_chase_kernel): one word per
// workgroup, each on a 64-byte line of its own; value t = "y_0 .. y_t of this workgroup's 16 rows are in memory".  Cleared with the
// exchange records (same memset).
constexpr int PX_FLAG_STRIDE = 16;    // dwords
constexpr size_t PX_FLAG_BYTES = (size_t)PX_MAX_NBLK * PX_FLAG_STRIDE * 4;
constexpr size_t PX_BYTES = PX_REC1_BYTES + PX_REC2_BYTES + PX_FLAG_BYTES;     // 1 MiB + 16 KiB: a multiple of 16 (memset price, Guideline 16)

struct PersistX {
  unsigned long long* rec1;   // [2][PX_MAX_NBLK][128] granules {value (low dword), tag (high dword)}
  unsigned long long* rec2;   // [2][PX_MAX_NBLK][128]: the row sum as published by each workgroup
  unsigned* yflag;            // [PX_MAX_NBLK][PX_FLAG_STRIDE]: forward progress (see PX_FLAG_STRIDE)
};
static inline PersistX persist_x_at(void* base) {
  PersistX x;
  x.rec1 = reinterpret_cast<unsigned long long*>(base);
  x.rec2 = x.rec1 + (size_t)2 * PX_MAX_NBLK * PX_COLS;
  x.yflag = reinterpret_cast<unsigned*>(x.rec2 + (size_t)2 * PX_MAX_NBLK * PX_COLS);
  return x;
}

__device__ __forceinline__ u32x4 px_ld(__amdgpu_buffer_rsrc_t r, unsigned byte_off) {
  return __builtin_amdgcn_raw_buffer_load_b128(r, byte_off, 0, 16);      // aux 16 = sc1 (agent scope, bypasses L1)
}
__device__ __forceinline__ void px_st(__amdgpu_buffer_rsrc_t r, unsigned byte_off, u32x4 v) {
  __builtin_amdgcn_raw_buffer_store_b128(v, r, byte_off, 0, 16);          // write-through
}
__device__ __forceinline__ void px_st_l2(__amdgpu_buffer_rsrc_t r, unsigned byte_off, u32x4 v) {
  __builtin_amdgcn_raw_buffer_store_b128(v, r, byte_off, 0, 0);           // plain: the line stays in this XCD's L2
}

// Publish two values as two granules with ONE 16-byte store (each 8-byte half is self-validating).
__device__ __forceinline__ void px_publish2(__amdgpu_buffer_rsrc_t r, unsigned granule, float v0, float v1, unsigned tag,
                                            bool xcd_local = false) {
  u32x4 g;
  g[0] = __float_as_uint(v0); g[1] = tag;
  g[2] = __float_as_uint(v1); g[3] = tag;
  if (xcd_local) px_st_l2(r, granule * 8u, g);
  else px_st(r, granule * 8u, g);
}

// The 16 x 16 grid by XCD affinity.  Workgroups b and b + 8 share an XCD under the dispatcher's round-robin placement (observed,
// never promised): with nblk a multiple of 128 the workgroups are renumbered L = (b % 8) (nblk / 8) + b / 8, so that the 16
// workgroups of a ROW (hop 1) are workgroups of one residue class.  The renumbering itself is placement-independent: it only
// decides which records a workgroup sums, in which order.
#ifndef PX_AFFINE_GRID      // the product keeps the plain 16 x 16 grid (see "measured, not shipped" below); px_test.hip defines it
#define PX_AFFINE_GRID 0
#endif
__device__ __forceinline__ bool px_affine(int nblk) { return PX_AFFINE_GRID && (nblk & 127) == 0; }
__device__ __forceinline__ int px_logical(int b, int nblk) { return px_affine(nblk) ? (b & 7) * (nblk >> 3) + (b >> 3) : b; }
__device__ __forceinline__ int px_physical(int L, int nblk) {
  if (!px_affine(nblk)) return L;
  const int cs = nblk >> 3;
  return (L % cs) * 8 + L / cs;
}

// Hop 1 through the XCD's own L2 (round 4: built, verified, measured -- NOT shipped: the kernels in dec_persist.hip do not call
// px_announce and pass no decision word).  In the harness the exchange drops from 3.3-3.9 to 2.4-2.9 us and all 256 workgroups
// decide for the L2 path; in the train step the same change bought 4 us of 1590 (1.5896 -> 1.5854 ms, 3 x 300 steps): the forward
// rollout already hides hop 1 under a hidden-side product, and in both rollouts the wait is for the LAST workgroup to publish,
// not for the fabric.  A quarter of a percent does not pay for a second memory-ordering regime in the kernels that everything
// else depends on.  What was built:
// A record published with PLAIN stores stays in the publishing XCD's L2, where the
// sc1 loads of a reader ON THE SAME XCD find it without the fabric round trip (MI355X_MICROARCH.md, stores / loads table:
// "plain KEEP the line in the XCD's L2", "sc1 loads bypass L1 only"): publish -> row sums visible ~1.0 -> ~0.4 us, the exchange
// 3.3-3.9 -> 2.4-2.9 us (gpurun_tools/px_test.hip, last argument).  A reader on ANOTHER XCD would never see such a record, and
// placement is not promised -- so it is verified, per launch and per row, with what the hardware reports:
//   * every workgroup announces the XCC it runs on (HW_REG_XCC_ID, + 1 so that the cleared word means "not yet") in word 1 of its
//     flag line, write-through, acknowledged (vmcnt(0)) in front of the workgroup barrier that precedes its first publish;
//   * the FIRST exchange of a launch is published write-through (the placement-independent form);
//   * behind its first hop 1 -- every row mate has published, hence announced -- a workgroup reads its 16 row mates' words: all
//     equal => this row's records go through L2 from now on.  All 16 see the same 16 words: the row decides as one, and the
//     readers' loads are the same sc1 loads either way.  Rows that are not on one XCD (placement other than round-robin, nblk not
//     a multiple of 128) simply stay write-through.  Hop 2 crosses XCDs by construction and is always write-through.
// The decision lives in one LDS word of the kernel (*xl: -1 undecided, 0 write-through, 1 through L2).
__device__ __forceinline__ void px_announce(const PersistX& x, int b, int* xl) {      // ONE lane, in front of a workgroup barrier
  unsigned xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  *xl = -1;
  __hip_atomic_store(x.yflag + (size_t)b * PX_FLAG_STRIDE + 1, (xcc & 0xfu) + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}
__device__ __forceinline__ int px_row_on_one_xcd(const PersistX& x, int nblk, int b) {
  if (!px_affine(nblk)) return 0;
  const int row = px_logical(b, nblk) / PX_GROUP;
  unsigned first = 0u;
  bool same = true;
#pragma unroll
  for (int m = 0; m < PX_GROUP; ++m) {
    const unsigned v = __hip_atomic_load(x.yflag + (size_t)px_physical(row * PX_GROUP + m, nblk) * PX_FLAG_STRIDE + 1,
                                         __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (m == 0) first = v;
    same = same && v == first && v != 0u;
  }
  return same ? 1 : 0;
}

// ---- the cluster kernels' exchange (gru.hip: gru_cluster_*_kernel, dec_rollout.hip: dec_cluster_fwd_kernel) ----------------------
// A "row record" holds 16 batch rows x (16 NT) columns as granules {value, tag}, TILE-MAJOR: granule ((tile 16 + row) 4 + q) 4 + e is
// column 16 tile + 4 q + e of that row.  The producer of a tile -- lane (row i, q) of the wave that holds the 16 x 16 result as an
// MFMA accumulator -- writes its four granules as 32 contiguous bytes, the whole tile as 2 KiB; a consumer sweeps whole tiles with
// fully used 16-byte lanes (chunk c of a tile: row c / 8, q = (c / 2) % 4, e = 2 (c % 2)) and scatters them into the MFMA B-fragment
// layout in LDS (xs[tile][q 16 + row], a float4 per lane = the k-step `tile` fragment).  sc1 loads are served at the fabric at
// ~25 GB/s per CU (measured with in-kernel stamps: a 26 KiB sweep whose lanes used half of every line took 2.1 us with the data
// already there), so the bytes a workgroup requests per step are what an exchange costs.
__device__ __forceinline__ bool cx_give_up(unsigned& spins, unsigned* fault) {
  ++spins;
  if (spins > 4000000u ||
      ((spins & 4095u) == 0 && fault && __hip_atomic_load(fault, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u)) {
    if (fault) __hip_atomic_store(fault, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return true;
  }
  return false;
}
__device__ __forceinline__ unsigned cx_tile_granule(int tile, int i, int q) { return (unsigned)(((tile * 16 + i) * 4 + q) * 4); }
// lane (i, q) publishes v[0..3] = columns 16 tile + 4 q .. + 3 of row i
// xcd_local: every reader of the record runs on the publisher's XCD (cx_cluster_on_one_xcd below) -- a PLAIN store then, which
// leaves the line in that XCD's L2, where the readers' sc1 loads (they bypass L1 only) find it without the fabric round trip.
__device__ __forceinline__ void cx_publish4(__amdgpu_buffer_rsrc_t rr, unsigned rec_granule0, int tile, int i, int q, const float* v,
                                            unsigned tag, bool xcd_local = false) {
  u32x4 a, b;
  a[0] = __float_as_uint(v[0]); a[1] = tag; a[2] = __float_as_uint(v[1]); a[3] = tag;
  b[0] = __float_as_uint(v[2]); b[1] = tag; b[2] = __float_as_uint(v[3]); b[3] = tag;
  const unsigned off = (rec_granule0 + cx_tile_granule(tile, i, q)) * 8u;
  if (xcd_local) {
    px_st_l2(rr, off, a);
    px_st_l2(rr, off + 16u, b);
  } else {
    px_st(rr, off, a);
    px_st(rr, off + 16u, b);
  }
}
// Do the `n` workgroups of this cluster (the ones that exchange row records with each other) run on ONE XCD?  Every workgroup
// announces the XCC it runs on (HW_REG_XCC_ID + 1, write-through, into word `me` of `words`: zeroed with the exchange records) and
// reads all n words: all equal => the cluster's records may go through that XCD's L2.  All n workgroups read the same n words, so
// the cluster decides as one; placement is observed (round-robin over the linear workgroup id), never promised -- a cluster that
// is not on one XCD simply keeps the write-through stores.  Called by every thread of the workgroup; `flag` is one LDS word.
__device__ __forceinline__ bool cx_cluster_on_one_xcd(unsigned* words, int n, int me, int* flag, int tid, unsigned* fault) {
  if (tid == 0) {
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    __hip_atomic_store(words + me, (xcc & 0xfu) + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    bool same = true;
    unsigned first = 0u;
    for (int m = 0; m < n; ++m) {
      unsigned v, spins = 0;
      for (;;) {
        v = __hip_atomic_load(words + m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (v != 0u) break;
        __builtin_amdgcn_s_sleep(1);
        if (cx_give_up(spins, fault)) break;
      }
      if (m == 0) first = v;
      same = same && v == first && v != 0u;
    }
    *flag = same ? 1 : 0;
  }
  lds_barrier();
  return *flag != 0;
}
// One wave sweeps NTL tiles t0, t0 + tstr, ... (those < nt) of a row record into xs (LDS: [tile][64] float4, fragment layout).
// Rows >= nrows and columns >= K are never published: zeros are written for them.  First a SENTINEL poll -- lane j watches the
// first granule pair of tile j, one 16-byte load per round -- and only then the bulk: a round of the full sweep costs 1.2-2.7 us
// at the fabric (in-kernel stamps), so a round issued before the producers have published is that much lost.  The sentinel proves
// only itself: every granule of the bulk is still validated (and re-read while stale).
template <int NTL>
__device__ __forceinline__ void cx_sweep_tiles(__amdgpu_buffer_rsrc_t rr, unsigned rec_granule0, int t0, int tstr, int nt, int nrows,
                                               int K, unsigned tag, float4* xs, int lane, unsigned* fault,
                                               unsigned long long* dbg = nullptr) {
  if (dbg && lane == 0) dbg[0] = __builtin_amdgcn_s_memtime();
  unsigned spins = 0;
  {
    const int tile = t0 + lane * tstr;
    const bool watch = lane < NTL && tile < nt;      // (row 0 and the tile's first columns always exist)
    const unsigned off = (rec_granule0 + (unsigned)((watch ? tile : 0) * 256)) * 8u;
    for (;;) {
      u32x4 sgl = (u32x4){0u, tag, 0u, tag};
      if (watch) sgl = px_ld(rr, off);
      const bool ok = sgl[1] == tag && sgl[3] == tag;
      if (__builtin_amdgcn_ballot_w64(!ok) == 0ull) break;
      __builtin_amdgcn_s_sleep(1);
      if (cx_give_up(spins, fault)) break;
    }
  }
  if (dbg && lane == 0) dbg[1] = __builtin_amdgcn_s_memtime();
  u32x4 g[NTL][2];
  bool need[NTL][2];
#pragma unroll
  for (int j = 0; j < NTL; ++j) {
    const int tile = t0 + j * tstr;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int c = h * 64 + lane;                       // chunk of the tile: row c / 8, q = (c / 2) % 4, e = 2 (c % 2)
      const int i = c >> 3, q = (c >> 1) & 3;
      need[j][h] = tile < nt && i < nrows && 16 * tile + 4 * q < K;
      if (need[j][h]) g[j][h] = px_ld(rr, (rec_granule0 + (unsigned)(tile * 256)) * 8u + (unsigned)c * 16u);
      else g[j][h] = (u32x4){0u, tag, 0u, tag};
    }
  }
  for (;;) {
    bool ok = true;
#pragma unroll
    for (int j = 0; j < NTL; ++j) ok &= g[j][0][1] == tag && g[j][0][3] == tag && g[j][1][1] == tag && g[j][1][3] == tag;
    if (dbg && lane == 0 && spins < 12) dbg[2 + spins] = __builtin_amdgcn_s_memtime();
    if (ok) break;
    __builtin_amdgcn_s_sleep(1);
#pragma unroll
    for (int j = 0; j < NTL; ++j) {
      const int tile = t0 + j * tstr;
#pragma unroll
      for (int h = 0; h < 2; ++h)
        if (g[j][h][1] != tag || g[j][h][3] != tag)
          g[j][h] = px_ld(rr, (rec_granule0 + (unsigned)(tile * 256)) * 8u + (unsigned)(h * 64 + lane) * 16u);
    }
    if (cx_give_up(spins, fault)) break;
  }
  if (dbg && lane == 0) dbg[15] = spins;
#pragma unroll
  for (int j = 0; j < NTL; ++j) {
    const int tile = t0 + j * tstr;
    if (tile < nt) {      // (uniform)
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int c = h * 64 + lane;
        const int i = c >> 3, q = (c >> 1) & 3, e = (c & 1) * 2;
        // (through scalars: a bit cast straight on an ext-vector element reads element 0 under ROCm 7.2's clang)
        const unsigned lo = g[j][h][0], hi = g[j][h][2];
        float* dst = reinterpret_cast<float*>(xs + (size_t)tile * 64 + (q * 16 + i)) + e;
        *reinterpret_cast<float2*>(dst) = make_float2(__uint_as_float(lo), __uint_as_float(hi));      // (zeros where nothing exists)
      }
    }
  }
}

// One wave adds the 16 x 16 tile `tile` of NP row records -- record p0, p0 + pstr, ... (those < np; `rec_stride` granules apart,
// record 0 at `rec_granule0`) -- in ascending order and leaves the sum in dst (LDS: [64] float4, accumulator layout q 16 + row;
// zeros where nothing exists).  A producer's tile is 128 chunks of 16 bytes: the lane takes chunks lane and 64 + lane.
template <int NP>
__device__ __forceinline__ void cx_sweep_tile_sum(__amdgpu_buffer_rsrc_t rr, unsigned rec_granule0, unsigned rec_stride, int p0, int pstr,
                                                  int np, int tile, int nrows, int K, unsigned tag, float4* dst, int lane,
                                                  unsigned* fault) {
  u32x4 g[NP][2];
  bool need[2];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int c = h * 64 + lane;
    need[h] = (c >> 3) < nrows && 16 * tile + 4 * ((c >> 1) & 3) < K;
  }
#pragma unroll
  for (int j = 0; j < NP; ++j)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int p = p0 + j * pstr;
      if (need[h] && p < np) g[j][h] = px_ld(rr, (rec_granule0 + (unsigned)p * rec_stride + (unsigned)(tile * 256)) * 8u + (unsigned)(h * 64 + lane) * 16u);
      else g[j][h] = (u32x4){0u, tag, 0u, tag};
    }
  unsigned spins = 0;
  for (;;) {
    bool ok = true;
#pragma unroll
    for (int j = 0; j < NP; ++j) ok &= g[j][0][1] == tag && g[j][0][3] == tag && g[j][1][1] == tag && g[j][1][3] == tag;
    if (ok) break;
    __builtin_amdgcn_s_sleep(1);
#pragma unroll
    for (int j = 0; j < NP; ++j)
#pragma unroll
      for (int h = 0; h < 2; ++h)
        if (g[j][h][1] != tag || g[j][h][3] != tag)
          g[j][h] = px_ld(rr, (rec_granule0 + (unsigned)(p0 + j * pstr) * rec_stride + (unsigned)(tile * 256)) * 8u + (unsigned)(h * 64 + lane) * 16u);
    if (cx_give_up(spins, fault)) break;
  }
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    float s0 = 0.f, s1 = 0.f;
#pragma unroll
    for (int j = 0; j < NP; ++j) {
      const unsigned lo = g[j][h][0], hi = g[j][h][2];
      s0 += __uint_as_float(lo);
      s1 += __uint_as_float(hi);
    }
    const int c = h * 64 + lane;
    float* d = reinterpret_cast<float*>(dst + (((c >> 1) & 3) * 16 + (c >> 3))) + (c & 1) * 2;
    *reinterpret_cast<float2*>(d) = make_float2(s0, s1);
  }
}

// Column sums of `n` (<= 16) records of 128 granules each -> tot[128] (LDS), by all 256 threads, fixed order.
// Thread (m = tid >> 4, c = tid & 15) owns granule pairs c, c+16, c+32, c+48 of record m and re-reads the ones whose
// tag is stale.  red: LDS [16][128].  Ends with a barrier; tot is valid for every thread afterwards.
// Record m of the sweep is record `rec_of(m)` of `base` (a [PX_MAX_NBLK][128] array).
// `filler()` runs between the first requests and the first look at what they returned: one fabric round trip (~1800 cycles)
// in which the wave would only wait -- the one place in these kernels where independent VALU / LDS work is free (beside the
// fp32 MFMAs it is not: measured, every filler instruction costs its full issue time there).  No vector-memory instruction
// in a filler: vmcnt counts in order, the sweep's wait would include it.
#ifdef G2V_PERSIST_DEVICE_CODE
struct PxNoFiller {
  __device__ __forceinline__ void operator()() const {}
};
template <class RecOf, class Filler = PxNoFiller>
__device__ __forceinline__ void px_sweep_sum(const unsigned long long* base, int n, RecOf rec_of, unsigned tag, float* red,
                                             float* tot, int tid, int stamp_off = 0, Filler&& filler = Filler()) {
  __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned long long*>(base), 0,
                                                                PX_MAX_NBLK * PX_COLS * 8, 0x00020000);
  const int m = tid >> 4, c = tid & 15;
  const int rec = rec_of(m < n ? m : 0);
  float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  PX_STAMP(stamp_off + 0);
  u32x4 g[4];
  if (m < n) {
#pragma unroll
    for (int k = 0; k < 4; ++k) g[k] = px_ld(rr, (unsigned)((rec * PX_COLS + 2 * (c + 16 * k)) * 8));
  }
  filler();      // (every lane: outside the m < n test)
  if (m < n) {
    unsigned spins = 0;
    for (;;) {
      bool ok = true;
#pragma unroll
      for (int k = 0; k < 4; ++k) ok &= (g[k][1] == tag) && (g[k][3] == tag);
      if (ok) break;
      __builtin_amdgcn_s_sleep(1);
#pragma unroll
      for (int k = 0; k < 4; ++k)
        if (g[k][1] != tag || g[k][3] != tag) g[k] = px_ld(rr, (unsigned)((rec * PX_COLS + 2 * (c + 16 * k)) * 8));
      ++spins;
      if (spins > 4000000u || ((spins & 4095u) == 0 &&
                               __hip_atomic_load(&g2v_persist_fault, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u)) {
        // a workgroup of this launch is not resident / died: latch the fault and stop waiting (see the header comment)
        __hip_atomic_store(&g2v_persist_fault, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        break;
      }
    }
    // NB: copy the vector elements to scalars first.  `__builtin_bit_cast(float, g[k][2])` straight on an ext-vector
    // element is compiled by ROCm 7.2's clang as a cast of ELEMENT 0 (seen in the IR: both floats come from lane 0 of
    // the vector) -- every odd column of the sums silently became a copy of its even neighbour.
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const unsigned lo = g[k][0], hi = g[k][2];
      v[2 * k] = __uint_as_float(lo);
      v[2 * k + 1] = __uint_as_float(hi);
    }
  }
#pragma unroll
  for (int k = 0; k < 4; ++k)
    *reinterpret_cast<float2*>(red + m * PX_COLS + 2 * (c + 16 * k)) = make_float2(v[2 * k], v[2 * k + 1]);
  lds_barrier();
  PX_STAMP(stamp_off + 1);
  if (tid < PX_COLS) {
    // every read requested before the first add (same order of the adds: bitwise the same sums): as `t += red[...]` the compiler
    // waited out one LDS round trip per pair of records, ~1000 cycles per sweep with half of the workgroup idle at the barrier
    float v16[16];
#pragma unroll
    for (int g = 0; g < 16; ++g) v16[g] = red[g * PX_COLS + tid];
    asm volatile("" : "+v"(v16[0]), "+v"(v16[1]), "+v"(v16[2]), "+v"(v16[3]), "+v"(v16[4]), "+v"(v16[5]), "+v"(v16[6]), "+v"(v16[7]));
    asm volatile("" : "+v"(v16[8]), "+v"(v16[9]), "+v"(v16[10]), "+v"(v16[11]), "+v"(v16[12]), "+v"(v16[13]), "+v"(v16[14]), "+v"(v16[15]));
    float t = 0.f;
#pragma unroll
    for (int g = 0; g < 16; ++g) t += v16[g];
    tot[tid] = t;
  }
  lds_barrier();
  PX_STAMP(stamp_off + 2);
}

// The exchange of one step for workgroup b, in two calls so that independent work can sit between the hops.  This
// workgroup's 128 partial sums were already published into rec1[par][b] (px_publish2 from the producing lanes).
//   px_hop1 : sum the records of this workgroup's row (workgroups 16 r .. 16 r + 15) and publish the row sum as rec2[par][b].
//   px_hop2 : sum one published copy of every row's sum: row r' is read from the workgroup of that row in this workgroup's
//             column (column c modulo the row's length: the last row may be short).  On return tot[128] holds the sums
//             over all nblk workgroups, bit-identical in every workgroup.
template <class Filler = PxNoFiller>
__device__ __forceinline__ void px_hop1(const PersistX& x, int par, unsigned tag, int nblk, int b, float* red, float* tot, int tid,
                                        int* xl = nullptr, Filler&& filler = Filler()) {
  const int row = px_logical(b, nblk) / PX_GROUP;
  const int n = min(PX_GROUP, nblk - row * PX_GROUP);
  px_sweep_sum(x.rec1 + (size_t)par * PX_MAX_NBLK * PX_COLS, n, [&](int m) { return px_physical(row * PX_GROUP + m, nblk); }, tag,
               red, tot, tid, 0, static_cast<Filler&&>(filler));
  if (xl != nullptr && *xl < 0) {      // (uniform) the launch's first exchange: may this row's records go through L2 from now on?
    const int v = px_row_on_one_xcd(x, nblk, b);
    lds_barrier();                     // every thread has read *xl
    if (tid == 0) *xl = v;             // (visible behind the barrier below)
  }
  if (tid < 64) {
    __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc(x.rec2 + ((size_t)par * PX_MAX_NBLK + b) * PX_COLS, 0,
                                                                  PX_COLS * 8, 0x00020000);
    px_publish2(rr, 2u * tid, tot[2 * tid], tot[2 * tid + 1], tag);
  }
  lds_barrier();     // tot / red are rewritten by the second sweep: every reader above is done
}
template <class Filler = PxNoFiller>
__device__ __forceinline__ void px_hop2(const PersistX& x, int par, unsigned tag, int nblk, int b, float* red, float* tot, int tid,
                                        Filler&& filler = Filler()) {
  const int nrow = (nblk + PX_GROUP - 1) / PX_GROUP;
  const int col = px_logical(b, nblk) % PX_GROUP;
  px_sweep_sum(x.rec2 + (size_t)par * PX_MAX_NBLK * PX_COLS, nrow,
               [&](int m) { return px_physical(m * PX_GROUP + col % min(PX_GROUP, nblk - m * PX_GROUP), nblk); }, tag, red, tot,
               tid, 3, static_cast<Filler&&>(filler));
}
// One row of workgroups only (nblk <= PX_GROUP, i.e. batches up to 256 rows -- the reference's own 128): the row sum hop 1
// leaves in tot IS the total (hop 2 would add one record to zero: the same bits), so the second hop is skipped.
__device__ __forceinline__ bool px_two_hops(int nblk) { return nblk > PX_GROUP; }
__device__ __forceinline__ void px_exchange(const PersistX& x, int par, unsigned tag, int nblk, int b, float* red, float* tot,
                                            int tid, int* xl = nullptr) {
  px_hop1(x, par, tag, nblk, b, red, tot, tid, xl);
  if (px_two_hops(nblk)) px_hop2(x, par, tag, nblk, b, red, tot, tid);
}

#endif      // G2V_PERSIST_DEVICE_CODE

}  // namespace g2v
