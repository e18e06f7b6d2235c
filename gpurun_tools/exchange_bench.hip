// exchange_bench.hip -- price of the per-step all-to-all exchange a PERSISTENT decoder rollout would need on MI355X.
//
// 256 workgroups (one per CU, 256 threads) stay resident for T steps.  Per step every workgroup publishes a 128-float
// record (the BatchNorm partial sums of its 16 rows) and needs the column sums over ALL workgroups' records before it
// can continue.  Variants (argv[1]):
//   0  no exchange (compute phases only)                                  -> baseline
//   1  flat, all threads poll : sc1 record stores -> vmcnt(0) -> barrier -> one sc1 flag store per workgroup;
//                     every thread polls one flag, then the 256 records are read with sc1 loads      (round 2a: 10 us)
//   3  flat, ONE wave polls (64 lanes x 16 B = all 256 flags in one request per lane) with s_sleep, then as 1
//   4  two-level tree, groups of G (argv[6], 8/16/32): members publish record + flag; the group's first workgroup sums
//      its G records and publishes a level-2 record + flag; everybody polls the 256/G leader flags (one wave, one load)
//      and sums the 256/G level-2 records.  Fixed summation tree: deterministic.
//   5  as 4 but poll only (no record reads) -> isolates the flag latency of the two hops
//   6  flat with fences: plain record stores -> vmcnt(0) -> barrier -> lane 0 release fence (agent) -> flag;
//      consumer: one wave polls, one agent acquire, barrier, PLAIN loads of the 256 records
// argv[2] = MFMAs per wave per step in the emulated compute phase, argv[3] = KB of plain stores per workgroup and step,
// argv[4] = steps, argv[5] = skew (1: some workgroups sleep extra each step).
// Every value received is checked against the closed form; a bounded spin turns a protocol bug into an error count.
//   hipcc --offload-arch=gfx950 -O3 -o exchange_bench exchange_bench.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
constexpr int NWG = 256, NCOL = 128;
constexpr int SPIN_MAX = 400000;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ float rec_val(int b, int e, int s) { return (float)((b * 131 + e * 7 + s * 13) % 1000); }

__device__ __forceinline__ u32x4 ld_sc1(__amdgpu_buffer_rsrc_t r, unsigned byte_off) {
  return __builtin_amdgcn_raw_buffer_load_b128(r, byte_off, 0, 16);
}
__device__ __forceinline__ void st_sc1(__amdgpu_buffer_rsrc_t r, unsigned byte_off, u32x4 v) {
  __builtin_amdgcn_raw_buffer_store_b128(v, r, byte_off, 0, 16);
}

struct Bufs {
  float* rec;        // [2][NWG][128]   level-1 records
  float* rec2;       // [2][32][128]    level-2 records
  unsigned* flags;   // [2][NWG]        level-1 flags (group-contiguous)
  unsigned* flags2;  // [2][32]         leader flags
};

// column sums of `n` records (n <= 256, multiple of 8) read with sc1 (or plain) loads -> tot[128] in LDS
template <bool PLAIN>
__device__ __forceinline__ void sum_records(const float* base, int n, float (*red)[NCOL], float* tot, int tid) {
  __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, n * NCOL * 4, 0x00020000);
  const int seg = tid >> 5, col = tid & 31;
  const int per = n >> 3;                 // records per segment
  f32x4 sum = {0.f, 0.f, 0.f, 0.f};
  for (int k0 = 0; k0 < per; k0 += 16) {
    u32x4 v[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const int k = seg * per + min(k0 + j, per - 1);
      if (PLAIN) v[j] = *reinterpret_cast<const u32x4*>(base + (size_t)k * NCOL + 4 * col);
      else v[j] = ld_sc1(rr, (unsigned)((k * NCOL + 4 * col) * 4));
    }
#pragma unroll
    for (int j = 0; j < 16; ++j)
      if (k0 + j < per) sum += __builtin_bit_cast(f32x4, v[j]);
  }
  for (int r = 0; r < 4; ++r) red[seg][4 * col + r] = sum[r];
  __syncthreads();
  if (tid < NCOL) {
    float t = 0.f;
    for (int g = 0; g < 8; ++g) t += red[g][tid];
    tot[tid] = t;
  }
  __syncthreads();
}

// wave 0 polls `n` consecutive flags (n <= 256, multiple of 4) until all equal s; everybody leaves through a barrier
__device__ __forceinline__ unsigned wait_flags(const unsigned* f, int n, unsigned s, int tid) {
  unsigned fail = 0;
  if (tid < 64) {
    __amdgpu_buffer_rsrc_t fr = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned*>(f), 0, n * 4, 0x00020000);
    const bool active = 4 * tid < n;
    for (int spins = 0;; ++spins) {
      bool ok = true;
      if (active) {
        const u32x4 v = ld_sc1(fr, (unsigned)(tid * 16));
        ok = (v[0] == s) && (v[1] == s) && (v[2] == s) && (v[3] == s);
      }
      if (__all(ok)) break;
      if (spins > SPIN_MAX) { fail = 1000000; break; }
      __builtin_amdgcn_s_sleep(2);
    }
  }
  __syncthreads();
  return fail;
}

__global__ __launch_bounds__(256) void bench_kernel(Bufs B, float* sink, float* junk, unsigned* errs,
                                                    int variant, int nmfma, int store_kb, int T, int skew, int G) {
  __shared__ float red[8][NCOL];
  __shared__ float tot[NCOL];
  const int tid = threadIdx.x, b = blockIdx.x;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  float a = 1.0f + tid * 1e-3f, bb = 0.5f;
  unsigned nerr = 0;
  const int NL = NWG / G;       // leaders
  for (int s = 1; s <= T; ++s) {
    // ---- emulated compute phase
    for (int m = 0; m < nmfma; ++m) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bb, acc, 0, 0, 0);
    for (int j = 0; j < store_kb / 4; ++j)      // 4 KB per iteration: 256 threads x 16 B, plain stores
      reinterpret_cast<f32x4*>(junk)[((size_t)b * 64 + (j & 63)) * 256 + tid] = acc;
    if (skew && ((b * 7 + s) & 15) == 0) __builtin_amdgcn_s_sleep(100);
    if (variant == 0) continue;
    const int par = s & 1;
    float* rec = B.rec + (size_t)par * NWG * NCOL;
    unsigned* flags = B.flags + par * NWG;
    // ---- publish this workgroup's record
    if (variant == 6) {
      if (tid < 32) {
        f32x4 v;
        for (int r = 0; r < 4; ++r) v[r] = rec_val(b, 4 * tid + r, s);
        *reinterpret_cast<f32x4*>(rec + (size_t)b * NCOL + 4 * tid) = v;
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (tid == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_store(flags + b, (unsigned)s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    } else {
      __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc(rec, 0, NWG * NCOL * 4, 0x00020000);
      if (tid < 32) {
        f32x4 v;
        for (int r = 0; r < 4; ++r) v[r] = rec_val(b, 4 * tid + r, s);
        st_sc1(rr, (unsigned)(b * NCOL * 4 + tid * 16), __builtin_bit_cast(u32x4, v));
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (tid == 0) __hip_atomic_store(flags + b, (unsigned)s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // ---- gather
    if (variant == 1) {
      int spins = 0;
      for (;;) {
        const unsigned f = __hip_atomic_load(flags + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int ok = __syncthreads_and(f == (unsigned)s);
        if (ok) break;
        if (++spins > SPIN_MAX) { nerr += 1000000; break; }
      }
      sum_records<false>(rec, NWG, red, tot, tid);
    } else if (variant == 3) {
      nerr += wait_flags(flags, NWG, (unsigned)s, tid);
      sum_records<false>(rec, NWG, red, tot, tid);
    } else if (variant == 6) {
      nerr += wait_flags(flags, NWG, (unsigned)s, tid);     // ends in a barrier
      if (tid == 0) {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      __syncthreads();
      sum_records<true>(rec, NWG, red, tot, tid);
    } else {   // 4, 5: two-level tree
      float* rec2 = B.rec2 + (size_t)par * 32 * NCOL;
      unsigned* flags2 = B.flags2 + par * 32;
      const int grp = b / G;
      if (b == grp * G) {     // leader of its group
        nerr += wait_flags(flags + grp * G, G, (unsigned)s, tid);
        if (variant == 4) sum_records<false>(rec + (size_t)grp * G * NCOL, G, red, tot, tid);
        else { if (tid < NCOL) tot[tid] = 0.f; __syncthreads(); }
        __amdgpu_buffer_rsrc_t r2 = __builtin_amdgcn_make_buffer_rsrc(rec2, 0, 32 * NCOL * 4, 0x00020000);
        if (tid < 32) {
          f32x4 v;
          for (int r = 0; r < 4; ++r) v[r] = tot[4 * tid + r];
          st_sc1(r2, (unsigned)(grp * NCOL * 4 + tid * 16), __builtin_bit_cast(u32x4, v));
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) __hip_atomic_store(flags2 + grp, (unsigned)s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      nerr += wait_flags(flags2, NL, (unsigned)s, tid);
      if (variant == 4) sum_records<false>(rec2, NL, red, tot, tid);
    }
    if (variant != 5 && tid < NCOL) {
      float expect = 0.f;
      for (int k = 0; k < NWG; ++k) expect += rec_val(k, tid, s);
      if (tot[tid] != expect) ++nerr;
      a += tot[tid] * 1e-12f;      // the next compute phase depends on the exchanged value
    }
  }
  if (nerr) atomicAdd(errs, nerr);
  sink[(size_t)b * 256 + tid] = acc[0] + acc[1] + acc[2] + acc[3];
}

int main(int argc, char** argv) {
  const int variant = argc > 1 ? atoi(argv[1]) : 1, nmfma = argc > 2 ? atoi(argv[2]) : 168, store_kb = argc > 3 ? atoi(argv[3]) : 64;
  const int T = argc > 4 ? atoi(argv[4]) : 33, skew = argc > 5 ? atoi(argv[5]) : 0, G = argc > 6 ? atoi(argv[6]) : 16;
  Bufs B;
  float *sink, *junk;
  unsigned* errs;
  char* state;
  const size_t state_bytes = 2 * NWG * NCOL * 4 + 2 * 32 * NCOL * 4 + 2 * NWG * 4 + 2 * 32 * 4;
  CK(hipMalloc(&state, state_bytes));
  B.rec = (float*)state;
  B.rec2 = B.rec + 2 * NWG * NCOL;
  B.flags = (unsigned*)(B.rec2 + 2 * 32 * NCOL);
  B.flags2 = B.flags + 2 * NWG;
  CK(hipMalloc(&sink, NWG * 256 * 4));
  CK(hipMalloc(&junk, (size_t)NWG * 64 * 4096));
  CK(hipMalloc(&errs, 4));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float best = 1e30f, sum = 0.f;
  const int reps = 20;
  unsigned herr = 0;
  for (int r = 0; r < reps + 3; ++r) {
    CK(hipMemset(state, 0, state_bytes));
    CK(hipMemset(errs, 0, 4));
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(bench_kernel, dim3(NWG), dim3(256), 0, 0, B, sink, junk, errs, variant, nmfma, store_kb, T, skew, G);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    unsigned e;
    CK(hipMemcpy(&e, errs, 4, hipMemcpyDeviceToHost));
    herr += e;
    if (r >= 3) { best = ms < best ? ms : best; sum += ms; }
  }
  printf("variant %d G %2d nmfma %3d store_kb %2d T %d skew %d : avg %6.2f us/step  best %6.2f us/step  errors %u\n",
         variant, G, nmfma, store_kb, T, skew, sum / reps * 1e3f / T, best * 1e3f / T, herr);
  return herr ? 2 : 0;
}
