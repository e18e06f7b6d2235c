// exchange_bench.hip -- price of the per-step all-to-all exchange a PERSISTENT decoder rollout would need on MI355X.
//
// 256 workgroups (one per CU, 256 threads) stay resident for T steps.  Per step every workgroup publishes a 128-float
// record (the BatchNorm partial sums of its 16 rows) and needs the column sums over ALL workgroups' records before it
// can continue.  Variants (argv[1]):
//   0  no exchange (compute phases only)                                  -> baseline
//   1  flags + data : sc1 record stores -> vmcnt(0) -> barrier -> one sc1 flag store per workgroup;
//                     readers poll the 256 flags (one per thread), then read the 256 records with sc1 loads
//   2  self-tagged  : records as 16-byte granules {v0, v1, v2, tag = step}; readers sweep the granules with sc1 loads
//                     and re-load the ones whose tag is stale (no flag, no second round trip)
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
constexpr int NWG = 256, NCOL = 128, NG = 44;   // 44 granules x 3 floats >= 128

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ float rec_val(int b, int e, int s) { return (float)((b * 131 + e * 7 + s * 13) % 1000); }

__device__ __forceinline__ u32x4 ld_sc1(__amdgpu_buffer_rsrc_t r, unsigned byte_off) {
  return __builtin_amdgcn_raw_buffer_load_b128(r, byte_off, 0, 16);
}
__device__ __forceinline__ void st_sc1(__amdgpu_buffer_rsrc_t r, unsigned byte_off, u32x4 v) {
  __builtin_amdgcn_raw_buffer_store_b128(v, r, byte_off, 0, 16);
}

__global__ __launch_bounds__(256) void bench_kernel(float* rec, unsigned* flags, float* sink, float* junk, unsigned* errs,
                                                    int variant, int nmfma, int store_kb, int T, int skew) {
  __shared__ float red[8][NCOL];
  __shared__ float tot[NCOL];
  __shared__ int ok_s;
  const int tid = threadIdx.x, b = blockIdx.x, lane = tid & 63;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  float a = 1.0f + tid * 1e-3f, bb = 0.5f;
  unsigned nerr = 0;
  const size_t rec_bytes_a = (size_t)NWG * NCOL * 4, rec_bytes_b = (size_t)NWG * NG * 16;
  for (int s = 1; s <= T; ++s) {
    // ---- emulated compute phase
    for (int m = 0; m < nmfma; ++m) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bb, acc, 0, 0, 0);
    for (int j = 0; j < store_kb / 4; ++j)      // 4 KB per iteration: 256 threads x 16 B, plain stores
      reinterpret_cast<f32x4*>(junk)[((size_t)b * 64 + (j & 63)) * 256 + tid] = acc;
    if (skew && ((b * 7 + s) & 15) == 0) __builtin_amdgcn_s_sleep(100);
    if (variant == 0) continue;
    const int par = s & 1;
    if (variant == 1) {
      __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc(rec + (size_t)par * NWG * NCOL, 0, (int)rec_bytes_a, 0x00020000);
      if (tid < 32) {
        f32x4 v;
        for (int r = 0; r < 4; ++r) v[r] = rec_val(b, 4 * tid + r, s);
        st_sc1(rr, (unsigned)(b * NCOL * 4 + tid * 16), __builtin_bit_cast(u32x4, v));
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (tid == 0) __hip_atomic_store(flags + par * NWG + b, (unsigned)s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      // poll: thread t watches workgroup t's flag
      int spins = 0;
      for (;;) {
        const unsigned f = __hip_atomic_load(flags + par * NWG + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int ok = __syncthreads_and(f == (unsigned)s);
        if (ok) break;
        if (++spins > 2000000) { nerr += 1000000; break; }
      }
      const int seg = tid >> 5, col = tid & 31;
      f32x4 sum = {0.f, 0.f, 0.f, 0.f};
      for (int k0 = 0; k0 < 32; k0 += 16) {
        u32x4 v[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) v[j] = ld_sc1(rr, (unsigned)(((seg * 32 + k0 + j) * NCOL + 4 * col) * 4));
#pragma unroll
        for (int j = 0; j < 16; ++j) sum += __builtin_bit_cast(f32x4, v[j]);
      }
      for (int r = 0; r < 4; ++r) red[seg][4 * col + r] = sum[r];
      __syncthreads();
      if (tid < NCOL) {
        float t = 0.f;
        for (int g = 0; g < 8; ++g) t += red[g][tid];
        tot[tid] = t;
      }
      __syncthreads();
    } else {
      __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc(rec + (size_t)par * NWG * NG * 4, 0, (int)rec_bytes_b, 0x00020000);
      if (tid < NG) {
        u32x4 v;
        for (int r = 0; r < 3; ++r) {
          const int e = 3 * tid + r;
          v[r] = __builtin_bit_cast(unsigned, e < NCOL ? rec_val(b, e, s) : 0.f);
        }
        v[3] = (unsigned)s;
        st_sc1(rr, (unsigned)((b * NG + tid) * 16), v);
      }
      // sweep: thread (seg, col) sums granule `col` of workgroups seg*52 .. seg*52+51 (5 segments x 44 columns)
      const int seg = tid / NG, col = tid - seg * NG;
      float s0 = 0.f, s1 = 0.f, s2 = 0.f;
      if (seg < 5) {
        const int kb = seg * 52, ke = min(NWG, kb + 52);
        for (int k0 = kb; k0 < ke; k0 += 13) {
          u32x4 v[13];
#pragma unroll
          for (int j = 0; j < 13; ++j) v[j] = ld_sc1(rr, (unsigned)((min(k0 + j, ke - 1) * NG + col) * 16));
#pragma unroll
          for (int j = 0; j < 13; ++j) {
            if (k0 + j >= ke) continue;
            int spins = 0;
            while (v[j][3] != (unsigned)s) {
              v[j] = ld_sc1(rr, (unsigned)(((k0 + j) * NG + col) * 16));
              if (++spins > 2000000) { nerr += 1000000; break; }
            }
            s0 += __builtin_bit_cast(float, v[j][0]);
            s1 += __builtin_bit_cast(float, v[j][1]);
            s2 += __builtin_bit_cast(float, v[j][2]);
          }
        }
      }
      __syncthreads();
      if (seg < 5) {
        if (3 * col < NCOL) red[seg][3 * col] = s0;
        if (3 * col + 1 < NCOL) red[seg][3 * col + 1] = s1;
        if (3 * col + 2 < NCOL) red[seg][3 * col + 2] = s2;
      }
      __syncthreads();
      if (tid < NCOL) {
        float t = 0.f;
        for (int g = 0; g < 5; ++g) t += red[g][tid];
        tot[tid] = t;
      }
      __syncthreads();
    }
    if (tid < NCOL) {
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
  const int T = argc > 4 ? atoi(argv[4]) : 33, skew = argc > 5 ? atoi(argv[5]) : 0;
  float *rec, *sink, *junk;
  unsigned *flags, *errs;
  CK(hipMalloc(&rec, 2 * NWG * NG * 16 + 2 * NWG * NCOL * 4));
  CK(hipMalloc(&flags, 2 * NWG * 4));
  CK(hipMalloc(&sink, NWG * 256 * 4));
  CK(hipMalloc(&junk, (size_t)NWG * 64 * 4096));
  CK(hipMalloc(&errs, 4));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float best = 1e30f, sum = 0.f;
  const int reps = 20;
  unsigned herr = 0;
  for (int r = 0; r < reps + 3; ++r) {
    CK(hipMemset(rec, 0, 2 * NWG * NG * 16 + 2 * NWG * NCOL * 4));
    CK(hipMemset(flags, 0, 2 * NWG * 4));
    CK(hipMemset(errs, 0, 4));
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(bench_kernel, dim3(NWG), dim3(256), 0, 0, rec, flags, sink, junk, errs, variant, nmfma, store_kb, T, skew);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    unsigned e;
    CK(hipMemcpy(&e, errs, 4, hipMemcpyDeviceToHost));
    herr += e;
    if (r >= 3) { best = ms < best ? ms : best; sum += ms; }
  }
  printf("variant %d nmfma %d store_kb %d T %d skew %d : avg %.2f us/step  best %.2f us/step  (kernel avg %.1f us)  errors %u\n",
         variant, nmfma, store_kb, T, skew, sum / reps * 1e3f / T, best * 1e3f / T, sum / reps * 1e3f, herr);
  return herr ? 2 : 0;
}
