// Diagnostic (round 3): does it matter how many separate ALLOCATIONS (pages / translations) a CU's request burst touches?
//   192 KB per workgroup, 8 waves x 24 requests of 1 KB, taken from NB buffers that were hipMalloc'ed separately (4 MB each).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
__device__ unsigned long long g_stamps[8];
struct Ptrs { const float4* p[24]; };
template <int NB>
__global__ __launch_bounds__(512) void burst_kernel(Ptrs ps, float* out) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  float4 v[24];
#pragma unroll
  for (int u = 0; u < 24; ++u) {
    const int blk = wave * 24 + u;                 // 0..191
    const int b = u % NB;                          // buffer of this request (compile-time), block offset inside it
    v[u] = ps.p[b][(size_t)(blk / NB) * 64 + lane];
  }
  __builtin_amdgcn_sched_barrier(0);
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int u = 0; u < 24; ++u) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
  __syncthreads();
  const unsigned long long t2 = __builtin_amdgcn_s_memtime();
  if (blockIdx.x == 100 && threadIdx.x == 0) { g_stamps[0] = t1 - t0; g_stamps[1] = t2 - t0; }
  if (acc.x + acc.y + acc.z + acc.w == 123.456f) out[blockIdx.x * 512 + threadIdx.x] = acc.x;
}
template <class F>
static void timeit(const char* name, F launch) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int i = 0; i < 20; ++i) launch();
  (void)hipEventRecord(e0, 0);
  const int reps = 200;
  for (int i = 0; i < reps; ++i) launch();
  (void)hipEventRecord(e1, 0);
  (void)hipEventSynchronize(e1);
  float ms;
  (void)hipEventElapsedTime(&ms, e0, e1);
  unsigned long long st[8];
  (void)hipMemcpyFromSymbol(st, HIP_SYMBOL(g_stamps), sizeof(st));
  printf("%-30s %6.2f us per launch; in-kernel: issued %5llu, landed %5llu cycles -> %5.1f B/clk\n", name, ms * 1e3 / reps, st[0], st[1],
         192 * 1024.0 / (double)st[1]);
}
int main() {
  Ptrs ps; float* out;
  std::vector<float> h(1 << 20, 1.0f);
  for (int b = 0; b < 24; ++b) {
    void* p; (void)hipMalloc(&p, 4 << 20);
    (void)hipMemcpy(p, h.data(), 4 << 20, hipMemcpyHostToDevice);
    ps.p[b] = (const float4*)p;
  }
  (void)hipMalloc(&out, 256 * 512 * 4);
  timeit("1 buffer", [&] { hipLaunchKernelGGL((burst_kernel<1>), dim3(256), dim3(512), 0, 0, ps, out); });
  timeit("2 buffers", [&] { hipLaunchKernelGGL((burst_kernel<2>), dim3(256), dim3(512), 0, 0, ps, out); });
  timeit("4 buffers", [&] { hipLaunchKernelGGL((burst_kernel<4>), dim3(256), dim3(512), 0, 0, ps, out); });
  timeit("8 buffers", [&] { hipLaunchKernelGGL((burst_kernel<8>), dim3(256), dim3(512), 0, 0, ps, out); });
  timeit("12 buffers", [&] { hipLaunchKernelGGL((burst_kernel<12>), dim3(256), dim3(512), 0, 0, ps, out); });
  timeit("24 buffers", [&] { hipLaunchKernelGGL((burst_kernel<24>), dim3(256), dim3(512), 0, 0, ps, out); });
  // one buffer, but blocks 16 KB apart (strided inside one allocation)
  return 0;
}
