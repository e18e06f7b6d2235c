// Diagnostic (round 3): how fast can ONE workgroup per CU bring a 192 KB L2-resident image onto the CU, by how the requests are
// spread over waves?  Every request is issued before the first use (the fused VQ kernel's pattern).
//   hipcc --offload-arch=gfx950 -O3 gpurun_tools/l2_stream_bench2.hip -o gpurun_tools/l2_stream_bench2
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

__device__ unsigned long long g_stamps[8];

// NW waves, each L loads of 1 KB (all issued up front into registers), NW * L = 192
template <int NW, int L>
__global__ __launch_bounds__(NW * 64) void burst_kernel(const float4* __restrict__ img, float* out) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  float4 v[L];
#pragma unroll
  for (int u = 0; u < L; ++u) v[u] = img[(size_t)(wave * L + u) * 64 + lane];
  __builtin_amdgcn_sched_barrier(0);
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int u = 0; u < L; ++u) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
  __syncthreads();
  const unsigned long long t2 = __builtin_amdgcn_s_memtime();
  if (blockIdx.x == 100 && threadIdx.x == 0) { g_stamps[0] = t1 - t0; g_stamps[1] = t2 - t0; }
  if (acc.x + acc.y + acc.z + acc.w == 123.456f) out[blockIdx.x * NW * 64 + threadIdx.x] = acc.x;
}

// the same bytes by LDS-DMA (global_load_lds_dwordx4): NW waves, L pieces of 1 KB each; 192 KB > 160 KB of LDS, so 128 KB only
template <int NW, int L>
__global__ __launch_bounds__(NW * 64) void dma_kernel(const float4* __restrict__ img, float* out) {
  extern __shared__ __attribute__((aligned(16))) float4 lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll
  for (int u = 0; u < L; ++u) {
    const int blk = wave * L + u;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(img + (size_t)blk * 64 + lane),
                                     (__attribute__((address_space(3))) void*)(lds + (size_t)blk * 64), 16, 0, 0);
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  const unsigned long long t2 = __builtin_amdgcn_s_memtime();
  float4 a = lds[(threadIdx.x * 7) % (NW * L * 64)];
  if (blockIdx.x == 100 && threadIdx.x == 0) { g_stamps[0] = t1 - t0; g_stamps[1] = t2 - t0; }
  if (a.x == 123.456f) out[blockIdx.x * NW * 64 + threadIdx.x] = a.x;
}

template <class F>
static void timeit(const char* name, F launch, int kb) {
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
  printf("%-34s %4d KB per WG: %6.2f us per launch; in-kernel (WG 100, wave 0): issued %5llu, all landed (workgroup) %5llu cycles -> %5.1f B/clk\n",
         name, kb, ms * 1e3 / reps, st[0], st[1], kb * 1024.0 / (double)st[1]);
}

int main() {
  float4* img; float* out;
  (void)hipMalloc(&img, 512 * 1024);
  (void)hipMalloc(&out, 256 * 1024 * 4);
  std::vector<float> h(512 * 256, 1.0f);
  (void)hipMemcpy(img, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  timeit("registers, 4 waves x 48", [&] { hipLaunchKernelGGL((burst_kernel<4, 48>), dim3(256), dim3(256), 0, 0, img, out); }, 192);
  timeit("registers, 8 waves x 24", [&] { hipLaunchKernelGGL((burst_kernel<8, 24>), dim3(256), dim3(512), 0, 0, img, out); }, 192);
  timeit("registers, 16 waves x 12", [&] { hipLaunchKernelGGL((burst_kernel<16, 12>), dim3(256), dim3(1024), 0, 0, img, out); }, 192);
  timeit("registers, 8 waves x 16", [&] { hipLaunchKernelGGL((burst_kernel<8, 16>), dim3(256), dim3(512), 0, 0, img, out); }, 128);
  timeit("registers, 16 waves x 8", [&] { hipLaunchKernelGGL((burst_kernel<16, 8>), dim3(256), dim3(1024), 0, 0, img, out); }, 128);
  timeit("registers, 4 waves x 32", [&] { hipLaunchKernelGGL((burst_kernel<4, 32>), dim3(256), dim3(256), 0, 0, img, out); }, 128);
  timeit("registers, 8 waves x 8", [&] { hipLaunchKernelGGL((burst_kernel<8, 8>), dim3(256), dim3(512), 0, 0, img, out); }, 64);
  timeit("registers, 16 waves x 4", [&] { hipLaunchKernelGGL((burst_kernel<16, 4>), dim3(256), dim3(1024), 0, 0, img, out); }, 64);
  (void)hipFuncSetAttribute((const void*)dma_kernel<8, 16>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
  (void)hipFuncSetAttribute((const void*)dma_kernel<16, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
  (void)hipFuncSetAttribute((const void*)dma_kernel<4, 32>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
  timeit("LDS-DMA, 4 waves x 32", [&] { hipLaunchKernelGGL((dma_kernel<4, 32>), dim3(256), dim3(256), 128 * 1024, 0, img, out); }, 128);
  timeit("LDS-DMA, 8 waves x 16", [&] { hipLaunchKernelGGL((dma_kernel<8, 16>), dim3(256), dim3(512), 128 * 1024, 0, img, out); }, 128);
  timeit("LDS-DMA, 16 waves x 8", [&] { hipLaunchKernelGGL((dma_kernel<16, 8>), dim3(256), dim3(1024), 128 * 1024, 0, img, out); }, 128);
  // one CU alone (no other CU competes for the L2): is the limit on the CU or in the L2?
  timeit("registers, 8 waves x 24, ONE WG", [&] { hipLaunchKernelGGL((burst_kernel<8, 24>), dim3(101), dim3(512), 0, 0, img, out); }, 192);
  return 0;
}
